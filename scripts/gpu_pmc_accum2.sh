#!/bin/bash
# Round 2: per-instruction rate of k_msm_accumulate_fb against the mixed-addition microbenchmark (k_bench_madd) under the same counters.
# Each --pmc pass also records the kernel trace, so cycles / duration gives the clock the kernel actually ran at.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > scripts/_pmc_tmp.py <<'PY'
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
os.environ["ROFL_FEMUL_MODE"] = "2"
R.bench_femul(64)  # ROFL_FEMUL_MODE: 2 = cache-resident entries, 3 = random gathers from ROFL_FEMUL_TABLE entries
vals, bl = bench.synth_client(1)
for i in range(3):
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
PY
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  ROFL_FEMUL_MODE=2 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/_pmc_tmp.py > /dev/null 2> gpurun_out/pmc_$tag.err
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); nd = collections.Counter()
want = ("k_msm_accumulate_fb", "k_bench_madd_l1", "k_bench_madd_gather")
for f in glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("rofl::", "")
        if k not in want: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob("gpurun_out/pmc_$tag/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("rofl::", "")
        if k not in want: continue
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9; nd[k] += 1
for k, v in acc.items(): print(k, "launches", nd[k], "dur_ms %.3f" % (dur[k] * 1e3), {c: "%.4g" % x for c, x in v.items()})
PY
  rm -rf gpurun_out/pmc_$tag
done
rm -f scripts/_pmc_tmp.py
