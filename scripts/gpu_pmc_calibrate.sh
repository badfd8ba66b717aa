#!/bin/bash
# FETCH_SIZE calibration for the access pattern of k_msm_accumulate_fb: every lane pulls its own 128-byte record with
# eight global_load_dwordx4, records uniformly at random from a table far larger than L2 + Infinity Cache.  rofl_bench_femul mode 3 issues a
# KNOWN number of such gathers (threads x iterations); the ratio (bytes gathered) / (FETCH_SIZE x 1024) is the correction factor for this
# pattern -- the guide's x2 was calibrated on coalesced 16 B/lane streams only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_cal; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import rofl_project_code_amd as R
R.set_device(0)
print("%.4e" % R.bench_femul(int(os.environ.get("ITERS", "64"))))
PY
for TAB in 8388608 33554432; do
  for ITERS in 64 128; do
    d=$OUT/t${TAB}_i${ITERS}
    ROFL_FEMUL_MODE=3 ROFL_FEMUL_TABLE=$TAB ROFL_FEMUL_LDS=40960 ITERS=$ITERS rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 $OUT/run.py > $d.out 2> $d.err
  done
done
python3 - <<PY
import csv, glob, json, os
res = []
for d in sorted(glob.glob("$OUT/t*_i*")):
    if not os.path.isdir(d): continue
    tab, iters = [int(x[1:]) for x in os.path.basename(d).split("_")]
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "k_bench_madd_gather" in r["Kernel_Name"]:
                rows.append(float(r["Counter_Value"]))
    threads = 256 * 8 * 256
    # two dispatches per run: the warm-up (8 iterations) and the measured one
    rows.sort()
    for val, it in zip(rows, (8, iters)):
        expect = threads * it * 128
        res.append({"table_bytes": tab * 128, "iterations": it, "gathers": threads * it, "bytes_gathered": expect, "FETCH_SIZE_KB": val,
                    "bytes_per_FETCH_SIZE_byte": expect / (val * 1024.0)})
json.dump({"pattern": "one 128-byte record per lane (eight global_load_dwordx4), uniformly random over the table; 4 waves/SIMD", "runs": res}, open("$R/gpurun_out/pmc_calibration.json", "w"), indent=1)
for r in res: print(r)
PY
