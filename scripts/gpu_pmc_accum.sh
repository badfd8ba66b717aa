#!/bin/bash
# Where do the wave cycles of k_msm_accumulate go?  SQ counters of one sequential client (own --pmc passes, no tracing).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
sed -e 's/os.environ\["ROFL_TRACE"\] = "2"/pass/' scripts/gpu_trace1.py > scripts/_pmc_tmp.py
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_$tag
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_$tag -- python3 scripts/_pmc_tmp.py > /dev/null 2> gpurun_out/pmc_$tag.err
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("rofl::", "")
        if k not in ("k_msm_accumulate", "k_fold_gens_tab", "k_fold_gens", "k_msm_scatter_lds"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items(): print(k, {c: "%.4g" % x for c, x in v.items()})
PY
  rm -rf gpurun_out/pmc_$tag
done
rm -f scripts/_pmc_tmp.py
