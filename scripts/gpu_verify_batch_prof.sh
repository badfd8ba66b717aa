#!/bin/bash
# kernel stats of the 48-client batch verification (verify_batch = 2): rocprofv3 --kernel-trace --stats over scripts/gpu_verify_batch.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/vbprof
MODES=2 TAMPER=0 REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vbprof -o vb -- python3 $R/scripts/gpu_verify_batch.py > $R/gpurun_out/vbprof/out.json 2> $R/gpurun_out/vbprof/err.txt
f=$(find $R/gpurun_out/vbprof -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/vb_kernel_stats.csv
find $R/gpurun_out/vbprof -name "*.csv" ! -name "*kernel_stats.csv" -delete
find $R/gpurun_out/vbprof -name "*.db" -delete
cut -c1-60,200- $R/gpurun_out/vb_kernel_stats.csv | head -5
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/vb_kernel_stats.csv")))
for r in rows[:25]:
    print("%-40s calls %5s total %9.3f ms avg %9.3f ms" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
cat $R/gpurun_out/vbprof/out.json
