#!/bin/bash
# k_msm_accumulate_fb with six clients in flight: kernel durations (rocprofv3 --kernel-trace --stats) and HBM fetch traffic
# (separate --pmc FETCH_SIZE pass) of the same command, next to the uncontended single-client figures of profiles/r02_*.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/contended; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 scripts/gpu_throughput.py lanes 6 > $OUT/run_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- python3 scripts/gpu_throughput.py lanes 6 > $OUT/run_pmc.log 2>&1
python3 - <<'PY'
import csv, glob, json
out = {}
f = glob.glob("gpurun_out/contended/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    if "accumulate_fb" in n or "fold_gens_tab" in n or "bin_l1" in n:
        out[n] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
acc = {}
for f in glob.glob("gpurun_out/contended/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        n = r["Kernel_Name"].split("(")[0]
        if n in out: a = acc.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
for n, (c, v) in acc.items():
    out[n]["fetch_mb_per_launch_raw"] = v / c / 1024; out[n]["fetch_mb_per_launch_x2"] = 2 * v / c / 1024
    out[n]["fetch_TBps_at_avg_duration"] = 2 * v / c * 1024 / (out[n]["avg_ms"] * 1e-3) / 1e12
json.dump({"command": "scripts/gpu_throughput.py lanes 6 (six cfg-2 clients in flight, create + verify)", "kernels": out}, open("gpurun_out/contended/r02_contended_accumulate.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -2 $OUT/run_stats.log
rm -rf $OUT/stats $OUT/pmc
