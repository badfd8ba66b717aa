#!/bin/bash
# A/B of k_msm_accumulate_fb between the shipped library and a variant build (ROFL_BUILD_VARIANT, default "nochunk"): kernel time from
# rocprofv3 --kernel-trace --stats and HBM traffic from a separate --pmc FETCH_SIZE pass, both over four sequential cfg-2 creates.
V=${1:-nochunk}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/acc_ab; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import rofl_project_code_amd as R
import bench
R.set_device(0)
vals, bl = bench.synth_client(1)
for i in range(5):
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(b"\x01" * 32), fp=(32, 7))
    assert R.range_proof_vec.verify_rangeproof(pr, cm, 32, fp=(32, 7))
PY
for tag in new $V; do
  lib=""; [ $tag != new ] && lib=$R/rofl_project_code_amd/build/librofl_zk_$tag.so
  ROFL_ZK_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$tag -- python3 $OUT/run.py > /dev/null 2> $OUT/st_$tag.err
  ROFL_ZK_LIB=$lib rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pm_$tag -- python3 $OUT/run.py > /dev/null 2> $OUT/pm_$tag.err
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for tag in ("new", "$V"):
    st = {}
    for f in glob.glob("$OUT/st_%s/**/*kernel_stats.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Name"].split("(")[0].replace("rofl::", "")
            st[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6)
    fe = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob("$OUT/pm_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "FETCH_SIZE": continue
            k = r["Kernel_Name"].split("(")[0].replace("rofl::", ""); fe[k][0] += 1; fe[k][1] += float(r["Counter_Value"])
    res[tag] = {k: {"calls": st[k][0], "avg_ms": round(st[k][1], 4), "total_ms": round(st[k][2], 2), "fetch_GB_per_launch_x2": round(2 * fe[k][1] * 1024 / max(fe[k][0], 1) / 1e9, 3) if k in fe else None}
                for k in ("k_msm_accumulate_fb", "k_msm_accumulate_gen", "k_msm_small", "k_fold_gens_tab", "k_fold_gens4", "k_msm_reduce_level", "k_msm_reduce_fused") if k in st}
    res[tag]["all_kernels_total_ms"] = round(sum(v[2] for v in st.values()), 2)
json.dump(res, open("$R/gpurun_out/acc_ab.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $OUT/st_* $OUT/pm_*
