#!/bin/bash
# reduction-kernel knobs, cfg 2 single client (bench headline only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --no-extras --steps 8 --warmup 2 > gpurun_out/red_$tag.json 2>/dev/null; python3 - <<PY
import json
j=json.load(open("gpurun_out/red_$tag.json"))
k={r["kernel"]:round(r["ms_per_client"],3) for r in j["kernels"]["top"]}
print("$tag", round(j["median_ms_per_step"],2), k.get("k_msm_reduce_level+fused"), k.get("k_msm_small"))
PY
}
run base A=1
run fused256 ROFL_RED_FUSED_T=256
run split ROFL_RED_SPLIT=1
run split256 ROFL_RED_SPLIT=1 ROFL_RED_FUSED_T=256
run group ROFL_MSM_GROUP_REDUCE=1
