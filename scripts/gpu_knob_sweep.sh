#!/bin/bash
# schedule knobs re-swept with the round-2 kernels, cfg 2 single client (bench headline only)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --no-extras --steps 8 --warmup 2 > gpurun_out/knob_$tag.json 2>/dev/null; python3 - <<PY
import json
try:
    j=json.load(open("gpurun_out/knob_$tag.json"))
    print("$tag", round(j["median_ms_per_step"],2), round(j["breakdown_ms_per_client"]["create"],2), round(j["breakdown_ms_per_client"]["verify"],2))
except Exception as e: print("$tag", "ERR", e)
PY
}
run base A=1
run t1_2 ROFL_FOLD_T1=2
run t1_4 ROFL_FOLD_T1=4
run t_3 ROFL_FOLD_T=3
run min512 ROFL_FOLD_MIN=512
run min2048 ROFL_FOLD_MIN=2048
run min4096 ROFL_FOLD_MIN=4096
run sets1 ROFL_MSM_FB_SETS=1
run sets4 ROFL_MSM_FB_SETS=4
run small8192 ROFL_MSM_SMALL_MAX=8192
run small2048 ROFL_MSM_SMALL_MAX=2048
run host16 ROFL_HOST_THREADS=16
run host4 ROFL_HOST_THREADS=4
