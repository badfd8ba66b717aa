#!/bin/bash
# k_msm_accumulate block mapping: plain descending order (0) against equal-work blocks (1), and the grid sizes around the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --no-extras --steps 8 --warmup 2 > gpurun_out/bal_$tag.json 2>/dev/null; python3 - <<PY
import json
j=json.load(open("gpurun_out/bal_$tag.json"))
k={r["kernel"][:18]:round(r["ms_per_client"],2) for r in j["kernels"]["top"]}
print("$tag", round(j["median_ms_per_step"],2), k)
PY
}
run b0 ROFL_ACC_BALANCE=0
run b1 ROFL_ACC_BALANCE=1
run b1_256k ROFL_ACC_BALANCE=1 ROFL_MSM_FB_THREADS=262144
