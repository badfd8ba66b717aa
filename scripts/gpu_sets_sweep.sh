#!/bin/bash
# fixed-base MSM: 4 bucket sets per problem (512 k accumulate threads, two batches of blocks) against 2 (256 k, one batch)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --no-extras --steps 8 --warmup 2 > gpurun_out/sets_$tag.json 2>/dev/null; python3 - <<PY
import json
j=json.load(open("gpurun_out/sets_$tag.json"))
k={r["kernel"][:18]:round(r["ms_per_client"],2) for r in j["kernels"]["top"]}
print("$tag", round(j["median_ms_per_step"],2), k)
PY
}
run t512k
run t256k ROFL_MSM_FB_THREADS=262144
run t512kb
run t256kb ROFL_MSM_FB_THREADS=262144
