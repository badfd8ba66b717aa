#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --no-extras --steps 8 --warmup 2 > gpurun_out/fold_$tag.json 2>/dev/null; python3 - <<PY
import json
j=json.load(open("gpurun_out/fold_$tag.json"))
k={r["kernel"]:round(r["ms_per_client"],3) for r in j["kernels"]["top"]}
print("$tag", round(j["median_ms_per_step"],2), "fold", k.get("k_fold_gens"), "fold_tab", k.get("k_fold_gens_tab"))
PY
}
run regs A=1
run noregs ROFL_FOLD_REGS=0
run regs_k1 ROFL_FOLD_K=1
run regs_k4 ROFL_FOLD_K=4
