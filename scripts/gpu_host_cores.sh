#!/bin/bash
# what one rank does on the host budget it gets when eight ranks share a node -- bench.py --host-cores K pins the
# process to K cores before any GPU call; cfg 2 at n_partition 4 and 64; ms per step and host cores busy.  -> gpurun_out/host_cores.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "[" > gpurun_out/host_cores.json
first=1
for P in 4 64; do
  for K in 2 4 8 16; do
    line=$(python bench.py --host-cores $K --n-partition $P --steps ${STEPS:-12} --warmup 4 --no-extras 2>/dev/null | tail -1)
    [ $first = 1 ] || echo "," >> gpurun_out/host_cores.json
    first=0
    python3 - "$line" $K $P >> gpurun_out/host_cores.json <<'PY'
import json, sys
j = json.loads(sys.argv[1]); c = j["config"]
print(json.dumps({"host_cores_pinned": int(sys.argv[2]), "n_partition": int(sys.argv[3]), "ms_per_step": round(j["ms_per_step"], 2), "median_ms_per_step": round(j["median_ms_per_step"], 2),
                  "elements_per_s": round(j["value"]), "host_cores_busy": c["host_cores_busy"], "wait_policy": c["wait_policy"], "create_ms": round(j["breakdown_ms_per_client"]["create"], 2), "verify_ms": round(j["breakdown_ms_per_client"]["verify"], 2)}))
PY
  done
done
echo "]" >> gpurun_out/host_cores.json
cat gpurun_out/host_cores.json
