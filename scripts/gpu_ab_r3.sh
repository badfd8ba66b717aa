#!/bin/bash
# Same-box A/B of the round-3 library (build/librofl_zk_r3.so, built from the round-3 tree and kept for this) against the shipped one:
# scripts/gpu_lat.py (sequential warm clients, cfg 2 shape), alternating three times at P = 4, twice at P = 64 -> gpurun_out/ab_r3.txt
cd $GRAFT_REPO_ROOT
R3=rofl_project_code_amd/build/librofl_zk_r3.so
{
for rep in 1 2 3; do
  echo "round 3: $(ROFL_ZK_LIB=$R3 python scripts/gpu_lat.py 4 12 | sed 's/.*create/create/')"
  echo "round 4: $(python scripts/gpu_lat.py 4 12 | sed 's/.*create/create/')"
done
for rep in 1 2; do
  echo "round 3, P = 64: $(ROFL_ZK_LIB=$R3 python scripts/gpu_lat.py 64 12 | sed 's/.*create/create/')"
  echo "round 4, P = 64: $(python scripts/gpu_lat.py 64 12 | sed 's/.*create/create/')"
done
uptime
} | tee gpurun_out/ab_r3.txt
