#!/bin/bash
# single-client latency (cfg 2 shape) against the fold schedule knobs, P = 4 and P = 64
for P in 4 64; do
  for kv in "ROFL_FOLD_T1=3" "ROFL_FOLD_T1=2" "ROFL_FOLD_T1=4" "ROFL_FOLD_T=1" "ROFL_FOLD_T=3" "ROFL_FOLD_T1=2 ROFL_FOLD_T=3" "ROFL_FOLD_MIN=256" "ROFL_FOLD_MIN=4096" "ROFL_MSM_FB_THREADS=1048576" "ROFL_MSM_FB_THREADS=262144"; do
    echo "P=$P $kv: $(env $kv python scripts/gpu_lat.py $P 10 | sed 's/.*create/create/')"
  done
done
