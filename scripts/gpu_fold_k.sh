#!/bin/bash
# single-client latency (cfg 2 shape) against the number of bit segments per folded output (K threads share one output: redundant doublings for parallelism)
for P in 4 64; do
  for kv in "ROFL_FOLD_K=0" "ROFL_FOLD_K=1" "ROFL_FOLD_K=2" "ROFL_FOLD_K=4" "ROFL_FOLD_THREADS=65536" "ROFL_FOLD_THREADS=262144" "ROFL_FOLD_K=0"; do
    echo "P=$P $kv: $(env $kv python scripts/gpu_lat.py $P 10 | sed 's/.*create/create/')"
  done
done
