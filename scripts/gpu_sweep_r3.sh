#!/bin/bash
# host pool: threads x idle polling window (latency of sequential clients)
for P in 4 64; do
  for ht in 14 16; do for us in 150 400; do ROFL_HOST_THREADS=$ht ROFL_POOL_SPIN_US=$us python scripts/gpu_lat.py $P 20; done; done
done
