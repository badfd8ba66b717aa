#!/bin/bash
# knob sweep (latency of sequential clients): how long idle pool workers poll
for P in 4 64; do
  for us in 150 400 1000 3000; do ROFL_POOL_SPIN_US=$us python scripts/gpu_lat.py $P 16; done
done
