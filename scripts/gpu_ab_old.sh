#!/bin/bash
# same-box A/B of the single-client latency: the library built from an older tree (build_old/) against the current one, interleaved
mkdir -p gpurun_out/r5r
OLD=$PWD/build_old/rofl_project_code_amd/librofl_zk.so
for i in 1 2 3 4; do
  echo -n "old "; ROFL_ZK_LIB=$OLD timeout 120 python scripts/gpu_lat.py 4 24
  echo -n "new "; timeout 120 python scripts/gpu_lat.py 4 24
done 2>&1 | tee gpurun_out/r5r/ab.txt
uptime; grep -m1 "model name" /proc/cpuinfo
