#!/bin/bash
# The library's single-client latency as a compiled C99 host sees it (integration/c_host/fl_round.c bench: system HIP runtime, no interpreter),
# alternating with the Python latency script on the same box -> gpurun_out/c_host_bench.json
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_c_host.py -q -m "not gpu" >/dev/null 2>&1      # builds integration/c_host/build/fl_round
B=integration/c_host/build/fl_round
{
for rep in 1 2 3; do
  $B bench 25000 32 4 40
  echo "{\"python_gpu_lat_P4\": \"$(python scripts/gpu_lat.py 4 12 | sed 's/.*create/create/')\"}"
done
$B bench 25000 32 64 40
$B bench 55000 32 4 20
$B bench 5000 8 4 40
} | tee gpurun_out/c_host_bench.json
uptime
