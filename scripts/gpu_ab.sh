#!/bin/bash
# same-box A/B of the freshly built library against rofl_project_code_amd/build/librofl_zk_prev.so (ROFL_ZK_LIB): core parity tests on the new one,
# then alternating sequential-client latencies (cfg 2 shape) at P = 4 and P = 64
mkdir -p gpurun_out
uptime | sed "s/.*load/load/"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact_vs_oracle or golden or format_and_identity or batch_verify or full_size_properties_cfg2 or extreme or l2_path" > gpurun_out/ab_tests.log 2>&1
tail -2 gpurun_out/ab_tests.log
PREV=$PWD/rofl_project_code_amd/build/librofl_zk_prev.so
for i in 1 2 3; do
  python scripts/gpu_lat.py 4 12 | sed 's/^/new  /'
  ROFL_ZK_LIB=$PREV python scripts/gpu_lat.py 4 12 | sed 's/^/prev /'
done
python scripts/gpu_lat.py 64 8 | sed 's/^/new  /'
ROFL_ZK_LIB=$PREV python scripts/gpu_lat.py 64 8 | sed 's/^/prev /'
