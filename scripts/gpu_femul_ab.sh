#!/bin/bash
# field-arithmetic microbenchmarks for the new library and for build/librofl_zk_prev.so
PREV=$PWD/rofl_project_code_amd/build/librofl_zk_prev.so
for lib in new prev; do
  for cfg in "0 -" "1 -" "2 -" "3 16384" "3 2097152" "3 8388608"; do
    set -- $cfg
    if [ $lib = prev ]; then export ROFL_ZK_LIB=$PREV; else unset ROFL_ZK_LIB; fi
    echo -n "$lib "; ROFL_FEMUL_MODE=$1 ROFL_FEMUL_TABLE=$2 python scripts/gpu_femul.py
  done
done
