#!/bin/bash
# usage: gpu_env_sweep.sh "A=1 B=2" "C=3" ...   -> last-rep create/verify for each env setting
for e in "$@"; do
  echo "== $e"
  env $e timeout 300 python scripts/gpu_big.py 25000 32 4 4 2>&1 | grep "^rep" | tail -2
done
