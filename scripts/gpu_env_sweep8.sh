#!/bin/bash
for e in "$@"; do
  echo "== $e"
  env $e timeout 300 python scripts/gpu_big.py 25000 8 4 4 2>&1 | grep "^rep" | tail -2
done
