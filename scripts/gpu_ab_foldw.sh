#!/bin/bash
# same-box A/B of the fold table width through bench.py itself (interleaved): ROFL_FOLD_W=8 (512 slices, 25.6 GB at cfg 2) against 9 (1 024 slices, 51.2 GB)
mkdir -p gpurun_out/r5u
for i in 1 2 3; do
  for w in ${WIDTHS:-8 9}; do
    ROFL_FOLD_W=$w timeout 300 python3 bench.py --no-extras 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('W=$w', round(j['ms_per_step'],2), round(j['median_ms_per_step'],2), {k:round(v,2) for k,v in j['breakdown_ms_per_client'].items()}, 'build_ms', round(j['cold']['gens_tables_build_ms']))"
  done
done 2>&1 | tee gpurun_out/r5u/ab_foldw.txt
uptime
