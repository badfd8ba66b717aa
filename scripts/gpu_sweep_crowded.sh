#!/bin/bash
# Throughput (6 clients in flight) against the fold schedule / MSM knobs: the latency-optimal settings need not be the throughput-optimal ones.
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 150 python bench.py --no-cpu-baseline --no-l2 --steps 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$*', round(d['value']), round(d['ms_per_step'],1), round(d['single_client']['ms_create_plus_verify'],2))"; }
run A=0
run ROFL_FOLD_T1=2
run ROFL_FOLD_T=3
run ROFL_FOLD_T=1
run ROFL_FOLD_MIN=4096
run ROFL_FOLD_MIN=256
run ROFL_FOLD_T1=4 ROFL_FOLD_T=3
run ROFL_MSM_FB_SETS=1
run ROFL_MSM_FB_SETS=4
run A=1
