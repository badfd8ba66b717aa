#!/bin/bash
# Which launches of the shipped configurations overflow a fixed-size structure and get repeated on the next variant?  (ROFL_TRACE=1 prints
# one line per MSM with its overflow flag.)  Shapes: cfg 1, cfg 2 at P = 4 / 64, cfg 4 at P = 4 / 64 (C host, 12 clients each), then the batched
# calls of bench.py --config 4 and the L2 composite of --config 5 (two rounds each).
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_c_host.py -q -m "not gpu" >/dev/null 2>&1
B=integration/c_host/build/fl_round
audit() { echo "== $1"; grep "^\[rofl\] msm" | sed 's/overflow=\([0-9]*\)/overflow=\1/' | awk '{k=$3" "$4" "$5" "$6" "$7" "$8; tot[k]++; if ($9!="overflow=0") bad[k]++} END {for (k in tot) printf "  %-60s launches %5d  overflowed %d\n", k, tot[k], bad[k]+0}' | sort; }
ROFL_TRACE=1 $B bench 5000 8 4 12 2>&1 >/dev/null | audit "cfg1 d=5000 8-bit P=4"
ROFL_TRACE=1 $B bench 25000 32 4 12 2>&1 >/dev/null | audit "cfg2 d=25000 P=4"
ROFL_TRACE=1 $B bench 25000 32 64 12 2>&1 >/dev/null | audit "cfg2 P=64"
ROFL_TRACE=1 $B bench 55000 32 4 12 2>&1 >/dev/null | audit "cfg4 d=55000 P=4"
ROFL_TRACE=1 $B bench 55000 32 64 12 2>&1 >/dev/null | audit "cfg4 P=64"
ROFL_TRACE=1 python bench.py --config 4 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 >/dev/null | audit "bench --config 4 (48 clients, batched create, one check)"
ROFL_TRACE=1 python bench.py --config 5 --steps 2 --warmup 1 --no-extras --no-cpu-baseline 2>&1 >/dev/null | audit "bench --config 5 (L2 composite)"
