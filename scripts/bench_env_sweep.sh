#!/bin/bash
# usage: bench_env_sweep.sh "A=1 B=2" ... -> headline value of bench.py (3 clients in flight) for each env setting
for e in "$@"; do
  echo "== $e"
  env $e timeout 600 python bench.py --no-cpu-baseline --no-l2 --steps ${STEPS:-6} --warmup 2 2>/dev/null | tail -1 | python3 -c "import sys,json; b=json.loads(sys.stdin.read()); print(round(b['value']), round(b['ms_per_step'],2), round(b['single_client']['ms_create_plus_verify'],2), 'cpu', b['config']['host_cores_busy'])"
done
