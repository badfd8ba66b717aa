#!/bin/bash
# Run the headline N times with default settings and print per-run step time and per-kernel-kind times (is the run-to-run spread in one kernel kind?)
cd $GRAFT_REPO_ROOT
n=${1:-8}
for i in $(seq 1 $n); do
  timeout 300 python bench.py --no-extras --steps 30 --warmup 6 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); k={t['kernel'].split(' ')[0][:22]: t.get('ms_per_client') for t in j['kernels']['top']}
b=j.get('breakdown_ms_per_client') or {}
print('run $i: median %.2f min %.2f | create %.2f verify %.2f host %.2f | ' % (j['median_ms_per_step'], j['min_ms_per_step'], b.get('create',0), b.get('verify',0), b.get('host_instrumented',0)) + ' '.join('%s %.2f' % (a, v) for a, v in k.items()))"
done
