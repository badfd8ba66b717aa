#!/bin/bash
# kernel timeline of a cfg-4 round (48 clients, d = 55 000) with ONE and with THREE batched create calls in flight.
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/cfg4_tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mi in 1 3; do
  rm -rf $OUT/kt$mi
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt$mi -- python3 bench.py --config 4 --steps 1 --warmup 1 --multi-inflight $mi --no-cpu-baseline --hip-runtime process > $OUT/bench_mi$mi.json 2> $OUT/kt$mi.err
  python3 scripts/kt_overlap.py $OUT/kt$mi 0.45 > $OUT/${TAG}_cfg4_timeline_inflight$mi.txt 2>&1
  rm -rf $OUT/kt$mi
done
for mi in 1 3; do
  timeout 600 python3 bench.py --config 4 --steps 3 --warmup 1 --multi-inflight $mi --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('inflight $mi:', round(j['value']), j.get('breakdown_ms_per_step_rank0'))" >> $OUT/${TAG}_cfg4_inflight_ab.txt
done
cat $OUT/${TAG}_cfg4_timeline_inflight1.txt $OUT/${TAG}_cfg4_timeline_inflight3.txt $OUT/${TAG}_cfg4_inflight_ab.txt
