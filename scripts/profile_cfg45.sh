#!/bin/bash
# The multi-client modes on one GPU: rocprofv3 kernel stats of bench.py --config 4 and --config 5 (two timed rounds each), then the bench
# lines themselves (cfg 4 at n_partition 4 and 64, cfg 5, cfg 4 as one host process driving two logical devices) -> profiles/<tag>_*
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg45
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 4 5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$c -- python3 bench.py --config $c --steps 2 --warmup 1 --no-extras --hip-runtime process > $OUT/bench_cfg${c}_under_rocprof.json 2> $OUT/stats$c.err
  cp $(find $OUT/stats$c -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg${c}_kernel_stats.csv
  rm -rf $OUT/stats$c
done
timeout 900 python3 bench.py --config 4 --steps 8 --warmup 3 2>$OUT/b4.err | tail -1 > $OUT/${TAG}_bench_cfg4.json
timeout 900 python3 bench.py --config 4 --n-partition 64 --steps 8 --warmup 3 2>$OUT/b4p64.err | tail -1 > $OUT/${TAG}_bench_cfg4_p64.json
timeout 900 python3 bench.py --config 5 --steps 8 --warmup 3 2>$OUT/b5.err | tail -1 > $OUT/${TAG}_bench_cfg5.json
timeout 900 python3 bench.py --config 4 --one-process --gpus 2 --steps 8 --warmup 3 2>$OUT/b4op.err | tail -1 > $OUT/${TAG}_bench_cfg4_one_process_2dev.json
cp $OUT/${TAG}_*.json $OUT/${TAG}_*.csv profiles/ 2>/dev/null
for f in $OUT/${TAG}_bench_*.json; do python3 -c "
import json,sys
j=json.load(open('$f')); print('$f'.split('/')[-1], round(j['value']), j.get('breakdown_ms_per_step_rank0') or j.get('breakdown_ms_per_step'))"; done
