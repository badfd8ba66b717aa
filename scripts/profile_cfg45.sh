#!/bin/bash
# rocprofv3 kernel stats of the multi-client modes (one GPU): bench.py --config 4 and --config 5, two timed rounds each
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cfg45
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats$c -- python3 bench.py --config $c --steps 2 --warmup 1 > $OUT/bench_cfg${c}_under_rocprof.json 2> $OUT/stats$c.err
  cp $(find $OUT/stats$c -name "*kernel_stats.csv" | head -1) $OUT/r03_cfg${c}_kernel_stats.csv
  rm -rf $OUT/stats$c
done
head -6 $OUT/r03_cfg5_kernel_stats.csv | cut -c1-90
