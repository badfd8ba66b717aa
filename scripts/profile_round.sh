#!/bin/bash
# Run on the GPU box (via gpurun): bench line, rocprofv3 kernel stats and the two PMC passes of the same command.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="bench.py --steps 8 --warmup 3 --no-extras --hip-runtime process"      # the headline only: one client per step (BASELINE cfg 2)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $CMD > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $CMD > /dev/null 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $CMD > /dev/null 2> $OUT/pmc_write.err
python3 scripts/pmc_to_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
cp $OUT/${TAG}_pmc_traffic.json profiles/      # bench.py reads the newest traffic file
cp $OUT/${TAG}_bench_kernel_stats.csv $OUT/bench_under_rocprof.json profiles/ 2>/dev/null; mv profiles/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
timeout 600 bash scripts/gpu_kt1.sh 4 > /dev/null 2>&1; cp gpurun_out/kt1.txt profiles/${TAG}_single_client_kernel_trace.txt
timeout 900 python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err; cp $OUT/${TAG}_bench.json profiles/
tail -c 3000 $OUT/${TAG}_bench.json
