#!/bin/bash
# VALU issue counters of the headline command (one client per step, BASELINE cfg 2), their own rocprofv3 --pmc passes (no tracing beside them).
# -> gpurun_out/prof_valu/<tag>_pmc_valu.json ; copied to profiles/ by the caller.  bench.py reads the newest profiles/*_pmc_valu.json.
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_valu
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="bench.py --steps 4 --warmup 2 --no-extras --hip-runtime process"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $OUT/p1 -- python3 $CMD > /dev/null 2> $OUT/p1.err
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/p2 -- python3 $CMD > /dev/null 2> $OUT/p2.err
python3 scripts/pmc_any_to_json.py $OUT/${TAG}_pmc_valu.json $OUT/p1 $OUT/p2
tail -3 $OUT/p1.err $OUT/p2.err
rm -rf $OUT/p1 $OUT/p2
cp $OUT/${TAG}_pmc_valu.json profiles/
