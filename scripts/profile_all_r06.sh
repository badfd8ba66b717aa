#!/bin/bash
# Round 6, one gpurun call: every profile the docs cite, from the final tree.  Everything lands under gpurun_out/r6p/ (merged back by gpurun);
# the caller copies gpurun_out/r6p/profiles/* into profiles/.
cd $GRAFT_REPO_ROOT
export TAG=r06
P=gpurun_out/r6p
mkdir -p $P
bash scripts/profile_round.sh r06 > $P/profile_round.log 2>&1                 # bench under rocprofv3 --kernel-trace --stats, FETCH_SIZE / WRITE_SIZE passes, single-client kernel trace, the bench line
bash scripts/profile_valu.sh r06 > $P/profile_valu.log 2>&1                   # VALU issue counters (their own --pmc passes)
bash scripts/profile_cfg45.sh r06 > $P/profile_cfg45.log 2>&1                 # cfg 4 / cfg 5 rounds: kernel stats + bench lines (+ P = 64, + one process on two logical devices)
bash scripts/profile_cfg4_timeline.sh r06 > $P/profile_cfg4_timeline.log 2>&1  # who runs beside whom in a cfg-4 round, one and three calls in flight
cp gpurun_out/cfg4_tl/r06_cfg4_*.txt profiles/ 2>/dev/null
timeout 900 python scripts/gpu_configs.py > $P/configs.log 2>&1; cp gpurun_out/r06_configs.json profiles/ 2>/dev/null
STEPS=12 timeout 1200 bash scripts/gpu_host_cores.sh > $P/host_cores.log 2>&1; cp gpurun_out/host_cores.json profiles/r06_host_cores.json
timeout 600 bash scripts/gpu_c_host_bench.sh > $P/c_host.log 2>&1; cp gpurun_out/c_host_bench.json profiles/r06_c_host_bench.json
# ONE client over several devices / ranks on the one GPU of the box: rehearsals of the control flow, NOT scaling numbers
timeout 600 python bench.py --one-process --gpus 2 --steps 8 --warmup 3 --compact-tables 2>$P/op2.err | tail -1 > profiles/r06_rehearsal_split_one_process_2dev.json
timeout 600 python bench.py --one-process --gpus 4 --n-partition 64 --steps 8 --warmup 3 --compact-tables 2>$P/op4.err | tail -1 > profiles/r06_rehearsal_split_one_process_4dev_p64.json
ROFL_BENCH_BACKEND=gloo ROFL_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 2 --split-chunks --steps 8 --warmup 3 --compact-tables 2>$P/ranks2.err | tail -1 > profiles/r06_rehearsal_split_ranks2.json
ROFL_BENCH_BACKEND=gloo ROFL_BENCH_SAME_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 8 --warmup 3 --compact-tables 2>$P/ranks2w.err | tail -1 > profiles/r06_rehearsal_ranks2_weak_with_split_extra.json
( cd scripts && ./ubench_fe9 ) > profiles/r06_fe9.txt 2>&1
mkdir -p $P/profiles; cp profiles/r06_* $P/profiles/ 2>/dev/null
ls $P $P/profiles
