#!/bin/bash
# Round 5, one gpurun call: every profile the docs cite, from the final tree.  Outputs land in profiles/ (copied from gpurun_out/ by the caller).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
bash scripts/profile_round.sh r05 > gpurun_out/r5p/profile_round.log 2>&1
bash scripts/profile_cfg45.sh r05 > gpurun_out/r5p/profile_cfg45.log 2>&1
timeout 900 python scripts/gpu_configs.py > gpurun_out/r5p/configs.log 2>&1
STEPS=12 timeout 1200 bash scripts/gpu_host_cores.sh > gpurun_out/r5p/host_cores.log 2>&1; cp gpurun_out/host_cores.json gpurun_out/r5p/r05_host_cores.json
timeout 600 bash scripts/gpu_c_host_bench.sh > gpurun_out/r5p/c_host.log 2>&1; cp gpurun_out/c_host_bench.json gpurun_out/r5p/r05_c_host_bench.json
timeout 200 python scripts/gpu_cfg5_verify.py > gpurun_out/r5p/r05_cfg5_verify_legs.json 2>/dev/null
timeout 200 ./scripts/stage_bw 16 > gpurun_out/r5p/r05_stage_bw.txt 2>&1
mkdir -p gpurun_out/r5p/profiles; cp profiles/r05_* gpurun_out/r5p/profiles/ 2>/dev/null; cp gpurun_out/r05_configs.json gpurun_out/r5p/profiles/ 2>/dev/null
ls gpurun_out/r5p gpurun_out/r5p/profiles
