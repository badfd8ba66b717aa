#!/bin/bash
# kernel traces of one warm client at n_partition = 64, window tables at c = 13 and c = 16 -> gpurun_out/kt_p64_c{13,16}.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in 13 16; do
  export ROFL_MSM_FB_C=$c
  bash scripts/gpu_kt1.sh 64 > /dev/null 2>&1
  cp gpurun_out/kt1.txt gpurun_out/kt_p64_c$c.txt
done
tail -30 gpurun_out/kt_p64_c13.txt
