#!/bin/bash
# host share of the hops (ROFL_TRACE=2 "[rofl-hops]" lines of warm clients) for the new library and build/librofl_zk_prev.so
PREV=$PWD/rofl_project_code_amd/build/librofl_zk_prev.so
for P in 4 64; do for i in 1 2; do
  echo "new  P=$P $(ROFL_TRACE=2 python scripts/gpu_lat.py $P 4 2>&1 >/dev/null | grep rofl-hops | tail -3 | sed 's/.*msm calls://' | tr '\n' '|')"
  echo "prev P=$P $(ROFL_ZK_LIB=$PREV ROFL_TRACE=2 python scripts/gpu_lat.py $P 4 2>&1 >/dev/null | grep rofl-hops | tail -3 | sed 's/.*msm calls://' | tr '\n' '|')"
done; done
