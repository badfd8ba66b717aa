#!/bin/bash
# quick GPU check used while iterating: core parity tests, a per-phase host trace of one warm client (P = 4 and P = 64), host microbench on the box's CPU
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bit_exact_vs_oracle or golden or format_and_identity or batch_verify or full_size_properties_cfg2 or extreme or l2_path" > gpurun_out/r3_tests.log 2>&1
tail -3 gpurun_out/r3_tests.log
python scripts/gpu_trace1.py 4 > gpurun_out/trace_p4.log 2>&1
python scripts/gpu_trace1.py 64 > gpurun_out/trace_p64.log 2>&1
grep "=== " gpurun_out/trace_p4.log | tail -4
grep "=== " gpurun_out/trace_p64.log | tail -4
python - <<'PY' > gpurun_out/hostbench.log 2>&1
import ctypes,sys
sys.path.insert(0,'.')
from rofl_project_code_amd import api
L=api.lib(); ns=ctypes.c_double(); out=[]
for what,it,name in ((0,20000,'keccak-f'),(1,50000,'gdouble'),(2,50000,'gadd'),(3,2000,'encode+add'),(4,2000,'fixed mul'),(5,500,'sc invert'),(6,8192,'V append')):
    best=1e9
    for r in range(3):
        L.rofl_dbg_host_bench(what, it, ctypes.byref(ns)); best=min(best,ns.value)
    out.append('%s %.0f'%(name,best))
print(' | '.join(out))
PY
cat gpurun_out/hostbench.log
