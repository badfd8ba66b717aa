#!/bin/bash
# host timeline of a single-client verification (ROFL_TRACE=2) + the transcript-prefix micro-benchmark + latency
mkdir -p gpurun_out/r5p
python3 - <<'P' > gpurun_out/r5p/prefix.txt 2>&1
import ctypes
from rofl_project_code_amd import api
L = api.lib()
ns = ctypes.c_double(); r = []
for _ in range(20):
    L.rofl_dbg_host_bench(6, 8192, ctypes.byref(ns)); r.append(ns.value)
print("prefix ns/commitment min/med", min(r), sorted(r)[10])
P
cat gpurun_out/r5p/prefix.txt
ROFL_TRACE=2 timeout 120 python scripts/gpu_lat.py 4 2 > gpurun_out/r5p/trace.txt 2>&1
tail -60 gpurun_out/r5p/trace.txt
for i in 1 2 3; do timeout 120 python scripts/gpu_lat.py 4 20; done 2>&1 | tee gpurun_out/r5p/lat.txt
