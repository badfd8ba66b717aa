#!/bin/bash
# keccak_f1600_zmm against the scalar permutation on the GPU box's host, then single-client latency with and without it (interleaved)
mkdir -p gpurun_out/r5o
python3 - <<'P' > gpurun_out/r5o/keccak.txt 2>&1
import ctypes
from rofl_project_code_amd import api
L = api.lib()
a = ctypes.c_double(); b = ctypes.c_double(); za = []; sb = []
for _ in range(40):
    rc = L.rofl_dbg_host_keccak_zmm_selftest(8, 2000, ctypes.byref(a), ctypes.byref(b)); za.append(a.value); sb.append(b.value)
print("rc", rc, "zmm min/med ns", min(za), sorted(za)[20], "scalar min/med ns", min(sb), sorted(sb)[20])
ns = ctypes.c_double(); r = []
for _ in range(10):
    L.rofl_dbg_host_bench(6, 8192, ctypes.byref(ns)); r.append(ns.value)
print("prefix ns/commitment min/med", min(r), sorted(r)[5])
P
grep -m1 "model name" /proc/cpuinfo >> gpurun_out/r5o/keccak.txt
cat gpurun_out/r5o/keccak.txt
for i in 1 2 3; do
  ROFL_KECCAK_ZMM=0 timeout 120 python scripts/gpu_lat.py 4 20
  ROFL_KECCAK_ZMM=1 timeout 120 python scripts/gpu_lat.py 4 20
  ROFL_MSM_FB_THREADS=262144 timeout 120 python scripts/gpu_lat.py 4 20
done 2>&1 | tee gpurun_out/r5o/lat_ab.txt
uptime
