#!/bin/bash
# knob sweep (latency of sequential clients) at n_partition = 64
python scripts/gpu_lat.py 64 16
ROFL_MSM_T10=8192 python scripts/gpu_lat.py 64 16
ROFL_MSM_T10=2048 python scripts/gpu_lat.py 64 16
ROFL_MSM_DEV_HORNER_MIN=1000 python scripts/gpu_lat.py 64 16
ROFL_FOLD_MIN=4096 python scripts/gpu_lat.py 64 16
ROFL_FOLD_T=3 python scripts/gpu_lat.py 64 16
ROFL_FOLD_T1=4 python scripts/gpu_lat.py 64 16
ROFL_FOLD_T1=2 python scripts/gpu_lat.py 64 16
ROFL_POOL_SPIN_US=0 python scripts/gpu_lat.py 64 16
python scripts/gpu_lat.py 4 16
ROFL_POOL_SPIN_US=0 python scripts/gpu_lat.py 4 16
