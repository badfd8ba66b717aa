"""What would locality of the window-table gathers buy the fixed-base accumulation?  Timing experiment (WRONG results: ROFL_DBG_IDX_MASK confines
the gathers of k_msm_accumulate_fb to a prefix of the window table): average launch time of the kernel at BASELINE cfg 2 for prefixes of
64 MB ... the whole 1 GB table.  usage: ROFL_DBG_IDX_MASK=0x7ffff python scripts/gpu_acc_locality.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import rofl_project_code_amd as R
R.set_device(0); R.api.set_fp(32, 7)
rng = np.random.default_rng(3); d = int(os.environ.get("D", "25000"))
mx = np.float32(16777216.0)
vals = np.clip(rng.uniform(-mx, mx, d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 15
R.api.bp_gens_prepare(32, (1 << (d - 1).bit_length()) // 4)
R.set_timing(1)
acc = []
for it in range(6):
    R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(bytes([it]) * 32))
    k = R.last_kernel_times()["k_msm_accumulate_fb"]
    if it >= 2: acc.append(k["ms"] / max(k["launches"], 1))
mask = os.environ.get("ROFL_DBG_IDX_MASK", "none")
print("idx mask %s (%s MB of table): k_msm_accumulate_fb %.3f ms per launch" % (mask, (int(mask, 0) + 1) * 128 >> 20 if mask != "none" else "whole", sum(acc) / len(acc)))
