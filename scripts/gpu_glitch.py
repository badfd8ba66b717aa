"""Latency glitches: distribution of create times of a small shape over many calls (env ROFL_HOST_THREADS selects the host pool size)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
R.set_device(0); api.set_fp(16, 7)
rng = np.random.default_rng(3)
d, nb = 5000, 8
mx = np.float32(((1 << (nb - 1)) - 1) / 128.0)
vals = rng.uniform(-mx, mx, d).astype(np.float32) * np.float32(0.99)
bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
ts = []
for i in range(203):
    t = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
    ts.append((time.perf_counter() - t) * 1e3)
ts = np.array(ts[3:])
print("host threads", os.environ.get("ROFL_HOST_THREADS", "default"), "median %.2f p90 %.2f p99 %.2f max %.2f ms; >1.3x median: %d of %d" % (
    np.median(ts), np.percentile(ts, 90), np.percentile(ts, 99), ts.max(), int((ts > 1.3 * np.median(ts)).sum()), ts.size))
