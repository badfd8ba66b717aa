"""cfg 5's server leg under the verifier seeds bench.py uses for its steps (bytes([s]) * 32, s = 0..15): EncParamsL2.verify_batch ms per seed, twice.
A seed-dependent outlier would be a weight that lands badly in the batched checks."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("ROFL_LANES", "12"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
import rofl_project_code_amd as R
from rofl_project_code_amd import params, api
FP = (32, 7); D = 55000; NC = 48; P = 4
R.set_device(0)
api.bp_gens_prepare(8, 65536 // P); api.bp_gens_prepare(32, 1)
ups = []
for c in range(12):
    rng = np.random.default_rng(1000 * c)
    x = (rng.integers(-3, 4, size=D) / 128.0).astype(np.float32)
    bl = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    ups.append(params.EncParamsL2.encrypt(x, bl, 8, P, 32, nonce_seed=bytes([c + 1]) * 32, rand_scalars=r2, fp=FP))
blobs = [ups[c % len(ups)].serialize(as_array=True).copy() for c in range(NC)]
R.set_option("verify_batch", 2)
U = [params.EncParamsL2.deserialize(b, copy=False) for b in blobs]
params.EncParamsL2.verify_batch(U, verifier_seed=b"\x07" * 32, fp=FP)
for rep in range(2):
    row = []
    for s in range(16):
        seed = bytes([s]) * 32
        t = time.perf_counter(); ok = params.EncParamsL2.verify_batch(U, verifier_seed=seed, fp=FP); dt = (time.perf_counter() - t) * 1e3
        assert all(ok)
        row.append(round(dt, 1))
    print("rep", rep, row, "msm retries", api.msm_retries(), flush=True)
