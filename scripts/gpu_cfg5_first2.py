"""Time every step of EncParamsL2.encrypt for the first client after a verify phase (six clients: six encrypts, six verifies)."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
R.set_device(0); api.set_fp(32, 7)
d = 25000
ins = []
for c in range(6):
    rng = np.random.default_rng(77 + c)
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    ins.append((vals, r1, r2))
marks = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); marks.append((name, round((time.perf_counter() - t0) * 1e3, 2))); return r
    setattr(obj, name, g)
wrap(params, "witness_digest"); wrap(params, "_concurrently"); wrap(params.range_proof_vec, "clip_f32_to_range_vec")
wrap(params.conversion32, "f32_to_scalar_vec"); wrap(params.pedersen_ops, "commit_vec")
if os.environ.get("NOGC"): gc.disable()
keep = []
for s in range(7):
    outs = []
    if os.environ.get('KEEP'): keep.append(outs)
    for k, (vals, r1, r2) in enumerate(ins):
        marks.clear(); t0 = time.perf_counter()
        outs.append(params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2))
        tot = (time.perf_counter() - t0) * 1e3
        if s >= 4 and k < 2: print("sample %d client %d: %.1f ms" % (s, k, tot), marks, "gc", gc.get_count())
    for upd in outs:
        assert upd.verify(verifier_seed=b"\x04" * 32)
