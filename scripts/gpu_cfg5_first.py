"""Which of the three proofs of the FIRST composite after a verify phase is slow?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
R.set_device(0); api.set_fp(32, 7)
d = 25000
rng = np.random.default_rng(77)
vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
orig = params._concurrently
log = []
T0 = [0.0]
def timed(*thunks):
    def wrap(k, t):
        def f():
            t0 = time.perf_counter(); r = t(); log.append((k, round((t0 - T0[0]) * 1e3, 1), (time.perf_counter() - t0) * 1e3)); return r
        return f
    T0[0] = time.perf_counter()
    r = orig(*[wrap(k, t) for k, t in enumerate(thunks)])
    log.append(("all", round((time.perf_counter() - T0[0]) * 1e3, 1), 0.0))
    return r
params._concurrently = timed
for s in range(6):
    outs = []
    for c in range(3):
        log.clear(); t0 = time.perf_counter()
        outs.append(params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2))
        print("sample %d encrypt %d: %.1f ms" % (s, c, (time.perf_counter() - t0) * 1e3), [(k, st, round(v, 1)) for k, st, v in log])
    for upd in outs:
        log.clear(); t0 = time.perf_counter()
        assert upd.verify(verifier_seed=b"\x04" * 32)
        print("sample %d verify: %.1f ms" % (s, (time.perf_counter() - t0) * 1e3), [(k, st, round(v, 1)) for k, st, v in log])
print("---- prelude pieces of the first encrypt after a verify phase")
from rofl_project_code_amd.api import range_proof_vec, pedersen_ops, conversion32
for s in range(4):
    upd = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2)
    assert upd.verify(verifier_seed=b"\x04" * 32)
    t0 = time.perf_counter(); wd = params.witness_digest(vals, r1, r2)
    t1 = time.perf_counter(); clipped = range_proof_vec.clip_f32_to_range_vec(vals, 8)
    t2 = time.perf_counter(); scal = conversion32.f32_to_scalar_vec(clipped)
    t3 = time.perf_counter(); com = pedersen_ops.commit_vec(scal, r1)
    t4 = time.perf_counter(); com = pedersen_ops.commit_vec(scal, r1)
    t5 = time.perf_counter()
    print("digest %.2f clip %.2f to-scalar %.2f commit_vec %.2f commit_vec again %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
