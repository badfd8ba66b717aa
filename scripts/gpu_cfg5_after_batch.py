"""Does a batched cfg-4 call earlier in the process change the L2 composite's latency afterwards?  (leak check for the in-flight counters)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
import bench
R.set_device(0); api.set_fp(32, 7)
d = 55000
def composite(tag, reps=8):
    rng = np.random.default_rng(77)
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    tc = []
    for i in range(reps + 3):
        t0 = time.perf_counter()
        upd = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2)
        t1 = time.perf_counter()
        assert upd.verify(verifier_seed=b"\x04" * 32)
        if i >= 3: tc.append((t1 - t0) * 1e3)
    tc.sort(); print(tag, "create med %.2f min %.2f max %.2f" % (tc[len(tc) // 2], tc[0], tc[-1]), flush=True)
composite("fresh process        ")
rpv = R.range_proof_vec
ins = [bench.synth_multi(4, c, 0) for c in range(6)]
res = rpv.create_rangeproof_batch([x[0] for x in ins], [x[1] for x in ins], 32, 4, nonces=[R.Nonce.seeded(bytes([c + 1]) * 32) for c in range(6)], fp=(32, 7))
oks = rpv.verify_rangeproof_batch([r[0] for r in res], [r[1] for r in res], 32, verifier_seed=b"\x02" * 32, fp=(32, 7))
assert all(oks)
composite("after a cfg-4 batch  ")
pr, cm = rpv.create_rangeproof(ins[0][0], ins[0][1], 32, 64, nonce=R.Nonce.seeded(b"\x09" * 32), fp=(32, 7))
assert rpv.verify_rangeproof(pr, cm, 32, verifier_seed=b"\x02" * 32, fp=(32, 7))
composite("after a P=64 client  ")
