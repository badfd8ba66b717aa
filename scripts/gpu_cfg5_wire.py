"""cfg 5 per-client costs around the proofs: serialize / deserialize of the EncParamsL2 wire message (d = 55 000)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
import bench
R.set_device(0); api.set_fp(32, 7)
x = bench.synth_multi(5, 0, 0)
te, ts, td, tv, tb = [], [], [], [], []
for i in range(8):
    t0 = time.perf_counter(); upd = params.EncParamsL2.encrypt(x[0], x[1], 8, 4, 32, nonce_seed=bytes([i + 1]) * 32, rand_scalars=x[2], fp=(32, 7))
    t1 = time.perf_counter(); blob = upd.serialize()
    t2 = time.perf_counter(); arr = np.frombuffer(blob, dtype=np.uint8)
    t3 = time.perf_counter(); upd2 = params.EncParamsL2.deserialize(bytes(arr))
    t4 = time.perf_counter(); ok = upd2.verify(verifier_seed=b"\x05" * 32, fp=(32, 7))
    t5 = time.perf_counter(); assert ok
    if i >= 2: te.append(t1 - t0); ts.append(t2 - t1); tb.append(t3 - t2); td.append(t4 - t3); tv.append(t5 - t4)
m = lambda v: float(np.median(v)) * 1e3
print("encrypt %.2f | serialize %.2f (%d bytes) | frombuffer %.3f | bytes()+deserialize %.2f | verify %.2f ms" % (m(te), m(ts), len(blob), m(tb), m(td), m(tv)))
