"""Latency of a client after an idle gap: sequential warm clients (cfg 2) with a pause of G ms before each; host share from the library's timing."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
cl = [bench.synth_client(1000 * i) for i in range(4)]
R.set_timing(2)
gc.collect(); gc.disable()
def one(i):
    vals, bl = cl[i % 4]
    t0 = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(bytes([i % 256]) * 32))
    t1 = time.perf_counter(); tc = R.last_timing()
    assert R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=bytes([i % 256]) * 32)
    t2 = time.perf_counter()
    return (t2 - t0) * 1e3, tc["host_ms"], tc["msm_accumulate_ms"]
for i in range(6): one(i)
for gap in (0, 2, 5, 10, 20, 50, 200):
    rows = []
    for i in range(8):
        if gap: time.sleep(gap / 1e3)
        rows.append(one(i))
    a = np.array(rows)
    print("gap %3d ms: step median %.2f max %.2f | host share of create %.2f | fixed-base accumulation %.2f ms" % (gap, np.median(a[:, 0]), a[:, 0].max(), np.median(a[:, 1]), np.median(a[:, 2])))
