"""The first 30 clients of a process, one line each: step ms, create (host share), verify (host share) -- is there a slow step ~150 ms into the load, and whose time is it?"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
cl = [bench.synth_client(1000 * i) for i in range(30)]
api.bp_gens_prepare(32, 8192)
R.set_timing(2)
gc.collect(); gc.disable()
T0 = time.perf_counter(); out = []
for i in range(30):
    vals, bl = cl[i]
    t0 = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(bytes([i % 256]) * 32))
    t1 = time.perf_counter(); tc = R.last_timing()
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=bytes([i % 256]) * 32)
    t2 = time.perf_counter(); tv = R.last_timing()
    out.append("%2d t=%6.1f step %.1f = create %.1f (host %.2f acc_fb %.2f) + verify %.1f (host %.2f)" % (i, (t0 - T0) * 1e3, (t2 - t0) * 1e3, (t1 - t0) * 1e3, tc["host_ms"], tc["msm_accumulate_ms"], (t2 - t1) * 1e3, tv["host_ms"]))
print("\n".join(out))
