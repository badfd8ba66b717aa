"""Latency of sequential warm clients (cfg 2 shape): gpu_lat.py P [reps] -> median / min / max of create and verify (ms).  Knobs come from the environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
api.bp_gens_prepare(32, R.range_proof_vec.next_pow2(int(os.environ.get("LAT_D", "25000"))) // min(P, R.range_proof_vec.next_pow2(int(os.environ.get("LAT_D", "25000")))))      # complete tables before timing
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
d = int(os.environ.get("LAT_D", "25000"))
vals, bl = bench.synth_client(1)
vals, bl = vals[:d], bl[:d]
tc, tv = [], []
for i in range(reps + 3):
    t = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, P, nonce=R.Nonce.seeded(bytes([i + 1]) * 32))
    t1 = time.perf_counter()
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=b"\x02" * 32)
    t2 = time.perf_counter()
    assert ok
    if i >= 3:
        tc.append((t1 - t) * 1e3); tv.append((t2 - t1) * 1e3)
tc.sort(); tv.sort()
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("ROFL_"))
print("P=%d d=%d %-40s create med %.2f min %.2f max %.2f | verify med %.2f min %.2f max %.2f | sum med %.2f" % (
    P, d, tag, tc[len(tc) // 2], tc[0], tc[-1], tv[len(tv) // 2], tv[0], tv[-1], tc[len(tc) // 2] + tv[len(tv) // 2]), flush=True)
