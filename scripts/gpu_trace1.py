"""One warm client (cfg 2) with the per-phase host trace (ROFL_TRACE=2) to see where a sequential proof spends its wall time."""
import os, sys, time
os.environ["ROFL_TRACE"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
vals, bl = bench.synth_client(1)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
api.bp_gens_prepare(32, max(R.range_proof_vec.next_pow2(bench.D) // P, 1))      # the full fold table (a warm client: not the compact table of a first call)
for i in range(3):
    sys.stderr.write("=== create %d\n" % i)
    t = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, P, nonce=R.Nonce.seeded(b"\x01" * 32))
    t1 = time.perf_counter()
    sys.stderr.write("=== verify %d (create %.2f ms)\n" % (i, (t1 - t) * 1e3))
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=b"\x02" * 32)
    sys.stderr.write("=== done (verify %.2f ms) ok=%s\n" % ((time.perf_counter() - t1) * 1e3, ok))
