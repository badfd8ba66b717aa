"""Is k_msm_accumulate bound by its gathers or by the VALU?  Same launches, with the point gathers confined to a cache-resident
prefix of the table (ROFL_DBG_IDX_MASK; results are wrong, only the event times matter)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7); R.set_timing(True)
vals, bl = bench.synth_client(1)
for i in range(3):
    t = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
    dt = (time.perf_counter() - t) * 1e3
    tm = R.last_timing()
print("mask", os.environ.get("ROFL_DBG_IDX_MASK", "none"), "create %.2f ms" % dt, "accumulate %.3f ms in %d launches" % (tm["msm_accumulate_ms"], tm["msm_accumulate_launches"]), "fold %.3f" % tm["fold_ms"])
