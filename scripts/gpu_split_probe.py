"""Does splitting ONE client's chunks into groups that run on separate lanes shorten the create?  cfg 2 (d = 25 000, P = 4) as one call
against the same 4 chunks as 2 concurrent calls of 2 chunks / 4 concurrent calls of 1 chunk (same (n, m) tables)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ROFL_LANES", "4")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import rofl_project_code_amd as R
import bench
R.set_device(0)
rpv = R.range_proof_vec; FP = bench.FP
vals, bl = bench.synth_client(3)
vals = np.concatenate([vals, vals[:7768]]); bl = np.concatenate([bl, bl[:7768]])      # 32 768 values: four full chunks
ex = ThreadPoolExecutor(max_workers=4)


PT = int(sys.argv[1]) if len(sys.argv) > 1 else 4      # n_partition of the whole client


def run(groups):
    per = 32768 // groups
    def one(g):
        return rpv.create_rangeproof(vals[g * per:(g + 1) * per], bl[g * per:(g + 1) * per], 32, PT // groups, nonce=R.Nonce.seeded(bytes([g + 1]) * 32), fp=FP)
    ts = []
    for rep in range(8):
        t0 = time.perf_counter()
        if groups == 1:
            one(0)
        else:
            list(ex.map(one, range(groups)))
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = sorted(ts[2:])
    print("groups=%d  create median %.2f ms  min %.2f  max %.2f" % (groups, ts[len(ts) // 2], ts[0], ts[-1]), flush=True)


for g in (1, 2, 4, 1, 2, 4):
    run(g)
