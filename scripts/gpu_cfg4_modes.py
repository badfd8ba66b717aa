"""cfg 4 shape (d = 55 000, 32-bit, P = 4), one GPU's share of clients: batched calls against C clients in flight on C lanes."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 6
NCL = int(sys.argv[2]) if len(sys.argv) > 2 else 12
os.environ.setdefault("ROFL_LANES", str(max(3, C)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, C + 2)))
import numpy as np
import rofl_project_code_amd as R
import bench
R.set_device(0)
rpv = R.range_proof_vec; FP = bench.FP
cl = [bench.synth_multi(4, c, 0)[:2] for c in range(NCL)]
nonces = [R.Nonce.seeded(bytes([j + 1]) * 32) for j in range(NCL)]
def batch_mode():
    res = []
    for g0 in range(0, NCL, 6):
        g = list(range(g0, min(g0 + 6, NCL)))
        res += rpv.create_rangeproof_batch([cl[k][0] for k in g], [cl[k][1] for k in g], 32, 4, nonces=[nonces[k] for k in g], fp=FP)
    return res
ex = ThreadPoolExecutor(max_workers=C)
def lanes_mode():
    return list(ex.map(lambda k: rpv.create_rangeproof(cl[k][0], cl[k][1], 32, 4, nonce=nonces[k], fp=FP), range(NCL)))
for name, f in (("batch", batch_mode), ("lanes", lanes_mode), ("batch", batch_mode), ("lanes", lanes_mode)):
    f()
    t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    print("%s C=%d: %.1f ms per client (create only), %.0f el/s" % (name, C, dt / NCL * 1e3, NCL * 55000 / dt), flush=True)
