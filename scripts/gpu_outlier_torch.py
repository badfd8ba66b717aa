"""Is the one ~31 ms step per bench.py run tied to torch?  Sequential cfg-2 clients with wall-clock stamps since process start; MODE:
none = no torch; import = torch imported only; cuda = torch.cuda initialised (set_device + synchronize), as bench.py does."""
import os, sys, time
T0 = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
if mode != "none":
    import torch
    if mode == "cuda":
        torch.cuda.set_device(0); torch.cuda.synchronize()
import numpy as np, gc
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.bp_gens_prepare(32, 8192)
vals, bl = bench.synth_client(1)
gc.collect(); gc.disable()
rows = []
for i in range(60):
    t = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(bytes([i + 1]) * 32), fp=(32, 7))
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=b"\x02" * 32, fp=(32, 7))
    rows.append((round(t - T0, 2), round((time.perf_counter() - t) * 1e3, 1)))
slow = [r for r in rows[3:] if r[1] > 26.5]
ms = sorted(r[1] for r in rows[3:])
hip = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
print(hip)
print(mode, "median %.1f" % ms[len(ms) // 2], "slow steps (t since start, ms):", slow, "first create at t=%.2f" % rows[0][0])
