"""Throughput modes on one GPU, cfg 2 clients (create + verify): C lanes in flight (one host thread per client) vs one batched call vs
T host threads with batched calls of C / T clients.  Usage: gpu_throughput.py MODE C [T]   MODE = lanes | batch | tbatch"""
import os, sys, time, resource
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, C = sys.argv[1], int(sys.argv[2]); T = int(sys.argv[3]) if len(sys.argv) > 3 else 2
os.environ.setdefault("ROFL_LANES", str(max(3, C if mode == "lanes" else T)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, C + 2)))
import numpy as np
import rofl_project_code_amd as R
import bench
R.set_device(0)
if os.environ.get("THR_D"): bench.D = int(os.environ["THR_D"])      # e.g. 55000: the cfg 4 client size
rpv = R.range_proof_vec; FP = bench.FP; D = bench.D
cl = [bench.synth_client(5000 + j) for j in range(C)]
nonces = [R.Nonce.seeded(bytes([j + 1]) * 32) for j in range(C)]
def one(j, tag):
    pr, cm = rpv.create_rangeproof(cl[j][0], cl[j][1], 32, 4, nonce=nonces[j], fp=FP)
    assert rpv.verify_rangeproof(pr, cm, 32, verifier_seed=bytes([tag % 256]) * 32, fp=FP)
def batch(js, tag):
    res = rpv.create_rangeproof_batch([cl[j][0] for j in js], [cl[j][1] for j in js], 32, 4, nonces=[nonces[j] for j in js], fp=FP)
    assert all(rpv.verify_rangeproof_batch([r[0] for r in res], [r[1] for r in res], 32, verifier_seed=bytes([tag % 256]) * 32, fp=FP))
ex = ThreadPoolExecutor(max_workers=max(C, T))
def step(tag):
    if mode == "lanes": list(ex.map(lambda j: one(j, tag), range(C)))
    elif mode == "batch": batch(list(range(C)), tag)
    else:
        per = C // T
        list(ex.map(lambda t: batch(list(range(t * per, (t + 1) * per)), tag), range(T)))
step(0); step(1)
ru0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter(); n = 6
for k in range(n): step(k + 2)
el = time.perf_counter() - t0; ru1 = resource.getrusage(resource.RUSAGE_SELF)
print("%-7s C=%d T=%d %s  %.0f el/s  %.1f ms/step  host cores busy %.2f" % (mode, C, T, " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("ROFL_") and k != "ROFL_LANES"),
      n * C * D / el, el / n * 1e3, ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / el))
