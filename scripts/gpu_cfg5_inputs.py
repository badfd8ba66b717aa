"""L2 composite (d = 55 000) create latency: the same client over and over against six different clients in turn (what gpu_configs.py and bench.py --config 5 do)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
R.set_device(0); api.set_fp(32, 7)
d = 55000
ins = []
for c in range(6):
    rng = np.random.default_rng(77 + c)
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    ins.append((vals, r1, r2))
def run(tag, pick, keep):
    tc = []; outs = []
    for i in range(27):
        vals, r1, r2 = ins[pick(i)]
        t0 = time.perf_counter()
        upd = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2)
        t1 = time.perf_counter()
        if keep: outs.append(upd); outs = outs[-6:]
        if i >= 3: tc.append((t1 - t0) * 1e3)
    tc.sort(); print(tag, "create med %.2f min %.2f p90 %.2f max %.2f" % (tc[len(tc) // 2], tc[0], tc[int(len(tc) * 0.9)], tc[-1]), flush=True)
print("load", open("/proc/loadavg").read().split()[:3])
run("same client, result dropped     ", lambda i: 0, False)
run("six clients in turn, dropped    ", lambda i: i % 6, False)
run("six clients in turn, six kept   ", lambda i: i % 6, True)
run("same client, six kept           ", lambda i: 0, True)
for nb, m in ((32, 8192), (32, 512), (8, 8192), (32, 16384), (32, 1024)):
    api.bp_gens_prepare(nb, m)
    print("tables of", (nb, m), api.bp_gens_table_bytes(nb, m) >> 20, "MiB")
run("same client, after other tables ", lambda i: 0, False)
rpv = R.range_proof_vec
import bench
x = bench.synth_multi(4, 0, 0)
for P in (4, 64):
    for k in range(3):
        pr, cm = rpv.create_rangeproof(x[0], x[1], 32, P, nonce=R.Nonce.seeded(b"\x09" * 32), fp=(32, 7))
        assert rpv.verify_rangeproof(pr, cm, 32, verifier_seed=b"\x02" * 32, fp=(32, 7))
run("same client, after cfg-4 clients", lambda i: 0, False)
