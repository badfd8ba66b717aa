"""What the Python binding adds to a cfg-2 step (run on the GPU box): the api wrappers (fresh output arrays per call) against the same two
C calls through ctypes with preallocated outputs, interleaved, median of 30."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
import bench
R.set_device(0); api.set_fp(32, 7)
vals, bl = bench.synth_client(1)
api.bp_gens_prepare(32, 8192)
L = api.lib(); sz = ctypes.c_size_t
d = vals.size
proofs = np.empty((4, 1440), np.uint8); commits = np.empty((d, 32), np.uint8)
ns = R.Nonce.seeded(b"\x01" * 32)._struct()
pl, npf, ok = sz(), sz(), ctypes.c_int()
seed = b"\x02" * 32


def direct():
    rc = L.rofl_create_rangeproof(vals.ctypes.data_as(ctypes.c_void_p), sz(d), bl.ctypes.data_as(ctypes.c_void_p), sz(d), sz(32), sz(4), 32, 7, ctypes.byref(ns),
                                  proofs.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pl), ctypes.byref(npf), commits.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    rc = L.rofl_verify_rangeproof(proofs.ctypes.data_as(ctypes.c_void_p), sz(1440), sz(4), commits.ctypes.data_as(ctypes.c_void_p), sz(d), sz(32), 32, 7, seed, ctypes.byref(ok))
    assert rc == 0 and ok.value == 1


def wrapped():
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 32, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
    assert R.range_proof_vec.verify_rangeproof(pr, cm, 32, verifier_seed=seed)


for f in (direct, wrapped, direct, wrapped):
    f()
td, tw = [], []
for i in range(30):
    t = time.perf_counter(); direct(); td.append((time.perf_counter() - t) * 1e3)
    t = time.perf_counter(); wrapped(); tw.append((time.perf_counter() - t) * 1e3)
td.sort(); tw.sort()
print("direct ctypes, preallocated outputs: median %.3f ms (min %.3f)   api wrappers: median %.3f ms (min %.3f)" % (td[15], td[0], tw[15], tw[0]))
