"""Soak: a few thousand calls through the C ABI (single create / verify, batched calls with good and bad members, the L2 composite, several shapes so the
table cache turns over) while watching host RSS and free device memory -- a leak shows as a trend after the warm-up.  gpu_soak.py [minutes]"""
import ctypes, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, psutil
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params

hip = ctypes.CDLL("libamdhip64.so.7")      # the runtime the library is already bound to


def dev_free_mb():
    fr, tot = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
    return fr.value / 2 ** 20


def inputs(seed, d, nb, fp):
    rng = np.random.default_rng(seed)
    mx = np.float32(((1 << (nb - 1)) - 1) / float(1 << fp[1]))
    vals = np.clip(rng.uniform(-mx, mx, d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
    bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    return vals, bl


minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
R.set_device(0)
proc = psutil.Process()
rpv = R.range_proof_vec
shapes = [(5000, 8, 4, (16, 7)), (3000, 32, 4, (32, 7)), (700, 8, 1, (16, 7)), (5000, 8, 64, (16, 7))]
samples, it, calls = [], 0, 0
t_end = time.time() + 60 * minutes
while time.time() < t_end:
    d, nb, P, fp = shapes[it % len(shapes)]
    vals, bl = inputs(it, d, nb, fp)
    pr, cm = rpv.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(bytes([it % 255 + 1]) * 32), fp=fp)
    assert rpv.verify_rangeproof(pr, cm, nb, fp=fp) is True
    calls += 2
    if it % 7 == 0:      # batched calls, one bad member
        ins = [inputs(1000 + it + k, d, nb, fp) for k in range(4)]
        res = rpv.create_rangeproof_batch([i[0] for i in ins], [i[1] for i in ins], nb, P, nonces=[R.Nonce.seeded(bytes([k + 1]) * 32) for k in range(4)], fp=fp)
        prs = [r[0].copy() for r in res]; cms = [r[1] for r in res]
        prs[2][0, 4 * 32 + 3] ^= 1
        for vb in (1, 2):
            R.set_option("verify_batch", vb)
            assert rpv.verify_rangeproof_batch(prs, cms, nb, fp=fp) == [True, True, False, True]
        R.set_option("verify_batch", 1)
        calls += 3
    if it % 11 == 0:     # the L2 composite
        v2, b2 = inputs(5000 + it, 1000, 8, (32, 7))
        rs = np.random.default_rng(it).integers(0, 256, size=(1000, 32), dtype=np.uint8); rs[:, 31] &= 0x0F
        upd = params.EncParamsL2.encrypt(v2, b2, 8, 4, 32, nonce_seed=bytes([it % 250 + 1]) * 32, rand_scalars=rs, fp=(32, 7))
        assert params.EncParamsL2.deserialize(upd.serialize()).verify(fp=(32, 7))
        calls += 2
    if it % 13 == 0:     # round 5: a round of L2 updates through encrypt_batch / verify_batch (three streams, two MSM workspaces, staging ring), one member tampered
        cl = []
        for k in range(5):
            v2, b2 = inputs(9000 + it + k, 2000, 8, (32, 7)); v2 = (np.round(v2 * 128) / 128 / 64).astype(np.float32)
            rs = np.random.default_rng(it + k).integers(0, 256, size=(2000, 32), dtype=np.uint8); rs[:, 31] &= 0x0F
            cl.append((v2, b2, rs))
        ups = params.EncParamsL2.encrypt_batch(cl, 8, 4, 32, nonce_seeds=[bytes([k + 1]) * 32 for k in range(5)], fp=(32, 7))
        blobs = [u.serialize(as_array=True) for u in ups]
        U = [params.EncParamsL2.deserialize(b, copy=False) for b in blobs]
        U[3] = params.EncParamsL2.deserialize(blobs[3]); U[3].square_proofs[7, 70] ^= 1
        for vb in (2, 1):
            R.set_option("verify_batch", vb)
            assert params.EncParamsL2.verify_batch(U, fp=(32, 7)) == [True, True, True, False, True]
        R.set_option("verify_batch", 1)
        calls += 8
    if it % 50 == 0:
        samples.append({"iteration": it, "calls": calls, "rss_mb": round(proc.memory_info().rss / 2 ** 20, 1), "device_free_mb": round(dev_free_mb(), 1)})
    it += 1
samples.append({"iteration": it, "calls": calls, "rss_mb": round(proc.memory_info().rss / 2 ** 20, 1), "device_free_mb": round(dev_free_mb(), 1)})
warm = samples[len(samples) // 3]      # after the first third everything that is going to be cached is cached
late = samples[3 * len(samples) // 4]  # a leak is a TREND: it shows in the last quarter as well (workspaces grow in rare steps: a data-dependent path met for the first time)
last = samples[-1]
out = {"minutes": minutes, "iterations": it, "calls": calls, "first": samples[0], "after_warm_up": warm, "last": last,
       "rss_growth_mb_after_warm_up": round(last["rss_mb"] - warm["rss_mb"], 1), "device_growth_mb_after_warm_up": round(warm["device_free_mb"] - last["device_free_mb"], 1),
       "samples": samples[:: max(1, len(samples) // 12)]}
out["rss_growth_mb_last_quarter"] = round(last["rss_mb"] - late["rss_mb"], 1); out["device_growth_mb_last_quarter"] = round(late["device_free_mb"] - last["device_free_mb"], 1)
print(json.dumps({k: out[k] for k in ("rss_growth_mb_last_quarter", "device_growth_mb_last_quarter")}), file=sys.stderr)
assert out["rss_growth_mb_last_quarter"] < 16 and out["device_growth_mb_last_quarter"] < 16, "memory keeps growing"
print(json.dumps(out))
