"""All five BASELINE.json configs on ONE MI355X, with the reference's bench protocol (rofl_crypto/benches/rangeproof_bench.rs:53-85,
bench_constants.rs:10: one warm-up, 4 timed samples, median), one client at a time through the C ABI.  The first call of a shape
is reported separately as `cold_*` (it builds the generator tables the reference recomputes on every call,
range_proof_vec/mod.rs:126,201).  Configs 4 and 5 are 48-client jobs over 8 GPUs: one GPU's share (6 clients) is run here.
Writes one JSON object per config to stdout / gpurun_out/<TAG>_configs.json (TAG from the environment, default r06)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params

R.set_device(0)
SAMPLES = 4


def clip_inputs(rng, d, nb, fp_frac):
    """values ~ U[fp_min, fp_max) f32 (benches/rangeproof_bench.rs:41,48-50), blindings canonical (< 2^252)"""
    mx = np.float32(((1 << (nb - 1)) - 1) / float(1 << fp_frac))
    vals = rng.uniform(-mx, mx, size=d).astype(np.float32)
    vals = np.clip(vals, -mx, np.nextafter(mx, np.float32(0)))
    bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    return vals, bl


def med(xs):
    return float(np.median(xs))


def linf(name, d, nb, P, fp_bits, fp_frac, clients=1, batch_verify=False):
    api.set_fp(fp_bits, fp_frac)
    ins = [clip_inputs(np.random.default_rng(1000 * c), d, nb, fp_frac) for c in range(clients)]
    t = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(*ins[0], nb, P, nonce=R.Nonce.seeded(b"\x07" * 32))
    cold_c = (time.perf_counter() - t) * 1e3
    t = time.perf_counter(); ok = R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x08" * 32); cold_v = (time.perf_counter() - t) * 1e3
    assert ok
    # the first call ran on the compact fold table; the full one is built in the first quiet moment -- here: now (steady state is what the samples measure)
    t = time.perf_counter(); api.bp_gens_prepare(nb, max(R.range_proof_vec.next_pow2(d) // P, 1)); full_tab = (time.perf_counter() - t) * 1e3
    tc, tv, tb, tcb, tb2 = [], [], [], [], []
    for s in range(SAMPLES):
        prs, cms = [], []
        t0 = time.perf_counter()
        for c in range(clients):
            pr, cm = R.range_proof_vec.create_rangeproof(*ins[c], nb, P, nonce=R.Nonce.seeded(bytes([s + 1, c]) * 16))
            prs.append(pr); cms.append(cm)
        t1 = time.perf_counter()
        oks = [R.range_proof_vec.verify_rangeproof(prs[c], cms[c], nb, verifier_seed=bytes([s]) * 32) for c in range(clients)]
        t2 = time.perf_counter()
        assert all(oks)
        tc.append((t1 - t0) * 1e3 / clients); tv.append((t2 - t1) * 1e3 / clients)
        if batch_verify:
            t3 = time.perf_counter(); okb = R.range_proof_vec.verify_rangeproof_batch(prs, cms, nb, verifier_seed=bytes([s]) * 32); tb.append((time.perf_counter() - t3) * 1e3 / clients)
            assert all(okb)
            R.set_option("verify_batch", 2)      # one random-weighted check for the six clients (the server role)
            t3 = time.perf_counter(); okb = R.range_proof_vec.verify_rangeproof_batch(prs, cms, nb, verifier_seed=bytes([s]) * 32); tb2.append((time.perf_counter() - t3) * 1e3 / clients)
            R.set_option("verify_batch", 1)
            assert all(okb)
            t4 = time.perf_counter()
            res = R.range_proof_vec.create_rangeproof_batch([i[0] for i in ins], [i[1] for i in ins], nb, P, nonces=[R.Nonce.seeded(bytes([s + 1, c]) * 16) for c in range(clients)])
            tcb.append((time.perf_counter() - t4) * 1e3 / clients)
            assert all((res[c][0] == prs[c]).all() and (res[c][1] == cms[c]).all() for c in range(clients))      # bit-identical to the per-client calls
    out = {"config": name, "d": d, "prove_range": nb, "n_partition": P, "fp": [fp_bits, fp_frac], "clients_on_this_gpu": clients,
           "create_ms_per_client": med(tc), "verify_ms_per_client": med(tv),
           "create_elements_per_s": d / med(tc) * 1e3, "verify_elements_per_s": d / med(tv) * 1e3,
           "create_plus_verify_elements_per_s": d / (med(tc) + med(tv)) * 1e3,
           "cold_create_ms": cold_c, "cold_verify_ms": cold_v, "full_fold_table_ms": full_tab,
           "protocol": "1 warm-up (the cold call: served from the compact fold table), rofl_bp_gens_prepare (the full table, full_fold_table_ms), 4 samples, median; sequential clients"}
    if tb:
        out["batch_verify_ms_per_client"] = med(tb); out["batch_verify_elements_per_s"] = d / med(tb) * 1e3
        out["batch_verify_one_check_ms_per_client"] = med(tb2); out["batch_verify_one_check_elements_per_s"] = d / med(tb2) * 1e3
        out["batch_create_ms_per_client"] = med(tcb); out["batch_create_plus_batch_verify_elements_per_s"] = d / (med(tcb) + med(tb)) * 1e3
    return out


def l2(name, d, P, clients=1):
    """EncParamsL2::encrypt / verify (rofl_service/src/flserver/params.rs:608-646, 206-234): value_range 8 L-inf proof + l2_value_range 32
    sum proof + per-element square proofs (cifar_large.yml:41-43)."""
    api.set_fp(32, 7)
    ins = []
    for c in range(clients):
        rng = np.random.default_rng(77 + c)
        vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
        r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
        r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
        ins.append((vals, r1, r2))
    tc, tv = [], []
    cold = None
    WARM = 3      # the three proofs of a composite land on different lanes from call to call: every lane's workspace has to grow once
    for s in range(SAMPLES + WARM):
        t0 = time.perf_counter(); outs = []
        for vals, r1, r2 in ins:
            outs.append(params.EncParamsL2.encrypt(vals, r1, 8, P, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2))
        t1 = time.perf_counter()
        for upd in outs:
            assert upd.verify(verifier_seed=b"\x04" * 32)
        t2 = time.perf_counter()
        if s == 0:
            cold = ((t1 - t0) * 1e3, (t2 - t1) * 1e3)
            api.bp_gens_prepare(8, max(R.range_proof_vec.next_pow2(d) // P, 1)); api.bp_gens_prepare(32, 1)      # full fold tables before the samples
        if s < WARM:
            continue
        tc.append((t1 - t0) * 1e3 / clients); tv.append((t2 - t1) * 1e3 / clients)
    return {"config": name, "d": d, "n_partition": P, "fp": [32, 7], "clients_on_this_gpu": clients, "value_range": 8, "l2_value_range": 32,
            "create_ms_per_client": med(tc), "verify_ms_per_client": med(tv), "create_plus_verify_elements_per_s": d / (med(tc) + med(tv)) * 1e3,
            "cold_create_ms": cold[0], "cold_verify_ms": cold[1], "protocol": "3 warm-ups (the first is the cold pass), 4 samples, median; sequential clients, the three proofs of a client on separate lanes (EncParamsL2.encrypt / verify)"}


res = [
    linf("cfg1: L-inf 8-bit, d=5000 (mnist_dev_intrinsic_5k), fp16/frac7 as the published bench files", 5000, 8, 4, 16, 7),
    linf("cfg2: L-inf 32-bit, d=25000 (resnet18_intrinsic_25k), P=4 [headline]", 25000, 32, 4, 32, 7),
    linf("cfg2 with the e2e partition count P=64 (cifar_large.yml:39-46)", 25000, 32, 64, 32, 7),
    l2("cfg3: L2 composite, d=25000, P=4", 25000, 4),
    linf("cfg4: L-inf 32-bit, d=55000 (resnet18_intrinsic_55k), 6 of 48 clients (one GPU's share of 8), P=4", 55000, 32, 4, 32, 7, clients=6, batch_verify=True),
    linf("cfg4 with P=64", 55000, 32, 64, 32, 7, clients=6, batch_verify=True),
    l2("cfg5: L2 composite, d=55000, 6 of 48 clients (one GPU's share of 8), P=4", 55000, 4, clients=6),
]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", os.environ.get("TAG", "r06") + "_configs.json"), "w") as f:
    json.dump(res, f, indent=1)
for r in res:
    print(json.dumps(r))
