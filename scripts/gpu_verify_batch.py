"""Server role of BASELINE cfg 4: NC clients of d = 55 000 (32-bit, n_partition P) verified by rofl_verify_rangeproof_batch --
verify_batch = 1 (one check per client, groups of six per call as bench.py --config 4 did in round 3) against verify_batch = 2 (one
random-weighted check for the whole batch).  Prints ms per round of NC clients and elements/s; ROFL_TRACE=2 adds the phase times."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rofl_project_code_amd as R
from rofl_project_code_amd import api

NC = int(os.environ.get("NC", "48")); D = int(os.environ.get("D", "55000")); P = int(os.environ.get("P", "4")); REPS = int(os.environ.get("REPS", "5"))
MODES = os.environ.get("MODES", "1,2").split(",")
FP = (32, 7); NB = 32
R.set_device(0)
rpv = R.range_proof_vec
mx = np.float32(16777216.0)
proofs, commits = [], []
t0 = time.perf_counter()
for g0 in range(0, NC, 6):
    ins = []
    for c in range(g0, min(g0 + 6, NC)):
        rng = np.random.default_rng(1000 * c)
        vals = np.clip(rng.uniform(-mx, mx, size=D).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
        bl = rng.integers(0, 256, size=(D, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
        ins.append((vals, bl))
    res = rpv.create_rangeproof_batch([i[0] for i in ins], [i[1] for i in ins], NB, P, nonces=[R.Nonce.seeded(bytes([c % 251 + 1]) * 32) for c in range(g0, g0 + len(ins))], fp=FP)
    for r in res:
        proofs.append(r[0]); commits.append(r[1])
print("created %d clients in %.2f s" % (NC, time.perf_counter() - t0), file=sys.stderr)
out = {}
for mode in MODES:
    vb = int(mode)
    R.set_option("verify_batch", vb)
    grp = 6 if vb == 1 else NC
    def round_(tag):
        oks = []
        for g0 in range(0, NC, grp):
            oks += rpv.verify_rangeproof_batch(proofs[g0:g0 + grp], commits[g0:g0 + grp], NB, verifier_seed=bytes([tag % 256]) * 32, fp=FP)
        assert all(oks), oks
    round_(0); round_(1)
    ts = []
    for r in range(REPS):
        t = time.perf_counter(); round_(r + 2); ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    out["verify_batch=%d" % vb] = {"ms_per_round_median": ts[len(ts) // 2], "ms_min": ts[0], "verify_only_elements_per_s": NC * D / (ts[len(ts) // 2] * 1e-3), "calls_per_round": (NC + grp - 1) // grp}
    if os.environ.get("KTIMES", "1") == "1":      # one fully instrumented round: HIP events around every instrumented launch
        R.set_timing(1)
        oks = rpv.verify_rangeproof_batch(proofs[:grp], commits[:grp], NB, verifier_seed=b"\x55" * 32, fp=FP)
        kt = R.last_kernel_times(); tm = R.last_timing()
        R.set_timing(0)
        out["verify_batch=%d" % vb]["kernels_ms_one_call_of_%d_clients" % grp] = {k: round(v["ms"], 3) for k, v in kt.items() if v["launches"]}
        out["verify_batch=%d" % vb]["device_span_ms"] = round(tm["total_ms"], 3); out["verify_batch=%d" % vb]["host_ms"] = round(tm["host_ms"], 3)
    if vb == 2 and os.environ.get("TAMPER", "1") == "1":      # the closer look: one bad client among NC
        bad = [p.copy() for p in proofs]; bad[NC // 3][1, 70] ^= 1
        t = time.perf_counter()
        oks = rpv.verify_rangeproof_batch(bad, commits, NB, verifier_seed=b"\x77" * 32, fp=FP)
        dt = (time.perf_counter() - t) * 1e3
        assert oks == [i != NC // 3 for i in range(NC)], oks
        out["verify_batch=2 with one bad client"] = {"ms": dt}
R.set_option("verify_batch", 1)
print(json.dumps({"clients": NC, "d": D, "n_partition": P, **out}))
