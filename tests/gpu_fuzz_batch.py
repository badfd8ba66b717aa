"""Randomised stress of the round-4 paths (run on the GPU box, time-boxed): several host threads, two logical devices on GPU 0, batches of
random size and shape through rofl_verify_rangeproof_batch with verify_batch 1 and 2 and the "devices" option on and off -- random members
tampered, every verdict list compared with per-client verification and (sampled) with the oracle; batch creates compared with single creates."""
import sys, os, time, threading, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc
import rofl_project_code_amd as R
from rofl_project_code_amd import api
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
R.set_device(0); api.map_device(1, 0)
rpv = R.range_proof_vec
t0 = time.time(); done = [0, 0]; errs = []
lock = threading.Lock()      # the options are process-wide: a round sets them and runs under the lock; creates run outside it, concurrently


def worker(tid):
    rng = np.random.default_rng(seed0 * 100 + tid)
    R.set_device(tid % 2)
    try:
        while time.time() - t0 < budget:
            nb = int(rng.choice([8, 16, 32])); fp = (32, int(rng.integers(0, 8)))
            P = int(rng.choice([1, 2, 4, 8])); d = int(rng.integers(3, 2500)); nc = int(rng.integers(1, 12))
            mn, mx = R.conversion32.get_clip_bounds(nb, fp=fp)
            ins = []
            for c in range(nc):
                v = np.clip(rng.uniform(mn, mx, size=d).astype(np.float32), mn, np.nextafter(np.float32(mx), np.float32(0)))
                ins.append((v, orc.rand_scalars(rng, d), bytes(rng.integers(0, 256, 32, dtype=np.uint8))))
            res = rpv.create_rangeproof_batch([i[0] for i in ins], [i[1] for i in ins], nb, P, nonces=[R.Nonce.seeded(i[2]) for i in ins], fp=fp)
            k = int(rng.integers(0, nc))
            one = rpv.create_rangeproof(ins[k][0], ins[k][1], nb, P, nonce=R.Nonce.seeded(ins[k][2]), fp=fp)
            assert (one[0] == res[k][0]).all() and (one[1] == res[k][1]).all(), ("batch create", nb, P, d, nc, k)
            proofs = [r[0].copy() for r in res]; commits = [r[1].copy() for r in res]
            want = [True] * nc
            for c in range(nc):
                u = rng.random()
                if u < 0.15: proofs[c][rng.integers(0, proofs[c].shape[0]), rng.integers(0, proofs[c].shape[1])] ^= 1 << int(rng.integers(0, 8)); want[c] = None
                elif u < 0.2: commits[c][rng.integers(0, d)] = commits[c][rng.integers(0, d)] if d > 1 else commits[c][0]; want[c] = None
            single = []
            for c in range(nc):
                try: single.append(rpv.verify_rangeproof(proofs[c], commits[c], nb, fp=fp))
                except R.RoflError: single.append(False)
            for c in range(nc):
                if want[c] is True: assert single[c] is True, ("untampered client rejected", nb, P, d, nc, c)
            if nc and rng.random() < 0.3:
                c = int(rng.integers(0, nc)); orc_rc, orc_ok = orc.verify_rangeproof(proofs[c], commits[c], nb, fp[0], fp[1])
                assert (orc_rc == 0 and orc_ok) == single[c], ("oracle disagrees", nb, P, d, c)
            with lock:
                for vb in (1, 2):
                    for devs in (0, 0b11):
                        R.set_option("verify_batch", vb); R.set_option("devices", devs)
                        got = rpv.verify_rangeproof_batch(proofs, commits, nb, verifier_seed=bytes(rng.integers(0, 256, 32, dtype=np.uint8)), fp=fp)
                        assert got == single, ("batch verdicts", vb, devs, nb, P, d, nc, got, single)
                R.set_option("verify_batch", 1); R.set_option("devices", 0)
            done[tid % 2] += 1
    except BaseException as e:      # noqa: BLE001
        errs.append(repr(e))


ts = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
for t in ts: t.start()
for t in ts: t.join()
assert not errs, errs
print(f"batch fuzz ok: {sum(done)} rounds in {time.time() - t0:.0f} s")
