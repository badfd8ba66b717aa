"""SURVEY 8(e), the half of the row that round 5 left unbuilt: ONE client's chunks over several devices / ranks ("cfg 2/3 at > 1 GPU ->
chunks over ranks").  The reference proves and verifies a client's chunks independently of each other on its rayon pool
(range_proof_vec/mod.rs:54-78, 168-181: a transcript and a generator set per chunk), so any run of chunks can be taken by another device:

* rofl_create_rangeproof_chunks / rofl_verify_rangeproof_chunks -- the per-rank unit (one process per GPU): every partition of the chunks
  into runs reproduces the unsplit bytes and the oracle's, padding-only runs and explicit nonce streams included;
* rofl_set_option("devices", mask) -- one process: the unchanged single-client calls deal the chunks to 2 and 4 logical devices
  (rofl_dbg_map_device: full device contexts on the one GPU of the box); at BASELINE cfg 2's full size with P = 4 (whole proof against the
  oracle) and P = 64 (sampled chunks against the oracle, everything against the one-device call).
Bit-exact: integer work."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def R():
    import rofl_project_code_amd as R
    from rofl_project_code_amd import build
    build.build()
    R.set_device(0)
    return R


def _inputs(seed, d, nb, fp):
    rng = np.random.default_rng(seed)
    mx = np.float32(((1 << (nb - 1)) - 1) / float(1 << fp[1]))
    vals = np.clip(rng.uniform(-mx, mx, d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
    return vals, orc.rand_scalars(rng, d)


@pytest.mark.parametrize("d,nb,P,fp", [(40, 8, 4, (16, 7)), (5, 8, 8, (16, 7)), (100, 16, 8, (16, 7)), (33, 32, 64, (32, 7)), (1000, 32, 2, (32, 12))])
def test_runs_of_chunks_reproduce_the_unsplit_call(R, d, nb, P, fp):
    vals, bl = _inputs(7 * d + nb, d, nb, fp)
    seed = bytes([d % 251]) * 32
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed), fp=fp)
    rc, opr, ocm = orc.create_rangeproof(vals, bl, nb, P, fp[0], fp[1], seed=seed)
    assert rc == 0 and (opr == pr).all() and (ocm == cm).all()
    np_, m = R.range_proof_vec.chunk_geometry(d, P)
    assert np_ == pr.shape[0]
    for cuts in ([0, np_], [0, 1, np_], [0, np_ // 2, np_], list(range(np_ + 1))):
        cuts = sorted(set(cuts))
        ps, cs, oks = [], [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            p, c = R.range_proof_vec.create_rangeproof_chunks(vals, bl, nb, P, a, b - a, nonce=R.Nonce.seeded(seed), fp=fp)
            assert p.shape == (b - a, pr.shape[1]) and c.shape[0] == max(0, min(d, b * m) - min(d, a * m))
            ps.append(p); cs.append(c)
            oks.append(R.range_proof_vec.verify_rangeproof_chunks(p, np_, a, c, d, nb, verifier_seed=b"\x03" * 32, fp=fp))
        assert (np.concatenate(ps) == pr).all() and (np.concatenate(cs) == cm).all(), cuts
        assert all(oks)
    # a tampered proof fails its own run only
    for c in range(np_):
        bad = pr[c:c + 1].copy(); bad[0, 40] ^= 1
        lo, hi = min(d, c * m), min(d, (c + 1) * m)
        assert R.range_proof_vec.verify_rangeproof_chunks(bad, np_, c, cm[lo:hi], d, nb, verifier_seed=b"\x03" * 32, fp=fp) is False
        assert R.range_proof_vec.verify_rangeproof_chunks(pr[c:c + 1], np_, c, cm[lo:hi], d, nb, verifier_seed=b"\x03" * 32, fp=fp) is True
    # a run's proof is not another run's: chunk 0's proof against chunk 1's commitments (when both hold real elements)
    if np_ > 1 and d > m:
        assert R.range_proof_vec.verify_rangeproof_chunks(pr[0:1], np_, 1, cm[m:min(d, 2 * m)], d, nb, verifier_seed=b"\x03" * 32, fp=fp) is False


def test_runs_with_an_explicit_nonce_stream_and_errors(R):
    """mode 0 (64-byte wide scalars in upstream's draw order): a run reads its own part of the client's stream; the errors of the run are
    those of the elements it reads."""
    fp = (16, 7); d, nb, P = 24, 8, 4
    vals, bl = _inputs(99, d, nb, fp)
    np_, m = R.range_proof_vec.chunk_geometry(d, P)
    per = m * (2 * nb + 4)
    stream = np.random.default_rng(5).integers(0, 256, size=(np_ * per, 64), dtype=np.uint8)
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.stream(stream), fp=fp)
    rc, opr, ocm = orc.create_rangeproof(vals, bl, nb, P, fp[0], fp[1], stream=stream)
    assert rc == 0 and (opr == pr).all() and (ocm == cm).all()
    for a, n in ((0, 1), (1, 2), (3, 1), (2, 2)):
        p, c = R.range_proof_vec.create_rangeproof_chunks(vals, bl, nb, P, a, n, nonce=R.Nonce.stream(stream), fp=fp)
        assert (p == pr[a:a + n]).all() and (c == cm[min(d, a * m):min(d, (a + n) * m)]).all()
    with pytest.raises(R.RoflError) as e:      # a stream that is too short for the CLIENT is rejected whichever run is asked for
        R.range_proof_vec.create_rangeproof_chunks(vals, bl, nb, P, 0, 1, nonce=R.Nonce.stream(stream[:-1]), fp=fp)
    assert e.value.code == 12
    vbad = vals.copy(); vbad[d - 1] = np.float32(1e6)
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.create_rangeproof_chunks(vbad, bl, nb, P, (d - 1) // m, 1, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp)
    assert e.value.code == 2
    p0, _ = R.range_proof_vec.create_rangeproof_chunks(vbad, bl, nb, P, 0, 1, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp)      # run 0 does not read the bad element
    assert p0.shape[0] == 1
    for a, n in ((4, 1), (3, 2), (0, 5)):
        with pytest.raises(R.RoflError) as e:
            R.range_proof_vec.create_rangeproof_chunks(vals, bl, nb, P, a, n, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp)
        assert e.value.code == 11


def _run_worker(npz, nd, P, timeout=600):
    env = dict(os.environ); env.pop("ROFL_DEVICE_MAP", None)
    env.update({"ROFL_FOLD_TAB_MB": "2048", "ROFL_LANES": "2"})      # four device contexts on one GPU: small fold tables (same results)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_split_worker.py"), npz, str(nd), str(P)], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "split ok:" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("nd", [2, 4])
def test_cfg2_full_size_chunks_over_logical_devices(full_oracle, tmp_path, nd):
    """BASELINE cfg 2 (d = 25 000, 32-bit, P = 4) with the client's four chunks on 2 and on 4 logical devices: proofs and commitments equal
    the oracle's whole proof (and the one-device call's); verdicts, tamper, FormatError and ValueOutOfRangeError as the unsplit call's."""
    c = full_oracle.case("cfg2")
    npz = str(tmp_path / "cfg2.npz")
    np.savez(npz, vals=c["vals"], bl=c["bl"], seed=np.frombuffer(c["seed"], np.uint8), nb=c["nb"], fp=np.array(c["fp"]), opr=c["opr"], ocm=c["ocm"])
    _run_worker(npz, nd, 4)


@pytest.mark.parametrize("nd", [2, 4])
def test_cfg2_e2e_partition_chunks_over_logical_devices(full_oracle, tmp_path, nd):
    """The same client at n_partition = 64 (cifar_large.yml:39-46): 64 chunks of 512 values in runs of 32 / 16 per device; chunks 0, 15, 16,
    48 and 63 (run boundaries, and a padding-only chunk at the end) against the oracle proving exactly that chunk (orc.prove_chunk), all 64
    against the one-device call."""
    c = full_oracle.inputs("cfg2")
    d, nb, P = c["d"], c["nb"], 64
    dp = 1 << (d - 1).bit_length(); m = dp // P
    vp = np.zeros(dp, np.float32); vp[:d] = c["vals"]
    bp = np.zeros((dp, 32), np.uint8); bp[:d] = c["bl"]
    idx, proofs = [0, 15, 16, 48, 63], []
    for k in idx:
        lo, hi = k * m, min((k + 1) * m, d)
        rc, op, _ = orc.prove_chunk(vp[k * m:(k + 1) * m], bp[k * m:(k + 1) * m], nb, k, c["fp"][1], c["seed"], n_real=max(hi - lo, 0))
        assert rc == 0
        proofs.append(op)
    npz = str(tmp_path / "cfg2_p64.npz")
    np.savez(npz, vals=c["vals"], bl=c["bl"], seed=np.frombuffer(c["seed"], np.uint8), nb=nb, fp=np.array(c["fp"]), ochunk_idx=np.array(idx), ochunk_proofs=np.stack(proofs))
    _run_worker(npz, nd, 64)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_sigma_vector_runs_of_elements_reproduce_the_unsplit_call(R, kind):
    """The per-element Sigma-proofs of a vector are independent (rand_proof_vec/mod.rs:45-58, square_rand_proof_vec/mod.rs:45-58): runs of
    elements proved by rofl_create_sigmaproof_vec_range concatenate to the bytes of the unsplit call -- and of the oracle --, seeded and with an
    explicit nonce stream, with and without existing commitments; every run verifies on its own sub-arrays."""
    from rofl_project_code_amd import api
    fp = (32, 7); d = 333
    rng = np.random.default_rng(900 + kind)
    vals = (rng.integers(-400, 400, size=d) / 128.0).astype(np.float32)
    r1, r2 = orc.rand_scalars(rng, d), orc.rand_scalars(rng, d)
    nn = 2 if kind == 0 else 3
    mods = {0: R.rand_proof_vec, 1: R.square_rand_proof_vec, 2: R.square_proof_vec}
    def whole(nonce, existing):
        if kind == 0:
            return R.rand_proof_vec.create_randproof_vec(vals, r1, nonce=nonce, existing=existing, fp=fp)
        return mods[kind].create_l2rangeproof_vec(vals, r1, r2, nonce=nonce, existing=existing, fp=fp)
    verify = {0: R.rand_proof_vec.verify_randproof_vec, 1: R.square_rand_proof_vec.verify_l2rangeproof_vec, 2: R.square_proof_vec.verify_l2rangeproof_vec}[kind]
    stream = rng.integers(0, 256, size=(nn * d, 64), dtype=np.uint8)
    for nonce_of, okw in ((lambda: R.Nonce.seeded(b"\x33" * 32), dict(seed=b"\x33" * 32)), (lambda: R.Nonce.stream(stream), dict(stream=stream))):
        pr, cm = whole(nonce_of(), None)
        rc, opr, ocm = orc.sigma_create(kind, vals, r1, r2 if kind else None, fp[0], fp[1], **okw)
        assert rc == 0 and (opr == pr).all() and (ocm == cm).all()
        for existing in (None, cm[:, :32].copy()):
            want_p, want_c = whole(nonce_of(), existing)
            for cuts in ([0, d], [0, 1, d], [0, 100, 101, 250, d]):
                ps, cs = [], []
                for a, b in zip(cuts[:-1], cuts[1:]):
                    p, c = api.create_sigmaproof_vec_range(kind, vals, r1, r2 if kind else None, a, b - a, nonce=nonce_of(), existing=existing, fp=fp)
                    assert verify(p, c) is True
                    ps.append(p); cs.append(c)
                assert (np.concatenate(ps) == want_p).all() and (np.concatenate(cs) == want_c).all(), (kind, cuts, existing is not None)
    bad = pr[100:250].copy(); bad[7, bad.shape[1] - 60] ^= 1      # a response scalar (a flipped point byte would be the vector's FormatError)
    assert verify(bad, cm[100:250]) is False
    with pytest.raises(R.RoflError) as e:
        api.create_sigmaproof_vec_range(kind, vals, r1, r2 if kind else None, 300, 40, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp)
    assert e.value.code == 11
    with pytest.raises(R.RoflError) as e:      # a stream too short for the VECTOR is refused whichever run is asked for
        api.create_sigmaproof_vec_range(kind, vals, r1, r2 if kind else None, 0, 10, nonce=R.Nonce.stream(stream[:-1]), fp=fp)
    assert e.value.code == 12


def test_l2_update_over_logical_devices(tmp_path):
    """BASELINE cfg 3 with ONE client over 2 and 4 devices: EncParamsL2.encrypt / verify (rofl_service/src/flserver/params.rs:608-646, 204-232) under
    rofl_set_option("devices") -- the 8-bit range leg split by chunks, the square proofs by runs of elements -- returns the bytes of the one-device
    update, which verify on one device and split; a tampered square proof fails."""
    env = dict(os.environ); env.pop("ROFL_DEVICE_MAP", None)
    env.update({"ROFL_FOLD_TAB_MB": "2048", "ROFL_LANES": "3"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_split_worker.py"), "l2", "4"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "split ok:" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
