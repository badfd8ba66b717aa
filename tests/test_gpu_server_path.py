"""The server role at the size it runs at (BASELINE cfg 4: L-inf 32-bit, d = 55 000, a round of clients verified as a batch;
rofl_service/src/flserver/server.rs:474-484 rejects the round when any client fails, :656-687 verifies every client of the round).
rofl_set_option("verify_batch", 2) + ("devices", 0b11): ONE random-weighted equation per device share, groups and single members only
when it fails.  A batch verifier is defined by what it rejects: twelve clients at full size with none / one / three bad members and a
member whose scalars are built to collide (the coarse bins of the two-level sort overflow and its check is repeated on the slot path)
-- every verdict list must equal the per-client check's (verify_batch = 1, one device), and the oracle must reject each bad member.
The randomised stress of the same paths (tests/gpu_fuzz_batch.py: threads x logical devices x verify_batch) runs here too, time-boxed."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP = (32, 7)
D, NB = 55000, 32
NC = 12


@pytest.fixture(scope="module")
def R():
    import rofl_project_code_amd as R
    from rofl_project_code_amd import api, build
    build.build()
    R.set_device(0)
    api.map_device(1, 0)              # the box has one GPU: logical device 1 is a second full context on it
    yield R
    R.set_option("devices", 0); R.set_option("verify_batch", 1)
    R.set_device(0)


def _inputs(seed):
    rng = np.random.default_rng(seed)
    mx = np.float32(((1 << (NB - 1)) - 1) / float(1 << FP[1]))
    vals = np.clip(rng.uniform(-mx, mx, D).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
    return vals, orc.rand_scalars(rng, D)


def _round(R, P, seed0):
    """NC clients of cfg 4's shape, proved in two batched calls of six (what bench.py --config 4 does per rank)."""
    ins = [_inputs(seed0 + i) for i in range(NC)]
    nonces = [R.Nonce.seeded(bytes([(seed0 + i) % 251]) * 32) for i in range(NC)]
    out = []
    for k in range(0, NC, 6):
        out += R.range_proof_vec.create_rangeproof_batch([v for v, _ in ins[k:k + 6]], [b for _, b in ins[k:k + 6]], NB, P, nonces=nonces[k:k + 6], fp=FP)
    return [p for p, _ in out], [c for _, c in out]


def _both(R, proofs, commits, seed=b"\x17" * 32):
    """verdicts of the server path (one equation per batch, two devices) and of the per-client path; they must be the same list"""
    R.set_option("verify_batch", 2); R.set_option("devices", 0b11)
    server = R.range_proof_vec.verify_rangeproof_batch(proofs, commits, NB, verifier_seed=seed, fp=FP)
    R.set_option("verify_batch", 1); R.set_option("devices", 0)
    per_client = R.range_proof_vec.verify_rangeproof_batch(proofs, commits, NB, verifier_seed=seed, fp=FP)
    assert server == per_client, (server, per_client)
    return server


def _oracle_rejects_chunk(proofs, commits, c, chunk):
    """the oracle on the one chunk that was tampered with (a full chunk: its commitments are a padded vector of their own)"""
    lo = c * chunk
    assert lo + chunk <= D
    rc, ok = orc.verify_rangeproof(proofs[c:c + 1].copy(), commits[lo:lo + chunk].copy(), NB, FP[0], FP[1])
    return rc != 0 or ok is False


def test_round_of_twelve_at_cfg4_size_rejects_what_the_per_client_check_rejects(R):
    from rofl_project_code_amd import api
    P = 4; chunk = 65536 // P; lg = (NB * chunk).bit_length() - 1
    P0, C0 = _round(R, P, 4000)
    try:
        assert _both(R, P0, C0) == [True] * NC
        # the oracle accepts an untampered chunk of this round (the rejections below are the tampering's, not the shape's)
        assert orc.verify_rangeproof(P0[5][1:2].copy(), C0[5][chunk:2 * chunk].copy(), NB, FP[0], FP[1]) == (0, True)
        # one bad member: a flipped byte in the LAST round's R of chunk 2 (the point the verifier weighs with u_(lg-1)^-2)
        p = [x.copy() for x in P0]
        p[11][2, 7 * 32 + 64 * (lg - 1) + 32 + 5] ^= 0x10
        assert _both(R, p, C0) == [True] * 11 + [False]
        assert _oracle_rejects_chunk(p[11], C0[11], 2, chunk)
        # three bad members in three different groups of the closer look: a late L, two commitments swapped, a non-canonical scalar
        p = [x.copy() for x in P0]; c = [x.copy() for x in C0]
        p[1][0, 7 * 32 + 64 * (lg - 2) + 9] ^= 0x01
        c[6][[chunk + 3, chunk + 4]] = c[6][[chunk + 4, chunk + 3]]
        p[10][1, 128:160] = 0xFF                       # t_x >= l: RangeProof::from_bytes rejects the member; the batch goes on without it
        want = [True] * NC
        for i in (1, 6, 10): want[i] = False
        assert _both(R, p, c) == want
        assert _oracle_rejects_chunk(p[1], c[1], 0, chunk) and _oracle_rejects_chunk(p[6], c[6], 1, chunk) and _oracle_rejects_chunk(p[10], c[10], 1, chunk)
        # scalars built to collide: a = 0 in every chunk of client 3 gives all 2^19 G terms of its own check ONE scalar per window -- the batch
        # equation fails, the closer look reaches the client alone, its coarse bins overflow and its generator MSM is repeated on the slot path
        p = [x.copy() for x in P0]
        p[3][:, -64:-32] = 0
        before = api.msm_retries()
        want = [True] * NC; want[3] = False
        assert _both(R, p, C0) == want
        after = api.msm_retries()
        assert after["bin_overflow_to_slots"] > before["bin_overflow_to_slots"], (before, after)
        assert _oracle_rejects_chunk(p[3], C0[3], 0, chunk)
        # the lanes are healthy afterwards and an all-good round is still one equation per share
        assert _both(R, P0, C0, seed=b"\x18" * 32) == [True] * NC
    finally:
        R.set_option("devices", 0); R.set_option("verify_batch", 1)


def test_round_of_twelve_at_the_e2e_partition_count(R):
    """n_partition = 64 (ansible/experiments/cifar_large.yml:39-46): 64 proofs of 1 024 values per client, 768 proofs in the batch equation."""
    P = 64; chunk = 65536 // P; lg = (NB * chunk).bit_length() - 1
    P0, C0 = _round(R, P, 4100)
    try:
        assert _both(R, P0, C0) == [True] * NC
        p = [x.copy() for x in P0]; c = [x.copy() for x in C0]
        p[4][40, 7 * 32 + 64 * (lg - 1) + 32 + 1] ^= 0x04          # late R of chunk 40
        c[9][7] = c[9][8]                                           # a commitment replaced by its neighbour
        want = [True] * NC; want[4] = False; want[9] = False
        assert _both(R, p, c) == want
        assert _oracle_rejects_chunk(p[4], c[4], 40, chunk) and _oracle_rejects_chunk(p[9], c[9], 0, chunk)
    finally:
        R.set_option("devices", 0); R.set_option("verify_batch", 1)


def test_a_batch_of_one_gives_a_verdict_not_an_error(R):
    """A malformed member of a batch fails ALONE (include/rofl_zk.h) -- also when it is the only member: a non-canonical scalar in the proof of a
    one-client batch is ok = 0, not the FormatError that rofl_verify_rangeproof raises for a single set (found by tests/gpu_fuzz_batch.py at
    seed 424242, 300 s: the batch entry treated n_clients == 1 as the single call).  Both verify_batch modes, devices on and off."""
    rng = np.random.default_rng(77)
    d, nb, P, fp = 300, 16, 2, (16, 7)
    mn, mx = R.conversion32.get_clip_bounds(nb, fp=fp)
    ins = [(np.clip(rng.uniform(mn, mx, size=d).astype(np.float32), mn, np.nextafter(np.float32(mx), np.float32(0))), orc.rand_scalars(rng, d)) for _ in range(2)]
    res = [R.range_proof_vec.create_rangeproof(v, b, nb, P, nonce=R.Nonce.seeded(bytes([k + 1]) * 32), fp=fp) for k, (v, b) in enumerate(ins)]
    bad = res[0][0].copy(); bad[1, 128 + 31] |= 0xF0          # t_x of the second chunk: not a canonical scalar any more
    with pytest.raises(R.RoflError):
        R.range_proof_vec.verify_rangeproof(bad, res[0][1], nb, fp=fp)
    for vb in (1, 2):
        for devs in (0, 0b11):
            R.set_option("verify_batch", vb); R.set_option("devices", devs)
            assert R.range_proof_vec.verify_rangeproof_batch([bad], [res[0][1]], nb, verifier_seed=b"\x09" * 32, fp=fp) == [False]
            assert R.range_proof_vec.verify_rangeproof_batch([res[0][0]], [res[0][1]], nb, verifier_seed=b"\x09" * 32, fp=fp) == [True]
            assert R.range_proof_vec.verify_rangeproof_batch([bad, res[1][0]], [res[0][1], res[1][1]], nb, verifier_seed=b"\x09" * 32, fp=fp) == [False, True]
    R.set_option("verify_batch", 1); R.set_option("devices", 0)
    # the create side: a client whose value is out of range is ITS outcome (rc_out), also when it is the only client of the batch
    vbad = ins[0][0].copy(); vbad[5] = np.float32(1e9)
    out = R.range_proof_vec.create_rangeproof_batch([vbad], [ins[0][1]], nb, P, nonces=[R.Nonce.seeded(b"\x01" * 32)], fp=fp)
    assert isinstance(out[0], R.RoflError) and out[0].code == 2
    out = R.range_proof_vec.create_rangeproof_batch([vbad, ins[1][0]], [ins[0][1], ins[1][1]], nb, P, nonces=[R.Nonce.seeded(b"\x01" * 32), R.Nonce.seeded(b"\x02" * 32)], fp=fp)
    assert isinstance(out[0], R.RoflError) and out[0].code == 2 and (out[1][0] == res[1][0]).all()
    with pytest.raises(R.RoflError):
        R.range_proof_vec.create_rangeproof(vbad, ins[0][1], nb, P, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp)


def test_batch_paths_fuzz():
    """tests/gpu_fuzz_batch.py, 30 s here (the long form is the script itself), fixed seed: three host threads over two logical devices, random shapes (8 / 16 / 32 bits, 1-8 chunks,
    1-11 clients), random members tampered; batched creates against single creates, verify_batch 1 / 2 x devices on / off against per-client
    verification, sampled members against the oracle."""
    env = dict(os.environ); env.pop("ROFL_DEVICE_MAP", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_fuzz_batch.py"), "30", "20261003"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "batch fuzz ok:" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
