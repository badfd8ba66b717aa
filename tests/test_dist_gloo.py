"""world_size-2 gloo test of the N>1 path used by bench.py: clients sharded round-robin, proofs produced per rank
(by the oracle here -- no GPU in this container), all-gathered, and every rank verifies the other's proofs."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import orc
    from rofl_project_code_amd import dist as rd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_clients, d, nb, P, fb, ff = 5, 6, 8, 2, 16, 7
    mine = rd.shard_clients(n_clients, rank, world)
    assert mine == list(range(rank, n_clients, world))
    res = {}
    for c in range(n_clients):          # every rank can recompute any client's inputs (seeded)
        rng = np.random.default_rng(1000 * c)
        res[c] = (rng.uniform(-0.9, 0.9, d).astype(np.float32), orc.rand_scalars(rng, d))
    plen = 32 * (9 + 2 * 5)
    slots = (n_clients + world - 1) // world
    buf_p = np.zeros((slots, P, plen), np.uint8); buf_c = np.zeros((slots, d, 32), np.uint8)
    ok_local = True
    for s, c in enumerate(mine):
        rc, pr, cm = orc.create_rangeproof(res[c][0], res[c][1], nb, P, fb, ff, seed=bytes([c]) * 32)
        assert rc == 0
        buf_p[s], buf_c[s] = pr, cm
        ok_local &= orc.verify_rangeproof(pr, cm, nb, fb, ff) == (0, True)
    all_p = rd.gather_bytes(buf_p, "cpu"); all_c = rd.gather_bytes(buf_c, "cpu")
    assert len(all_p) == world
    # verify what the OTHER ranks produced
    for r in range(world):
        pp = all_p[r].reshape(slots, P, plen); cc = all_c[r].reshape(slots, d, 32)
        for s, c in enumerate(rd.shard_clients(n_clients, r, world)):
            assert orc.verify_rangeproof(pp[s], cc[s], nb, fb, ff) == (0, True)
    assert rd.all_verified(ok_local, "cpu") is True
    assert rd.all_verified(rank != 1, "cpu") is False     # one failing rank fails the round (server.rs:474-484)
    # the same round as ONE collective (what bench.py times): verify bit + both payloads of every rank in one all-gather
    ok_all, per_rank = rd.exchange_round([buf_p, buf_c], ok_local, "cpu")
    assert ok_all is True and len(per_rank) == world
    for r in range(world):
        assert (per_rank[r][0] == np.asarray(all_p[r]).reshape(-1)).all() and (per_rank[r][1] == np.asarray(all_c[r]).reshape(-1)).all()
    ok_all, _ = rd.exchange_round([buf_p, buf_c], rank != 1, "cpu")
    assert ok_all is False
    dist.barrier(); dist.destroy_process_group()
    q.put(rank)


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


def _comm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from rofl_project_code_amd import dist as rd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    notes = []
    # no GPU here: the ranks AGREE (over the gloo control group) that the library's RCCL communicator cannot be formed -- rank 0 cannot draw a
    # unique id -- and every rank ends up on the same fallback; nobody is left waiting inside ncclCommInitRank
    comm = rd.make_comm(rank, world, torch.device("cpu"), prefer_lib=True, torch_backend="gloo", log=notes.append)
    assert isinstance(comm, rd.TorchComm) and comm.backend == "gloo" and comm.world == world and comm.rank == rank
    ok, per = comm.exchange_round([np.full(40, rank, np.uint8), np.arange(5, dtype=np.uint8) + rank], rank == 0)
    assert ok is False and [int(p[0][0]) for p in per] == list(range(world)) and [int(p[1][4]) for p in per] == [4 + r for r in range(world)]
    assert comm.all_verified(True) is True and comm.all_verified(rank != 1) is False
    assert comm.reduce([float(rank + 1), 10.0], "sum").tolist() == [3.0, 20.0] and comm.reduce([float(rank)], "max")[0] == 1.0 and comm.reduce([float(rank)], "min")[0] == 0.0
    comm.barrier(); comm.close()
    one = rd.make_comm(0, 1, torch.device("cpu"))
    assert isinstance(one, rd.LocalComm) and one.exchange_round([np.zeros(3, np.uint8)], True)[0] is True and one.reduce([2.0], "max")[0] == 2.0
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, bool(notes) if rank == 0 else True))


def test_communicator_choice_is_agreed_across_ranks():
    """rofl_project_code_amd.dist.make_comm: LibComm (the library's RCCL) when every rank can form it, otherwise ONE fallback for all ranks.
    On the CPU rank 0 cannot draw the unique id; both ranks must land on TorchComm(gloo) and run the round's collectives there."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = sorted(q.get() for _ in range(2))
    assert got == [(0, True), (1, True)]      # (rank 0 logged why the library's communicator was not used)


def _split_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import orc
    from rofl_project_code_amd import dist as rd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = rd.TorchComm(torch.device("cpu"))
    assert rd.chunk_runs(4, 2) == [(0, 2), (2, 2)] and rd.chunk_runs(8, 3) == [(0, 2), (2, 3), (5, 3)] and rd.chunk_runs(2, 4) == [(0, 1), (1, 1), (0, 0), (0, 0)]
    # ONE client (same seeded inputs on every rank), d = 11 of d_pad = 16 in 8 chunks of m = 2: chunk 5 is half padding, chunks 6 and 7 padding only
    d, nb, P, fb, ff = 11, 8, 8, 16, 7
    rng = np.random.default_rng(77)
    vals = rng.uniform(-0.4, 0.4, d).astype(np.float32); bl = orc.rand_scalars(rng, d)
    seed = b"\x2b" * 32
    rc, want_p, want_c = orc.create_rangeproof(vals, bl, nb, P, fb, ff, seed=seed)      # the unsplit answer
    assert rc == 0
    n_chunks, plen = want_p.shape; m = 16 // n_chunks
    vp = np.zeros(16, np.float32); vp[:d] = vals
    bp = np.zeros((16, 32), np.uint8); bp[:d] = bl
    calls = []

    def create_run(first, count):      # the rank's run, chunk by chunk, from each chunk's own place in the client's nonce space
        calls.append((first, count))
        ps = []
        for c in range(first, first + count):
            lo, hi = c * m, min((c + 1) * m, d)
            rc, pr, _ = orc.prove_chunk(vp[c * m:(c + 1) * m], bp[c * m:(c + 1) * m], nb, c, ff, seed, n_real=max(hi - lo, 0))
            assert rc == 0
            ps.append(pr)
        lo, hi = min(d, first * m), min(d, (first + count) * m)
        return np.stack(ps), want_c[lo:hi]      # (commitments: C_j = f32_to_scalar(x_j) B + r_j B~, independent of the split)

    proofs, commits = rd.split_create(comm, rank, world, n_chunks, m, d, plen, create_run)
    assert calls == [rd.chunk_runs(n_chunks, world)[rank]]
    assert (proofs == want_p).all() and (commits == want_c).all()

    # The oracle on a run: the run's chunks as a proof set of their own.  verify_rangeproof shifts every commitment it is given up by 2^(n-1) B
    # and pads with the identity AFTERWARDS (range_proof_vec/mod.rs:155-167), so the padding elements of a run are handed over as
    # C_pad = -2^(n-1) B, whose shifted form is the identity the prover committed to.
    L = (1 << 252) + 27742317777372353535851937790883648493
    c_pad = orc.commit_vec(np.frombuffer((L - (1 << (nb - 1))).to_bytes(32, "little"), np.uint8).reshape(1, 32), np.zeros((1, 32), np.uint8))[0]

    def verify_run(first, pr, cm):
        cnt = pr.shape[0]
        full = np.tile(c_pad, (cnt * m, 1)); full[:cm.shape[0]] = cm
        return orc.verify_rangeproof(pr, full, nb, fb, ff) == (0, True)

    assert rd.split_verify(comm, rank, world, proofs, commits, m, verify_run) is True
    bad = proofs.copy(); bad[n_chunks - 1, 40] ^= 1      # a bad proof in the LAST run: whichever rank checks that run fails the client
    assert rd.split_verify(comm, rank, world, bad, commits, m, verify_run) is False
    dist.barrier(); dist.destroy_process_group()
    q.put(rank)


def test_one_client_split_over_two_ranks_gloo():
    """SURVEY 8(e) "cfg 2/3 at > 1 GPU -> chunks over ranks" on CPU: two ranks prove contiguous runs of ONE client's chunks (the oracle
    as the prover), one all-gather assembles proofs and commitments, every rank holds the unsplit bytes; the verdict is the MIN over runs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]


def test_split_helpers_with_more_ranks_than_chunks():
    """dist.split_create / split_verify in a world of EIGHT with four chunks (BASELINE cfg 2 at P = 4 on an 8-GPU node) and with 64: ranks without
    a run only join the collectives, every run is created once, checked once (by the rank before its owner), and the assembled arrays are the
    whole client's.  Threads and an in-process communicator stand in for the ranks: no GPU, no process group."""
    import threading
    sys.path.insert(0, ROOT)
    from rofl_project_code_amd import dist as rd

    class ThreadComm:
        def __init__(self, shared, rank, world): self.s, self.rank, self.world = shared, rank, world

        def _all(self, item):
            self.s["slots"][self.rank] = item
            self.s["bar"].wait()
            got = list(self.s["slots"])
            self.s["bar"].wait()
            return got

        def exchange_round(self, payloads, ok_local):
            got = self._all((bool(ok_local), [np.ascontiguousarray(p, dtype=np.uint8).reshape(-1).copy() for p in payloads]))
            return all(g[0] for g in got), [g[1] for g in got]

        def all_verified(self, ok_local): return all(self._all(bool(ok_local)))

    for n_chunks, d, m, plen in ((4, 25000, 8192, 1440), (64, 25000, 512, 1184), (4, 9000, 8192, 1440)):      # (the last: chunks 2 and 3 hold no real element)
        world = 8
        rng = np.random.default_rng(n_chunks + d)
        want_p = rng.integers(0, 256, size=(n_chunks, plen), dtype=np.uint8)
        want_c = rng.integers(0, 256, size=(d, 32), dtype=np.uint8)
        shared = {"slots": [None] * world, "bar": threading.Barrier(world)}
        created, checked, out, errs = [], [], {}, []

        def rank_main(r):
            try:
                comm = ThreadComm(shared, r, world)

                def create_run(first, count):
                    created.append((r, first, count))
                    return want_p[first:first + count], want_c[min(d, first * m):min(d, (first + count) * m)]

                def verify_run(first, pr, cm):
                    checked.append((r, first, pr.shape[0]))
                    return bool((pr == want_p[first:first + pr.shape[0]]).all() and (cm == want_c[min(d, first * m):min(d, (first + pr.shape[0]) * m)]).all())

                p, c = rd.split_create(comm, r, world, n_chunks, m, d, plen, create_run)
                ok = rd.split_verify(comm, r, world, p, c, m, verify_run)
                bad = p.copy(); bad[n_chunks - 1, 5] ^= 1
                ok_bad = rd.split_verify(comm, r, world, bad, c, m, verify_run)
                out[r] = (p, c, ok, ok_bad)
            except BaseException as e:      # noqa: BLE001
                errs.append(e); shared["bar"].abort()

        ths = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in ths: t.start()
        for t in ths: t.join()
        assert not errs, errs
        if n_chunks == 4 and d == 25000:      # the same for the ELEMENTS of a Sigma-proof vector (192-byte proofs, 96-byte commitments): runs of >= 2 048
            ep = rng.integers(0, 256, size=(d, 192), dtype=np.uint8); ec = rng.integers(0, 256, size=(d, 96), dtype=np.uint8)
            assert rd.elem_runs(5000, 8) == [(0, 2500), (2500, 2500)] + [(0, 0)] * 6 and rd.elem_runs(100, 4) == [(0, 100)] + [(0, 0)] * 3
            eout, eerrs = {}, []

            def elem_main(r):
                try:
                    comm = ThreadComm(shared, r, world)
                    p, c = rd.split_create_elems(comm, r, world, d, 192, 96, lambda f, n: (ep[f:f + n], ec[f:f + n]))
                    chk = lambda pr, cm: bool(any((pr == ep[f:f + pr.shape[0]]).all() and (cm == ec[f:f + pr.shape[0]]).all() for f, _ in rd.elem_runs(d, world)))
                    ok = rd.split_verify_elems(comm, r, world, p, c, chk)
                    bad = p.copy(); bad[d - 1, 3] ^= 1
                    eout[r] = (p, c, ok, rd.split_verify_elems(comm, r, world, bad, c, chk))
                except BaseException as e:      # noqa: BLE001
                    eerrs.append(e); shared["bar"].abort()

            ths = [threading.Thread(target=elem_main, args=(r,)) for r in range(world)]
            for t in ths: t.start()
            for t in ths: t.join()
            assert not eerrs, eerrs
            for r in range(world):
                assert (eout[r][0] == ep).all() and (eout[r][1] == ec).all() and eout[r][2] is True and eout[r][3] is False
        runs = rd.chunk_runs(n_chunks, world)
        assert sum(c for _, c in runs) == n_chunks and sorted(created) == sorted((r, f, c) for r, (f, c) in enumerate(runs) if c)
        assert len([x for x in checked]) == 2 * len([1 for _, c in runs if c])      # every run checked once per split_verify call
        for r in range(world):
            p, c, ok, ok_bad = out[r]
            assert (p == want_p).all() and (c == want_c).all() and ok is True and ok_bad is False
