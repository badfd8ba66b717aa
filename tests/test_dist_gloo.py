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
