"""Multi-GPU sharding of the proof path: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" in the CPU tests).  The path shards by independent units (clients); the only exchange
steps are an all-gather of proof bytes + commitments and a MIN all-reduce of verify bits.  The shared generator
table is deterministic (SHAKE256 chain) and is rebuilt locally per rank (6 ms on an MI355X) instead of being
broadcast: a 50 MB broadcast would cost more than regenerating it."""
import numpy as np
import torch
import torch.distributed as dist


def shard_clients(n_clients, rank, world):
    """Round-robin client -> rank assignment (clients are independent: server.rs:656-687)."""
    return list(range(rank, n_clients, world))


def gather_bytes(local_u8, device, group=None):
    """All-gather equally sized uint8 payloads (proof bytes or compressed commitments) from every rank.
    Returns a list (one numpy array per rank)."""
    t = torch.from_numpy(np.ascontiguousarray(local_u8).reshape(-1)).to(device)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [t.cpu().numpy()]
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)
    return [o.cpu().numpy() for o in outs]


def all_verified(ok_local, device, group=None, force_collective=False):
    """MIN all-reduce of verify bits: True iff every rank's local proofs verified.  force_collective: run the collective even in
    a group of one (the RCCL smoke test on a single GPU)."""
    t = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=device)
    if dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective):
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())


# ---------------------------------------------------------------------------------------------------------------- ONE client over the ranks
# "cfg 2/3 at > 1 GPU -> chunks over ranks".  The reference proves / verifies a client's chunks independently on its rayon pool
# (range_proof_vec/mod.rs:54-78, 168-181); here rank r takes the r-th contiguous run of the chunks (the same deal as the library's one-process
# split, rofl_set_option("devices")), one all-gather assembles the client's proof set and commitment vector on every rank, and the verdict is
# the MIN over the ranks' runs.  No collective inside a proof.
def chunk_runs(n_chunks, world):
    """[(first, count)] per rank: contiguous runs, as even as they come; ranks beyond the chunk count get (0, 0) and only join the collectives."""
    nd = min(world, n_chunks)
    runs = [(k * n_chunks // nd, (k + 1) * n_chunks // nd - k * n_chunks // nd) for k in range(nd)]
    return runs + [(0, 0)] * (world - nd)


def split_create(comm, rank, world, n_chunks, m, d, proof_len, create_run):
    """create_run(first, count) -> (proofs u8[count, proof_len], commitments u8[k, 32] of the run's own elements) -- e.g.
    range_proof_vec.create_rangeproof_chunks.  Returns (proofs u8[n_chunks, proof_len], commitments u8[d, 32]) assembled from ONE
    all-gather, identical on every rank and byte for byte what the unsplit create returns."""
    runs = chunk_runs(n_chunks, world)
    cmax = max(c for _, c in runs)
    first, count = runs[rank]
    buf_p = np.zeros((cmax, proof_len), np.uint8); buf_c = np.zeros((cmax * m, 32), np.uint8)
    if count:
        pr, cm = create_run(first, count)
        buf_p[:count] = pr; buf_c[:cm.shape[0]] = cm
    _, per_rank = comm.exchange_round([buf_p, buf_c], True)
    proofs = np.zeros((n_chunks, proof_len), np.uint8); commits = np.zeros((d, 32), np.uint8)
    for r, (f, c) in enumerate(runs):
        if not c:
            continue
        proofs[f:f + c] = per_rank[r][0].reshape(cmax, proof_len)[:c]
        lo, hi = min(d, f * m), min(d, (f + c) * m)
        commits[lo:hi] = per_rank[r][1].reshape(cmax * m, 32)[:hi - lo]
    return proofs, commits


def split_verify(comm, rank, world, proofs, commits, m, verify_run, shift=1):
    """verify_run(first, proofs_run, commits_run) -> bool for a run of the client's proofs -- e.g. range_proof_vec.verify_rangeproof_chunks.
    Rank r checks the run that rank (r + shift) created (so the bytes it reads really crossed the collective); the verdict is the MIN over
    the ranks (range_proof_vec/mod.rs:182-190 ANDs the chunks' bits)."""
    n_chunks, d = proofs.shape[0], commits.shape[0]
    first, count = chunk_runs(n_chunks, world)[(rank + shift) % world]
    ok = True
    if count:
        lo, hi = min(d, first * m), min(d, (first + count) * m)
        ok = bool(verify_run(first, proofs[first:first + count], commits[lo:hi]))
    return comm.all_verified(ok)


def elem_runs(d, world, min_run=2048):
    """[(first, count)] per rank for the ELEMENTS of a Sigma-proof vector (one proof per element, independent of the others:
    rand_proof_vec/mod.rs:45-58): contiguous runs of at least min_run elements -- the deal of the library's one-process split; ranks
    beyond d // min_run get (0, 0)."""
    nd = max(1, min(world, d // min_run))
    return [(k * d // nd, (k + 1) * d // nd - k * d // nd) for k in range(nd)] + [(0, 0)] * (world - nd)


def split_create_elems(comm, rank, world, d, proof_len, commit_len, create_run, min_run=2048):
    """create_run(first, count) -> (proofs u8[count, proof_len], commitments u8[count, commit_len]) -- e.g. api.create_sigmaproof_vec_range.
    One all-gather; returns the whole vector's (proofs u8[d, proof_len], commitments u8[d, commit_len]) on every rank, byte for byte the
    unsplit create_*_vec result."""
    runs = elem_runs(d, world, min_run)
    cmax = max(c for _, c in runs)
    first, count = runs[rank]
    buf_p = np.zeros((cmax, proof_len), np.uint8); buf_c = np.zeros((cmax, commit_len), np.uint8)
    if count:
        pr, cm = create_run(first, count)
        buf_p[:count] = pr; buf_c[:count] = cm
    _, per_rank = comm.exchange_round([buf_p, buf_c], True)
    proofs = np.zeros((d, proof_len), np.uint8); commits = np.zeros((d, commit_len), np.uint8)
    for r, (f, c) in enumerate(runs):
        if c:
            proofs[f:f + c] = per_rank[r][0].reshape(cmax, proof_len)[:c]
            commits[f:f + c] = per_rank[r][1].reshape(cmax, commit_len)[:c]
    return proofs, commits


def split_verify_elems(comm, rank, world, proofs, commits, verify_run, shift=1, min_run=2048):
    """verify_run(proofs_run, commits_run) -> bool -- the ordinary verify_*_vec call on a run's sub-arrays.  Rank r checks the run of rank
    r + shift; the vector's verdict is the MIN over the ranks."""
    first, count = elem_runs(proofs.shape[0], world, min_run)[(rank + shift) % world]
    ok = bool(verify_run(proofs[first:first + count], commits[first:first + count])) if count else True
    return comm.all_verified(ok)


_round_bufs = {}


def exchange_round(payloads, ok_local, device, group=None, force_collective=False):
    """One collective per round instead of three: every rank contributes [verify bit | payload_0 | payload_1 ...] (equal sizes on all
    ranks), the pieces are all-gathered into ONE preallocated device buffer and come back in one copy.  Returns (ok_all, per_rank) with
    ok_all = MIN over the ranks' verify bits (server.rs:474-484: one failing client fails the round) and per_rank[r] = the list of rank r's
    payloads as numpy arrays.  force_collective: go through the backend's all-gather even in a group of one (tests/test_gpu_dist.py
    forms a world-size-1 `nccl` group on the GPU so that the first multi-GPU run is not also RCCL's first run)."""
    parts = [np.ascontiguousarray(p, dtype=np.uint8).reshape(-1) for p in payloads]
    sizes = [p.size for p in parts]
    n = 1 + sum(sizes)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    key = (n, world, str(device))
    if key not in _round_bufs:
        _round_bufs[key] = (torch.empty(n, dtype=torch.uint8, device=device), torch.empty(n * world, dtype=torch.uint8, device=device),
                            torch.empty(n, dtype=torch.uint8).pin_memory() if str(device) != "cpu" else torch.empty(n, dtype=torch.uint8))
    loc, allb, stage = _round_bufs[key]
    host = stage.numpy()
    host[0] = 1 if ok_local else 0
    o = 1
    for p in parts:
        host[o:o + p.size] = p; o += p.size
    loc.copy_(stage, non_blocking=True)
    if world == 1 and not (force_collective and dist.is_initialized()):
        allb.copy_(loc)
    elif dist.get_backend(group) == "gloo":
        outs = list(allb.view(world, n).unbind(0))
        dist.all_gather(outs, loc, group=group)
    else:
        try:
            dist.all_gather_into_tensor(allb, loc, group=group)
        except (RuntimeError, AttributeError):      # a backend without the flat form: per-rank views of the same buffer
            dist.all_gather(list(allb.view(world, n).unbind(0)), loc, group=group)
    # (a CPU device -- the gloo tests -- would hand out views of the cached buffer, which the next round overwrites: copy there;
    #  from the GPU, .cpu() is a fresh tensor already)
    got = (allb.cpu() if allb.is_cuda else allb.clone()).numpy().reshape(world, n)
    ok_all = bool(got[:, 0].min() == 1)
    per_rank = []
    for r in range(world):
        o = 1; lst = []
        for sz in sizes:
            lst.append(got[r, o:o + sz]); o += sz
        per_rank.append(lst)
    return ok_all, per_rank


# ---------------------------------------------------------------------------------------------------------------- communicators
# The exchange steps of a round behind one small interface, so that bench.py (and any other multi-rank host) is written once:
#   LibComm   -- the library's own RCCL communicator (rofl_comm_*, include/rofl_zk.h): the collectives run on the HIP runtime the proofs run
#                on and need no torch; the default of a real multi-GPU run
#   TorchComm -- torch.distributed ("nccl" = the RCCL torch bundles, or "gloo" for the CPU tests and for several ranks on ONE GPU, which
#                RCCL refuses); the fallback when the library's communicator cannot be formed
#   LocalComm -- a group of one
class LocalComm:
    backend = None; world = 1; rank = 0

    def exchange_round(self, payloads, ok_local):
        return bool(ok_local), [[np.ascontiguousarray(p, dtype=np.uint8).reshape(-1) for p in payloads]]

    def all_verified(self, ok_local): return bool(ok_local)
    def barrier(self): pass
    def reduce(self, values, op="sum"): return np.asarray(values, dtype=np.float64).reshape(-1).copy()
    def close(self): pass


class TorchComm:
    def __init__(self, device, group=None, force_collective=False):
        self.device, self.group, self.force = device, group, force_collective
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def exchange_round(self, payloads, ok_local): return exchange_round(payloads, ok_local, self.device, self.group, self.force)
    def all_verified(self, ok_local): return all_verified(ok_local, self.device, self.group, self.force)
    def barrier(self): dist.barrier(group=self.group)

    def reduce(self, values, op="sum"):
        t = torch.tensor(np.asarray(values, dtype=np.float64).reshape(-1), dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op], group=self.group)
        return t.cpu().numpy()

    def close(self): pass


class LibComm:
    """rofl_comm_* (RCCL loaded by librofl_zk.so itself).  `uid` = the 128 bytes rank 0 drew with LibComm.unique_id(), handed to the
    other ranks by the caller's control channel (bench.py: the launcher's store through a gloo group)."""

    def __init__(self, uid, rank, world):
        from . import api
        self._c = api.comm
        self._c.init(uid, rank, world)
        self.world, self.rank = world, rank
        i = self._c.info()
        self.backend = "rccl %d via librofl_zk rofl_comm_* (%s)" % (i["rccl_version"], i["library"])

    @staticmethod
    def unique_id():
        from . import api
        return api.comm.unique_id()

    def exchange_round(self, payloads, ok_local):
        parts = [np.ascontiguousarray(p, dtype=np.uint8).reshape(-1) for p in payloads]
        sizes = [p.size for p in parts]
        buf = np.empty(1 + sum(sizes), dtype=np.uint8)
        buf[0] = 1 if ok_local else 0
        o = 1
        for p in parts:
            buf[o:o + p.size] = p; o += p.size
        got = self._c.allgather(buf, self.world)
        per_rank = []
        for r in range(self.world):
            o = 1; lst = []
            for sz in sizes:
                lst.append(got[r, o:o + sz]); o += sz
            per_rank.append(lst)
        return bool(got[:, 0].min() == 1), per_rank

    def all_verified(self, ok_local): return bool(self._c.allreduce([1.0 if ok_local else 0.0], "min")[0] == 1.0)
    def barrier(self): self._c.barrier()
    def reduce(self, values, op="sum"): return self._c.allreduce(values, op)
    def close(self): self._c.destroy()


def make_comm(rank, world, device, control_group=None, prefer_lib=True, torch_backend="nccl", log=None, force_lib=False):
    """The communicator of a multi-rank run.  world == 1: LocalComm.  Otherwise the library's RCCL communicator when every rank can form
    it (the unique id travels over `control_group`, a gloo group; success is agreed by a MIN over the ranks -- a rank that cannot load
    librccl, or two ranks on one GPU, must not leave the others waiting inside ncclCommInitRank forever: RCCL itself reports duplicate
    devices to every rank), else torch.distributed with `torch_backend`."""
    if world == 1:      # force_lib: the RCCL path in a group of one (the single-GPU rehearsal of a multi-GPU run)
        return LibComm(LibComm.unique_id(), 0, 1) if force_lib else LocalComm()
    cpu = torch.device("cpu")
    if prefer_lib:      # every rank must be able to load librccl before anyone enters ncclCommInitRank (a rank that never arrives would hang the rest)
        try:
            from . import api
            info = api.comm.info()      # loads librccl (ROFL_RCCL_LIB); raises when it cannot be loaded
            if info["world"] != 0:
                raise RuntimeError("this process already holds a communicator")
            api.set_device(api.get_device())      # the device context rofl_comm_init binds to comes up HERE (a rank whose GPU cannot be initialised must say so before anybody enters ncclCommInitRank)
            can = 1
        except Exception as e:      # noqa: BLE001
            can = 0
            if log: log("librccl not usable on rank %d: %r" % (rank, e))
        t = torch.tensor([can], dtype=torch.int32); dist.all_reduce(t, op=dist.ReduceOp.MIN, group=control_group)
        prefer_lib = int(t.item()) == 1
    if prefer_lib:
        uid = torch.zeros(128, dtype=torch.uint8)
        ok = 1
        if rank == 0:
            try:
                uid = torch.from_numpy(np.frombuffer(LibComm.unique_id(), dtype=np.uint8).copy())
            except Exception as e:      # noqa: BLE001
                ok = 0
                if log: log("rofl_comm_unique_id failed: %r" % (e,))
        flag = torch.tensor([ok], dtype=torch.int32); dist.broadcast(flag, src=0, group=control_group)
        if int(flag.item()) == 1:
            dist.broadcast(uid, src=0, group=control_group)
            c = None
            # ncclCommInitRank is collective and has no timeout of its own: a rank that dies between the agreement above and its own call would
            # leave the others inside it forever.  A watchdog ends this process (non-zero) when the call has not returned in time; the launcher
            # then stops the remaining ranks (bench.py launch_ranks, torchrun).
            import os, threading
            limit = float(os.environ.get("ROFL_COMM_INIT_TIMEOUT_S", "180"))
            def _give_up():
                if log: log("rofl_comm_init did not return within %.0f s on rank %d: leaving" % (limit, rank))
                os._exit(17)
            dog = threading.Timer(limit, _give_up); dog.daemon = True; dog.start()
            try:
                c = LibComm(uid.numpy().tobytes(), rank, world)
            except Exception as e:      # noqa: BLE001
                if log: log("rofl_comm_init failed on rank %d: %r" % (rank, e))
            finally:
                dog.cancel()
            good = torch.tensor([1 if c else 0], dtype=torch.int32); dist.all_reduce(good, op=dist.ReduceOp.MIN, group=control_group)
            if int(good.item()) == 1:
                return c
            if c:
                c.close()
    if torch_backend == "gloo":
        return TorchComm(cpu, control_group)
    g = dist.new_group(backend=torch_backend)
    return TorchComm(device, g)
