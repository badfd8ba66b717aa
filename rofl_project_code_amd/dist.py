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
