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


def all_verified(ok_local, device, group=None):
    """MIN all-reduce of verify bits: True iff every rank's local proofs verified."""
    t = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=device)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item())
