"""The path's one exchange step: finished self-play samples go to rank 0.

Self-play shards by game slot with no communication during search; once per drain each rank
contributes its new PlayHistory rows.  Row counts are exchanged with one all_gather of a scalar,
then the rows travel with a padded gather (rank 0 receives, everyone else sends) — on GPUs this is
RCCL over xGMI (`backend="nccl"`), in the CPU tests it is gloo.  Messages are small
(Connect4: 712 B/row), so no ring tuning is involved.
"""
import torch
import torch.distributed as dist


def gather_rows_to_rank0(parts, rank, world, group=None):
    """parts: list of tensors with the same leading dimension n_local (may be 0).
    Returns on rank 0 a list of concatenated tensors (rank order), elsewhere None."""
    dev = parts[0].device
    n_local = parts[0].shape[0]
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=dev), group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n_max == 0:
        return [p[:0].clone() for p in parts] if rank == 0 else None
    out = []
    for p in parts:
        padded = torch.zeros((n_max,) + tuple(p.shape[1:]), dtype=p.dtype, device=dev)
        padded[:n_local] = p
        if rank == 0:
            bufs = [torch.empty_like(padded) for _ in range(world)]
            dist.gather(padded, bufs, dst=0, group=group)
            out.append(torch.cat([b[:c] for b, c in zip(bufs, counts)], 0))
        else:
            dist.gather(padded, None, dst=0, group=group)
    return out if rank == 0 else None


def gather_history_to_rank0(pm, dev, rank, world):
    """Gathers the rows `pm` has finished since the previous call. Returns total rows on rank 0."""
    c, v, p, meta = pm.take_history_device(dev)      # the unread rows of the engine's ring, released once copied out
    parts = [c, v, p]
    res = gather_rows_to_rank0(parts, rank, world)
    if rank == 0:
        pm._gathered = getattr(pm, "_gathered", [])
        pm._gathered.append(res)
        return int(res[0].shape[0])
    return 0
