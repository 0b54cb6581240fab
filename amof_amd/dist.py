"""Multi-GPU plumbing: one process per GPU, torch.distributed (RCCL or gloo).

Frames are independent for RDF / BAD / CN, atoms are independent for MSD
(SURVEY 8e).  The only data-path collective is the final sum of the integer
histograms (or of the S x W float64 MSD partial sums): one all-reduce.
"""

import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:
        return None
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


def world(group=None):
    """(rank, world_size) of the default (or given) process group; (0, 1)
    when torch.distributed is not initialised."""
    d = _dist()
    if d is None:
        return 0, 1
    return d.get_rank(group), d.get_world_size(group)


def shard_range(n, rank, world_size):
    """Contiguous shard ``[lo, hi)`` of ``n`` units: unit u goes to rank
    floor(u * G / n)."""
    lo = (n * rank + world_size - 1) // world_size
    hi = (n * (rank + 1) + world_size - 1) // world_size
    return lo, min(hi, n)


def all_reduce_sum(x, group=None):
    """Sum ``x`` over ranks and return it (same type as given).

    numpy arrays travel through a CPU tensor (gloo) or, when the backend is
    nccl (= RCCL on ROCm), through a CUDA tensor on the current device.
    uint64 counts are reinterpreted as int64 (sums stay far below 2^63)."""
    d = _dist()
    if d is None:
        return x
    import torch
    if isinstance(x, np.ndarray):
        was_u64 = x.dtype == np.uint64
        t = torch.from_numpy(np.ascontiguousarray(x.view(np.int64) if was_u64 else x).copy())
        backend = d.get_backend(group)
        if backend == "nccl":
            t = t.cuda()
        d.all_reduce(t, op=d.ReduceOp.SUM, group=group)
        out = t.cpu().numpy()
        return out.view(np.uint64) if was_u64 else out
    d.all_reduce(x, op=d.ReduceOp.SUM, group=group)
    return x


def all_reduce_min(value, group=None):
    d = _dist()
    if d is None:
        return value
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    if d.get_backend(group) == "nccl":
        t = t.cuda()
    d.all_reduce(t, op=d.ReduceOp.MIN, group=group)
    return float(t.cpu()[0])


def all_gather_rows(x, group=None):
    """Concatenate per-rank row blocks (numpy 2-D, possibly different row
    counts) in rank order."""
    d = _dist()
    if d is None:
        return x
    gathered = [None] * d.get_world_size(group)
    d.all_gather_object(gathered, x, group=group)
    return np.concatenate(gathered, axis=0)
