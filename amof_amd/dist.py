"""Multi-GPU plumbing: one process per GPU, torch.distributed (RCCL or gloo).

Frames are independent for RDF / BAD / CN, atoms are independent for MSD
(SURVEY 8e).  The only data-path collective is the final sum of the integer
histograms (or of the S x W float64 MSD partial sums): one all-reduce.
"""

import os

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


def merging(world_size):
    """True when results must be merged over the ranks.  ``AMOF_DIST_FORCE_MERGE=1`` also runs the whole sharding +
    collective path in a one-rank group -- the only way to drive the real RCCL calls on a single-GPU test box
    (RCCL refuses two ranks on one device)."""
    return world_size > 1 or (os.environ.get("AMOF_DIST_FORCE_MERGE") == "1" and _dist() is not None)


def shard_range(n, rank, world_size):
    """Contiguous shard ``[lo, hi)`` of ``n`` units: unit u goes to rank
    floor(u * G / n)."""
    lo = (n * rank + world_size - 1) // world_size
    hi = (n * (rank + 1) + world_size - 1) // world_size
    return lo, min(hi, n)


def device_collectives(group=None):
    """True when the group's backend moves CUDA tensors (nccl = RCCL): results can then stay in HBM from the
    kernel that produced them through the all-reduce (no D2H / H2D hop in between)."""
    d = _dist()
    return d is not None and d.get_backend(group) == "nccl"


def all_reduce_sum(x, group=None, device=None):
    """Sum ``x`` over ranks and return it (same type as given).

    A torch tensor is reduced in place where it lives (a CUDA tensor over RCCL: the device-resident merge of the
    histograms written by the "_dev" entry points).

    numpy arrays travel through a CPU tensor (gloo) or, when the backend is
    nccl (= RCCL on ROCm), through a CUDA tensor on the current device.
    uint64 counts are reinterpreted as int64 (sums stay far below 2^63).
    ``device``: GPU index for the staging tensor under nccl (default: torch's current device -- pass the
    context's device when the caller may not have called ``torch.cuda.set_device``)."""
    d = _dist()
    if d is None:
        return x
    import torch
    if isinstance(x, np.ndarray):
        was_u64 = x.dtype == np.uint64
        t = torch.from_numpy(np.ascontiguousarray(x.view(np.int64) if was_u64 else x).copy())
        backend = d.get_backend(group)
        if backend == "nccl":
            t = t.cuda(device)
        d.all_reduce(t, op=d.ReduceOp.SUM, group=group)
        out = t.cpu().numpy()
        return out.view(np.uint64) if was_u64 else out
    d.all_reduce(x, op=d.ReduceOp.SUM, group=group)
    return x


def all_reduce_min(value, group=None, device=None):
    d = _dist()
    if d is None:
        return value
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    if d.get_backend(group) == "nccl":
        t = t.cuda(device)
    d.all_reduce(t, op=d.ReduceOp.MIN, group=group)
    return float(t.cpu()[0])


def all_gather_rows(x, group=None, device=None):
    """Concatenate per-rank row blocks (numpy 2-D, possibly different row
    counts) in rank order."""
    d = _dist()
    if d is None:
        return x
    world = d.get_world_size(group)
    if d.get_backend(group) == "nccl" and x.dtype.kind in "iu" and x.dtype.itemsize == 8:
        # two tensor all-gathers over RCCL (row counts, then the blocks padded to the largest count) instead of
        # pickled objects
        import torch
        n = torch.tensor([x.shape[0]], dtype=torch.int64).cuda(device)
        counts = torch.empty(world, dtype=torch.int64, device=n.device)
        d.all_gather_into_tensor(counts, n, group=group)
        counts = counts.cpu().numpy()
        rows = int(counts.max())
        cols = int(np.prod(x.shape[1:]))
        mine = torch.zeros((rows, cols), dtype=torch.int64)
        mine[:x.shape[0]] = torch.from_numpy(np.ascontiguousarray(x).view(np.int64).reshape(x.shape[0], cols))
        mine = mine.cuda(device)
        full = torch.empty((world * rows, cols), dtype=torch.int64, device=mine.device)
        d.all_gather_into_tensor(full, mine, group=group)
        full = full.cpu().numpy().reshape(world, rows, cols)
        out = np.concatenate([full[r, :counts[r]] for r in range(world)], axis=0)
        return out.view(x.dtype).reshape((out.shape[0],) + x.shape[1:])
    gathered = [None] * world
    d.all_gather_object(gathered, x, group=group)
    return np.concatenate(gathered, axis=0)
