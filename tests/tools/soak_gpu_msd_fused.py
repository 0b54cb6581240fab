"""Randomized soak of the fused window-MSD form and its atom-sharded halves (csrc/msd.hip msd_seg_kernel / msd_fused_kernel,
amof_msd_shard_begin / _finish) against the numpy restatement and the transposed forms: diagonal cells (constant and per
frame), evenly spaced windows, ragged segment counts, open axes, unwrapped input, gases (the flag -> fallback).  Run by hand
on a GPU box: `python tests/tools/soak_gpu_msd_fused.py SECONDS` (not collected by pytest)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from amof_amd import _hip
from amof_amd.frames import PackedTrajectory
from oracle import numpy_oracle as no

ctx = _hip.get_context(0)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 120)
bad = n = 0
paths = {}
seed = 31000
worst = 0.0
trace = open(os.environ["AMOF_SOAK_TRACE"], "w") if os.environ.get("AMOF_SOAK_TRACE") else None
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    d = int(rng.choice([16, 17, 31, 50, 64, 100, 129, 200, 256]))
    nq = int(rng.integers(2, 106))
    F = int(min(6000, max(64, nq * d - int(rng.integers(0, d)))))
    W = int(rng.integers(2, 33))
    N = int(rng.choice([1, 3, 8, 9, 40, 65, 130, 300]))
    S = int(rng.integers(1, 5))
    numbers = rng.choice([1, 6, 8, 30][:S], size=N)
    cell = np.diag(rng.uniform(7.0, 15.0, 3))
    cells = np.array([cell * (1 + 0.003 * rng.normal()) for _ in range(F)]) if seed % 3 == 0 else cell
    sigma = float(rng.choice([0.01, 0.05, 0.2]))
    walk = np.cumsum(rng.normal(scale=sigma, size=(F, N, 3)), axis=0) + rng.uniform(0, 10, (1, N, 3))
    if seed % 11 == 0:
        walk = rng.uniform(0, 1, (F, N, 3)) * np.diag(cell)            # a gas: the flag must send it to the transposed forms
    c0 = np.diagonal(cells, axis1=-2, axis2=-1)
    dd = c0[:, None, :] if c0.ndim == 2 else c0
    pos = walk - np.floor(walk / dd) * dd if seed % 5 else walk        # wrapped into the cell, or left unwrapped
    pbc = (True, True, seed % 13 != 0)
    packed = PackedTrajectory(pos, cells, numbers, pbc=pbc)
    window = np.array([w * d for w in range(W) if w * d < F], dtype=np.int32)
    if len(window) < 2:
        continue
    if trace:                                                           # (a GPU fault ends the process: what was running?)
        trace.seek(0)
        trace.write("seed %d F %d d %d W %d N %d cells %d wrapped %d pbc %s\n" % (seed, F, d, len(window), N, np.ndim(cells) - 1, seed % 5 != 0, pbc))
        trace.flush()
    got, kinds = ctx.msd_window(packed, window)
    p = ctx.last_path()
    paths[p] = paths.get(p, 0) + 1
    elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window, pbc=pbc)
    for e, r in zip(elements, ref):
        g = got[kinds.index(int(e))] / (packed.numbers == e).sum() / (F - window)
        err = float(np.max(np.where(np.abs(r) > 1e-9, np.abs(g - r) / np.maximum(np.abs(r), 1e-9), 0.0)))   # (lag 0 is exactly 0)
        worst = max(worst, err)
        n += 1
        if not np.allclose(g, r, rtol=1e-9, atol=1e-12):
            bad += 1
            print("MSD MISMATCH seed", seed, F, d, W, N, p, err, flush=True)
    os.environ["AMOF_MSD_NOFUSED"] = "1"
    old, _ = ctx.msd_window(packed, window)
    os.environ.pop("AMOF_MSD_NOFUSED")
    n += 1
    if not np.allclose(got, old, rtol=1e-10, atol=1e-9):
        bad += 1
        print("FUSED vs TRANSPOSED MISMATCH seed", seed, F, d, W, N, p, flush=True)
    if N >= 2:
        # the sharded halves, as two ranks would call them (one context, one share after the other)
        dev = packed.to_device(0)
        cut = int(rng.integers(1, N))
        shares = ((0, cut), (cut, N))
        try:
            tabs = []
            for r_ in shares:
                t = torch.empty((F, 3), dtype=torch.float64, device="cuda:0")
                ctx.msd_shard_begin(dev, window, r_, t)
                tabs.append(t)
            total = tabs[0] + tabs[1]
            out = torch.zeros((len(kinds), len(window)), dtype=torch.float64, device="cuda:0")
            for r_ in shares:
                ctx.msd_shard_begin(dev, window, r_, torch.empty((F, 3), dtype=torch.float64, device="cuda:0"))
                ctx.msd_shard_finish(dev, window, r_, total, out)
            n += 1
            if not np.allclose(out.cpu().numpy(), got, rtol=1e-10, atol=1e-9):
                bad += 1
                print("SHARDED MISMATCH seed", seed, F, d, W, N, cut, flush=True)
        except _hip.Unsupported:
            pass
    if n % 400 < 4:
        print("progress: %d comparisons, %d mismatches, worst relative error %.2e, paths %s" % (n, bad, worst, paths), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches, worst relative error %.2e" % (n, bad, worst))
