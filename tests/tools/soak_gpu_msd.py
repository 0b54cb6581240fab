"""Randomized MSD parity soak (comb / generic / long-trajectory kernels, unwrap on/off, changing cells) against the
numpy restatement.  Run by hand on a GPU box: `python tests/tools/soak_gpu_msd.py SECONDS` (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from amof_amd import _hip
from amof_amd.frames import PackedTrajectory
from oracle import numpy_oracle as no

ctx = _hip.get_context(0)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 120)
bad = n = 0
paths = {}


def _note():
    k = ctx.last_path()
    paths[k] = paths.get(k, 0) + 1


seed = 9000
worst = 0.0
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    F = int(rng.choice([2, 3, 11, 64, 257, 1000, 2500, 20500]))
    N = int(rng.choice([1, 5, 33, 130])) if F < 20000 else int(rng.choice([1, 4]))
    S = int(rng.integers(1, 4))
    numbers = rng.choice([1, 8, 30][:S], size=N)
    cell = np.diag(rng.uniform(7.0, 15.0, 3))
    if seed % 2:
        cell = cell + np.tril(rng.uniform(-2.0, 2.0, (3, 3)), k=-1)
    cells = np.array([cell * (1 + 0.003 * rng.normal()) for _ in range(F)]) if seed % 3 == 0 else cell
    walk = np.cumsum(rng.normal(scale=rng.choice([0.02, 0.3]), size=(F, N, 3)), axis=0) + rng.uniform(0, 10, (1, N, 3))
    c0 = cells[0] if cells.ndim == 3 else cells
    s = walk @ np.linalg.inv(c0)
    pos = (s - np.floor(s)) @ c0 if seed % 5 else walk          # wrapped into the first cell, or left unwrapped
    packed = PackedTrajectory(pos, cells, numbers)
    kind = seed % 4
    if kind == 0:
        d = int(rng.integers(1, max(2, F // 3)))
        window = np.arange(0, max(1, F // 2), d)[:32]
    elif kind == 1:
        window = np.unique(rng.integers(0, F, size=int(rng.integers(1, 40))))
    elif kind == 2:
        window = np.arange(0, F // 2 + 1)[:300]
    else:
        window = np.array([0])
    window = window.astype(np.int32)
    unwrap = bool(seed % 7 == 0)
    sumsq, kinds = ctx.msd_window(packed, window, unwrap=unwrap)
    _note()
    elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window, unwrap=unwrap)
    for e, r in zip(elements, ref):
        got = sumsq[kinds.index(int(e))] / (packed.numbers == e).sum() / (F - window)
        err = np.max(np.abs(got - r) / np.maximum(np.abs(r), 1e-12)) if len(r) else 0.0
        worst = max(worst, float(err))
        n += 1
        if not np.allclose(got, r, rtol=1e-9, atol=1e-12):
            bad += 1
            print("MSD MISMATCH seed", seed, F, N, len(window), unwrap, float(err), flush=True)
    if N >= 2:            # atom-sharded calls (what the ranks of a multi-GPU run make): the two ranges add up to the whole
        cut = int(rng.integers(1, N))
        parts = ctx.msd_window(packed, window, unwrap=unwrap, atom_range=(0, cut))[0] + \
            ctx.msd_window(packed, window, unwrap=unwrap, atom_range=(cut, N))[0]
        n += 1
        if not np.allclose(parts, sumsq, rtol=1e-11, atol=1e-13):
            bad += 1
            print("MSD ATOM-RANGE MISMATCH seed", seed, F, N, cut, unwrap, flush=True)
    if n % 300 < 3:
        print("progress: %d comparisons, %d mismatches, worst relative error %.2e" % (n, bad, worst), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches, worst relative error %.2e" % (n, bad, worst))
