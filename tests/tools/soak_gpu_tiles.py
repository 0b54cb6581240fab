"""Randomized parity soak for the multi-tile machinery (slab culling, LDS-DMA double buffering, sub-tile loop,
range kernel, 1-D cell lists of CN/BAD): medium-size random systems against the C oracle's cell-list variant.
Run by hand on a GPU box: `python tests/tools/soak_gpu_tiles.py SECONDS` (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from amof_amd import _hip
from amof_amd.frames import PackedTrajectory
from oracle import clib
from tests import helpers as H

ctx = _hip.get_context(0)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 240)
bad = n = 0
paths = {}


def _note():
    k = ctx.last_path()
    paths[k] = paths.get(k, 0) + 1


seed = 5000
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 5))
    N = int(rng.choice([600, 1500, 3000, 5000]))
    rho = rng.uniform(0.03, 0.08)
    shape = rng.choice([1.0, 1.0, 2.0, 4.0], 3)
    L = shape * (N / rho / shape.prod()) ** (1 / 3)
    cell = np.diag(L)
    if seed % 2:
        cell = cell + np.tril(rng.uniform(-0.2, 0.2, (3, 3)) * L[:, None], k=-1)
    F = int(rng.integers(1, 4))
    kinds = [1, 6, 7, 30][:S]
    numbers = rng.choice(kinds, size=N, p=np.array([6, 6, 4, 1][:S]) / sum([6, 6, 4, 1][:S]))
    numbers[:S] = kinds
    pos = (rng.uniform(0, 1, (F, N, 3)) + rng.integers(-1, 2, (F, N, 3))) @ cell
    cells = np.array([cell * (1 + 0.01 * rng.normal()) for _ in range(F)]) if seed % 3 == 0 and F > 1 else cell
    packed = PackedTrajectory(pos, cells, numbers)
    kinds_s, sp = H.species_of(packed.numbers)
    hmin = min(1.0 / np.linalg.norm(np.linalg.inv(c), axis=0).max() for c in packed.cell)
    rmax = float(hmin / 2 * rng.choice([1.0, 0.999, 0.7, 0.35, 0.2]))
    rmax = min(rmax, 14.0)
    nb = int(rng.choice([50, 700, 2310, 5000]))
    h, _, _ = ctx.rdf_accumulate(packed, rmax, nb)
    _note()
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds_s), rmax, nb, cell_list=True)
    n += 1
    if not np.array_equal(h, ref):
        bad += 1
        print("RDF MISMATCH seed", seed, N, F, rmax, nb, flush=True)
    rcm = rng.uniform(1.0, 3.0, (len(kinds_s), len(kinds_s)))
    rcm = np.maximum(rcm, rcm.T)
    sets = [(a, b) for a in range(len(kinds_s)) for b in range(len(kinds_s))]
    s1 = ctx.cn_count(packed, rcm, sets)
    _note()
    s2 = clib.cn_counts(packed.pos, packed.cell, sp, len(kinds_s), rcm, sets)
    n += 1
    if not np.array_equal(s1, s2):
        bad += 1
        print("CN MISMATCH seed", seed, flush=True)
    if s2.max() / max(1, N) < 20:
        edges = np.arange(int(180 // 1.0) + 2) * 1.0
        triples = [(a, b) for a in range(len(kinds_s)) for b in range(-1, len(kinds_s))][:6]
        try:
            hr, ar = clib.bad_hist(packed.pos, packed.cell, sp, len(kinds_s), rcm, triples, edges)
            hg, ag = ctx.bad_hist(packed, rcm, triples, edges)
            _note()
            n += 1
            if not (np.array_equal(hr, hg) and np.array_equal(ar, ag)):
                bad += 1
                print("BAD MISMATCH seed", seed, flush=True)
        except (ZeroDivisionError, _hip.AmofError) as exc:
            print("skip BAD seed", seed, type(exc).__name__, flush=True)
    print("progress: %d comparisons, %d mismatches (seed %d, N=%d)" % (n, bad, seed, N), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches" % (n, bad))
