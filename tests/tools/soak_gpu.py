"""Randomized parity soak, run by hand on a GPU box (`python tests/tools/soak_gpu.py SECONDS`; not collected by pytest): many more seeds than tests/test_gpu_random.py,
plus adversarial lattice cases where distances sit on bin edges / cutoffs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from amof_amd import _hip
from amof_amd.frames import PackedTrajectory
from oracle import clib
from tests import helpers as H
from tests.test_gpu_random import _case

ctx = _hip.get_context(0)
t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 240
bad = 0
paths = {}


def _note():
    k = ctx.last_path()
    paths[k] = paths.get(k, 0) + 1


n = 0
seed = 1000
while time.time() < t_end:
    seed += 1
    rng, packed = _case(seed)
    kinds, sp = H.species_of(packed.numbers)
    hmin = min(1.0 / np.linalg.norm(np.linalg.inv(c), axis=0).max() for c in packed.cell)
    rmax = float(rng.uniform(0.2, 1.3) * hmin / 2 if seed % 4 else np.min(packed.cell_lengths()) / 2)
    nb = int(rng.choice([1, 7, 100, 999, 2310, 9000]))
    h, vol, _ = ctx.rdf_accumulate(packed, rmax, nb)
    ref, vref = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb)
    n += 1
    if not np.array_equal(h, ref):
        bad += 1
        print("RDF MISMATCH seed", seed, packed.n_atoms, rmax, nb, int(np.abs(h.astype(np.int64) - ref.astype(np.int64)).sum()), flush=True)
    # lattice case: integer grid in an integer (possibly sheared) cell, bins aligned with the grid
    g = np.arange(16) * 1.0
    pts = np.array([[x, y, z] for x in g for y in g for z in g])
    pick = rng.choice(len(pts), size=int(rng.choice([50, 300, 900])), replace=False)
    cell = np.diag([16.0, 16.0, 16.0 * int(rng.choice([1, 2]))])
    if seed % 2:
        cell = cell + np.array([[0, 0, 0], [float(rng.integers(-3, 4)), 0, 0], [float(rng.integers(-3, 4)), float(rng.integers(-3, 4)), 0]])
    numbers = rng.choice([1, 8], size=len(pick))
    shift = rng.choice([0.0, 0.5, 0.125])
    lat = PackedTrajectory((pts[pick] + shift)[None], cell, numbers)
    kinds, sp = H.species_of(lat.numbers)
    hmin = 1.0 / np.linalg.norm(np.linalg.inv(cell), axis=0).max()
    rmax = float(rng.choice([hmin / 2, 5.0, 7.0, 3.0]))
    rmax = min(rmax, hmin / 2) if seed % 3 else rmax
    nb = int(rng.choice([int(rmax * k) for k in (1, 2, 4, 10, 100)] + [333]))
    nb = max(nb, 1)
    h, _, _ = ctx.rdf_accumulate(lat, rmax, nb)
    _note()
    ref, _ = clib.rdf_hist(lat.pos, lat.cell, sp, len(kinds), rmax, nb)
    n += 1
    if not np.array_equal(h, ref):
        bad += 1
        print("LATTICE RDF MISMATCH seed", seed, len(pick), rmax, nb, cell.tolist(), flush=True)
    rc = float(rng.choice([1.0, np.sqrt(2.0), np.sqrt(3.0), 2.0, np.nextafter(2.0, 3.0), 2.5]))
    rcm = np.full((len(kinds), len(kinds)), rc)
    sets = [(a, b) for a in range(len(kinds)) for b in range(len(kinds))]
    if rc < hmin / 2:
        s1 = ctx.cn_count(lat, rcm, sets)
        _note()
        s2 = clib.cn_counts(lat.pos, lat.cell, sp, len(kinds), rcm, sets)
        n += 1
        if not np.array_equal(s1, s2):
            bad += 1
            print("LATTICE CN MISMATCH seed", seed, rc, flush=True)
    if n % 200 < 3:
        print("progress: %d comparisons, %d mismatches" % (n, bad), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches" % (n, bad))
