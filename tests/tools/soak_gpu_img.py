"""Randomized parity soak of the image-aware tile kernel: sheared (and changing) cells at the reference's default
cutoff -- half the shortest cell LENGTH, i.e. beyond half a cell HEIGHT -- and a little around it, several tiles per
species, against the C oracle's cell-list variant.  Run by hand on a GPU box: `python tests/tools/soak_gpu_img.py
SECONDS` (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from amof_amd import _hip
from amof_amd.frames import PackedTrajectory
from oracle import clib
from tests import helpers as H

ctx = _hip.get_context(0)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 180)
bad = n = 0
paths = {}
seed = 30000
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 5))
    N = int(rng.choice([300, 1200, 2600, 4200]))
    rho = rng.uniform(0.03, 0.08)
    shape = rng.choice([1.0, 1.0, 1.3, 2.0], 3)
    L = shape * (N / rho / shape.prod()) ** (1 / 3)
    eps = float(rng.choice([1e-9, 1e-4, 0.01, 0.03, 0.08, 0.2]))
    cell = np.diag(L) + np.tril(rng.uniform(-1, 1, (3, 3)) * eps * L[:, None], k=-1)
    if seed % 2:
        cell = cell.T.copy() if seed % 4 == 1 else cell          # upper- or lower-triangular
    if seed % 5 == 0:
        # exactly hexagonal (two equal vectors at 120 or 60 degrees, the third perpendicular, any order of the rows): the
        # exact-half x wrap of near mode 4 (csrc/rdf.hip tri_q_twin<HALF>)
        a_h = float(L[0])
        sgn = -0.5 if seed % 10 == 0 else 0.5
        hexc = np.array([[a_h, 0.0, 0.0], [sgn * a_h, np.sqrt(3.0) / 2.0 * a_h, 0.0], [0.0, 0.0, float(L[2])]])
        perm = [(0, 1, 2), (2, 0, 1), (1, 2, 0), (1, 0, 2)][(seed // 5) % 4]
        cell = hexc[list(perm)][:, list(perm)] if seed % 15 else hexc[list(perm)]
    F = int(rng.integers(1, 4))
    kinds = [1, 6, 7, 30][:S]
    numbers = rng.choice(kinds, size=N)
    numbers[:S] = kinds
    pos = (rng.uniform(0, 1, (F, N, 3)) + rng.integers(-1, 2, (F, N, 3))) @ cell
    cells = np.array([cell * (1 + 0.005 * rng.normal()) for _ in range(F)]) if seed % 3 == 0 and F > 1 else cell
    packed = PackedTrajectory(pos, cells, numbers)
    kinds_s, sp = H.species_of(packed.numbers)
    rmax = float(np.min(packed.cell_lengths()) / 2 * rng.choice([1.0, 1.0, 0.999, 1.01, 0.98]))
    nb = int(rng.choice([97, 800, 2310]))
    h, _, _ = ctx.rdf_accumulate(packed, rmax, nb)
    k = ctx.last_path()
    paths[k] = paths.get(k, 0) + 1
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds_s), rmax, nb, cell_list=True)
    n += 1
    if not np.array_equal(h, ref):
        bad += 1
        print("IMG RDF MISMATCH seed", seed, N, F, S, eps, rmax, nb, k, flush=True)
    if n % 50 == 0:
        print("progress: %d comparisons, %d mismatches, paths %s" % (n, bad, dict(sorted(paths.items()))), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches" % (n, bad))
