"""Randomized parity soak of the 3-D cell-list RDF kernel (forced), against the C oracle's cell-list variant.
Run by hand on a GPU box: `python tests/tools/soak_gpu_cell.py SECONDS` (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["AMOF_RDF_FORCE_CELL"] = "1"
from amof_amd import _hip
from amof_amd.frames import PackedTrajectory
from oracle import clib
from tests import helpers as H

ctx = _hip.get_context(0)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 180)
bad = n = 0
paths = {}


def _note():
    k = ctx.last_path()
    paths[k] = paths.get(k, 0) + 1


seed = 20000
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 7))
    # (AMOF_SOAK_BIG=1: frames of 2048 atoms and more only -- the one-kernel cell sort serves those)
    N = int(rng.choice([2048, 2049, 5000, 12000, 30000] if os.environ.get("AMOF_SOAK_BIG") else [64, 65, 200, 777, 2500, 6000]))
    rho = rng.uniform(0.02, 0.1)
    shape = rng.choice([1.0, 1.0, 1.5, 3.0], 3)
    L = shape * (N / rho / shape.prod()) ** (1 / 3)
    cell = np.diag(L)
    if seed % 2:
        cell = cell + np.tril(rng.uniform(-0.3, 0.3, (3, 3)) * L[:, None], k=-1)
    F = int(rng.integers(1, 4))
    kinds = [1, 6, 7, 8, 14, 30][:S]
    numbers = rng.choice(kinds, size=N)
    numbers[:S] = kinds
    if seed % 5 == 0:                           # clustered: very uneven cell populations
        centres = rng.uniform(0, 1, (8, 3))
        frac = (centres[rng.integers(0, 8, (F, N))] + rng.normal(scale=0.03, size=(F, N, 3))) % 1.0
    else:
        frac = rng.uniform(0, 1, (F, N, 3))
    pos = (frac + rng.integers(-1, 2, (F, N, 3))) @ cell
    cells = np.array([cell * (1 + 0.01 * rng.normal()) for _ in range(F)]) if seed % 3 == 0 and F > 1 else cell
    packed = PackedTrajectory(pos, cells, numbers)
    kinds_s, sp = H.species_of(packed.numbers)
    hmin = min(1.0 / np.linalg.norm(np.linalg.inv(c), axis=0).max() for c in packed.cell)
    rmax = float(hmin * rng.uniform(0.02, 0.39))
    nb = int(rng.choice([1, 13, 400, 999, 2000]))
    h, _, _ = ctx.rdf_accumulate(packed, rmax, nb)
    _note()
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds_s), rmax, nb, cell_list=True)
    n += 1
    if not np.array_equal(h, ref):
        bad += 1
        print("CELL RDF MISMATCH seed", seed, N, F, S, rmax, nb, flush=True)
    if n % 100 == 0:
        print("progress: %d comparisons, %d mismatches" % (n, bad), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches" % (n, bad))
