"""Randomized parity soak for the tile kernel's variant with f32 slab coordinates and the always-add histogram
("rdf_tile_zf": diagonal cells, with slab culling or -- near-cubic boxes -- without): long and near-cubic diagonal boxes, uniform / layered / lattice atom
distributions, one to four species (rare ones included: their centre sub-tiles span wide slab ranges and fall back to
the integer differences step by step), constant or per-frame cells -- against the C oracle and against the plain tile
kernel (AMOF_RDF_NOZF=1).  Run by hand on a GPU box: `python tests/tools/soak_gpu_zf.py SECONDS` (not collected by pytest)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from amof_amd import _hip
from amof_amd.frames import PackedTrajectory
from oracle import clib
from tests import helpers as H

ctx = _hip.get_context(0)
t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 240)
os.environ["AMOF_RDF_NOCELL"] = "1"
os.environ["AMOF_RDF_NORANGE"] = "1"
bad = n = 0
paths = {}
seed = 90000
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 5))
    N = int(rng.choice([700, 1500, 2600, 4000]))
    kind = ["uniform", "layers", "lattice"][seed % 3]
    F = int(rng.integers(1, 4))
    if kind == "lattice":
        a = float(rng.choice([1.0, 1.5, 2.0]))
        nx, ny = int(rng.integers(4, 8)), int(rng.integers(4, 8))
        nz = max(int(rng.integers(2, 4) * max(nx, ny) + 1), N // (nx * ny)) if seed % 2 else int(rng.integers(4, 9))
        g = np.array([[x, y, z] for x in range(nx) for y in range(ny) for z in range(nz)], dtype=float) * a
        N = len(g)
        L = np.array([nx, ny, nz]) * a
        pos = np.stack([g + rng.choice([0.0, 0.25, -3.0 * L[2]]) for _ in range(F)])
    else:
        rho = rng.uniform(0.03, 0.08)
        stretch = rng.uniform(2.2, 4.5) if seed % 2 else rng.uniform(0.9, 1.3)     # long boxes (culling) / near-cubic (none)
        Lx = (N / rho / stretch) ** (1 / 3) * rng.uniform(0.8, 1.25)
        Ly = (N / rho / stretch) ** (1 / 3) * rng.uniform(0.8, 1.25)
        L = np.array([Lx, Ly, N / rho / (Lx * Ly)])
        L = L[rng.permutation(3)]                      # the long axis is not always z
        if kind == "layers":
            ax = int(np.argmax(L))
            centres = rng.uniform(0, L[ax], int(rng.integers(2, 6)))
            frac = rng.uniform(0.5, 0.95)
            nl = int(N * frac)
            p = rng.uniform(0, 1, (F, N, 3)) * L
            p[:, :nl, ax] = rng.choice(centres, nl) + rng.normal(0, rng.uniform(0.05, 0.6), (F, nl))
            pos = p
        else:
            pos = rng.uniform(0, 1, (F, N, 3)) * L
        pos = pos + rng.integers(-1, 2, (F, N, 3)) * L     # unwrapped input
    kinds = [1, 6, 7, 30][:S]
    w = np.array([6, 6, 4, rng.choice([1, 0.1])][:S], dtype=float)
    numbers = rng.choice(kinds, size=N, p=w / w.sum())
    numbers[:S] = kinds
    cells = np.diag(L)
    if F > 1 and seed % 4 == 0:
        cells = np.array([np.diag(L * (1 + 0.01 * rng.normal(size=3))) for _ in range(F)])
    packed = PackedTrajectory(pos, cells, numbers)
    kinds_s, sp = H.species_of(packed.numbers)
    lmin = float(np.min(packed.cell_lengths()))
    rmax = lmin / 2 * float(rng.choice([1.0, 0.9999, 0.93, 0.6]))
    nb = int(rng.choice([50, 700, 2310, 5000, 12000])) if kind != "lattice" else int(round(rmax / rng.choice([0.01, 0.05, 0.25, 1.0])))
    nb = max(nb, 4)
    h, _, _ = ctx.rdf_accumulate(packed, rmax, nb)
    k = ctx.last_path()
    paths[k] = paths.get(k, 0) + 1
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds_s), rmax, nb, cell_list=True)
    n += 1
    if not np.array_equal(h, ref):
        bad += 1
        print("RDF MISMATCH seed", seed, kind, N, F, rmax, nb, k, int(np.abs(h.astype(np.int64) - ref).sum()), flush=True)
    if k == "rdf_tile_zf":
        os.environ["AMOF_RDF_NOZF"] = "1"
        h2, _, _ = ctx.rdf_accumulate(packed, rmax, nb)
        os.environ.pop("AMOF_RDF_NOZF")
        n += 1
        if not np.array_equal(h2, ref):
            bad += 1
            print("PLAIN MISMATCH seed", seed, flush=True)
    if n % 50 < 2:
        print("progress: %d comparisons, %d mismatches (seed %d, %s, N=%d)" % (n, bad, seed, kind, N), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches" % (n, bad))
