"""Randomized parity soak for the whole-frame-in-LDS tier of CN / BAD (cn_frame_kernel, lists_frame_kernel,
bad_rows_kernel; whole frames and z-slabs): random gases, jittered lattices (pairs near the cutoff) and ZIF-4 walks, orthorhombic / sheared /
breathing cells, sparse cutoff matrices, same-species pairs, 'X' triples, per-atom counts, BadByCn keys, uniform and
ragged bin edges -- against the C oracle.  Run by hand on a GPU box: `python tests/tools/soak_gpu_frame.py SECONDS`
(not collected by pytest)."""
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


seed = 91000
while time.time() < t_end:
    seed += 1
    rng = np.random.default_rng(seed)
    mode = seed % 4
    # every third case through the z-slab kernels whatever its size (the others: slabs only where a pair is too big for one
    # workgroup -- N = 14000 below)
    if seed % 3 == 1:
        os.environ["AMOF_NBR_SLABS"] = "1"
    else:
        os.environ.pop("AMOF_NBR_SLABS", None)
    if mode == 0:       # ZIF-4 supercell walk (the bench's system, smaller)
        rep = tuple(int(x) for x in rng.integers(1, 4, 3))
        packed = H.random_walk(H.replicate(H.zif4_frame(), rep), int(rng.integers(1, 4)), float(rng.choice([0.02, 0.1, 0.3])),
                               seed, ortho=bool(seed % 8 < 4), cell_jitter=0.01 if seed % 16 < 4 else 0.0)
    else:
        S = int(rng.integers(1, 5))
        N = int(rng.choice([300, 900, 2500, 6000, 9000, 14000]))
        rho = rng.uniform(0.03, 0.09)
        shape = rng.choice([1.0, 1.0, 2.0, 3.0], 3)
        L = shape * (N / rho / shape.prod()) ** (1 / 3)
        cell = np.diag(L)
        if mode == 2:
            cell = cell + np.tril(rng.uniform(-0.25, 0.25, (3, 3)) * L[:, None], k=-1)
        F = int(rng.integers(1, 4))
        kinds = [1, 6, 7, 30][:S]
        numbers = rng.choice(kinds, size=N, p=np.array([6, 6, 4, 1][:S]) / sum([6, 6, 4, 1][:S]))
        numbers[:S] = kinds
        if mode == 3:   # jittered simple cubic lattice: many pairs close to the cutoffs below
            m = int(np.ceil(N ** (1 / 3)))
            g = np.array([[x, y, z] for x in range(m) for y in range(m) for z in range(m)], dtype=float)[:N] / m
            frac = g[None] + rng.normal(scale=rng.choice([0.0, 1e-9, 1e-3]), size=(F, N, 3))
        else:
            frac = rng.uniform(0, 1, (F, N, 3))
        pos = (frac + rng.integers(-1, 2, (F, N, 3))) @ cell
        cells = np.array([cell * (1 + 0.01 * rng.normal()) for _ in range(F)]) if seed % 3 == 0 and F > 1 else cell
        packed = PackedTrajectory(pos, cells, numbers)
    kinds_s, sp = H.species_of(packed.numbers)
    S = len(kinds_s)
    N = packed.n_atoms
    hmin = min(1.0 / np.linalg.norm(np.linalg.inv(c), axis=0).max() for c in packed.cell)
    top = min(3.2, hmin / 3.05)
    rcm = rng.uniform(0.9, max(1.0, top), (S, S))
    if mode == 3:       # lattice constant and its sqrt(2), exactly and one ulp either side
        a0 = float(np.linalg.norm(packed.cell[0][0])) / int(np.ceil(N ** (1 / 3)))
        rcm[:] = min(top, float(rng.choice([a0, np.nextafter(a0, 9.0), np.nextafter(a0, 0.0), a0 * np.sqrt(2.0)])))
    rcm = np.maximum(rcm, rcm.T)
    rcm[rng.uniform(size=(S, S)) < 0.3] = 0.0        # sparse: some pairs carry no cutoff
    rcm = np.minimum(rcm, rcm.T)
    sets = [(a, b) for a in range(S) for b in range(S) if rcm[a, b] > 0]
    if sets:
        per_atom = bool(seed % 2)
        got = ctx.cn_count(packed, rcm, sets, per_atom=per_atom)
        _note()
        ref = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=per_atom)
        n += 1
        ok = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) if per_atom else np.array_equal(got, ref)
        if not ok:
            bad += 1
            print("CN MISMATCH seed", seed, N, flush=True)
    ref_cn = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, [(a, b) for a in range(S) for b in range(S)])
    if ref_cn.max() / max(1, N) < 12:
        edges = np.arange(int(180 // 1.0) + 2) * 1.0 if seed % 5 else np.sort(np.concatenate([[0.0, 180.0], rng.uniform(0, 180, 40)]))
        triples = [(a, b) for a in range(-1, S) for b in range(-1, S)]
        triples = [triples[k] for k in rng.permutation(len(triples))[:7]]
        try:
            if seed % 7 == 0:
                cn_max = int(rng.integers(2, 7))
                hr = clib.bad_hist_by_cn(packed.pos, packed.cell, sp, S, rcm, triples, edges, cn_max)
                hg = ctx.bad_hist_by_cn(packed, rcm, triples, edges, cn_max=cn_max)
                same = np.array_equal(hr[0], hg[0]) and np.array_equal(hr[1], hg[1])
            else:
                hr = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
                hg = ctx.bad_hist(packed, rcm, triples, edges)
                same = np.array_equal(hr[0], hg[0]) and np.array_equal(hr[1], hg[1])
            _note()
            n += 1
            if not same:
                bad += 1
                print("BAD MISMATCH seed", seed, N, triples, flush=True)
        except (ZeroDivisionError, _hip.AmofError) as exc:
            print("skip BAD seed", seed, type(exc).__name__, flush=True)
    if n % 20 < 2:
        print("progress: %d comparisons, %d mismatches (seed %d, N=%d) %s" % (n, bad, seed, N, dict(sorted(paths.items()))), flush=True)
print("kernel families exercised:", dict(sorted(paths.items())))
print("SOAK DONE: %d comparisons, %d mismatches" % (n, bad))
