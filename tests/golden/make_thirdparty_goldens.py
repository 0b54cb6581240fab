#!/usr/bin/env python
"""Third-party parity, one command away: golden vectors from the REAL ase + asap3 (+ the reference on top of them).

    python tests/golden/make_thirdparty_goldens.py [/path/to/amof-checkout]

Needs ``ase`` and ``asap3`` importable (the reference pins ase 3.20.1 / asap3 3.12.8, requirements.txt:1-2).  Neither
is installed in the build container of this repository and there is no network there, so the files this script
writes -- ``tests/golden/thirdparty_*.npz`` -- do not exist yet: RDF / CN / BAD parity is "unpinned" until a
maintainer with the two packages runs it once and commits the outputs.  ``tests/test_thirdparty_goldens.py`` picks
the files up when they exist (oracle on the CPU, HIP product on the GPU; skipped otherwise) and reports which of the
named assumptions A1 - A8 of DESIGN.md 5.1 each comparison settles.

Nothing is stubbed here.  What is written is data only: the inputs (positions, cells, atomic numbers, parameters)
and the outputs of
  * the third-party primitives the hot path sits on, each in isolation
      - asap3.analysis.rdf.RadialDistributionFunction: update() over the frames, get_rdf() total and per element pair
        (assumptions A1 shell volume, A2 partial normalisation, A3 bin of a pair, A4 periodic images, A8 volume)
      - ase.neighborlist.neighbor_list('ij', atoms, {(Z1, Z2): rc}) (A5: strict <, symmetric dict cutoffs, images)
      - ase.Atoms.get_angles(indices, mic=True) (A6)
      - ase.geometry.geometry.wrap_positions(d, cell, center=(0, 0, 0)) (A7)
      - ase.data.atomic_masses (A10)
  * and, when an amof checkout is given (or /root/reference exists), the reference's own classes run UNMODIFIED on
    lists of ase.Atoms: Rdf, CoordinationNumber, Bad, WindowMsd (unwrap off / on).
"""

import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "golden")

try:
    import ase
    import ase.data
    from ase import Atoms
    from ase.geometry.geometry import wrap_positions
    from ase.neighborlist import neighbor_list
    from asap3.analysis.rdf import RadialDistributionFunction
    import asap3
except ImportError as exc:          # the normal outcome in the build container
    sys.stderr.write("make_thirdparty_goldens: %s -- install ase (3.20.1) and asap3 (3.12.8) and run again; "
                     "nothing was written\n" % exc)
    sys.exit(2)

sys.path.insert(0, ROOT)
from tests.helpers import read_extxyz          # noqa: E402  (plain-text reader of the fixture; no ase involved)

VERSIONS = json.dumps({"ase": ase.__version__, "asap3": getattr(asap3, "__version__", "?"), "numpy": np.__version__})


def walk(base, F, sigma, rng, cell_jitter=0.0):
    """list of ase.Atoms: base + cumulative Gaussian steps, wrapped into the (possibly breathing) cell"""
    frames, pos = [], base.get_positions()
    cell0 = np.array(base.get_cell())
    for _ in range(F):
        cell = cell0 * (1.0 + cell_jitter * rng.normal()) if cell_jitter else cell0
        s = np.linalg.solve(cell.T, pos.T).T
        frames.append(Atoms(numbers=base.get_atomic_numbers(), positions=(s - np.floor(s)) @ cell, cell=cell, pbc=True))
        pos = pos + rng.normal(scale=sigma, size=pos.shape)
    return frames


def pack(frames):
    return dict(pos=np.array([f.get_positions() for f in frames]), cell=np.array([np.array(f.get_cell()) for f in frames]),
                numbers=np.array(frames[0].get_atomic_numbers()), versions=VERSIONS)


def main():
    rng = np.random.default_rng(20261004)
    fx = read_extxyz(os.path.join(OUT, "ZIF-4.xyz"), 0)
    zif = Atoms(numbers=fx.numbers, positions=fx.positions, cell=fx.cell, pbc=True)
    small = Atoms(numbers=[30] * 4 + [7] * 10 + [6] * 8, positions=rng.uniform(0, 1, (22, 3)) @ np.diag([6.5, 7.0, 7.5]),
                  cell=np.array([[6.5, 0, 0], [1.2, 7.0, 0], [-0.8, 1.5, 7.5]]), pbc=True)

    # ---- asap3 RDF in isolation: fixture at the default cutoff, and a small sheared cell whose cutoff needs images
    for tag, base, rmax, nbins, F, jitter in (("zif4", zif, 7.7021, 770, 3, 0.0), ("zif4_npt", zif, 6.0, 120, 3, 0.01),
                                              ("small_images", small, 5.9, 59, 2, 0.0)):
        frames = walk(base, F, 0.04, rng, jitter)
        obj = None
        for a in frames:
            if obj is None:
                obj = RadialDistributionFunction(a, rmax, nbins)
            obj.atoms = a
            obj.update()
        kinds = sorted(set(int(z) for z in base.get_atomic_numbers()))
        out = pack(frames)
        out.update(rmax=rmax, nbins=nbins, kinds=np.array(kinds), total=np.array(obj.get_rdf(groups=0)))
        for a in kinds:
            for b in kinds:
                out["partial_%d_%d" % (a, b)] = np.array(obj.get_rdf(elements=(a, b), groups=0))
        np.savez_compressed(os.path.join(OUT, "thirdparty_asap3_rdf_%s.npz" % tag), **out)

    # ---- ase neighbour list, angles, wrap
    frames = walk(zif, 2, 0.05, rng) + walk(small, 2, 0.05, rng)
    for k, a in enumerate(frames):
        cut = {(30, 7): 2.5, (6, 7): 1.6, (6, 6): 1.7} if k < 2 else {(30, 7): 3.4, (7, 7): 3.1, (6, 7): 3.0}
        i, j = neighbor_list('ij', a, cut)
        trip = []
        nl = [[] for _ in range(len(a))]
        for x, y in zip(i, j):
            nl[x].append(y)
        for c, lst in enumerate(nl):
            trip += [[lst[u], c, lst[v]] for u in range(len(lst)) for v in range(u + 1, len(lst))]
        trip = np.array(trip[:4000], dtype=int).reshape(-1, 3)
        ang = a.get_angles(trip, mic=True) if len(trip) else np.zeros(0)
        np.savez_compressed(os.path.join(OUT, "thirdparty_ase_neighbours_%d.npz" % k), pos=a.get_positions()[None],
                            cell=np.array(a.get_cell())[None], numbers=np.array(a.get_atomic_numbers()),
                            cutoffs=json.dumps({"%d-%d" % kk: v for kk, v in cut.items()}), i=np.array(i), j=np.array(j),
                            triples=trip, angles=np.array(ang), versions=VERSIONS)
    d = rng.normal(scale=4.0, size=(500, 3))
    d[:20] = (np.array([0.5, -0.5, 0.5 - 1e-7])[None] + rng.integers(-2, 3, (20, 3))) @ np.array(small.get_cell())   # on the faces
    np.savez_compressed(os.path.join(OUT, "thirdparty_ase_wrap.npz"), d=d, cell=np.array(small.get_cell()),
                        wrapped=wrap_positions(d, np.array(small.get_cell()), center=(0, 0, 0)),
                        masses=np.array(ase.data.atomic_masses[:100]), versions=VERSIONS)

    # ---- the reference's classes, unmodified, on real ase.Atoms
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if not os.path.isdir(os.path.join(ref, "amof")):
        print("no amof checkout at %s: class-level goldens skipped" % ref)
        return
    sys.path.insert(0, ref)
    import amof.rdf, amof.cn, amof.bad, amof.msd          # noqa: E401,E402

    def df(d):
        return {"columns": np.array(list(d.columns)), "values": np.array(d.values, dtype=np.float64)}

    frames = walk(zif, 4, 0.03, rng)
    out = pack(frames)
    np.savez_compressed(os.path.join(OUT, "thirdparty_e2e_rdf.npz"), dr=0.01, rmax="half_cell",
                        **out, **df(amof.rdf.Rdf.from_trajectory([f.copy() for f in frames]).data))
    cut = {'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}
    np.savez_compressed(os.path.join(OUT, "thirdparty_e2e_cn.npz"), cutoffs=json.dumps(cut), **out,
                        **df(amof.cn.CoordinationNumber.from_trajectory([f.copy() for f in frames], cut).data))
    cut = {'Zn-N': 2.5, 'C-N': 1.6}
    np.savez_compressed(os.path.join(OUT, "thirdparty_e2e_bad.npz"), cutoffs=json.dumps(cut), dtheta=0.5, **out,
                        **df(amof.bad.Bad.from_trajectory([f.copy() for f in frames], cut, dtheta=0.5).data))
    frames = walk(small, 24, 0.35, rng)
    out = pack(frames)
    for unwrap in (False, True):
        m = amof.msd.WindowMsd.from_trajectory([f.copy() for f in frames], delta_time=2, timestep=1, unwrap=unwrap)
        np.savez_compressed(os.path.join(OUT, "thirdparty_e2e_msd_%s.npz" % ("unwrap" if unwrap else "raw")),
                            delta_time=2, timestep=1, unwrap=unwrap, **out, **df(m.data))
    for fn in sorted(os.listdir(OUT)):
        if fn.startswith("thirdparty_"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
