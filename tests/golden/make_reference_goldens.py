#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE's own code in this container.

Run from the repo root, in the development container only (it needs
/root/reference, which does not travel to the GPU box):

    python tests/golden/make_reference_goldens.py

What is real and what is stubbed
--------------------------------
The reference (coudertlab/amof v1.1.0) imports ase / asap3 / xarray, which are
not installed here (ModuleNotFoundError, an ordinary Python error).  This script
registers stand-in modules for them and then imports the reference's
``amof.msd``, ``amof.trajectory``, ``amof.cn``, ``amof.bad`` and ``amof.rdf``
from /root/reference and runs them unmodified.

* ``reference_msd_of_m.npz`` / ``reference_construct_step.json``: pure
  reference-owned numpy (amof/msd.py:185-205, amof/trajectory.py:244-283); no
  stub participates in the numbers.  These PIN the oracle.
* ``reference_e2e_*.npz``: the reference's class-level drivers
  (WindowMsd.from_trajectory, CoordinationNumber.from_trajectory,
  Bad.from_trajectory, Rdf.from_trajectory) executed on in-repo Frame objects.
  Reference-owned logic (window arithmetic, COM removal, per-element split,
  column naming and order, np.histogram density, 'A-X' sums, formula-weighted
  'X' column) is real; the third-party calls underneath are served by the
  oracle's restatements of their published behaviour ([3P-memory]):
  ase wrap_positions / neighbor_list / get_angles and asap3's RDF
  accumulate + normalise.  They pin the host-side plumbing, not the 3P maths.

Only data (inputs + outputs) is written; no reference source is copied.
"""

import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from amof_amd import data as eldata                      # element tables (public facts)
from amof_amd.frames import Frame                        # Atoms look-alike
from tests.helpers import read_extxyz
from oracle import numpy_oracle as no                    # restated 3P behaviour


# ------------------------------------------------------------------ stubs --
def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _neighbor_list(quantities, atoms, cutoff):
    assert quantities == 'ij' and isinstance(cutoff, dict)
    Z = np.array(atoms.get_atomic_numbers())
    kinds = sorted(set(Z.tolist()))
    idx = {z: k for k, z in enumerate(kinds)}
    S = len(kinds)
    rcm = np.zeros((S, S))
    for (z1, z2), c in cutoff.items():
        if z1 in idx and z2 in idx:
            rcm[idx[z1], idx[z2]] = c
            rcm[idx[z2], idx[z1]] = c
    species = np.array([idx[z] for z in Z])
    nl = no.neighbour_lists(atoms.get_positions(), np.array(atoms.get_cell()), species, S, rcm, atoms.pbc)
    ii, jj = [], []
    for i, lst in enumerate(nl):
        for (j, _) in lst:
            ii.append(i); jj.append(j)
    return np.array(ii, dtype=int), np.array(jj, dtype=int)


class GoldenFrame(Frame):
    """Frame + ase.Atoms.get_angles(mic=True) ([3P-memory] of ase 3.20.1)."""

    def get_angles(self, indices, mic=False):
        indices = np.array(indices)
        v12 = self.positions[indices[:, 0]] - self.positions[indices[:, 1]]
        v32 = self.positions[indices[:, 2]] - self.positions[indices[:, 1]]
        if mic:
            v12, v32 = self._mic(v12), self._mic(v32)
        n1 = np.linalg.norm(v12, axis=1)[:, None]
        n2 = np.linalg.norm(v32, axis=1)[:, None]
        if (n1 <= 0).any() or (n2 <= 0).any():
            raise ZeroDivisionError('Undefined angle')
        v12 = v12 / n1
        v32 = v32 / n2
        return 180 / np.pi * np.arccos(np.einsum('ij,ij->i', v12, v32).clip(-1.0, 1.0))

    def _mic(self, v):
        s = np.linalg.solve(self.cell.T, v.T).T
        s -= np.round(s)
        best = s @ self.cell
        bestn = (best ** 2).sum(axis=1)
        for n in np.ndindex(3, 3, 3):
            t = (np.array(n) - 1) @ self.cell
            cand = s @ self.cell + t
            cn = (cand ** 2).sum(axis=1)
            better = cn < bestn - 1e-12
            best[better] = cand[better]; bestn[better] = cn[better]
        return best

    def copy(self):
        return GoldenFrame(self.numbers, self.positions, self.cell, self.pbc, self._masses)


class _StubRDF(object):
    """asap3.analysis.rdf.RadialDistributionFunction ([3P-memory])."""

    shell = "exact"

    def __init__(self, atoms, rMax, nBins, groups=None, interval=1, restart=None, verbose=0):
        self.atoms = atoms
        self.natoms = len(atoms)
        self.rMax = rMax * 1.0
        self.nBins = nBins
        self.dr = self.rMax / nBins
        self.volume = 0.0
        self.countRDF = 0
        Z = np.array(atoms.get_atomic_numbers())
        self.kinds = sorted(set(Z.tolist()))
        self.idx = {z: k for k, z in enumerate(self.kinds)}
        self.hist = np.zeros((len(self.kinds), len(self.kinds), nBins), dtype=np.uint64)
        self.atomcounts = np.zeros(len(self.kinds), dtype=np.int64)

    def update(self):
        a = self.atoms
        assert len(a) == self.natoms
        Z = np.array(a.get_atomic_numbers())
        sp = np.array([self.idx[z] for z in Z])
        self.hist += no.rdf_hist(a.get_positions(), np.array(a.get_cell()), sp, len(self.kinds),
                                 self.rMax, self.nBins, a.pbc)
        for k in range(len(self.kinds)):
            self.atomcounts[k] += (sp == k).sum()
        self.countRDF += 1
        self.volume += a.get_volume()

    def _normalize(self, h, ncount):
        vol = self.volume / self.countRDF
        r = (np.arange(self.nBins) + 0.5) * self.dr
        if _StubRDF.shell == "exact":          # the default since round 4
            shell = 4 * np.pi * self.dr * (r * r + self.dr * self.dr / 12.0)
        else:
            shell = 4 * np.pi * r * r * self.dr
        return h * (vol / (self.natoms * ncount)) / shell

    def get_rdf(self, groups=None, elements=None):
        if elements is None:
            return self._normalize(self.hist.sum(axis=(0, 1)).astype(np.float64), self.countRDF * self.natoms)
        a, b = elements
        return self._normalize(self.hist[self.idx[a], self.idx[b]].astype(np.float64),
                               self.atomcounts[self.idx[a]])


_mod("ase")
_mod("ase.data", chemical_symbols=eldata.chemical_symbols, atomic_numbers=eldata.atomic_numbers)
_mod("ase.atoms")
_mod("ase.neighborlist", neighbor_list=_neighbor_list)
_mod("ase.io")
_mod("ase.geometry")
_mod("ase.geometry.geometry", wrap_positions=no.wrap_positions)
_mod("asap3")
_mod("asap3.analysis")
_mod("asap3.analysis.rdf", RadialDistributionFunction=_StubRDF)
_mod("xarray")
for parent, child in [("ase", "data"), ("ase", "atoms"), ("ase", "neighborlist"), ("ase", "io"),
                      ("ase", "geometry"), ("asap3", "analysis")]:
    setattr(sys.modules[parent], child, sys.modules[parent + "." + child])
sys.modules["ase.geometry"].geometry = sys.modules["ase.geometry.geometry"]
sys.modules["asap3.analysis"].rdf = sys.modules["asap3.analysis.rdf"]

sys.path.insert(0, "/root/reference")
import amof.msd            # noqa: E402  (the reference)
import amof.trajectory     # noqa: E402
import amof.cn             # noqa: E402
import amof.bad            # noqa: E402
import amof.rdf            # noqa: E402


# ------------------------------------------------ 1. pure reference numpy --
def gold_msd_of_m():
    rng = np.random.default_rng(0)
    cases = {}
    k = 0
    for (F, n) in [(12, 5), (30, 7), (64, 3), (101, 16), (5, 1)]:
        base = [rng.normal(size=(n, 3)) for _ in range(F)]
        ms = sorted(set([0, 1, 2, 3, 4, F // 2, F - 2, F - 1]) & set(range(F)))
        vals = []
        for m in ms:
            delta = [b.copy() for b in base]
            vals.append(amof.msd.WindowMsd.compute_msd_of_m(delta, m))
        cases["delta_%d" % k] = np.array(base)
        cases["m_%d" % k] = np.array(ms)
        cases["msd_%d" % k] = np.array(vals)
        k += 1
    # ballistic: delta[k] = v for k >= 1
    F, n = 40, 4
    v = rng.normal(size=(n, 3))
    base = [rng.normal(size=(n, 3))] + [v.copy() for _ in range(F - 1)]
    ms = list(range(0, 20, 3))
    vals = [amof.msd.WindowMsd.compute_msd_of_m([b.copy() for b in base], m) for m in ms]
    cases["delta_%d" % k] = np.array(base); cases["m_%d" % k] = np.array(ms); cases["msd_%d" % k] = np.array(vals)
    # one shared delta list across successive m (the reference reuses and mutates it, msd.py:247)
    base = [rng.normal(size=(6, 3)) for _ in range(25)]
    shared = [b.copy() for b in base]
    ms = list(range(0, 12))
    vals = [amof.msd.WindowMsd.compute_msd_of_m(shared, m) for m in ms]
    cases["delta_shared"] = np.array(base); cases["m_shared"] = np.array(ms); cases["msd_shared"] = np.array(vals)
    cases["n_cases"] = np.array(k + 1)
    np.savez_compressed(os.path.join(OUT, "reference_msd_of_m.npz"), **cases)


def gold_construct_step():
    combos = [
        dict(delta_Step=1, first_frame=0, number_of_frames=10),
        dict(delta_Step=5, first_frame=100, number_of_frames=7),
        dict(delta_Step=2, first_frame=3, last_frame=20),
        dict(delta_Step=10, last_frame=1000, number_of_frames=5),
        dict(number_of_frames=6, first_frame=0, last_frame=50),
        dict(step=[3, 1, 4, 1, 5]),
        dict(step=("slice", 2, 30, 4)),
        dict(step=("slice", None, 5, None)),
        dict(delta_Step=3),
        dict(),
    ]
    out = []
    for kw in combos:
        call = dict(kw)
        if isinstance(call.get("step"), tuple):
            _, a, b, c = call["step"]
            call["step"] = slice(a, b, c)
        res = amof.trajectory.construct_step(**call)
        out.append({"kwargs": kw, "result": None if res is None else np.asarray(res).tolist(),
                    "dtype": None if res is None else str(np.asarray(res).dtype)})
    with open(os.path.join(OUT, "reference_construct_step.json"), "w") as fh:
        json.dump(out, fh, indent=1)


# ----------------------------------------------------- 2. class-level e2e --
def _df_to_npz(df):
    return {"columns": np.array(list(df.columns)), "values": np.array(df.values, dtype=np.float64)}


def random_walk_frames(base, F, sigma, rng, wrap=True, cell_jitter=0.0):
    frames = []
    pos = base.get_positions()
    cell0 = np.array(base.get_cell())
    for k in range(F):
        cell = cell0 * (1.0 + cell_jitter * rng.normal()) if cell_jitter else cell0
        p = pos.copy()
        if wrap:
            s = np.linalg.solve(cell.T, p.T).T
            s -= np.floor(s)
            p = s @ cell
        frames.append(GoldenFrame(base.numbers, p, cell, base.pbc))
        pos = pos + rng.normal(scale=sigma, size=pos.shape)
    return frames


def pack(frames):
    return (np.array([f.positions for f in frames]), np.array([f.cell for f in frames]),
            frames[0].numbers.copy())


def gold_e2e():
    rng = np.random.default_rng(20261003)
    zif = read_extxyz(os.path.join(OUT, "ZIF-4.xyz"), 0)
    zif = GoldenFrame(zif.numbers, zif.positions, zif.cell, zif.pbc)

    # --- MSD: small mixed-species system, wrapped walk, ortho + triclinic, unwrap on/off
    numbers = np.array([30] * 3 + [7] * 6 + [6] * 5 + [1] * 6)
    for tag, cell in [("ortho", np.diag([7.0, 8.0, 9.0])),
                      ("tri", np.array([[7.0, 0.0, 0.0], [1.5, 8.0, 0.0], [-1.0, 2.0, 9.0]]))]:
        base = GoldenFrame(numbers, rng.uniform(0, 1, size=(len(numbers), 3)) @ cell, cell)
        for unwrap in (False, True):
            frames = random_walk_frames(base, 24, 0.35, rng, wrap=True)
            pos, cells, Z = pack(frames)
            msd = amof.msd.WindowMsd.from_trajectory([f.copy() for f in frames], delta_time=2,
                                                     timestep=1, unwrap=unwrap)
            d = _df_to_npz(msd.data)
            np.savez_compressed(os.path.join(OUT, "reference_e2e_msd_%s_%s.npz" % (tag, "unwrap" if unwrap else "raw")),
                                pos=pos, cell=cells, numbers=Z, delta_time=2, timestep=1, unwrap=unwrap, **d)
    # the example script's call (examples/Compute structural properties.py:110-118): 11 rattled frames
    frames = [zif.copy()]
    for _ in range(10):
        f = frames[-1].copy()
        f.positions += rng.normal(scale=0.5, size=f.positions.shape)
        frames.append(f)
    pos, cells, Z = pack(frames)
    msd = amof.msd.WindowMsd.from_trajectory([f.copy() for f in frames], delta_time=1, timestep=1)
    np.savez_compressed(os.path.join(OUT, "reference_e2e_msd_zif4_rattle.npz"),
                        pos=pos, cell=cells, numbers=Z, delta_time=1, timestep=1, unwrap=False,
                        **_df_to_npz(msd.data))

    # --- CN / BAD / RDF on a short thermalised ZIF-4 fixture trajectory
    frames = random_walk_frames(zif, 4, 0.03, rng, wrap=True)
    pos, cells, Z = pack(frames)
    cn = amof.cn.CoordinationNumber.from_trajectory([f.copy() for f in frames], {'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3},
                                                    delta_Step=5, first_frame=10)
    np.savez_compressed(os.path.join(OUT, "reference_e2e_cn_zif4.npz"), pos=pos, cell=cells, numbers=Z,
                        cutoffs=json.dumps({'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}), delta_Step=5, first_frame=10,
                        **_df_to_npz(cn.data))
    for dtheta in (0.05, 0.5):
        bad = amof.bad.Bad.from_trajectory([f.copy() for f in frames], {'Zn-N': 2.5, 'C-N': 1.6}, dtheta=dtheta)
        np.savez_compressed(os.path.join(OUT, "reference_e2e_bad_zif4_dtheta%s.npz" % str(dtheta).replace('.', 'p')),
                            pos=pos, cell=cells, numbers=Z, cutoffs=json.dumps({'Zn-N': 2.5, 'C-N': 1.6}),
                            dtheta=dtheta, **_df_to_npz(bad.data))
    # the reference's Rdf class over the stand-in asap3 object, once per shell-volume convention (A1): `values` is the
    # default (exact shell), `values_midpoint` what AMOF_RDF_SHELL=midpoint must reproduce
    for name, kw in [("reference_e2e_rdf_zif4_default.npz", dict()),
                     ("reference_e2e_rdf_zif4_dr0p05_rmax6.npz", dict(dr=0.05, rmax=6.0))]:
        both = {}
        for shell in ("exact", "midpoint"):
            _StubRDF.shell = shell
            rdf = amof.rdf.Rdf.from_trajectory([f.copy() for f in frames], **kw)
            d = _df_to_npz(rdf.data)
            both["columns"] = d["columns"]
            both["values" if shell == "exact" else "values_midpoint"] = d["values"]
        _StubRDF.shell = "exact"
        np.savez_compressed(os.path.join(OUT, name), pos=pos, cell=cells, numbers=Z,
                            dr=kw.get("dr", 0.01), rmax=kw.get("rmax", "half_cell"), **both)


def gold_direct_msd():
    """DirectMsd is pure reference numpy on Atoms.get_cell()/get_positions(): no stub in the numbers."""
    rng = np.random.default_rng(7)
    numbers = np.array([30] * 2 + [7] * 5 + [6] * 4 + [1] * 6)
    cell = np.diag([6.0, 7.0, 8.0])
    base = GoldenFrame(numbers, rng.uniform(0, 1, size=(len(numbers), 3)) @ cell, cell)
    frames = random_walk_frames(base, 14, 0.6, rng, wrap=True, cell_jitter=0.004)
    pos, cells, Z = pack(frames)
    d = amof.msd.DirectMsd.from_trajectory([f.copy() for f in frames], delta_Step=2, first_frame=4)
    np.savez_compressed(os.path.join(OUT, "reference_e2e_directmsd_ortho.npz"), pos=pos, cell=cells, numbers=Z,
                        delta_Step=2, first_frame=4, **_df_to_npz(d.data))


if __name__ == "__main__":
    gold_direct_msd()
    gold_msd_of_m()
    gold_construct_step()
    gold_e2e()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))
