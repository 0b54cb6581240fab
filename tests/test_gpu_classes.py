"""Class-level parity on the GPU: the drop-in classes against the DataFrames
the reference's own drivers produced (tests/golden/reference_e2e_*.npz), and
feather round trips."""

import json
import os

import numpy as np
import pandas as pd
import pytest

from amof_amd.frames import Frame, PackedTrajectory
from amof_amd.rdf import Rdf
from amof_amd.msd import WindowMsd
from amof_amd.bad import Bad
from amof_amd.cn import CoordinationNumber
from tests import helpers as H
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

RTOL = 1e-6   # north_star: floats within 1e-6 relative


def _frames(g):
    cell = g["cell"]
    return [Frame(g["numbers"], g["pos"][k], cell[k if len(cell) > 1 else 0]) for k in range(len(g["pos"]))]


def _check_df(df, g, rtol=RTOL, atol=1e-12, values="values"):
    assert list(df.columns) == [str(c) for c in g["columns"]]
    np.testing.assert_allclose(df.values.astype(float), g[values], rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ["zif4_default", "zif4_dr0p05_rmax6"])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("shell", [None, "exact", "midpoint"])
def test_rdf_matches_reference_dataframe(name, packed, shell, monkeypatch):
    """The reference's own Rdf class over the stand-in asap3 object, under BOTH shell-volume conventions (assumption
    A1): the default (exact shell) and AMOF_RDF_SHELL=midpoint each reproduce their DataFrame -- and not the other's."""
    if shell is None:
        monkeypatch.delenv("AMOF_RDF_SHELL", raising=False)
    else:
        monkeypatch.setenv("AMOF_RDF_SHELL", shell)
    g = np.load(os.path.join(GOLDEN, "reference_e2e_rdf_%s.npz" % name))
    rmax = str(g["rmax"]) if g["rmax"].dtype.kind == "U" else float(g["rmax"])
    traj = _frames(g)
    if packed:
        traj = PackedTrajectory(g["pos"], g["cell"], g["numbers"])
    rdf = Rdf.from_trajectory(traj, dr=float(g["dr"]), rmax=rmax)
    _check_df(rdf.data, g, values="values_midpoint" if shell == "midpoint" else "values")
    with pytest.raises(AssertionError):      # dr^2 / (12 r^2) at the first populated bins: far above 1e-6
        _check_df(rdf.data, g, values="values" if shell == "midpoint" else "values_midpoint")
    # integer sum rule and symmetry of the raw counts
    assert np.array_equal(rdf.hist, rdf.hist.transpose(1, 0, 2))


@pytest.mark.parametrize("name", ["ortho_raw", "ortho_unwrap", "tri_raw", "tri_unwrap", "zif4_rattle"])
def test_msd_matches_reference_dataframe(name):
    g = np.load(os.path.join(GOLDEN, "reference_e2e_msd_%s.npz" % name))
    frames = _frames(g)
    before = [f.positions.copy() for f in frames]
    msd = WindowMsd.from_trajectory(frames, delta_time=int(g["delta_time"]), timestep=int(g["timestep"]),
                                    unwrap=bool(g["unwrap"]))
    _check_df(msd.data, g, rtol=1e-9)
    for f, b in zip(frames, before):           # unlike the reference, inputs are not mutated
        assert np.array_equal(f.positions, b)


def test_cn_matches_reference_dataframe():
    g = np.load(os.path.join(GOLDEN, "reference_e2e_cn_zif4.npz"))
    cn = CoordinationNumber.from_trajectory(_frames(g), json.loads(str(g["cutoffs"])),
                                            delta_Step=int(g["delta_Step"]), first_frame=int(g["first_frame"]))
    assert list(cn.data.columns) == [str(c) for c in g["columns"]]
    assert np.array_equal(cn.data.values.astype(float), g["values"])      # integer counts / N_A: exact


@pytest.mark.parametrize("tag", ["0p05", "0p5"])
def test_bad_matches_reference_dataframe(tag):
    g = np.load(os.path.join(GOLDEN, "reference_e2e_bad_zif4_dtheta%s.npz" % tag))
    bad = Bad.from_trajectory(_frames(g), json.loads(str(g["cutoffs"])), dtheta=float(g["dtheta"]))
    _check_df(bad.data, g, rtol=1e-12, atol=0)
    assert "Zn-N-Zn" not in bad.data.columns                   # empty columns are omitted (amof/bad.py:159)


def test_bad_with_all_species_uses_X(zif4):
    packed = H.random_walk(zif4, 2, 0.02, 5)
    cut = {'Zn-N': 2.5, 'C-H': 1.3, 'C-N': 1.6}
    bad = Bad.from_trajectory(packed, cut, dtheta=0.5)
    assert "X-X-X" in bad.data.columns and "X-Zn-X" in bad.data.columns and "N-Zn-N" in bad.data.columns
    # every Zn neighbour is an N here, so X-Zn-X == N-Zn-N
    assert np.array_equal(bad.data["X-Zn-X"].values, bad.data["N-Zn-N"].values)
    w = np.diff(np.arange(int(180 // 0.5) + 2) * 0.5)
    assert (bad.data["X-X-X"].values * w).sum() == pytest.approx(1.0, rel=1e-12)


def test_feather_round_trips(tmp_path, zif4):
    pytest.importorskip("pyarrow")       # feather backend; the reference uses it too (amof/rdf.py:118)
    packed = H.random_walk(zif4, 6, 0.05, 1)
    rdf = Rdf.from_trajectory(packed, dr=0.05)
    rdf.write_to_file(tmp_path / "a")
    assert (tmp_path / "a.rdf").exists()
    back = Rdf.from_file(tmp_path / "a")
    assert np.allclose(back.data, rdf.data) and list(back.data.columns) == list(rdf.data.columns)
    msd = WindowMsd.from_trajectory(packed, delta_time=1, timestep=1)
    msd.write_to_file(tmp_path / "m.msd")
    assert np.array_equal(WindowMsd.from_file(tmp_path / "m").data.values, msd.data.values)
    cn = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5})
    cn.write_to_file(tmp_path / "c")
    assert np.array_equal(CoordinationNumber.from_file(tmp_path / "c").data.values, cn.data.values)
    bad = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=1.0)
    bad.write_to_file(tmp_path / "b")
    assert np.array_equal(Bad.from_file(tmp_path / "b").data.values, bad.data.values)
    assert list(cn.data.columns) == ["Step", "Zn-N"] and (cn.data["Zn-N"] == 4.0).all()


def test_sum_rules_on_dataframe(zif4):
    packed = H.random_walk(zif4, 5, 0.05, 2)
    rdf = Rdf.from_trajectory(packed)
    d = rdf.data
    counts = packed.formula_count()
    N = packed.n_atoms
    # 'A-X' = sum_j g_Aj  and  g_XX = sum_a (N_a / N) g_aX
    for a in counts:
        np.testing.assert_allclose(d[a + "-X"], sum(d[a + "-" + b] for b in counts), rtol=1e-13)
    np.testing.assert_allclose(d["X-X"], sum(counts[a] / N * d[a + "-X"] for a in counts), rtol=1e-12)
    assert len(d) == int(rdf.rmax // 0.01) and d["r"][1] == 0.01


def test_torch_resident_trajectory_matches_host(zif4):
    import torch
    packed = H.random_walk(zif4, 8, 0.05, 3)
    dev = PackedTrajectory(torch.tensor(packed.pos, device="cuda:0"), packed.cell, packed.numbers)
    a = Rdf.from_trajectory(packed).hist
    b = Rdf.from_trajectory(dev).hist
    assert np.array_equal(a, b)
    m1 = WindowMsd.from_trajectory(packed, delta_time=1, timestep=1).data.values
    m2 = WindowMsd.from_trajectory(dev, delta_time=1, timestep=1).data.values
    assert np.array_equal(m1, m2)                 # same kernels, same order: bitwise


def test_direct_msd_matches_reference_dataframe():
    from amof_amd.msd import DirectMsd
    g = np.load(os.path.join(GOLDEN, "reference_e2e_directmsd_ortho.npz"))
    d = DirectMsd.from_trajectory(_frames(g), delta_Step=int(g["delta_Step"]), first_frame=int(g["first_frame"]))
    _check_df(d.data, g, rtol=1e-9)


def test_static_helpers_of_the_msd_classes_against_the_reference_vectors():
    """The reference's static helpers are part of its classes' surface (amof/msd.py:84-107, 185-205).  Here they run on the
    GPU: `WindowMsd.compute_msd_of_m` against the vectors the REFERENCE's own function produced
    (tests/golden/reference_msd_of_m.npz, made by tests/golden/make_reference_goldens.py -- pure reference numpy), single
    calls and the reference's pattern of successive calls on one list; `DirectMsd.compute_species_msd` against the class's
    columns, which `reference_e2e_directmsd_ortho.npz` ties to the reference."""
    from amof_amd.msd import DirectMsd
    g = np.load(os.path.join(GOLDEN, "reference_msd_of_m.npz"))
    for k in range(int(g["n_cases"])):
        base, ms, want = g["delta_%d" % k], g["m_%d" % k], g["msd_%d" % k]
        for m, w in zip(ms, want):
            got = WindowMsd.compute_msd_of_m([b.copy() for b in base], int(m))
            assert got == pytest.approx(w, rel=1e-12, abs=1e-300), (k, int(m))
    shared = [b.copy() for b in g["delta_shared"]]
    first = shared[0].copy()
    for m, w in zip(g["m_shared"], g["msd_shared"]):
        assert WindowMsd.compute_msd_of_m(shared, int(m)) == pytest.approx(w, rel=1e-12, abs=1e-300)
    assert np.array_equal(shared[0], first)                      # (documented deviation: the caller's list is not modified)
    with pytest.raises(ValueError):
        WindowMsd.compute_msd_of_m(shared, len(shared))
    e = np.load(os.path.join(GOLDEN, "reference_e2e_directmsd_ortho.npz"))
    frames = _frames(e)
    d = DirectMsd.from_trajectory(frames, delta_Step=int(e["delta_Step"]), first_frame=int(e["first_frame"]))
    assert np.array_equal(DirectMsd.compute_species_msd(frames), d.data["X"].values)
    for z in sorted(set(int(n) for n in frames[0].get_atomic_numbers())):
        from amof_amd import data as amdata
        assert np.array_equal(DirectMsd.compute_species_msd(frames, z), d.data[amdata.chemical_symbols[z]].values)
    with pytest.raises(ValueError):
        DirectMsd.compute_species_msd(frames, 92)


def test_to_device_round_trip(zif4):
    packed = H.random_walk(zif4, 5, 0.05, 9)
    dev = packed.to_device(0)
    assert dev.on_device and dev.to_device(0) is dev and not packed.on_device
    assert np.array_equal(dev.pos_host(), packed.pos)
    a = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5}).data.values
    b = CoordinationNumber.from_trajectory(dev, {'Zn-N': 2.5}).data.values
    assert np.array_equal(a, b)


def test_rdf_integration_coordination_number(zif4):
    # deprecated RDF-integration CN: must agree with the counting CN up to its integration error
    from amof_amd.rdf import CoordinationNumber as RdfCn
    packed = H.random_walk(zif4, 3, 0.02, 21)
    a = RdfCn.from_trajectory(packed, {'Zn-N': 2.5, 'C-H': 1.3}, delta_Step=2, first_frame=5, dr=0.001)
    b = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5, 'C-H': 1.3}, delta_Step=2, first_frame=5)
    assert list(a.data.columns) == ["Step", "Zn-N", "C-H"] and list(a.data["Step"]) == [5, 7, 9]
    np.testing.assert_allclose(a.data["Zn-N"].values, b.data["Zn-N"].values, rtol=0.01)
    # single-frame histograms at dr = 1e-3 are spikes; Simpson's alternating 4/3, 2/3 weights make the
    # integral noisy (the reference warns about exactly this, amof/rdf.py:139-141)
    np.testing.assert_allclose(a.data["C-H"].values, b.data["C-H"].values, rtol=0.34)
    rdf = Rdf.from_trajectory(packed, dr=0.001, rmax=6.0)
    rho = 272 / zif4.get_volume()
    assert rdf.get_coordination_number("Zn-N", 2.5, rho) == pytest.approx(4.0, rel=0.02)


@pytest.mark.parametrize("resident", [False, True])
def test_several_contexts_from_one_process(resident):
    # device=[0, 0]: a MultiContext (two contexts / streams / threads; a real multi-GPU node would list
    # different devices) shards frames (RDF, BAD, CN) or atoms (MSD) and merges on the host -- same results
    from amof_amd.bad import Bad, BadByCn
    from amof_amd.cn import CoordinationNumber
    from amof_amd.msd import WindowMsd
    from amof_amd.rdf import Rdf
    packed = H.random_walk(H.replicate(H.zif4_frame(), (1, 1, 2)), 11, 0.08, 123, cell_jitter=0.004)
    if resident:
        packed = packed.to_device(0)
    cut = {'Zn-N': 2.5, 'C-N': 1.6}
    for make in (lambda d: Rdf.from_trajectory(packed, dr=0.02, device=d, distributed=False),
                 lambda d: Bad.from_trajectory(packed, cut, dtheta=0.5, device=d, distributed=False),
                 lambda d: CoordinationNumber.from_trajectory(packed, cut, device=d, distributed=False)):
        one, two = make(0).data, make([0, 0]).data
        assert list(one.columns) == list(two.columns) and np.array_equal(one.values, two.values)
    m1 = WindowMsd.from_trajectory(packed, delta_time=1, timestep=1, device=0, distributed=False).data
    m2 = WindowMsd.from_trajectory(packed, delta_time=1, timestep=1, device=[0, 0], distributed=False).data
    np.testing.assert_allclose(m2.values, m1.values, rtol=1e-12, atol=1e-15)
    b1 = BadByCn.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=1.0, device=0, distributed=False)
    b2 = BadByCn.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=1.0, device=[0, 0], distributed=False)
    assert np.array_equal(b1.hist, b2.hist)


def test_classes_take_ase_shaped_atoms_lists(zif4):
    """the reference's calling convention -- a Python list of ase.Atoms -- with objects shaped like ASE's own
    (tests.helpers.AseLikeAtoms: Cell object, property-backed arrays, symbols.formula._count); results equal the
    packed-trajectory route, and the inputs are left untouched (the reference mutates them, amof/msd.py:230,237)"""
    from amof_amd.bad import Bad
    from amof_amd.msd import WindowMsd
    packed = H.random_walk(zif4, 7, 0.06, 41, cell_jitter=0.004)
    atoms = H.as_ase_like(packed)
    before = [a.get_positions() for a in atoms]
    cut = {'Zn-N': 2.5}
    pairs = [(Rdf.from_trajectory(atoms, dr=0.02), Rdf.from_trajectory(packed, dr=0.02)),
             (Bad.from_trajectory(atoms, cut, dtheta=0.5), Bad.from_trajectory(packed, cut, dtheta=0.5)),
             (CoordinationNumber.from_trajectory(atoms, cut), CoordinationNumber.from_trajectory(packed, cut)),
             (WindowMsd.from_trajectory(atoms, delta_time=1, timestep=1),
              WindowMsd.from_trajectory(packed, delta_time=1, timestep=1)),
             (WindowMsd.from_trajectory(atoms, delta_time=1, timestep=1, unwrap=True),
              WindowMsd.from_trajectory(packed, delta_time=1, timestep=1, unwrap=True))]
    for a, b in pairs:
        assert list(a.data.columns) == list(b.data.columns) and np.array_equal(a.data.values, b.data.values)
    assert all(np.array_equal(a.get_positions(), p) for a, p in zip(atoms, before))
