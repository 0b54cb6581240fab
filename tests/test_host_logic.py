"""Host-side logic and the C-ABI surface (no GPU needed)."""

import ctypes
import os
import re

import numpy as np
import pandas as pd
import pytest

import amof_amd
from amof_amd import _hip, atom, data, dist
from amof_amd.files.path import append_suffix
from amof_amd.frames import Frame, PackedTrajectory, pack_trajectory
from tests import helpers as H
from tests.conftest import ROOT


def test_library_loads_and_exports_every_declared_symbol():
    lib = _hip.load_library()
    header = open(os.path.join(ROOT, "include", "amof_hip.h")).read()
    declared = set(re.findall(r"\b(amof_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.amof_abi_version() == _hip.ABI_VERSION == 4


def test_struct_layout_matches_header():
    # 8 + 4 + 4 + 8 + 8*3 + 8 + 8 + 3 + 5 = 72 bytes, 8-byte aligned
    assert ctypes.sizeof(_hip.AmofTraj) == 72
    assert _hip.AmofTraj.cell.offset == 16 and _hip.AmofTraj.species.offset == 48
    assert _hip.AmofTraj.pbc.offset == 64


def test_product_fails_loudly_without_gpu():
    if _hip.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no GPU"):
        _hip.Context(0)
    from amof_amd.rdf import Rdf
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Rdf.from_trajectory(H.random_walk(H.zif4_frame(), 2, 0.05, 0))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "amof_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libamof_oracle" not in src, f


def test_append_suffix():
    assert str(append_suffix("a/b", "rdf")) == "a/b.rdf"
    assert str(append_suffix("a/b.rdf", ".rdf")) == "a/b.rdf"
    assert str(append_suffix("a/b.msd", "rdf")) == "a/b.msd.rdf"
    # outputs of the reference's own function on these inputs (amof/files/path.py:7-21, run in the build container)
    for path, suffix, want in [("a", "rdf", "a.rdf"), ("a.rdf", "rdf", "a.rdf"), ("dir/a.v2", "msd", "dir/a.v2.msd"),
                               ("a.tar.gz", "gz", "a.tar.gz"), ("a", "", "a"), ("a.b", "", "a.b"), ("y.", ".bad", "y..bad"),
                               (".hidden", "rdf", ".hidden.rdf"), ("a.rdf.bak", "rdf", "a.rdf.bak.rdf"), ("", "cn", ".cn")]:
        assert str(append_suffix(path, suffix)) == want, (path, suffix)
    assert str(append_suffix("x", "")) == "x"


def test_format_cutoff_and_matrix():
    d = atom.format_cutoff({'Zn-N': 2.5, 'N-Zn': 3.0, 'C-H': 1.2})
    assert d == {(30, 7): 2.5, (7, 30): 3.0, (6, 1): 1.2}
    assert atom.format_cutoff({'Zn-N': 2.5}, sort_pair=True) == {(7, 30): 2.5}
    m = atom.cutoff_matrix(d, [1, 6, 7, 30])
    assert m[3, 2] == m[2, 3] == 3.0 and m[0, 1] == m[1, 0] == 1.2 and m[0, 0] == 0
    assert atom.cutoff_matrix({(8, 1): 1.0}, [1, 6]).sum() == 0     # species absent


def test_element_tables():
    assert data.chemical_symbols[0] == 'X' and data.chemical_symbols[30] == 'Zn'
    assert data.atomic_numbers['N'] == 7 and len(data.chemical_symbols) == 119
    assert data.atomic_masses[1] == pytest.approx(1.008) and data.atomic_masses[30] == pytest.approx(65.38)


def test_bin_count_arithmetic_is_python_floor_division():
    # SURVEY TL;DR: never compute bin counts in C
    assert int(10 // 0.01) == 999 and int(7.5 // 0.01) == 749 and int(180 // 0.05) == 3599
    assert int(180 // 0.5) == 360


def test_frame_api(zif4):
    f = zif4
    assert len(f) == 272 and f.get_global_number_of_atoms() == 272
    assert f.get_volume() == pytest.approx(4380.4858, abs=1e-3)
    assert atom.get_number_density(f) == pytest.approx(0.0620936, abs=1e-6)
    assert f.symbols.formula._count == {'C': 96, 'H': 96, 'N': 64, 'Zn': 16}
    com = f.get_center_of_mass()
    g = f.copy(); g.translate(-com)
    np.testing.assert_allclose(g.get_center_of_mass(), 0, atol=1e-12)
    assert atom.select_species_positions(f, 30).shape == (16, 3)
    assert sorted(int(z) for z in atom.get_atomic_numbers_unique(f)) == [1, 6, 7, 30]
    la = f.get_cell_lengths_and_angles()
    assert la[0] == pytest.approx(15.4231) and la[3] == pytest.approx(90.0, abs=1e-2)


def test_pack_trajectory(zif4):
    frames = [zif4.copy() for _ in range(3)]
    frames[1].positions += 0.1
    p = pack_trajectory(frames)
    assert p.pos.shape == (3, 272, 3) and p.cell.shape == (1, 3, 3) and p.n_frames == 3
    assert pack_trajectory(p) is p
    frames[2].cell = frames[2].cell * 1.01
    assert pack_trajectory(frames).cell.shape == (3, 3, 3)
    with pytest.raises(ValueError):
        pack_trajectory([])
    with pytest.raises(ValueError):
        PackedTrajectory(np.zeros((2, 3, 2)), np.eye(3), [1, 1, 1])
    kinds, sp = _hip.species_index(p.numbers)
    assert kinds == [1, 6, 7, 30] and sp.dtype == np.int32 and sp.max() == 3


def test_normalize_rdf_formula(monkeypatch):
    from amof_amd.rdf import normalize_rdf
    monkeypatch.delenv("AMOF_RDF_SHELL", raising=False)
    nb, rmax, N, F, V = 10, 5.0, 100, 4, 1000.0
    h = np.arange(nb) * 7
    g = normalize_rdf(h, F * N, N, V, rmax, nb)
    d = rmax / nb
    r = (np.arange(nb) + 0.5) * d
    np.testing.assert_allclose(g, V * h / (4 * np.pi * d * (r ** 2 + d * d / 12) * N * F * N), rtol=1e-15)


def test_shard_range_partitions():
    for n in (0, 1, 7, 5000, 9792):
        for w in (1, 2, 3, 8):
            r = [dist.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    assert dist.world() == (0, 1)
    x = np.arange(4, dtype=np.uint64)
    assert dist.all_reduce_sum(x) is x


def test_simps_restatement_and_rdf_integration_cn():
    import scipy.integrate
    from amof_amd.rdf import simps, get_coordination_number
    rng = np.random.default_rng(0)
    x = np.sort(rng.uniform(0, 3, 41))
    y = np.sin(x) + x ** 2
    # odd number of samples: plain composite Simpson, identical in every scipy
    assert simps(y, x) == pytest.approx(scipy.integrate.simpson(y, x=x), rel=1e-13)
    # even number: scipy 1.7.1's 'avg' = mean of (Simpson on the first N-1 + trapezoid on the last interval)
    # and (trapezoid on the first interval + Simpson on the last N-1)
    xe, ye = x[:-1], y[:-1]
    first = scipy.integrate.simpson(ye[:-1], x=xe[:-1]) + 0.5 * (xe[-1] - xe[-2]) * (ye[-1] + ye[-2])
    last = scipy.integrate.simpson(ye[1:], x=xe[1:]) + 0.5 * (xe[1] - xe[0]) * (ye[1] + ye[0])
    assert simps(ye, xe) == pytest.approx(0.5 * (first + last), rel=1e-13)
    # CN of an ideal gas: 4 pi rho int_0^rc r^2 dr = rho * (4/3) pi rc^3
    r = np.arange(3000) * 0.001
    rho, rc = 0.05, 2.5
    assert get_coordination_number(r, np.ones_like(r), rc, rho) == pytest.approx(rho * 4 / 3 * np.pi * rc ** 3, rel=2e-3)


def test_traj_handle_of_a_host_trajectory_needs_no_gpu():
    """Context._traj (argument marshalling for the C ABI) on host arrays: no device check, no stream ordering"""
    from amof_amd import _hip
    from tests import helpers as H

    class FakeCtx(object):
        device = 0

        def wait_stream(self, ptr):
            raise AssertionError("host-resident positions need no stream ordering")

    packed = H.random_walk(H.zif4_frame(), 4, 0.05, 3)
    th = _hip.Context._traj(FakeCtx(), packed, (1, 3))
    assert th.c.n_frames == 2 and th.c.n_atoms == 272 and th.c.pos_on_device == 0 and th.S == 4
    assert th.kinds == [1, 6, 7, 30]
    assert packed._abi_species[1] is th.species          # cached on the trajectory


def test_simps_is_scipy_171_simps_even_avg():
    """rdf.simps restates scipy 1.7.1's simps(y, x) (default even='avg'), which the reference pins
    (requirements.txt:15) and calls at amof/rdf.py:226.  Pins: for an odd number of samples it is the composite
    Simpson rule -- equal to the installed scipy.integrate.simpson; for an even number it is the average of
    (Simpson on the first N-1 samples + trapezoid on the last interval) and (trapezoid on the first interval +
    Simpson on the last N-1 samples), each built here from the odd-count rule."""
    from scipy.integrate import simpson
    from amof_amd.rdf import simps
    rng = np.random.default_rng(5)
    for n in (3, 5, 101, 2499):                 # odd counts, non-uniform abscissae
        x = np.cumsum(rng.uniform(0.5, 1.5, n)) * 1e-3
        y = np.sin(40 * x) + x ** 2 + rng.normal(scale=0.01, size=n)
        np.testing.assert_allclose(simps(y, x), simpson(y, x=x), rtol=1e-13, atol=1e-16)
    for n in (4, 6, 100, 2500):                 # even counts: scipy 1.7.1's even='avg'
        x = np.cumsum(rng.uniform(0.5, 1.5, n)) * 1e-3
        y = np.cos(25 * x) + rng.normal(scale=0.01, size=n)
        first = simpson(y[:-1], x=x[:-1]) + 0.5 * (x[-1] - x[-2]) * (y[-1] + y[-2])
        last = 0.5 * (x[1] - x[0]) * (y[1] + y[0]) + simpson(y[1:], x=x[1:])
        np.testing.assert_allclose(simps(y, x), 0.5 * (first + last), rtol=1e-13, atol=1e-16)
    assert simps([1.0], [0.0]) == 0.0
    # exact for cubics on an odd number of uniform samples
    x = np.linspace(0, 2, 41)
    np.testing.assert_allclose(simps(x ** 3 - x, x), 2.0, rtol=1e-14)


def test_pack_trajectory_of_ase_shaped_objects():
    """pack_trajectory sees real ASE shapes: a Cell object (not an ndarray), property-backed positions / numbers"""
    from amof_amd.frames import pack_trajectory
    from tests import helpers as H
    packed = H.random_walk(H.zif4_frame(), 5, 0.05, 3, cell_jitter=0.01)
    atoms = H.as_ase_like(packed)
    assert not isinstance(atoms[0].cell, np.ndarray) and not isinstance(atoms[0].get_cell(), np.ndarray)
    again = pack_trajectory(atoms)
    assert np.array_equal(again.pos, packed.pos) and np.array_equal(again.cell, packed.cell)
    assert np.array_equal(again.numbers, packed.numbers) and np.array_equal(again.masses, packed.masses)
    assert again.pbc.all() and again.formula_count() == atoms[0].symbols.formula._count
    const = pack_trajectory(H.as_ase_like(H.random_walk(H.zif4_frame(), 3, 0.05, 4)))
    assert const.cell.shape == (1, 3, 3)                     # identical cells collapse to one record


def test_volume_sum_is_the_librarys_sum_bit_for_bit():
    """PackedTrajectory.volume_sum restates csrc/ctx.hip build_geometry (|det| by cofactors along the first row, added
    frame by frame); the oracle's geom_make follows the same sequence, so its volume_sum must agree to the last bit --
    for changing (sheared, jittered) cells, for a constant cell added thousands of times, and for frame sub-ranges.
    A frame-sharded RDF relies on it instead of all-reducing the ranks' partial float sums."""
    from oracle import clib
    rng = np.random.default_rng(5)
    base = np.array([[15.4231, 0, 0], [-0.0001882, 15.4042, 0], [-0.00057924, -0.00012873, 18.43789999]])
    cells = base[None] * (1.0 + 0.01 * rng.standard_normal((257, 1, 1))) + 0.3 * rng.standard_normal((257, 3, 3))
    pos = rng.random((257, 2, 3))
    for cell, F in ((cells, 257), (base, 257), (cells[:1], 1), (np.diag([46.2693, 46.2126, 73.7516]), 4999)):
        p = np.ascontiguousarray(np.broadcast_to(pos[:1], (F, 2, 3)))
        packed = PackedTrajectory(p, cell, [1, 1])
        _, vol = clib.rdf_hist(p, packed.cell, np.zeros(2, np.int32), 1, 0.5, 5)
        assert packed.volume_sum() == vol
        if packed.cell.shape[0] > 1:
            _, vol2 = clib.rdf_hist(p[31:200], packed.cell[31:200], np.zeros(2, np.int32), 1, 0.5, 5)
            assert packed.volume_sum((31, 200)) == vol2
    assert PackedTrajectory(pos[:0], base, [1, 1]).volume_sum() == 0.0


def test_rdf_shell_switch(monkeypatch):
    """AMOF_RDF_SHELL: the exact shell volume (4 pi / 3)(r_hi^3 - r_lo^3) (default since round 4) or the midpoint shell
    4 pi r^2 dr -- assumption A1 about asap3's get_rdf, a visible switch because asap3 cannot be run here."""
    from amof_amd.rdf import normalize_rdf, normalize_rdf_shell
    hist = np.arange(1, 51, dtype=np.uint64) * 1000
    args = (hist, 4 * 272.0, 272, 4380.486, 5.0, 50)
    monkeypatch.delenv("AMOF_RDF_SHELL", raising=False)
    default = normalize_rdf(*args)
    monkeypatch.setenv("AMOF_RDF_SHELL", "midpoint")
    mid = normalize_rdf(*args)
    monkeypatch.setenv("AMOF_RDF_SHELL", "exact")
    exact = normalize_rdf(*args)
    monkeypatch.setenv("AMOF_RDF_SHELL", "bogus")
    with pytest.raises(ValueError):
        normalize_rdf(*args)
    assert np.array_equal(default, exact)
    dr = 5.0 / 50
    r = (np.arange(50) + 0.5) * dr
    np.testing.assert_allclose(mid / exact, 1.0 + dr * dr / (12.0 * r * r), rtol=1e-14)
    lo, hi = r - dr / 2, r + dr / 2
    np.testing.assert_allclose(exact, hist * (4380.486 / (272 * 4 * 272.0)) / (4 * np.pi / 3 * (hi ** 3 - lo ** 3)), rtol=1e-13)
    assert np.array_equal(mid, normalize_rdf_shell(*args, "midpoint"))


def test_every_kernel_family_is_pinned_by_a_forced_path_test():
    """Variant sprawl guard: every path name the library can report (amof_last_path, the strings in csrc/*.hip) is
    documented in include/amof_hip.h AND asserted by at least one `-m gpu` test that forces it and compares with the
    oracle -- a kernel family nobody selects any more, or one nobody tests, fails here."""
    csrc = os.path.join(ROOT, "amof_amd", "csrc")
    emitted = set()
    for fn in os.listdir(csrc):
        if fn.endswith(".hip"):
            src = open(os.path.join(csrc, fn)).read()
            emitted |= set(re.findall(r'last_path = [^;]*?"([a-z_0-9]+)"', src))
            emitted |= set(re.findall(r'last_path = [^;]*\? "([a-z_0-9]+)" : "([a-z_0-9]+)"', src) and
                           [x for pair in re.findall(r'\? "([a-z_0-9]+)" : "([a-z_0-9]+)"', src) for x in pair])
            emitted |= set(re.findall(r'timing_dom_begin\(ctx, "([a-z_0-9]+)"', src))
            for args in re.findall(r'timing_dom_begin\(ctx, ([^;]*)\);', src):
                emitted |= set(re.findall(r'"([a-z_0-9]+)"', args))
    emitted = {p for p in emitted if re.match(r"(rdf|cn|bad|msd)_", p)}
    header = open(os.path.join(ROOT, "include", "amof_hip.h")).read()
    doc = header[header.index("kernel family that produced the result of the last call"):header.index("const char *amof_last_path")]
    documented = set(re.findall(r'"((?:rdf|cn|bad|msd)_[a-z_0-9]+)"', doc))
    assert emitted == documented, (sorted(emitted - documented), sorted(documented - emitted))
    tested = set()
    for fn in os.listdir(os.path.join(ROOT, "tests")):
        if fn.startswith("test_gpu") and fn.endswith(".py"):
            src = open(os.path.join(ROOT, "tests", fn)).read()
            for line in src.splitlines():
                if "last_path()" in line or "want" in line:
                    tested |= set(re.findall(r'"((?:rdf|cn|bad|msd)_[a-z_0-9]+)"', line))
    assert emitted <= tested, sorted(emitted - tested)


def test_constructors_start_with_the_reference_s_empty_frame_built_on_first_look():
    """The reference's constructors hold an empty DataFrame with the first column (amof/rdf.py:33-35,144-146, msd.py:64-66,152-154,
    bad.py:66-68, cn.py:30-32); here it is built when first looked at (amof_amd/_lazy.py) -- same columns, one object per
    instance, replaced by assignment, survives pickling"""
    import pickle
    from amof_amd.rdf import Rdf, CoordinationNumber as RdfCn
    from amof_amd.msd import WindowMsd, DirectMsd
    from amof_amd.bad import Bad, BadByCn
    from amof_amd.cn import CoordinationNumber
    for cls, col in ((Rdf, "r"), (RdfCn, "Step"), (WindowMsd, "Time"), (DirectMsd, "Step"), (Bad, "theta"), (CoordinationNumber, "Step")):
        a, b = cls(), cls()
        assert "_data" not in a.__dict__ or a.__dict__["_data"] is None
        assert list(a.data.columns) == [col] and len(a.data) == 0 and a.data.dtypes[col] == np.float64
        assert a.data is a.data and a.data is not b.data
        c = pickle.loads(pickle.dumps(a))
        assert list(c.data.columns) == [col]
        a.data = pd.DataFrame({col: [1.0, 2.0]})
        assert len(a.data) == 2 and len(b.data) == 0
    assert BadByCn().data is None          # (the reference starts with an empty xarray.DataArray, amof/bad.py:183-187; xarray is not installed here)
