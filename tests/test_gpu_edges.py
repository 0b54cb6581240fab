"""Edge cases and size-independent properties of the HIP path (GPU)."""

import numpy as np
import pytest

from amof_amd import _hip
from amof_amd.frames import Frame, PackedTrajectory
from amof_amd.rdf import Rdf
from amof_amd.msd import WindowMsd
from oracle import clib, numpy_oracle as no
from tests import helpers as H

pytestmark = pytest.mark.gpu


class H_env(object):
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _oracle_rdf(packed, rmax, nb):
    kinds, sp = H.species_of(packed.numbers)
    return clib.rdf_hist(packed.pos_host(), packed.cell, sp, len(kinds), rmax, nb, pbc=packed.pbc)[0]


def test_sc_lattice_and_exact_half_cell(hip_ctx):
    # 216-atom simple-cubic KAT of BASELINE cfg1: atoms exactly L/2 apart are NOT counted (strict <)
    a, n = 2.5, 6
    g = np.arange(n) * a
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    idx = np.array([[i, j, k] for i in range(n) for j in range(n) for k in range(n)])
    numbers = np.where(idx.sum(axis=1) % 2 == 0, 11, 17)
    packed = PackedTrajectory(pos[None], np.diag([n * a] * 3), numbers)
    rdf = Rdf.from_trajectory(packed)
    assert rdf.rmax == 7.5 and len(rdf.data) == 749
    assert np.array_equal(rdf.hist, _oracle_rdf(packed, 7.5, 749))
    tot = rdf.hist.sum(axis=(0, 1))
    # lattice vectors with |v| < 3a: 6 + 12 + 8 + 6 + 24 + 24 + 12 (k = 1,2,3,4,5,6,8); k = 9 is |v| = L/2: excluded
    assert tot.sum() == (6 + 12 + 8 + 6 + 24 + 24 + 12) * 216


@pytest.mark.parametrize("n_atoms", [1, 2, 63, 64, 65, 255, 256, 257, 513])
def test_ragged_tile_sizes(hip_ctx, n_atoms):
    rng = np.random.default_rng(n_atoms)
    numbers = rng.choice([1, 8, 14], size=n_atoms)
    packed = H.random_gas(n_atoms, [9.0, 10.0, 11.0], numbers, n_atoms, F=3)
    h, _, _ = hip_ctx.rdf_accumulate(packed, 4.5, 450)
    assert np.array_equal(h, _oracle_rdf(packed, 4.5, 450))


@pytest.mark.parametrize("env", [None, ("AMOF_QUANT_NOSTAGE", "1"), ("AMOF_QUANT_PER_SPECIES", "1")])
def test_species_segments_longer_than_the_quantisers_stage(hip_ctx, monkeypatch, env):
    """quantize_frame_kernel (csrc/quant.hip) places a species' records through an LDS stage of 4608 records and scatters a
    longer segment directly; 5200 atoms of one species beside two short ones, RDF and neighbour counts against the oracle
    (and the same with the stage switched off, and through the per-species kernel)"""
    if env:
        monkeypatch.setenv(*env)
    rng = np.random.default_rng(5200)
    numbers = np.concatenate([np.full(5200, 18), np.full(300, 8), np.full(7, 30)])
    rng.shuffle(numbers)
    packed = H.random_gas(len(numbers), [38.0, 41.0, 44.0], numbers, 52, F=2)
    h, _, _ = hip_ctx.rdf_accumulate(packed, 19.0, 380)
    assert hip_ctx.last_path().startswith("rdf_tile")
    kinds, sp = H.species_of(packed.numbers)
    ref = clib.rdf_hist(packed.pos_host(), packed.cell, sp, len(kinds), 19.0, 380, cell_list=True)[0]
    assert np.array_equal(h, ref)
    rcm = np.zeros((3, 3))
    rcm[kinds.index(18), kinds.index(18)] = 3.1
    rcm[kinds.index(8), kinds.index(18)] = rcm[kinds.index(18), kinds.index(8)] = 3.4
    sets = [(kinds.index(18), kinds.index(18)), (kinds.index(8), kinds.index(18))]
    got = hip_ctx.cn_count(packed, rcm, sets)
    assert np.array_equal(got, clib.cn_counts(packed.pos_host(), packed.cell, sp, 3, rcm, sets))


def test_empty_inputs(hip_ctx):
    z = H.zif4_frame()
    empty = PackedTrajectory(np.zeros((0, 272, 3)), z.cell, z.numbers)
    h, vol, _ = hip_ctx.rdf_accumulate(empty, 5.0, 50)
    assert h.sum() == 0 and vol == 0.0
    one = H.random_walk(z, 1, 0.0, 0)
    s, _ = hip_ctx.msd_window(one, [0])
    assert (s == 0).all()
    sums = hip_ctx.cn_count(one, np.zeros((4, 4)), [(0, 1)])
    assert sums.shape == (1, 1) and sums[0, 0] == 0


def test_argument_errors(hip_ctx):
    z = H.random_walk(H.zif4_frame(), 2, 0.01, 0)
    with pytest.raises(ValueError):
        hip_ctx.rdf_accumulate(z, -1.0, 10)
    with pytest.raises(ValueError):
        hip_ctx.rdf_accumulate(z, 5.0, 0)
    with pytest.raises(ValueError):
        hip_ctx.msd_window(z, [5])                        # window >= F
    bad_cell = PackedTrajectory(z.pos, np.zeros((3, 3)), z.numbers)
    with pytest.raises(_hip.AmofError) as e:
        hip_ctx.rdf_accumulate(bad_cell, 5.0, 10)
    assert e.value.code == _hip.AMOF_ESINGULAR
    with pytest.raises(ValueError):
        hip_ctx.cn_count(z, np.array([[0, 1.0], [2.0, 0]]).repeat(2, 0).repeat(2, 1), [(0, 1)])   # asymmetric


def test_undefined_angle_raises_like_ase(hip_ctx):
    pos = np.array([[[1.0, 1, 1], [2.0, 1, 1], [1.0, 1, 1]]])
    packed = PackedTrajectory(pos, np.diag([20.0, 20, 20]), [30, 7, 7])
    rcm = np.array([[0, 1.5], [1.5, 0]])
    with pytest.raises(ZeroDivisionError):
        hip_ctx.bad_hist(packed, rcm, [(1, 0)], np.arange(181.0))


def by_triples(by, kinds):
    """(centre, partner) species indices of a BadByCn result's columns 'B-A-B'"""
    from amof_amd import data as D
    out = []
    for name in by.columns:
        b, a, _ = name.split("-")
        out.append(tuple(-1 if x == "X" else kinds.index(D.atomic_numbers[x]) for x in (a, b)))
    return out


@pytest.mark.parametrize("path_env", [None, ("AMOF_NBR_NOFRAME", "1"), ("AMOF_NBR_SLABS", "1"), ("AMOF_NBR_KERNEL", "v1")])
def test_angles_that_sit_on_histogram_edges(hip_ctx, monkeypatch, path_env):
    """An exact simple cubic lattice with cutoff a sqrt(2): every angle is 45, 60, 90, 120, 135 or 180 degrees up to the last
    place, and the edges are whole degrees -- the bin then depends on the last bit of arccos.  numpy's, glibc's and the
    device library's arccos differ there (a soak case of round 4 found 73 angles binned differently); the kernels and the
    oracle evaluate one published algorithm (fdlibm's), so every kernel family agrees with the oracle bit for bit."""
    if path_env:
        monkeypatch.setenv(*path_env)
    m, a0 = 12, 2.7356674770047027
    g = np.array([[x, y, z] for x in range(m) for y in range(m) for z in range(m)], dtype=float) / m
    cell = np.diag([m * a0] * 3)
    rng = np.random.default_rng(3)
    numbers = np.where(rng.uniform(size=len(g)) < 0.5, 30, 7)
    packed = PackedTrajectory((g @ cell)[None], cell, numbers)
    kinds, sp = H.species_of(packed.numbers)
    for rc in (a0 * np.sqrt(2.0), np.nextafter(a0, 9.0)):
        rcm = np.full((2, 2), rc)
        rcm[1, 1] = 0.0
        triples = [(0, 0), (-1, 0), (-1, 1), (1, 0), (0, -1), (1, -1), (-1, -1)]
        edges = np.arange(182) * 1.0
        got = hip_ctx.bad_hist(packed, rcm, triples, edges)
        ref = clib.bad_hist(packed.pos, packed.cell, sp, 2, rcm, triples, edges)
        assert np.array_equal(got[1], ref[1])
        assert np.array_equal(got[0], ref[0]), (hip_ctx.last_path(), int(np.abs(got[0].astype(np.int64) - ref[0].astype(np.int64)).sum()))
        assert ref[0][:, [45, 60, 90, 120, 135]].sum() + ref[0][:, [44, 59, 89, 119, 134]].sum() > 0


def test_more_neighbours_than_the_lds_lists_hold(hip_ctx):
    """The reference has no limit on the neighbours of a centre (amof/bad.py:87-100).  A dense gas with ~100
    neighbours per atom exceeds the 32-entry LDS lists of the BAD kernels: the call goes through the big-list pass
    (lists in global memory) and still equals the oracle; BadByCn keys by the true neighbour count."""
    packed = H.random_gas(200, [6.0, 6.0, 6.0], np.array([1, 8] * 100), 5, F=3)
    kinds, sp = H.species_of(packed.numbers)
    rcm = np.array([[2.9, 2.2], [2.2, 2.9]])
    triples = [(0, 0), (1, 0), (0, -1), (-1, -1)]
    edges = np.arange(181.0)
    h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, 2, rcm, triples, edges)
    h_gpu, a_gpu = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_exact_biglist"
    assert np.array_equal(a_gpu, a_ref) and np.array_equal(h_gpu, h_ref)
    pa = clib.cn_counts(packed.pos, packed.cell, sp, 2, rcm, [(0, 0), (0, 1)], per_atom=True)[1]
    assert pa.max() > 32
    # a non-periodic axis and a cutoff that needs further images take the exact kernel first, then the same pass
    packed.pbc = np.array([True, True, False])
    h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, 2, rcm, triples, edges, pbc=packed.pbc)
    h_gpu, a_gpu = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_exact_biglist"
    assert np.array_equal(a_gpu, a_ref) and np.array_equal(h_gpu, h_ref)
    packed.pbc = np.array([True, True, True])
    cn_max = 150
    hb_ref, ab_ref = clib.bad_hist_by_cn(packed.pos, packed.cell, sp, 2, rcm, triples[:2], edges, cn_max)
    hb_gpu, ab_gpu = hip_ctx.bad_hist_by_cn(packed, rcm, triples[:2], edges, cn_max=cn_max)
    assert np.array_equal(ab_gpu, ab_ref) and np.array_equal(hb_gpu, hb_ref)
    assert ab_ref[:, 33:].any() and not ab_ref[:, cn_max].any()
    # the class retries with a larger cn_max instead of failing
    from amof_amd.bad import BadByCn
    from amof_amd.frames import Frame
    frames = [Frame(packed.numbers, packed.pos[k], packed.cell_of(k)) for k in range(2)]
    by = BadByCn.from_trajectory(frames, {'H-O': 2.9}, dtheta=1.0)
    assert max(by.bad["O-H-O"]) > 16              # ~47 O within 2.9 A of an H in this gas
    # ... ONCE: the slots of the second pass come from a count pass (largest neighbour count of any centre), not from x4 steps
    rc2 = np.array([[0.0, 2.9], [2.9, 0.0]])
    largest = clib.cn_counts(packed.pos[:2], packed.cell, sp, 2, rc2, [(0, 1), (1, 0)], per_atom=True)[1].max()
    assert by.passes == 2 and by.hist.shape[1] - 1 == largest + 1 and not by.n_angles[:, -1].any()
    hb_ref, ab_ref = clib.bad_hist_by_cn(packed.pos[:2], packed.cell, sp, 2, rc2,
                                         by_triples(by, kinds), np.arange(182) * 1.0, by.hist.shape[1] - 1)
    assert np.array_equal(by.hist, hb_ref) and np.array_equal(by.n_angles, ab_ref)


def test_more_angle_bins_than_lds_holds(hip_ctx):
    """dtheta = 0.005 degrees -> 36 000 bins (the LDS histogram holds 20 480): counted with global atomics"""
    packed = H.random_walk(H.zif4_frame(), 3, 0.05, 12)
    kinds, sp = H.species_of(packed.numbers)
    rcm = np.zeros((4, 4))
    rcm[kinds.index(30), kinds.index(7)] = rcm[kinds.index(7), kinds.index(30)] = 2.5
    bins = int(180 // 0.005)
    edges = np.arange(bins + 2) * 0.005
    triples = [(kinds.index(30), kinds.index(7)), (kinds.index(7), kinds.index(30))]
    h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, 4, rcm, triples, edges)
    h_gpu, a_gpu = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert len(edges) - 1 > 20480 and hip_ctx.last_path() in ("bad_fast", "bad_frame")
    assert np.array_equal(a_gpu, a_ref) and np.array_equal(h_gpu, h_ref)


def test_many_bins_uses_global_histogram_path(hip_ctx):
    packed = H.random_walk(H.zif4_frame(), 2, 0.05, 9)
    nb = _hip.load_library() and 40000                     # > AMOF_MAX_LDS_BINS
    h, _, _ = hip_ctx.rdf_accumulate(packed, 7.0, nb)
    assert hip_ctx.last_path() == "rdf_exact"               # (the global-histogram variant of the exact kernels)
    assert np.array_equal(h, _oracle_rdf(packed, 7.0, nb))


def test_non_periodic_axis(hip_ctx):
    packed = H.random_gas(150, [8.0, 9.0, 10.0], np.array([1, 8] * 75), 4, F=2)
    packed.pbc = np.array([True, False, True])
    h, _, _ = hip_ctx.rdf_accumulate(packed, 4.0, 200)
    assert np.array_equal(h, _oracle_rdf(packed, 4.0, 200))
    kinds, sp = H.species_of(packed.numbers)
    rcm = np.array([[0, 1.9], [1.9, 1.5]])
    sets = [(0, 1), (1, 0), (1, 1)]
    assert np.array_equal(hip_ctx.cn_count(packed, rcm, sets),
                          clib.cn_counts(packed.pos, packed.cell, sp, 2, rcm, sets, pbc=packed.pbc))


def test_non_finite_coordinates_never_count(hip_ctx):
    # a NaN or infinite coordinate makes every distance of that atom NaN in the canonical
    # arithmetic: it is never a neighbour and never binned; everything else is unaffected
    packed = H.random_walk(H.replicate(H.zif4_frame(), (1, 1, 2)), 3, 0.05, 77)
    pos = packed.pos.copy()
    pos[0, 5, 1] = np.nan
    pos[1, 300, 0] = np.inf
    pos[2, 17] = [-np.inf, np.nan, 1.0]
    bad = PackedTrajectory(pos, packed.cell, packed.numbers)
    kinds, sp = H.species_of(bad.numbers)
    S = len(kinds)
    h, _, _ = hip_ctx.rdf_accumulate(bad, 6.0, 600)
    assert np.array_equal(h, clib.rdf_hist(bad.pos, bad.cell, sp, S, 6.0, 600)[0])
    clean, _, _ = hip_ctx.rdf_accumulate(packed, 6.0, 600)
    assert 0 < int(clean.sum()) - int(h.sum()) < 3 * 2 * 400          # only the three atoms' pairs are gone
    rcm = np.full((S, S), 2.0)
    sets = [(a, b) for a in range(S) for b in range(S)]
    assert np.array_equal(hip_ctx.cn_count(bad, rcm, sets), clib.cn_counts(bad.pos, bad.cell, sp, S, rcm, sets))


def test_small_skewed_cell_counts_images(hip_ctx):
    cell = np.array([[4.0, 0, 0], [2.5, 3.5, 0], [1.0, 1.5, 3.0]])
    rng = np.random.default_rng(12)
    N = 40
    packed = PackedTrajectory((rng.uniform(-1, 2, (3, N, 3))) @ cell, cell, rng.choice([6, 7], N))
    for rmax, nb in [(2.0, 100), (5.5, 275)]:             # 5.5 > every perpendicular height
        h, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
        assert np.array_equal(h, _oracle_rdf(packed, rmax, nb))
    kinds, sp = H.species_of(packed.numbers)
    rcm = np.array([[3.2, 2.1], [2.1, 0.0]])
    sets = [(0, 0), (0, 1), (1, 0)]
    s_gpu, pa_gpu = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    s_cpu, pa_cpu = clib.cn_counts(packed.pos, packed.cell, sp, 2, rcm, sets, per_atom=True)
    assert np.array_equal(s_gpu, s_cpu) and np.array_equal(pa_gpu, pa_cpu)


def test_permutation_and_frame_order_invariance(hip_ctx):
    packed = H.random_walk(H.zif4_frame(), 6, 0.05, 13, cell_jitter=0.005)
    rng = np.random.default_rng(0)
    h0, v0, _ = hip_ctx.rdf_accumulate(packed, 7.0, 350)
    pa = rng.permutation(packed.n_atoms)
    pf = rng.permutation(packed.n_frames)
    shuffled = PackedTrajectory(packed.pos[pf][:, pa], packed.cell[pf], packed.numbers[pa])
    h1, v1, _ = hip_ctx.rdf_accumulate(shuffled, 7.0, 350)
    assert np.array_equal(h0, h1)
    # linearity in frames: hist(A + B) = hist(A) + hist(B)
    a, _, _ = hip_ctx.rdf_accumulate(packed, 7.0, 350, frame_range=(0, 2))
    b, _, _ = hip_ctx.rdf_accumulate(packed, 7.0, 350, frame_range=(2, 6))
    assert np.array_equal(a + b, h0)
    # idempotence: same call twice gives the same integers
    assert np.array_equal(hip_ctx.rdf_accumulate(packed, 7.0, 350)[0], h0)


@pytest.mark.parametrize("frames", [64, 69, 71, 79, 133])
def test_frame_counts_that_do_not_divide_by_the_xcds(hip_ctx, frames):
    """the tile kernel deals its frame chunks to the 8 XCDs and the frames % 8 behind them one grid row each
    (csrc/rdf.hip rdf_tile_kernel_fast): every frame exactly once -- against the oracle, and added up over ranges that take the
    unmapped grid (< 32 chunks)"""
    import torch
    packed = H.device_walk(torch.device("cuda", 0), (1, 2, 2), frames, 0.05, 77 + frames)       # 1088 atoms
    rmax = float(np.min(packed.cell_lengths()) / 2)
    nb = int(rmax // 0.02)
    full, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
    assert hip_ctx.last_path().startswith("rdf_tile")
    cut = frames - frames % 8 - 8
    parts = hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(0, cut))[0] + \
        sum(hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(k, min(k + 5, frames)))[0] for k in range(cut, frames, 5))
    assert np.array_equal(parts, full)
    with H_env(AMOF_RDF_NOTAIL="1"):
        assert np.array_equal(hip_ctx.rdf_accumulate(packed, rmax, nb)[0], full)
    assert np.array_equal(_oracle_rdf(packed, rmax, nb), full)


def test_msd_atom_sharding_adds_up(hip_ctx):
    packed = H.random_walk(H.zif4_frame(), 30, 0.2, 14)
    w = np.arange(0, 15, 2)
    full, _ = hip_ctx.msd_window(packed, w)
    parts = sum(hip_ctx.msd_window(packed, w, atom_range=r)[0] for r in [(0, 100), (100, 101), (101, 272)])
    np.testing.assert_allclose(parts, full, rtol=1e-13)


def test_msd_changing_cell_and_long_windows(hip_ctx):
    packed = H.random_walk(H.zif4_frame(), 70, 0.3, 15, cell_jitter=0.004)
    window, _ = no.msd_window_setup(70, delta_time=1, timestep=1)
    for unwrap in (False, True):
        sumsq, kinds = hip_ctx.msd_window(packed, window, unwrap=unwrap)
        elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window,
                                           unwrap=unwrap)
        for e, r in zip(elements, ref):
            got = sumsq[kinds.index(int(e))] / (packed.numbers == e).sum() / (70 - window)
            np.testing.assert_allclose(got, r, rtol=1e-9, atol=1e-12)


def test_msd_long_trajectory_uses_global_path(hip_ctx):
    # F + W beyond the LDS-resident limit (19 200): scan in global memory, windows from L2
    rng = np.random.default_rng(5)
    F, n = 21000, 6
    cell = np.diag([9.0, 10.0, 11.0])
    pos = np.cumsum(rng.normal(scale=0.2, size=(F, n, 3)), axis=0) + 4.0
    s = pos / np.diag(cell)
    packed = PackedTrajectory((s - np.floor(s)) * np.diag(cell), cell, [1, 1, 1, 8, 8, 30])
    window = np.array([0, 1, 7, 500, 9999, 20000, 20999], dtype=np.int32)
    for unwrap in (False, True):
        sumsq, kinds = hip_ctx.msd_window(packed, window, unwrap=unwrap)
        assert hip_ctx.last_path() == "msd_global"
        elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window, unwrap=unwrap)
        for e, r in zip(elements, ref):
            got = sumsq[kinds.index(int(e))] / (packed.numbers == e).sum() / (F - window)
            np.testing.assert_allclose(got, r, rtol=1e-9, atol=1e-12)
    # evenly spaced windows on the same long trajectory: the comb kernel on globally scanned columns
    for d, wmax in [(100, 105), (1000, 10), (7, 250), (333, 31)]:
        wap = (np.arange(wmax) * d).astype(np.int32)
        wap = wap[wap < F]
        for unwrap in (False, True):
            sumsq, kinds = hip_ctx.msd_window(packed, wap, unwrap=unwrap)
            assert hip_ctx.last_path() == "msd_comb_global"
            with H_env(AMOF_MSD_NOCOMB="1"):
                generic, _ = hip_ctx.msd_window(packed, wap, unwrap=unwrap)
                assert hip_ctx.last_path() == "msd_global"
            np.testing.assert_allclose(sumsq, generic, rtol=1e-11, atol=1e-12)
        elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, wap)
        sumsq, kinds = hip_ctx.msd_window(packed, wap)
        for e, r in zip(elements, ref):
            got = sumsq[kinds.index(int(e))] / (packed.numbers == e).sum() / (F - wap)
            np.testing.assert_allclose(got, r, rtol=1e-9, atol=1e-12)
    # and many windows on a shorter one (W > 32: generic LDS kernel)
    short = PackedTrajectory(packed.pos[:600], cell, packed.numbers)
    w2 = np.arange(0, 300, dtype=np.int32)
    sumsq, kinds = hip_ctx.msd_window(short, w2)
    elements, ref = no.window_msd_fast(short.pos, short.cell, short.numbers, short.masses, w2)
    for e, r in zip(elements, ref):
        np.testing.assert_allclose(sumsq[kinds.index(int(e))] / (short.numbers == e).sum() / (600 - w2), r, rtol=1e-9, atol=1e-12)


def test_headline_shape_properties(hip_ctx):
    """cfg3-sized frames (9792 atoms): oracle on 1 frame, properties on more."""
    base = H.replicate(H.zif4_frame(), (3, 3, 4))
    packed = H.random_walk(base, 6, 0.05, 16, ortho=True)
    rdf = Rdf.from_trajectory(packed)
    assert len(rdf.data) == 2310 and packed.n_atoms == 9792
    one = PackedTrajectory(packed.pos[:1], packed.cell, packed.numbers)
    kinds, sp = H.species_of(packed.numbers)
    h_cpu, _ = clib.rdf_hist(one.pos, one.cell, sp, 4, rdf.rmax, 2310, cell_list=True)
    h_gpu, _, _ = hip_ctx.rdf_accumulate(one, rdf.rmax, 2310)
    assert np.array_equal(h_gpu, h_cpu)
    # symmetry + checksum of checksums: per-frame histograms add up to the trajectory's
    acc = sum(hip_ctx.rdf_accumulate(packed, rdf.rmax, 2310, frame_range=(k, k + 1))[0] for k in range(6))
    assert np.array_equal(acc, rdf.hist) and np.array_equal(rdf.hist, rdf.hist.transpose(1, 0, 2))
    # MSD vs the vectorised oracle at full width
    msd = WindowMsd.from_trajectory(packed, delta_time=1, timestep=1)
    window, _ = no.msd_window_setup(6, 1, "half", 1)
    el, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window)
    from amof_amd import data as eldata
    for e, r in zip(el, ref):
        np.testing.assert_allclose(msd.data[eldata.chemical_symbols[int(e)]].values, r, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("jitter", [0.0, 0.003])
def test_msd_two_pass_form_and_its_rare_columns(hip_ctx, jitter):
    """Round 4: diagonal cells fold the centre of mass into the transposition (pos read once) and the window kernels subtract
    C[k] from the scanned RAW columns -- unless an entry lies within the centre-of-mass step of half the cell, in which case
    the column is corrected entry by entry with the wrap arithmetic.  A gas whose atoms jump anywhere in the box from frame
    to frame puts such entries into every column (and gives the centre of mass a large step); a quiet walk has none.  Every
    window-kernel family, against the numpy restatement and against the 3-pass form (AMOF_MSD_NOFOLD)."""
    rng = np.random.default_rng(77)
    F, N = 330, 40
    cell = np.diag([9.0, 11.0, 13.0])
    numbers = np.array([1, 8] * (N // 2))
    cells = np.array([cell * (1 + jitter * rng.normal()) for _ in range(F)]) if jitter else cell
    gas = PackedTrajectory(rng.uniform(0, 1, (F, N, 3)) @ cell, cells, numbers)
    walk = H.random_walk(Frame(numbers, rng.uniform(0, 1, (N, 3)) @ cell, cell), F, 0.05, 3, cell_jitter=jitter, ortho=True)
    for packed in (gas, walk):
        for window, want in ((np.arange(0, 160, 64), "msd_stream"), (np.arange(0, 160, 8), "msd_comb"),
                             (np.array([0, 3, 50, 161]), "msd_group")):
            window = window.astype(np.int32)
            got, kinds = hip_ctx.msd_window(packed, window)
            # round 5: evenly spaced windows (spacing >= 16) of a diagonal cell go to the fused form first; the gas raises
            # its flag (entries that wrap again under the centre-of-mass step) and is answered by the forms below
            fused = want == "msd_stream" and packed is walk
            assert hip_ctx.last_path() == ("msd_fused" if fused else want)
            if fused:
                with H_env(AMOF_MSD_NOFUSED="1"):
                    two_pass, _ = hip_ctx.msd_window(packed, window)
                    assert hip_ctx.last_path() == want
                np.testing.assert_allclose(got, two_pass, rtol=1e-11, atol=1e-9)
            with H_env(AMOF_MSD_NOFOLD="1", AMOF_MSD_NOFUSED="1"):
                old, _ = hip_ctx.msd_window(packed, window)
                assert hip_ctx.last_path() == want
            np.testing.assert_allclose(got, old, rtol=1e-11, atol=1e-9)
            elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window)
            for e, r in zip(elements, ref):
                g = got[kinds.index(int(e))] / (packed.numbers == e).sum() / (F - window)
                np.testing.assert_allclose(g, r, rtol=1e-9, atol=1e-12)
