"""Every kernel family has an exact (brute-force, canonical float64) path and a fast path
(fixed-point minimum image, slab culling, f32 prefilter with exact refinement).  Both must
give identical integers; the exact one is also what partially periodic / tiny cells use."""

import os

import numpy as np
import pytest

from amof_amd.frames import PackedTrajectory
from oracle import clib
from tests import helpers as H

pytestmark = pytest.mark.gpu


class _env(object):
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _traj(kind):
    z = H.zif4_frame()
    if kind == "elongated":       # 2x1x4 supercell: slab culling along z is active (2 rmax < Lz)
        return H.random_walk(H.replicate(z, (2, 1, 4)), 3, 0.05, 31, ortho=True), 6.0, 600
    if kind == "elongated_tri":
        return H.random_walk(H.replicate(z, (1, 1, 3)), 3, 0.05, 32), 5.0, 417
    if kind == "npt":
        return H.random_walk(H.replicate(z, (1, 1, 2)), 6, 0.05, 33, cell_jitter=0.01), 7.0, 700
    return H.random_walk(z, 4, 0.05, 34), 7.7, 770


@pytest.mark.parametrize("kind", ["elongated", "elongated_tri", "npt", "cubicish"])
def test_rdf_fast_equals_exact_equals_oracle(hip_ctx, kind):
    packed, rmax, nb = _traj(kind)
    with _env(AMOF_RDF_NOCELL="1", AMOF_RDF_NORANGE="1"):
        fast, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
        # diagonal cells: the variant with f32 slab coordinates and the always-add histogram; general cells: the same in
        # the orthogonalised lattice frame (round 4); AMOF_RDF_NOZF / AMOF_RDF_NOTRI name the plain tile kernel
        diagonal = bool(np.all(packed.cell == packed.cell * np.eye(3)))
        assert hip_ctx.last_path() == ("rdf_tile_zf" if diagonal else "rdf_tile_tri")
        with _env(AMOF_RDF_NOZF="1", AMOF_RDF_NOTRI="1"):
            nozf, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() == "rdf_tile"
        assert np.array_equal(fast, nozf)
        with _env(AMOF_RDF_NOCULL="1"):
            nocull, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() == ("rdf_tile_zf" if diagonal else "rdf_tile_tri")
            with _env(AMOF_RDF_NOZF="1", AMOF_RDF_NOTRI="1"):
                nocull_nozf, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
                assert hip_ctx.last_path() == "rdf_tile"
            assert np.array_equal(nocull, nocull_nozf)
    with _env(AMOF_RDF_KERNEL="v1"):
        exact, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
        assert hip_ctx.last_path() == "rdf_exact"
    kinds, sp = H.species_of(packed.numbers)
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
    assert np.array_equal(fast, ref) and np.array_equal(nocull, ref) and np.array_equal(exact, ref)


def test_rdf_lattice_pairs_on_bin_edges(hip_ctx):
    # every distance of a perfect lattice sits on (or within rounding of) a bin edge: the
    # fast path must push all of them through the exact refinement and still agree
    a, n = 2.0, 8
    g = np.arange(n) * a
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    packed = PackedTrajectory(np.stack([pos, pos + 0.25]), np.diag([n * a] * 3), np.ones(len(pos), int))
    for rmax, nb in [(7.9, 79), (8.0, 800), (6.0, 6)]:
        h, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, np.zeros(len(pos), np.int32), 1, rmax, nb)
        assert np.array_equal(h, ref)


def test_sheared_lattice_pairs_on_edges(hip_ctx):
    # integer grid 2*Z^3 inside a strongly sheared cell with integer cell vectors: the Cartesian
    # components are sums of large cancelling terms (the kappa factor of the f32 guard), and the
    # distances 2, 4, 6, 2*sqrt(2) ... sit exactly on bin edges / on the cutoff
    cell = np.array([[16.0, 0, 0], [8.0, 16.0, 0], [8.0, 8.0, 16.0]])
    g = np.arange(8) * 2.0
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    numbers = np.where(np.arange(len(pos)) % 3 == 0, 30, 7)
    packed = PackedTrajectory(np.stack([pos, pos + 0.375, pos - 7.0]), cell, numbers)
    kinds, sp = H.species_of(packed.numbers)
    for rmax, nb in [(6.0, 12), (6.5, 650), (6.9, 69)]:
        h, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, 2, rmax, nb)
        assert np.array_equal(h, ref)
    for rc in [2.0, np.sqrt(8.0), np.nextafter(np.sqrt(8.0), 9.0), 4.0, np.nextafter(4.0, 9.0)]:
        rcm = np.full((2, 2), rc)
        sets = [(0, 0), (0, 1), (1, 0), (1, 1)]
        assert np.array_equal(hip_ctx.cn_count(packed, rcm, sets), clib.cn_counts(packed.pos, packed.cell, sp, 2, rcm, sets))


@pytest.mark.parametrize("kind", ["elongated", "elongated_tri", "npt"])
def test_cn_bad_fast_equals_exact_equals_oracle(hip_ctx, kind):
    packed, _, _ = _traj(kind)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    zn, n, c, h = kinds.index(30), kinds.index(7), kinds.index(6), kinds.index(1)
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.5
    rcm[c, n] = rcm[n, c] = 1.6
    rcm[c, h] = rcm[h, c] = 1.3
    rcm[c, c] = 1.7
    sets = [(zn, n), (n, zn), (c, n), (c, c), (h, c)]
    triples = [(zn, n), (n, -1), (-1, -1), (c, c), (c, -1)]
    edges = np.arange(int(180 // 0.5) + 2) * 0.5
    with _env(AMOF_NBR_NOCELL="1"):                                  # 1-D slab list
        s_fast, pa_fast = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
        assert hip_ctx.last_path() == "cn_fast"
        h_fast, a_fast = hip_ctx.bad_hist(packed, rcm, triples, edges)
        assert hip_ctx.last_path() == "bad_fast"
    with _env(AMOF_NBR_FORCE_CELL="1"):                              # 3-D cell list (species-major cell sort)
        s_cell, pa_cell = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
        assert hip_ctx.last_path() == "cn_cell"
        h_cell, a_cell = hip_ctx.bad_hist(packed, rcm, triples, edges)
        assert hip_ctx.last_path() == "bad_cell"
    with _env(AMOF_NBR_KERNEL="v1"):
        s_ex, pa_ex = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
        assert hip_ctx.last_path() == "cn_exact"
        h_ex, a_ex = hip_ctx.bad_hist(packed, rcm, triples, edges)
        assert hip_ctx.last_path() == "bad_exact"
    s_frame, pa_frame = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)      # whole frame in LDS (the default tier)
    assert hip_ctx.last_path() == "cn_frame"
    h_frame, a_frame = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_frame"
    with _env(AMOF_BAD_ROWS_MB="1"):                                 # the same through many small frame batches of rows
        h_fb, a_fb = hip_ctx.bad_hist(packed, rcm, triples, edges)
        assert hip_ctx.last_path() == "bad_frame"
    assert np.array_equal(a_fb, a_frame) and np.array_equal(h_fb, h_frame)
    with _env(AMOF_BAD_NOMERGE="1"):                                 # one angle pass per triple instead of one per centre species
        h_nm, a_nm = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert np.array_equal(a_nm, a_frame) and np.array_equal(h_nm, h_frame)
    with _env(AMOF_BAD_EDGE_TABLE="1"):                              # bin edges read from the table instead of recomputed as k * step
        h_et, a_et = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert np.array_equal(a_et, a_frame) and np.array_equal(h_et, h_frame)
    s_ref, pa_ref = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)
    h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
    assert np.array_equal(s_fast, s_ref) and np.array_equal(pa_fast, pa_ref)
    assert np.array_equal(s_cell, s_ref) and np.array_equal(pa_cell, pa_ref)
    assert np.array_equal(s_ex, s_ref) and np.array_equal(pa_ex, pa_ref)
    assert np.array_equal(s_frame, s_ref) and np.array_equal(pa_frame, pa_ref)
    assert np.array_equal(a_fast, a_ref) and np.array_equal(h_fast, h_ref)
    assert np.array_equal(a_cell, a_ref) and np.array_equal(h_cell, h_ref)
    assert np.array_equal(a_ex, a_ref) and np.array_equal(h_ex, h_ref)
    assert np.array_equal(a_frame, a_ref) and np.array_equal(h_frame, h_ref)
    assert a_ref.sum() > 0


@pytest.mark.parametrize("poison", [0x00, 0xA5, 0xFF])
def test_cell_list_sparse_cutoff_matrix_on_poisoned_scratch(hip_ctx, poison):
    """a species WITH a cutoff followed, in sorted species order, by one WITHOUT (kinds H, C, N, Zn: 'C-N' alone
    leaves Zn unsorted behind N; 'C-H' alone leaves N and Zn): the cell kernels take the upper bound of a species'
    last cell row from the table entry behind it, which the skipped species never writes.  Scratch is poisoned
    before every call, so a stale or unwritten entry shows (zero: neighbours dropped; large: garbage ranges).
    Also per-atom counts of a set whose centre species carries no cutoff at all (its records must still be sorted)."""
    packed = H.random_walk(H.replicate(H.zif4_frame(), (2, 2, 1)), 3, 0.05, 77, ortho=True)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    h, c, n, zn = kinds.index(1), kinds.index(6), kinds.index(7), kinds.index(30)
    assert (h, c, n, zn) == (0, 1, 2, 3)
    edges = np.arange(int(180 // 0.5) + 2) * 0.5
    cases = [({(c, n): 1.6}, [(c, n), (n, c), (zn, n), (h, h)], [(c, n), (n, c), (n, -1)]),
             ({(h, c): 1.3}, [(h, c), (c, h), (zn, zn)], [(c, h), (c, -1)]),
             ({(n, n): 2.6}, [(n, n), (h, n)], [(n, n)]),
             ({(h, h): 2.0, (c, c): 1.7}, [(h, h), (c, c), (n, zn)], [(h, h), (c, c), (-1, -1)])]
    for cut, sets, triples in cases:
        rcm = np.zeros((S, S))
        for (x, y), rc in cut.items():
            rcm[x, y] = rcm[y, x] = rc
        s_ref, pa_ref = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)
        h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
        with _env(AMOF_NBR_FORCE_CELL="1"):
            hip_ctx.debug_poison(poison)
            s_cell, pa_cell = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
            assert hip_ctx.last_path() in ("cn_cell", "cn_frame")
            hip_ctx.debug_poison(poison)
            s_only = hip_ctx.cn_count(packed, rcm, sets)
            hip_ctx.debug_poison(poison)
            h_cell, a_cell = hip_ctx.bad_hist(packed, rcm, triples, edges)
            assert hip_ctx.last_path() == "bad_cell"
        assert np.array_equal(s_cell, s_ref) and np.array_equal(pa_cell, pa_ref), cut
        assert np.array_equal(s_only, s_ref), cut
        assert np.array_equal(a_cell, a_ref) and np.array_equal(h_cell, h_ref), cut
        assert s_ref.sum() > 0


def test_bad_fast_list_overflow_takes_the_lds_exact_kernel_first(hip_ctx):
    """a centre with 17 .. 32 neighbours overflows the fast kernels' 16-entry lists: the call is redone by the exact
    kernel with its 32-deep LDS lists ("bad_exact"), not by the count + global big-list passes; beyond 32: big list"""
    packed = H.random_walk(H.zif4_frame(), 2, 0.05, 78)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    edges = np.arange(int(180 // 1.0) + 2) * 1.0
    sets = [(a, b) for a in range(S) for b in range(S)]
    for rc, want in ((3.4, "bad_exact"), (5.0, "bad_exact_biglist")):
        rcm = np.full((S, S), rc)
        triples = [(-1, -1)]
        pa = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)[1]
        most = np.where(pa < 0, 0, pa).sum(axis=1).max()          # fullest centre, all partner species together
        assert (16 < most <= 32) if want == "bad_exact" else most > 32
        hb, ab = hip_ctx.bad_hist(packed, rcm, triples, edges)
        assert hip_ctx.last_path() == want, (rc, hip_ctx.last_path())
        hr, ar = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
        assert np.array_equal(ab, ar) and np.array_equal(hb, hr)


@pytest.mark.parametrize("kind", ["hot", "intact", "npt_tri"])
def test_bad_transposed_lists_equal_the_second_search(hip_ctx, kind):
    """both triples of a species pair (N-Zn-N and Zn-N-Zn, C-N-C and N-C-N ...): only the side with fewer centres is
    searched, the other side's angles come from the transposed lists -- identical counts to searching both sides
    (AMOF_BAD_NOTRANSPOSE=1) and to the oracle, on the cell-list and the slab-list kernels, with BadByCn's cn keys, on a
    hot system where N atoms hold 0 .. 3 Zn, and when a list overflows (generous cutoff: exact kernels take over)"""
    if kind == "hot":
        packed = H.random_walk(H.replicate(H.zif4_frame(), (2, 1, 2)), 5, 0.3, 61, ortho=True)
    elif kind == "intact":
        packed = H.random_walk(H.replicate(H.zif4_frame(), (2, 2, 1)), 3, 0.02, 62, ortho=True)
    else:
        packed = H.random_walk(H.replicate(H.zif4_frame(), (1, 2, 2)), 4, 0.2, 63, cell_jitter=0.01)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    h, c, n, zn = (kinds.index(z) for z in (1, 6, 7, 30))
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.6
    rcm[c, n] = rcm[n, c] = 1.7
    rcm[c, h] = rcm[h, c] = 1.35
    triples = [(zn, n), (n, zn), (c, n), (n, c), (h, c), (c, h), (n, -1)]
    edges = np.arange(int(180 // 1.0) + 2) * 1.0
    h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
    assert a_ref[1] > 0 or kind == "intact"                     # Zn-N-Zn angles exist in the hot systems
    for env, path in (({"AMOF_NBR_FORCE_CELL": "1"}, "bad_cell"), ({"AMOF_NBR_NOCELL": "1"}, "bad_fast")):
        with _env(**env):
            hip_ctx.debug_poison(0xA5)
            h_tr, a_tr = hip_ctx.bad_hist(packed, rcm, triples, edges)
            assert hip_ctx.last_path() == path
            with _env(AMOF_BAD_NOTRANSPOSE="1"):
                h_two, a_two = hip_ctx.bad_hist(packed, rcm, triples, edges)
            by_tr = hip_ctx.bad_hist_by_cn(packed, rcm, triples, edges, cn_max=6)
        assert np.array_equal(a_tr, a_ref) and np.array_equal(h_tr, h_ref), (kind, path)
        assert np.array_equal(a_two, a_ref) and np.array_equal(h_two, h_ref)
        by_ref = clib.bad_hist_by_cn(packed.pos, packed.cell, sp, S, rcm, triples, edges, 6)
        assert np.array_equal(by_tr[1], by_ref[1]) and np.array_equal(by_tr[0], by_ref[0]), (kind, path)
    # a list fuller than its capacity (17+ partners of the other species): the exact kernels answer
    big = np.zeros((S, S))
    big[h, c] = big[c, h] = 4.6
    hb, ab = hip_ctx.bad_hist(packed, big, [(h, c), (c, h)], edges, frame_range=(0, 1))
    _, pa = clib.cn_counts(packed.pos[:1], packed.cell[:1], sp, S, big, [(h, c), (c, h)], per_atom=True)
    if pa.max() > 16:
        assert hip_ctx.last_path() in ("bad_exact", "bad_exact_biglist")
    hr, ar = clib.bad_hist(packed.pos[:1], packed.cell[:1], sp, S, big, [(h, c), (c, h)], edges)
    assert np.array_equal(ab, ar) and np.array_equal(hb, hr)


def test_cell_list_neighbours_on_lattices_and_sheared_cells(hip_ctx):
    """the 3-D cell-list CN / BAD kernels where the cell grid is tight: atoms exactly on cell faces (integer
    lattice, cells a whole number of lattice constants thick), pairs exactly at the cutoff, a strongly sheared
    cell, three cells per axis (every neighbour cell distinct only just), several species and an 'X' triple"""
    a, n = 2.0, 9
    g = np.arange(n) * a
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    numbers = np.where(np.arange(len(pos)) % 3 == 0, 30, np.where(np.arange(len(pos)) % 3 == 1, 7, 6))
    edges = np.arange(int(180 // 1.0) + 2) * 1.0
    for cell in (np.diag([n * a] * 3), np.array([[n * a, 0, 0], [4 * a, n * a, 0], [2 * a, 6 * a, n * a]])):
        packed = PackedTrajectory(np.stack([pos, pos + 0.5, pos - 11.0]), cell, numbers)
        kinds, sp = H.species_of(packed.numbers)
        for rc in (a, np.nextafter(a, 9.0), a * np.sqrt(2.0), np.nextafter(a * np.sqrt(2.0), 9.0), 2.9):
            rcm = np.full((3, 3), rc)
            rcm[0, 0] = 0.0
            sets = [(x, y) for x in range(3) for y in range(3)]
            triples = [(2, 1), (1, -1), (-1, -1)]
            s_ref, pa_ref = clib.cn_counts(packed.pos, packed.cell, sp, 3, rcm, sets, per_atom=True)
            h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, 3, rcm, triples, edges)
            with _env(AMOF_NBR_FORCE_CELL="1"):
                s_cell, pa_cell = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
                path_cn = hip_ctx.last_path()
                h_cell, a_cell = hip_ctx.bad_hist(packed, rcm, triples, edges)
                path_bad = hip_ctx.last_path()
            assert np.array_equal(s_cell, s_ref) and np.array_equal(pa_cell, pa_ref), (rc, path_cn)
            assert np.array_equal(a_cell, a_ref) and np.array_equal(h_cell, h_ref), (rc, path_bad)
            if np.array_equal(cell, np.diag(np.diag(cell))):
                # (a centre with 17 .. 32 neighbours sends the call to the exact kernel's LDS lists, beyond that to the big list)
                assert path_cn == "cn_cell" and path_bad in ("bad_cell", "bad_exact", "bad_exact_biglist")
            # the whole-frame-in-LDS tier on the same inputs (sets with a cutoff only: a zero-cutoff set with per-atom
            # output keeps the gather kernels)
            live = [(x, y) for (x, y) in sets if rcm[x, y] > 0]
            s_fr, pa_fr = hip_ctx.cn_count(packed, rcm, live, per_atom=True)
            assert hip_ctx.last_path() == "cn_frame"
            keep = [k for k, xy in enumerate(sets) if rcm[xy] > 0]
            assert np.array_equal(s_fr, s_ref[:, keep]) and np.array_equal(pa_fr, pa_ref[:, keep]), rc
            h_fr, a_fr = hip_ctx.bad_hist(packed, rcm, triples, edges)
            assert hip_ctx.last_path() in ("bad_frame", "bad_exact", "bad_exact_biglist")
            assert np.array_equal(a_fr, a_ref) and np.array_equal(h_fr, h_ref), rc


def test_pairs_exactly_at_the_cutoff(hip_ctx):
    # simple cubic lattice, cutoff exactly the lattice constant: strict '<' excludes the shell;
    # one ulp more includes it.  The f32 prefilter cannot tell: the exact re-decision must.
    a, n = 2.0, 8
    g = np.arange(n) * a
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    packed = PackedTrajectory(pos[None] + 0.125, np.diag([n * a] * 3), np.ones(len(pos), int))
    for rc, want in [(a, 0), (np.nextafter(a, 10.0), 6), (a * np.sqrt(2.0), 6), (1.5 * a, 18)]:
        sums = hip_ctx.cn_count(packed, [[rc]], [(0, 0)])
        ref = clib.cn_counts(packed.pos, packed.cell, np.zeros(len(pos), np.int32), 1, [[rc]], [(0, 0)])
        assert sums[0, 0] == ref[0, 0]
        if want is not None and rc != a * np.sqrt(2.0):
            assert sums[0, 0] == want * len(pos)


def test_far_away_atoms_fall_back(hip_ctx):
    # coordinates > 1e4 cells from the origin defeat the fixed-point fold: the library must
    # notice and answer through the exact kernels
    packed = H.random_walk(H.zif4_frame(), 2, 0.05, 35)
    far = PackedTrajectory(packed.pos + np.array([2.0e5 * packed.cell[0, 0, 0], 0, 0]), packed.cell, packed.numbers)
    kinds, sp = H.species_of(packed.numbers)
    h, _, _ = hip_ctx.rdf_accumulate(far, 6.0, 60)
    assert hip_ctx.last_path() == "rdf_exact"
    ref, _ = clib.rdf_hist(far.pos, far.cell, sp, len(kinds), 6.0, 60)
    assert np.array_equal(h, ref)
    rcm = np.zeros((4, 4)); rcm[2, 3] = rcm[3, 2] = 2.5
    assert np.array_equal(hip_ctx.cn_count(far, rcm, [(3, 2)]), clib.cn_counts(far.pos, far.cell, sp, 4, rcm, [(3, 2)]))
    edges = np.arange(181.0)
    hb, ab = hip_ctx.bad_hist(far, rcm, [(3, 2)], edges)
    hr, ar = clib.bad_hist(far.pos, far.cell, sp, 4, rcm, [(3, 2)], edges)
    assert np.array_equal(ab, ar) and np.array_equal(hb, hr)


def test_bad_by_cn(hip_ctx):
    from amof_amd.bad import Bad, BadByCn
    packed = H.random_walk(H.replicate(H.zif4_frame(), (1, 1, 2)), 6, 0.25, 41)     # hot: mixed coordination numbers
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    zn, n, c = kinds.index(30), kinds.index(7), kinds.index(6)
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.6
    rcm[c, n] = rcm[n, c] = 1.7
    triples = [(zn, n), (n, -1), (-1, -1)]
    edges = np.arange(int(180 // 1.0) + 2) * 1.0
    for env in ({}, {"AMOF_NBR_KERNEL": "v1"}):
        with _env(**env):
            h, a = hip_ctx.bad_hist_by_cn(packed, rcm, triples, edges, cn_max=8)
        hr, ar = clib.bad_hist_by_cn(packed.pos, packed.cell, sp, S, rcm, triples, edges, 8)
        assert np.array_equal(a, ar) and np.array_equal(h, hr)
        assert (ar[:, 2:] > 0).sum() >= 4 and ar[:, :2].sum() == 0          # several cn values populated, none below 2
    # summing over cn gives the plain histogram
    h0, a0 = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert np.array_equal(h.sum(axis=1), h0) and np.array_equal(a.sum(axis=1), a0)
    # class level: 'partial' weights make the per-cn BADs add up to Bad's column
    cut = {'Zn-N': 2.6, 'C-N': 1.7}
    by = BadByCn.from_trajectory(packed, cut, dtheta=1.0, normalization='partial')
    plain = Bad.from_trajectory(packed, cut, dtheta=1.0)
    assert set(by.bad) == set(plain.data.columns) - {"theta"}
    for col, per_cn in by.bad.items():
        np.testing.assert_allclose(sum(per_cn.values()), plain.data[col].values, rtol=1e-12, atol=1e-15)
    tot = BadByCn.from_trajectory(packed, cut, dtheta=1.0)
    for col, per_cn in tot.bad.items():
        for cn, dens in per_cn.items():
            assert (dens * np.diff(np.arange(182) * 1.0)).sum() == pytest.approx(1.0, rel=1e-12)


def test_direct_msd_matches_reference_restatement(hip_ctx):
    from amof_amd.msd import DirectMsd
    from amof_amd import data as eldata
    from oracle import numpy_oracle as no
    packed = H.random_walk(H.zif4_frame(), 25, 0.3, 43, ortho=True, cell_jitter=0.005)
    d = DirectMsd.from_trajectory(packed, delta_Step=5, first_frame=100)
    assert hip_ctx.last_path() == "msd_direct"
    elements, ref = no.direct_msd(packed.pos, packed.cell, packed.numbers)
    assert list(d.data.columns)[:2] == ["Step", "X"] and d.data["Step"][3] == 115
    np.testing.assert_allclose(d.data["X"].values, ref[None], rtol=1e-9, atol=1e-12)
    for e in elements:
        np.testing.assert_allclose(d.data[eldata.chemical_symbols[int(e)]].values, ref[e], rtol=1e-9, atol=1e-12)
    # wrapped random walk: the running unwrap recovers the true displacement
    assert d.data["X"].values[-1] > 1.0


@pytest.mark.parametrize("tri", [False, True])
def test_rdf_range_kernel_two_level_cell_list(hip_ctx, tri):
    # long cell, small cutoff: the 2-level (slab x y-bin) range kernel is selected; it must agree
    # with the 1-D slab path, the exact kernels and the oracle
    base = H.replicate(H.zif4_frame(), (2, 2, 5))            # 5440 atoms, 30.8 x 30.8 x 92.2 A
    packed = H.random_walk(base, 3, 0.08, 61, ortho=not tri)
    kinds, sp = H.species_of(packed.numbers)
    for rmax, nb in [(5.0, 500), (9.5, 333)]:
        with _env(AMOF_RDF_FORCE_RANGE="1", AMOF_RDF_NOCELL="1"):
            got, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() == "rdf_range"
        with _env(AMOF_RDF_NORANGE="1", AMOF_RDF_NOCELL="1"):
            slab, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() in ("rdf_tile_tri", "rdf_tile_zf")
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
        assert np.array_equal(got, ref) and np.array_equal(slab, ref)
    # exactly three slabs (nz = 3): the forward-slab rule must not double count across the wrap
    small = H.random_walk(H.replicate(H.zif4_frame(), (2, 2, 2)), 2, 0.05, 62, ortho=not tri)   # 30.8 x 30.8 x 36.9 A
    for rmax, nb in [(5.0, 250), (12.0, 240), (12.29, 1229)]:        # nz = 7, 3, 3
        with _env(AMOF_RDF_FORCE_RANGE="1", AMOF_RDF_NOCELL="1"):
            got, _, _ = hip_ctx.rdf_accumulate(small, rmax, nb)
            assert hip_ctx.last_path() == "rdf_range"
        ref, _ = clib.rdf_hist(small.pos, small.cell, H.species_of(small.numbers)[1], 4, rmax, nb)
        assert np.array_equal(got, ref), rmax


def test_rdf_pipelined_host_staging(hip_ctx):
    # >= 1024 host-resident frames: the trajectory is copied batch by batch on a second stream
    # while the previous batch is computed; same integers as the device-resident call and the oracle
    import torch
    packed = H.random_walk(H.zif4_frame(), 1100, 0.03, 91)
    kinds, sp = H.species_of(packed.numbers)
    host, _, vol_h = hip_ctx.rdf_accumulate(packed, 7.0, 700)
    dev = PackedTrajectory(torch.as_tensor(packed.pos).cuda(), packed.cell, packed.numbers)
    on_dev, _, vol_d = hip_ctx.rdf_accumulate(dev, 7.0, 700)
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), 7.0, 700, cell_list=True)
    assert np.array_equal(host, ref) and np.array_equal(on_dev, ref) and vol_h == vol_d
    with _env(AMOF_RDF_KERNEL="v1"):                     # exact kernels: everything staged up front
        exact, _, _ = hip_ctx.rdf_accumulate(packed, 7.0, 700)
    assert np.array_equal(exact, ref)
    # CN / BAD share the pipelined staging
    S = len(kinds)
    zn, n = kinds.index(30), kinds.index(7)
    rcm = np.zeros((S, S)); rcm[zn, n] = rcm[n, zn] = 2.5
    edges = np.arange(int(180 // 0.5) + 2) * 0.5
    assert np.array_equal(hip_ctx.cn_count(packed, rcm, [(zn, n), (n, zn)]),
                          clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, [(zn, n), (n, zn)]))
    hb, ab = hip_ctx.bad_hist(packed, rcm, [(zn, n), (n, zn)], edges)
    hr, ar = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, [(zn, n), (n, zn)], edges)
    assert np.array_equal(hb, hr) and np.array_equal(ab, ar) and ar.sum() > 0


@pytest.mark.parametrize("F,d,W", [(7, 1, 3), (50, 3, 16), (333, 7, 24), (333, 11, 25), (1000, 100, 5), (1000, 31, 32),
                                   (257, 13, 12), (90, 100, 1), (64, 2, 29), (1201, 50, 21),
                                   (333, 3, 40), (1000, 7, 64), (2000, 9, 100), (700, 5, 33), (900, 11, 70),
                                   # window spacings of 64 .. 256 frames: the streaming kernel (one thread per residue
                                   # class; whole blocks, ragged comb ends, combs shorter than one block, 128 / 256 threads)
                                   (5000, 100, 25), (3000, 64, 20), (2000, 128, 12), (2600, 129, 9), (4100, 256, 8),
                                   (999, 77, 13), (1530, 100, 15), (4999, 100, 32), (650, 65, 10), (3333, 90, 24)])
def test_msd_comb_kernel_equals_generic_and_oracle(hip_ctx, F, d, W):
    # windows w*d: the comb kernel (every template bucket, ragged comb ends, F < d, the skipped origin)
    # against the generic LDS kernel and the numpy restatement
    from oracle import numpy_oracle as no
    rng = np.random.default_rng(F * 131 + d)
    n = 9
    cell = np.array([[9.0, 0, 0], [1.0, 10.0, 0], [0.5, -0.7, 11.0]])
    pos = np.cumsum(rng.normal(scale=0.25, size=(F, n, 3)), axis=0) + 4.0
    s = pos @ np.linalg.inv(cell)
    packed = PackedTrajectory((s - np.floor(s)) @ cell, cell, [1, 1, 1, 1, 8, 8, 30, 30, 30])
    window = np.array([w * d for w in range(W) if w * d < F], dtype=np.int32)
    comb, kinds = hip_ctx.msd_window(packed, window)
    streams = 64 <= d <= 256 and 2 <= len(window) <= 32
    assert hip_ctx.last_path() == ("msd_stream" if streams else ("msd_comb" if len(window) >= 2 else "msd_group"))
    if streams:
        with _env(AMOF_MSD_NOSTREAM="1"):
            blockform, _ = hip_ctx.msd_window(packed, window)
            assert hip_ctx.last_path() == "msd_comb"
        np.testing.assert_allclose(comb, blockform, rtol=1e-12, atol=1e-12)
    with _env(AMOF_MSD_NOCOMB="1"):
        generic, _ = hip_ctx.msd_window(packed, window)
        assert hip_ctx.last_path() == "msd_group"
    np.testing.assert_allclose(comb, generic, rtol=1e-12, atol=1e-12)
    elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window)
    for e, r in zip(elements, ref):
        got = comb[kinds.index(int(e))] / (packed.numbers == e).sum() / (F - window)
        np.testing.assert_allclose(got, r, rtol=1e-9, atol=1e-12)


def test_msd_device_side_merge_entry_points(hip_ctx):
    """amof_msd_com_dev + amof_msd_window_dev (what an atom-sharded RCCL run uses): the centre of mass filled in by
    frame shares into a zeroed table, the sums of two atom shares accumulated in a device buffer -- the same kernels as
    the plain call, so the same bits; a precomputed centre of mass is refused together with unwrap; volume_sum of the
    host restatement equals the library's."""
    import torch
    packed = H.random_walk(H.zif4_frame(), 600, 0.08, 91, cell_jitter=0.004)
    dev = packed.to_device(0)
    window = np.arange(0, 300, 64, dtype=np.int32)
    F, N = packed.n_frames, packed.n_atoms
    whole, kinds = hip_ctx.msd_window(dev, window)
    halves = [hip_ctx.msd_window(dev, window, atom_range=r)[0] for r in ((0, 100), (100, N))]
    com = torch.zeros((F, 3), dtype=torch.float64, device="cuda:0")
    for r in ((0, 7), (7, 311), (311, F)):
        hip_ctx.msd_com(dev, r, com)
        assert hip_ctx.last_path() == "msd_com"
    m = packed.masses
    np.testing.assert_allclose(com.cpu().numpy(), (packed.pos * m[None, :, None]).sum(axis=1) / m.sum(), rtol=1e-13)
    # a HOST trajectory with frame_begin > 0 (advisor, round 3: only the frames of the range are staged)
    com_h = torch.zeros((F, 3), dtype=torch.float64, device="cuda:0")
    for r in ((311, F), (5, 311), (0, 5)):
        hip_ctx.msd_com(packed, r, com_h)
    assert torch.equal(com_h, com)
    out = torch.zeros((len(kinds), len(window)), dtype=torch.float64, device="cuda:0")
    for r in ((0, 100), (100, N)):
        hip_ctx.msd_window(dev, window, atom_range=r, com=com, out=out)
    assert np.array_equal(out.cpu().numpy(), halves[0] + halves[1])
    np.testing.assert_allclose(out.cpu().numpy(), whole, rtol=1e-12)
    with pytest.raises(ValueError):
        hip_ctx.msd_window(dev, window, unwrap=True, com=com, out=out)
    # the frame-sharded RDF's host-side volume sum is the library's own, bit for bit
    for fr in ((0, F), (13, 402)):
        _, vol, _ = hip_ctx.rdf_accumulate(packed, 3.0, 30, frame_range=fr)
        assert vol == packed.volume_sum(fr)


def test_threads_shared_and_separate_contexts(hip_ctx):
    # a Context serialises its callers; two Contexts (own streams and scratch) run concurrently
    import threading
    from amof_amd import _hip
    trajs = [H.random_walk(H.zif4_frame(), 40 + 7 * k, 0.05, 100 + k) for k in range(4)]
    refs = []
    for tr in trajs:
        kinds, sp = H.species_of(tr.numbers)
        refs.append(clib.rdf_hist(tr.pos, tr.cell, sp, len(kinds), 7.0, 350, cell_list=True)[0])
    other = _hip.Context(0)
    errors = []

    def work(ctx, k):
        try:
            for _ in range(5):
                h, _, _ = ctx.rdf_accumulate(trajs[k], 7.0, 350)
                if not np.array_equal(h, refs[k]):
                    errors.append("mismatch in thread %d" % k)
        except Exception as exc:                       # noqa: BLE001 - reported below
            errors.append(repr(exc))

    threads = [threading.Thread(target=work, args=(hip_ctx if k < 2 else other, k)) for k in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    other.close()
    assert not errors, errors


@pytest.mark.parametrize("kind", ["ortho", "tri", "npt", "one_species", "tiny_cells"])
def test_rdf_cell_kernel_three_level_cell_list(hip_ctx, kind):
    # cutoffs far below the cell size: the 3-D cell-list kernel (forced), against the other fast paths,
    # the exact kernels and the oracle
    if kind == "one_species":
        packed = H.random_gas(3000, [40.0, 37.0, 45.0], np.full(3000, 8), 71, F=2)
        cases = [(4.0, 400), (7.3, 73)]
    elif kind == "tiny_cells":
        packed = H.random_gas(900, [30.0, 30.0, 30.0], np.array([1, 8, 30] * 300), 72, F=3)
        cases = [(1.2, 120), (5.9, 59)]
    else:
        base = H.replicate(H.zif4_frame(), (2, 2, 3))            # 3264 atoms, 30.8 x 30.8 x 55.3 A
        packed = H.random_walk(base, 3, 0.08, 73, ortho=(kind != "tri"), cell_jitter=0.01 if kind == "npt" else 0.0)
        cases = [(5.0, 500), (6.1, 2310)]
    kinds, sp = H.species_of(packed.numbers)
    for rmax, nb in cases:
        with _env(AMOF_RDF_FORCE_CELL="1"):
            got, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)           # (round 5: one wave per cell)
            assert hip_ctx.last_path() == "rdf_cell"
        with _env(AMOF_RDF_FORCE_CELL="1", AMOF_RDF_CELL_GATHER="1"):      # the per-lane gather form
            gathered, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() == "rdf_cell"
        assert np.array_equal(gathered, got), (kind, rmax, nb)
        with _env(AMOF_RDF_NOCELL="1"):
            other, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() in ("rdf_tile_tri", "rdf_tile_zf", "rdf_range")
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
        assert np.array_equal(got, ref), (kind, rmax, nb, int(got.sum()), int(ref.sum()))
        assert np.array_equal(other, ref)


def test_rdf_cell_kernel_lattice_on_bin_edges(hip_ctx):
    # integer lattice (atoms exactly on cell faces of the 3-D grid, every distance on a bin edge) in an
    # integer sheared cell: the cell-list kernel must find every pair once and refine all of them exactly
    n = 18
    g = np.arange(n) * 1.0
    pts = np.array([[x, y, z] for x in g for y in g for z in g])
    cell = np.array([[18.0, 0, 0], [3.0, 18.0, 0], [-2.0, 4.0, 18.0]])
    numbers = np.where((pts.sum(axis=1) % 2) == 0, 11, 17)           # rock-salt colouring
    packed = PackedTrajectory(np.stack([pts, pts + 0.5, pts - 3.25]), cell, numbers)
    kinds, sp = H.species_of(packed.numbers)
    for rmax, nb in [(3.0, 30), (3.0, 300), (2.0, 2), (3.5, 7)]:
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, 2, rmax, nb, cell_list=True)
        for gather in ("0", "1"):                                   # one wave per cell | one lane per centre
            with _env(AMOF_RDF_FORCE_CELL="1", **({"AMOF_RDF_CELL_GATHER": "1"} if gather == "1" else {})):
                got, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
                assert hip_ctx.last_path() == "rdf_cell"
            assert np.array_equal(got, ref), (rmax, nb, gather)


@pytest.mark.parametrize("eps,jitter", [(0.02, 0.0), (0.10, 0.0), (0.03, 0.004)])
def test_rdf_sheared_cell_default_cutoff_counts_images(hip_ctx, eps, jitter):
    # a slightly sheared cell with the reference's default rmax (half the shortest cell LENGTH) needs further
    # periodic images (rmax > half a cell HEIGHT): the image-aware tile kernel against the exact kernels and the oracle
    from amof_amd.frames import Frame
    base = H.replicate(H.zif4_frame(), (2, 2, 2))
    shear = np.eye(3) + np.array([[0, eps, eps / 2], [0, 0, eps], [0, 0, 0]])
    sheared = Frame(base.numbers, base.positions @ shear, base.cell @ shear)
    packed = H.random_walk(sheared, 3, 0.08, 97, cell_jitter=jitter)
    kinds, sp = H.species_of(packed.numbers)
    rmax = float(np.min(packed.cell_lengths()) / 2)
    for nb in (257, 1540):
        tri, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
        assert hip_ctx.last_path() == "rdf_tile_tri"          # (round 4: the orthogonalised-frame kernel takes these cells)
        with _env(AMOF_RDF_NOTRI="1"):
            got, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() == "rdf_tile_img"
            with _env(AMOF_RDF_NOIMG="1"):
                exact, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
                assert hip_ctx.last_path() == "rdf_exact"
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
        assert np.array_equal(exact, ref)
        assert np.array_equal(got, ref), (eps, nb, int(got.sum()), int(ref.sum()))
        assert np.array_equal(tri, ref), (eps, nb, int(tri.sum()), int(ref.sum()))


def test_rdf_image_aware_variant_on_a_plain_case(hip_ctx):
    # the image-aware variant forced on an input that does not need it: same integers as the plain tile kernel
    packed = H.random_walk(H.replicate(H.zif4_frame(), (2, 1, 2)), 3, 0.05, 98, ortho=True)
    with _env(AMOF_RDF_NOCELL="1", AMOF_RDF_NORANGE="1"):
        plain, _, _ = hip_ctx.rdf_accumulate(packed, 7.0, 700)
        assert hip_ctx.last_path() in ("rdf_tile", "rdf_tile_zf")
        with _env(AMOF_RDF_FORCE_IMG="1"):
            img, _, _ = hip_ctx.rdf_accumulate(packed, 7.0, 700)
            assert hip_ctx.last_path() == "rdf_tile_img"
    assert np.array_equal(plain, img)


def _zf_vs_oracle(hip_ctx, packed, rmax, nb, nocull_too=True):
    kinds, sp = H.species_of(packed.numbers)
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
    for env in ([{}, {"AMOF_RDF_NOCULL": "1"}] if nocull_too else [{}]):     # (culling forced off: the antipodal-band split)
        with _env(AMOF_RDF_NOCELL="1", AMOF_RDF_NORANGE="1", **env):
            zf, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert hip_ctx.last_path() == "rdf_tile_zf"
            with _env(AMOF_RDF_NOZF="1"):
                plain, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
                assert hip_ctx.last_path() == "rdf_tile"
        assert np.array_equal(plain, ref)
        assert np.array_equal(zf, ref), (env, rmax, nb, int(zf.sum()), int(ref.sum()), int(np.abs(zf.astype(np.int64) - ref).sum()))


def test_rdf_zf_lattice_pairs_on_bin_edges(hip_ctx):
    # the variant with f32 slab coordinates (diagonal cell, slab culling live): a perfect lattice in a long box puts
    # every distance on a bin edge, i.e. every in-range pair takes the provisional count + fix-up route
    a, n = 2.0, (6, 6, 18)
    pos = np.array([[x, y, z] for x in range(n[0]) for y in range(n[1]) for z in range(n[2])], dtype=float) * a
    numbers = np.where(np.arange(len(pos)) % 5 == 0, 30, np.where(np.arange(len(pos)) % 2 == 0, 7, 6))
    cell = np.diag([n[0] * a, n[1] * a, n[2] * a])
    packed = PackedTrajectory(np.stack([pos, pos + 0.25, pos - 31.0]), cell, numbers)
    for rmax, nb in [(5.9, 59), (6.0, 600), (6.0, 6), (5.999, 2310)]:
        _zf_vs_oracle(hip_ctx, packed, rmax, nb)


def test_rdf_zf_uneven_slabs_and_rare_species(hip_ctx):
    # atoms bunched into a few thin layers along the long axis plus a rare species spread over all of it: centre
    # sub-tiles of the rare species span more than 1/16 of the axis (those steps fall back to the integer slab
    # differences inside the same launch), bunched ones sit in a single slab; coincident atoms included
    rng = np.random.default_rng(77)
    L = np.array([14.0, 15.0, 48.0])
    N = 2600
    z = np.concatenate([rng.normal(c, 0.4, 600) for c in (3.0, 11.0, 30.0, 41.0)])
    z = np.concatenate([z, rng.uniform(0, L[2], N - len(z))])
    xy = rng.uniform(0, 1, (N, 2)) * L[:2]
    pos = np.column_stack([xy, z])
    pos[7] = pos[3]                                   # two atoms on top of each other
    numbers = np.where(np.arange(N) >= 2400, 30, np.where(np.arange(N) % 2 == 0, 6, 1))
    frames = np.stack([pos, pos + rng.normal(0, 0.05, pos.shape), pos + np.array([0.0, 0.0, 24.0])])
    packed = PackedTrajectory(frames, np.diag(L), numbers)
    for rmax, nb in [(7.0, 700), (6.5, 2310), (3.0, 50)]:
        _zf_vs_oracle(hip_ctx, packed, rmax, nb)


def test_rdf_zf_changing_diagonal_cells(hip_ctx):
    # NPT-like: a different diagonal cell per frame (per-frame scales, reach and guards)
    packed = H.random_walk(H.replicate(H.zif4_frame(), (1, 2, 4)), 5, 0.06, 91, cell_jitter=0.02, ortho=True)
    assert packed.cell.shape[0] == 5
    rmax = float(np.min(packed.cell_lengths()) / 2)
    _zf_vs_oracle(hip_ctx, packed, rmax, int(rmax // 0.01))


def test_rdf_zf_cubic_cells_without_culling(hip_ctx):
    # cubic cells at the default cutoff (half the cell): no slab culling is possible, every partner is visited, and the
    # f32 slab coordinates cover all of them but the band around the sub-tile's antipode -- a perfect lattice (every
    # distance on a bin edge, pairs at exactly half the cell), layered atoms with a rare species, and a random gas
    a, n = 2.0, 10
    g = np.arange(n) * a
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    numbers = np.where(np.arange(len(pos)) % 7 == 0, 30, np.where(np.arange(len(pos)) % 2 == 0, 7, 6))
    packed = PackedTrajectory(np.stack([pos, pos + 0.5, pos - 27.0]), np.diag([n * a] * 3), numbers)
    for rmax, nb in [(10.0, 1000), (10.0, 10), (9.99, 2310), (7.0, 70)]:
        _zf_vs_oracle(hip_ctx, packed, rmax, nb, nocull_too=False)
        assert hip_ctx.last_path() == "rdf_tile"      # (the last call of the helper ran with AMOF_RDF_NOZF)
    rng = np.random.default_rng(5)
    L, N = 24.0, 3000
    z = np.concatenate([rng.normal(c, 0.3, 700) for c in (2.0, 9.0, 14.0, 21.5)])
    z = np.concatenate([z, rng.uniform(0, L, N - len(z))])
    p = np.column_stack([rng.uniform(0, L, (N, 2)), z])
    numbers = np.where(np.arange(N) >= 2850, 30, np.where(np.arange(N) % 3 == 0, 6, 1))
    packed = PackedTrajectory(np.stack([p, p[:, [2, 0, 1]], p + 100.0]), np.diag([L] * 3), numbers)
    for rmax, nb in [(12.0, 1200), (11.5, 2310), (12.0, 37)]:
        _zf_vs_oracle(hip_ctx, packed, rmax, nb, nocull_too=False)


def test_frame_tier_on_a_host_trajectory_staged_in_batches(hip_ctx):
    """1700 host-resident frames reach the whole-frame kernels in staged batches of 512, 1024, ... frames: same counts as
    the device-resident copy, the oracle on both ends of the trajectory"""
    import torch
    packed = H.random_walk(H.replicate(H.zif4_frame(), (2, 1, 1)), 1700, 0.03, 5, ortho=True)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    zn, n, c = kinds.index(30), kinds.index(7), kinds.index(6)
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.5
    rcm[c, n] = rcm[n, c] = 1.6
    sets = [(zn, n), (n, zn), (c, n)]
    triples = [(zn, n), (n, zn), (n, -1), (-1, -1)]
    edges = np.arange(182) * 1.0
    dev = PackedTrajectory(torch.tensor(packed.pos, device="cuda:0"), packed.cell, packed.numbers)
    s_host = hip_ctx.cn_count(packed, rcm, sets)
    assert hip_ctx.last_path() == "cn_frame"
    assert np.array_equal(s_host, hip_ctx.cn_count(dev, rcm, sets))
    assert np.array_equal(s_host[:40], clib.cn_counts(packed.pos[:40], packed.cell, sp, S, rcm, sets))
    h_host, a_host = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_frame"
    h_dev, a_dev = hip_ctx.bad_hist(dev, rcm, triples, edges)
    assert np.array_equal(h_host, h_dev) and np.array_equal(a_host, a_dev)
    h_ref, a_ref = clib.bad_hist(packed.pos[1600:], packed.cell, sp, S, rcm, triples, edges)
    h_end, a_end = hip_ctx.bad_hist(packed, rcm, triples, edges, frame_range=(1600, 1700))
    assert np.array_equal(h_end, h_ref) and np.array_equal(a_end, a_ref)


@pytest.mark.parametrize("n_pair", [4096, 4097, 8192, 8193])
def test_frame_tier_atom_count_boundaries(hip_ctx, n_pair):
    """4096 / 4097 atoms in a pair switch between the 4- and the 8-atoms-per-thread kernels, 8192 / 8193 between one
    workgroup per frame and z-slabs of the frame: one species alone, and two species sharing the count unevenly"""
    rng = np.random.default_rng(n_pair)
    L = (n_pair / 0.06) ** (1 / 3)
    cell = np.diag([L, 1.1 * L, 0.9 * L])
    edges = np.arange(182) * 1.0
    for split in (n_pair, n_pair // 3):
        numbers = np.where(np.arange(n_pair) < split, 30, 7)
        pos = rng.uniform(0, 1, (2, n_pair, 3)) @ cell
        packed = PackedTrajectory(pos, cell, numbers)
        kinds, sp = H.species_of(packed.numbers)
        S = len(kinds)
        rcm = np.full((S, S), 2.2)
        sets = [(a, b) for a in range(S) for b in range(S)]
        got = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
        path = hip_ctx.last_path()
        # (two species: every pair of the call must fit -- the same-species pairs are smaller than the mixed one)
        assert path == ("cn_frame" if n_pair <= 8192 else "cn_frame_slabs"), (n_pair, split, path)
        ref = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]), (n_pair, split, path)
        triples = [(0, 0), (0, -1), (-1, -1)] if S == 1 else [(0, 1), (1, 0), (1, -1), (-1, -1)]
        hg = hip_ctx.bad_hist(packed, rcm, triples, edges)
        assert hip_ctx.last_path() == ("bad_frame" if n_pair <= 8192 else "bad_frame_slabs"), (n_pair, split, hip_ctx.last_path())
        hr = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
        assert np.array_equal(hg[0], hr[0]) and np.array_equal(hg[1], hr[1]), (n_pair, split)


@pytest.mark.parametrize("ortho", [True, False])
def test_frame_tier_in_slabs_forced_on_small_frames(hip_ctx, monkeypatch, ortho):
    """AMOF_NBR_SLABS=1 sends pairs that fit one workgroup through the streaming slab kernels (one or several z-slabs per
    pair): counts per atom and all 17 triples of three cutoffs equal the oracle's and the whole-frame kernels', on the
    orthorhombic and on the fixture's own triclinic lattice, frames in several batches"""
    packed = H.random_walk(H.replicate(H.zif4_frame(), (2, 2, 3)), 6, 0.04, 11, ortho=ortho)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    zn, n, c, h = (kinds.index(z) for z in (30, 7, 6, 1))
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.5
    rcm[c, n] = rcm[n, c] = 1.6
    rcm[c, h] = rcm[h, c] = 1.25
    rcm[c, c] = 1.6
    rcm[zn, zn] = 6.3
    sets = [(zn, n), (n, zn), (c, n), (n, c), (h, c), (c, c), (zn, zn)]
    triples = [(a, b) for a in range(S) for b in range(S)] + [(-1, -1)]
    edges = np.arange(182) * 1.0
    whole = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    assert hip_ctx.last_path() == "cn_frame"
    h_whole = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_frame"
    monkeypatch.setenv("AMOF_NBR_SLABS", "1")
    got = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    assert hip_ctx.last_path() == "cn_frame_slabs"
    ref = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert np.array_equal(got[0], whole[0]) and np.array_equal(got[1], whole[1])
    monkeypatch.setenv("AMOF_BAD_ROWS_MB", "8")             # (several batches of neighbour rows)
    hg = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_frame_slabs"
    hr = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
    assert np.array_equal(hg[0], hr[0]) and np.array_equal(hg[1], hr[1])
    assert np.array_equal(hg[0], h_whole[0]) and np.array_equal(hg[1], h_whole[1])


def test_frame_tier_keeps_a_4x4x4_supercell(hip_ctx):
    """Review item (round 3): 17 408 atoms (4 x 4 x 4 ZIF-4; C + H = 12 288 atoms, C + N = 10 240) stay on the frame tier,
    in z-slabs; counts and angles equal the oracle's"""
    packed = H.random_walk(H.replicate(H.zif4_frame(), (4, 4, 4)), 2, 0.05, 3, ortho=False)
    assert packed.pos.shape[1] == 17408
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    zn, n, c, h = (kinds.index(z) for z in (30, 7, 6, 1))
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.5
    rcm[c, n] = rcm[n, c] = 1.6
    rcm[c, h] = rcm[h, c] = 1.25
    sets = [(zn, n), (n, zn), (c, n), (n, c), (h, c), (c, h)]
    got = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    assert hip_ctx.last_path() == "cn_frame_slabs"
    ref = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    triples = [(a, b) for a in range(S) for b in range(S)] + [(-1, -1)]
    edges = np.arange(182) * 1.0
    hg = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_frame_slabs"
    hr = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
    assert np.array_equal(hg[0], hr[0]) and np.array_equal(hg[1], hr[1])


def test_frame_tier_slab_denser_than_its_records_falls_back(hip_ctx):
    """a slab sized for its expected share of the atoms (+ 20 % + 128) that meets a much denser layer raises the overflow
    flag; the gather kernels then answer, with the oracle's counts"""
    rng = np.random.default_rng(9)
    N, L = 12000, 60.0
    z = np.concatenate([rng.normal(30.0, 2.0, 8000), rng.uniform(0, L, N - 8000)]) % L
    pos = np.column_stack([rng.uniform(0, L, (N, 2)), z])[None]
    numbers = np.where(np.arange(N) % 2 == 0, 30, 7)
    packed = PackedTrajectory(pos, np.diag([L] * 3), numbers)
    kinds, sp = H.species_of(packed.numbers)
    rcm = np.full((2, 2), 1.5)
    sets = [(0, 1), (1, 1)]
    got = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    assert hip_ctx.last_path() in ("cn_cell", "cn_fast"), hip_ctx.last_path()
    ref = clib.cn_counts(packed.pos, packed.cell, sp, 2, rcm, sets, per_atom=True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    triples = [(0, 1), (1, 0), (-1, -1)]
    edges = np.arange(182) * 1.0
    hg = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert not hip_ctx.last_path().startswith("bad_frame"), hip_ctx.last_path()
    hr = clib.bad_hist(packed.pos, packed.cell, sp, 2, rcm, triples, edges)
    assert np.array_equal(hg[0], hr[0]) and np.array_equal(hg[1], hr[1])


@pytest.mark.parametrize("records", ["1", "0"])
def test_frame_tier_compact_records_grid_as_fine_as_the_cutoff_allows(hip_ctx, monkeypatch, records):
    """Advisor (round 3, medium): COMPACT 8-byte records take the cell of an atom from its coordinate TRUNCATED to 16 bits,
    which moves a centre against its partner by up to 2^-16 of the cell vector; with cells only 1e-5 thicker than the
    cutoff a partner just inside rc could land two cells from its centre and be missed.  Constructed case: a cubic cell
    with hmin = 7 rc (1 + 1.2e-5), axis-aligned pairs at rc (1 - 1e-7) whose centre sits 0 .. 2 units of 2^-16 above every
    cell edge of the 7-cell grid, along each axis, compact records forced (and, as a control, forbidden)."""
    monkeypatch.setenv("AMOF_NBR_COMPACT", records)
    L, nk = 21.0, 7
    rc = L / (nk * (1.0 + 1.2e-5))
    u = 2.0 ** -16
    rng = np.random.default_rng(77)
    nfill = 20
    numbers = np.array([30] + [7] + [30] * nfill + [7] * nfill)
    frames = []
    for axis in range(3):
        for c in range(nk):
            for e in range(64):
                s = np.full((2, 3), 0.37)
                s[0, axis] = c / nk + (e + 0.5) / 32.0 * u
                s[1, axis] = s[0, axis] + (rc / L) * (1.0 - 1e-7)
                fill = rng.uniform(0, 1, (2 * nfill, 3))
                frames.append(np.concatenate([s, fill]) % 1.0 * L)
    packed = PackedTrajectory(np.array(frames), np.diag([L, L, L]), numbers)
    kinds, sp = H.species_of(packed.numbers)
    rcm = np.array([[0.0, rc], [rc, 0.0]])
    sets = [(0, 1), (1, 0)]
    got = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    assert hip_ctx.last_path() == "cn_frame"
    ref = clib.cn_counts(packed.pos, packed.cell, sp, 2, rcm, sets, per_atom=True)
    a30 = int(np.nonzero(packed.numbers == 30)[0][0])
    assert ref[1][:, sets.index((kinds.index(30), kinds.index(7))), a30].min() >= 1      # (the constructed partner is in range)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    edges = np.arange(182) * 1.0
    triples = [(0, 1), (1, 0), (-1, -1)]
    hg = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_frame"
    hr = clib.bad_hist(packed.pos, packed.cell, sp, 2, rcm, triples, edges)
    assert np.array_equal(hg[0], hr[0]) and np.array_equal(hg[1], hr[1])


def _tri_cell(kind):
    """general cells for the orthogonalised-frame tile kernel ("rdf_tile_tri"), by what its selection has to do"""
    from amof_amd.frames import Frame
    z = H.zif4_frame()
    if kind == "fixture":           # the reference's own lattice (off-diagonals ~1e-5): second images only inside the guard band
        return H.replicate(z, (2, 2, 3))
    if kind == "equal_ab":          # two equal in-plane lengths, sheared: the x wrap takes the y term along
        base = H.replicate(z, (2, 2, 2))
        shear = np.eye(3) + np.array([[0, 0.02, 0.01], [0, 0, 0.02], [0, 0, 0]])
        return Frame(base.numbers, base.positions @ shear, base.cell @ shear)
    if kind == "short_c":           # a clearly shortest axis that is sheared against the others: near test on y
        base = H.replicate(z, (2, 2, 1))
        base = Frame(base.numbers, base.positions, np.diag(np.diag(base.cell)))
        shear = np.eye(3) + np.array([[0, 0.02, 0.01], [0, 0, 0.02], [0, 0, 0]])
        return Frame(base.numbers, base.positions @ shear, base.cell @ shear)
    if kind == "cubic":             # cubic, sheared: no culling, the slab axis needs the near test too
        cub = H.replicate(Frame(z.numbers, z.positions, np.diag([15.4, 15.4, 15.4])), (2, 2, 2))
        shear = np.eye(3) + np.array([[0, 0, 0], [0.03, 0, 0], [-0.015, 0.03, 0]])
        return Frame(cub.numbers, cub.positions @ shear, cub.cell @ shear)
    if kind == "hexagonal":         # a = b, gamma = 120 degrees: 15 % of the pairs have a second image along y (twin in the slow path)
        base = H.replicate(z, (2, 2, 2))
        d3 = np.diag(np.diag(base.cell))
        a_hex = float(np.sqrt(d3[0, 0] * d3[1, 1] / (np.sqrt(3.0) / 2.0)))
        hexc = np.array([[a_hex, 0.0, 0.0], [-0.5 * a_hex, np.sqrt(3.0) / 2.0 * a_hex, 0.0], [0.0, 0.0, d3[2, 2]]])
        frac = np.linalg.solve(np.asarray(base.cell).T, base.positions.T).T
        return Frame(base.numbers, frac @ hexc, hexc)
    if kind == "hexagonal60":       # the same lattice described with gamma = 60 degrees (c10 = + 1/2), the hexagonal plane = (b, c)
        base = H.replicate(z, (2, 2, 2))
        d3 = np.diag(np.diag(base.cell))
        a_hex = float(np.sqrt(d3[1, 1] * d3[2, 2] / (np.sqrt(3.0) / 2.0)))
        # (the first axis the longest: it becomes the slab axis, the hexagonal pair the in-plane one)
        hexc = np.array([[1.4 * a_hex, 0.0, 0.0], [0.0, a_hex, 0.0], [0.0, 0.5 * a_hex, np.sqrt(3.0) / 2.0 * a_hex]])
        frac = np.linalg.solve(np.asarray(base.cell).T, base.positions.T).T
        return Frame(base.numbers, frac @ hexc, hexc)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["hexagonal", "hexagonal60"])
def test_rdf_hexagonal_cells_take_the_exact_half_x_wrap(hip_ctx, kind, capfd):
    """c10 = -+ 1/2 exactly: near mode 4 with the x wrap's y term as a shift and the twin's x as a flipped top bit
    (csrc/rdf.hip tri_q_twin<HALF>: codes 10 / 11) -- against the oracle and against the float-product form (code 4 / 9)"""
    packed = H.random_walk(_tri_cell(kind), 3, 0.08, 43)
    kinds, sp = H.species_of(packed.numbers)
    rmax = float(np.min(packed.cell_lengths()) / 2)
    for nb in (1540, 311):
        with _env(AMOF_RDF_NOCELL="1", AMOF_RDF_NORANGE="1", AMOF_RDF_DEBUG="1"):
            capfd.readouterr()
            got, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            err = capfd.readouterr().err
            assert hip_ctx.last_path() == "rdf_tile_tri" and ("code 10 " in err or "code 11 " in err), err
            with _env(AMOF_RDF_NOHALF="1"):
                plain, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
                err = capfd.readouterr().err
                assert hip_ctx.last_path() == "rdf_tile_tri" and ("code 4 " in err or "code 9 " in err), err
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
        assert np.array_equal(got, ref) and np.array_equal(plain, ref)


@pytest.mark.parametrize("kind", ["fixture", "equal_ab", "short_c", "cubic", "hexagonal", "hexagonal60"])
@pytest.mark.parametrize("jitter", [0.0, 0.004])
def test_rdf_triangular_frame_kernel(hip_ctx, kind, jitter):
    """General cells at the reference's default cutoff (half the shortest cell LENGTH, amof/rdf.py:74): the tile kernel in
    the orthogonalised lattice frame against the oracle, with the image-aware / exact kernels it replaces beside it; a
    cutoff beyond half the x axis (only the C ABI allows it) is refused by the variant"""
    packed = H.random_walk(_tri_cell(kind), 3, 0.08, 41, cell_jitter=jitter)
    kinds, sp = H.species_of(packed.numbers)
    rmax = float(np.min(packed.cell_lengths()) / 2)
    for rm, nb in ((rmax, 1540), (rmax, 257), (0.8 * rmax, 500)):
        with _env(AMOF_RDF_NOCELL="1", AMOF_RDF_NORANGE="1"):
            got, _, _ = hip_ctx.rdf_accumulate(packed, rm, nb)
            assert hip_ctx.last_path() == "rdf_tile_tri", (kind, rm)
            with _env(AMOF_RDF_NOCULL="1"):
                nocull, _, _ = hip_ctx.rdf_accumulate(packed, rm, nb)
                assert hip_ctx.last_path() == "rdf_tile_tri"
            with _env(AMOF_RDF_NOTRI="1"):
                old, _, _ = hip_ctx.rdf_accumulate(packed, rm, nb)
                assert hip_ctx.last_path() in ("rdf_tile", "rdf_tile_img", "rdf_exact")
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rm, nb, cell_list=True)
        assert np.array_equal(got, ref), (kind, rm, nb, int(np.abs(got.astype(np.int64) - ref).sum()))
        assert np.array_equal(nocull, ref) and np.array_equal(old, ref)
    # a cutoff beyond half the shortest length (only the C ABI allows it): second images along that axis -- as y or z it
    # gets the near test, as x (both in-plane axes too short) the variant is refused
    big, _, _ = hip_ctx.rdf_accumulate(packed, 1.01 * rmax, 800)
    if kind in ("equal_ab", "cubic", "hexagonal", "hexagonal60"):
        assert hip_ctx.last_path() != "rdf_tile_tri"
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), 1.01 * rmax, 800, cell_list=True)
    assert np.array_equal(big, ref)


@pytest.mark.parametrize("cell,rmax", [(((5.0, 0, 0), (20.0, 7.0, 0), (0, 0, 30.0)), 2.4),       # the advisor's cell: |L10| = 4 L00
                                       (((6.0, 0, 0), (5.5, 6.5, 0), (0, 0, 28.0)), 2.9),        # |L10| ~ 0.92 L00
                                       (((9.0, 0, 0), (-8.0, 9.5, 0), (1.0, 2.0, 31.0)), 4.4)])  # negative, with a sheared slab axis
def test_rdf_skewed_non_reduced_cells_do_not_take_the_integer_x_wrap(hip_ctx, cell, rmax):
    """Cells whose second vector leans over the first by about its length or more (not Niggli-reduced: nothing in the
    reference asks for reduced cells).  The triangular-frame variants that wrap x in the integer domain form (int)(fy * c10),
    which saturates beyond 2^31 once |c10| (R / L11) >= 1/2: selection must refuse them there (advisor, round 4) and
    whatever kernel answers must equal the oracle."""
    rng = np.random.default_rng(5)
    cell = np.array(cell, dtype=float)
    n = 700
    pos = rng.uniform(0, 1, (3, n, 3)) @ cell
    packed = PackedTrajectory(pos, cell, [1, 8] * (n // 2))
    kinds, sp = H.species_of(packed.numbers)
    for nb in (240, 997):
        ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
        for env in ({}, {"AMOF_RDF_NOCELL": "1", "AMOF_RDF_NORANGE": "1"}, {"AMOF_RDF_NOCELL": "1", "AMOF_RDF_NORANGE": "1", "AMOF_RDF_NOCULL": "1"}):
            with _env(**env):
                got, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
            assert np.array_equal(got, ref), (env, hip_ctx.last_path(), int(np.abs(got.astype(np.int64) - ref.astype(np.int64)).sum()))


def test_rdf_triangular_frame_lattice_on_faces_and_edges(hip_ctx):
    """an integer lattice in integer sheared cells: every distance on a bin edge, pairs exactly on cell faces (both images
    at the same distance) -- every in-range pair goes through the queue of the canonical pass, which overflows into the
    in-place evaluation"""
    g12 = np.arange(12) * 1.0
    for cell in (np.array([[12.0, 0, 0], [1.0, 12.0, 0], [0.0, 1.0, 24.0]]), np.array([[12.0, 0, 0], [0.0, 12.0, 0], [1.0, -1.0, 12.0]])):
        pts = np.array([[x, y, zz] for x in g12 for y in g12 for zz in np.arange(int(cell[2, 2])) * 1.0])
        numbers = np.where((pts.sum(axis=1) % 2) == 0, 11, 17)
        packed = PackedTrajectory(np.stack([pts, pts + 0.5]), cell, numbers)
        kinds, sp = H.species_of(packed.numbers)
        for rm, nb in ((6.0, 60), (6.0, 600), (5.0, 50)):
            with _env(AMOF_RDF_NOCELL="1", AMOF_RDF_NORANGE="1"):
                got, _, _ = hip_ctx.rdf_accumulate(packed, rm, nb)
                assert hip_ctx.last_path() == "rdf_tile_tri"
            ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rm, nb, cell_list=True)
            assert np.array_equal(got, ref), (cell[2].tolist(), rm, nb)


@pytest.mark.parametrize("frames", [5, 19])
def test_rdf_cell_kernel_frame_chunks(hip_ctx, frames):
    """round 4: a workgroup of the cell-list kernel keeps its block of atoms over a chunk of frames (one histogram flush):
    chunk sizes that divide the launch, that leave a ragged last chunk, and that exceed it -- with (>= 16 frames) and without
    the XCD mapping of the frames"""
    packed = H.random_walk(H.replicate(H.zif4_frame(), (3, 3, 3)), frames, 0.05, 71)
    kinds, sp = H.species_of(packed.numbers)
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), 6.0, 600, cell_list=True)
    for fpc in ("1", "2", "3", "64"):
        with _env(AMOF_RDF_FORCE_CELL="1", AMOF_RDF_CELL_FPC=fpc):
            got, _, _ = hip_ctx.rdf_accumulate(packed, 6.0, 600)
            assert hip_ctx.last_path() == "rdf_cell"
        assert np.array_equal(got, ref), (frames, fpc)
        with _env(AMOF_RDF_FORCE_CELL="1", AMOF_RDF_CELL_FPC=fpc, AMOF_RDF_CELL_GATHER="1"):
            got, _, _ = hip_ctx.rdf_accumulate(packed, 6.0, 600)
        assert np.array_equal(got, ref), (frames, fpc, "gather")
    # round 5, one wave per cell: cell groups per workgroup that divide the grid, leave a ragged last group, exceed it
    for cpw in ("1", "3", "7", "1000"):
        with _env(AMOF_RDF_FORCE_CELL="1", AMOF_RDF_CELL_CPW=cpw, AMOF_RDF_CELL_FPC="2"):
            got, _, _ = hip_ctx.rdf_accumulate(packed, 6.0, 600)
            assert hip_ctx.last_path() == "rdf_cell"
        assert np.array_equal(got, ref), (frames, "cpw", cpw)


def test_rdf_range_kernel_is_what_a_thin_long_cell_selects(hip_ctx):
    """housekeeping (round-3 review): the 2-level range kernel is not dead code -- a slab-shaped cell, too thin along ONE axis
    for five cells of rmax / 2 (so no 3-D cell list) but many cutoffs long and wide, selects it WITHOUT any switch; the 1-D slab list beside it"""
    rng = np.random.default_rng(12)
    cell = np.diag([19.0, 120.0, 150.0])            # a slab: thin along x only
    N = 20000
    numbers = np.where(np.arange(N) % 3 == 0, 8, 1)
    packed = PackedTrajectory(rng.uniform(0, 1, (2, N, 3)) @ cell, cell, numbers)
    kinds, sp = H.species_of(packed.numbers)
    got, _, _ = hip_ctx.rdf_accumulate(packed, 9.0, 900)
    assert hip_ctx.last_path() == "rdf_range"
    with _env(AMOF_RDF_NORANGE="1"):
        slab, _, _ = hip_ctx.rdf_accumulate(packed, 9.0, 900)
        assert hip_ctx.last_path() == "rdf_tile_zf"
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), 9.0, 900, cell_list=True)
    assert np.array_equal(got, ref) and np.array_equal(slab, ref)
