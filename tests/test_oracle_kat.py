"""Analytic known-answer tests and cross-checks of the CPU oracle (RDF / CN /
BAD).  The reference holds no vectors for these; this is what pins them
(SURVEY 8c)."""

import numpy as np
import pytest

from oracle import clib, numpy_oracle as no
from tests import helpers as H


def sc_lattice(n, a):
    g = np.arange(n) * a
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    idx = np.array([[i, j, k] for i in range(n) for j in range(n) for k in range(n)])
    return pos, np.diag([n * a] * 3), idx


# number of lattice vectors of squared length k (simple cubic): r3(k)
SHELLS = {1: 6, 2: 12, 3: 8, 4: 6, 5: 24, 6: 24, 8: 12}


def test_sc_shell_multiplicities():
    a, n = 2.5, 6
    pos, cell, _ = sc_lattice(n, a)
    pos = pos + 0.123  # off the cell boundary
    rmax, nb = 7.4, 740
    h, vol = clib.rdf_hist(pos, cell, np.zeros(len(pos), np.int32), 1, rmax, nb)
    tot = h[0, 0]
    dr = rmax / nb
    for k, mult in SHELLS.items():
        r = a * np.sqrt(k)
        if r >= rmax:
            continue
        b = int(r / dr)
        # lattice distances carry ~1e-15 noise: accept the neighbouring bin
        assert tot[max(b - 1, 0):b + 2].sum() == mult * len(pos), k
    assert tot.sum() == sum(m for k, m in SHELLS.items() if a * np.sqrt(k) < rmax) * len(pos)
    assert vol == pytest.approx((n * a) ** 3)


def test_rock_salt_partials_and_sum_rule():
    a, n = 2.5, 6
    pos, cell, idx = sc_lattice(n, a)
    sp = (idx.sum(axis=1) % 2).astype(np.int32)
    h, _ = clib.rdf_hist(pos + 0.3, cell, sp, 2, 7.4, 740)
    dr = 7.4 / 740
    half = len(pos) // 2

    def shell(hh, k):
        b = int(a * np.sqrt(k) / dr)
        return hh[max(b - 1, 0):b + 2].sum()
    assert shell(h[0, 1], 1) == 6 * half and shell(h[0, 0], 1) == 0      # nearest neighbours: unlike
    assert shell(h[0, 0], 2) == 12 * half and shell(h[0, 1], 2) == 0     # second shell: like
    assert shell(h[1, 0], 3) == 8 * half
    assert np.array_equal(h[0, 1], h[1, 0])
    h1, _ = clib.rdf_hist(pos + 0.3, cell, np.zeros(len(pos), np.int32), 1, 7.4, 740)
    assert np.array_equal(h.sum(axis=(0, 1)), h1[0, 0])                  # sum_ab H_ab = H_total


def test_strict_cutoff_and_cn_values():
    a, n = 2.0, 5
    pos, cell, _ = sc_lattice(n, a)
    sp = np.zeros(len(pos), np.int32)
    for rc, want in [(1.1 * a, 6), (1.5 * a, 18), (a, 0), (np.nextafter(a, 10), 6)]:
        sums, pa = clib.cn_counts(pos, cell, sp, 1, [[rc]], [[0, 0]], per_atom=True)
        assert (pa == want).all(), rc
        assert sums[0, 0] == want * len(pos)


def test_small_cell_counts_periodic_images():
    # one atom in a cubic box of edge 1: neighbours are its own images
    pos = np.array([[0.2, 0.3, 0.4]])
    cell = np.eye(3)
    sums = clib.cn_counts(pos, cell, [0], 1, [[1.05]], [[0, 0]])
    assert sums[0, 0] == 6
    sums = clib.cn_counts(pos, cell, [0], 1, [[1.5]], [[0, 0]])
    assert sums[0, 0] == 18
    h, _ = clib.rdf_hist(pos, cell, [0], 1, 1.8, 180)
    assert h.sum() == 6 + 12 + 8
    # two atoms, box smaller than the cutoff: every image counted in both directions
    pos2 = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0]])
    h, _ = clib.rdf_hist(pos2, cell, [0, 1], 2, 1.2, 120)
    ref = no.rdf_hist(pos2, cell, [0, 1], 2, 1.2, 120)
    assert np.array_equal(h, ref)


def test_sc_angles():
    a, n = 2.0, 4
    pos, cell, _ = sc_lattice(n, a)
    sp = np.zeros(len(pos), np.int32)
    ang = clib.angles(pos, cell, sp, 1, [[1.2 * a]], 0, 0)
    assert len(ang) == 15 * len(pos)
    assert np.isclose(ang, 90).sum() == 12 * len(pos) and np.isclose(ang, 180).sum() == 3 * len(pos)
    # 180 degrees must land in the right-closed last bin of numpy.histogram edges
    for dtheta in (0.05, 0.5):
        bins = int(180 // dtheta)
        edges = np.arange(bins + 2) * dtheta
        hist, nang = clib.bad_hist(pos, cell, sp, 1, [[1.2 * a]], [[0, 0], [-1, -1]], edges)
        want = np.histogram(ang, bins=edges)[0]
        assert np.array_equal(hist[0], want) and np.array_equal(hist[1], want)
        assert nang[0] == len(ang) and hist[0].sum() == len(ang)


def test_tetrahedron_angles():
    v = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=float)
    pos = np.concatenate([[np.zeros(3)], v]) + 10.0
    cell = np.diag([30.0, 31.0, 32.0])
    sp = np.array([0, 1, 1, 1, 1], np.int32)
    rcm = np.array([[0, 2.0], [2.0, 0]])
    ang = clib.angles(pos, cell, sp, 2, rcm, 0, 1)
    assert len(ang) == 6
    np.testing.assert_allclose(ang, np.degrees(np.arccos(-1.0 / 3.0)), rtol=1e-13)


def test_undefined_angle_raises():
    pos = np.array([[1.0, 1, 1], [2.0, 1, 1], [1.0, 1, 1]])   # atom 2 sits on the centre atom 0
    cell = np.diag([20.0, 20, 20])
    sp = np.array([0, 1, 1], np.int32)
    rcm = np.array([[0, 1.5], [1.5, 0]])
    with pytest.raises(ZeroDivisionError):
        clib.angles(pos, cell, sp, 2, rcm, 0, 1)


def test_fixture_known_answers(zif4):
    kinds, sp = H.species_of(zif4.numbers)
    S = len(kinds)
    zn, n = kinds.index(30), kinds.index(7)
    rmax = min(zif4.get_cell_lengths_and_angles()[:3]) / 2
    assert rmax == pytest.approx(7.7021000006, abs=1e-9)
    nb = int(rmax // 0.01)
    assert nb == 770
    h, vol = clib.rdf_hist(zif4.positions, zif4.cell, sp, S, rmax, nb)
    assert h.sum() == 30968 and np.nonzero(h.sum(axis=(0, 1)))[0][0] == 107
    assert vol == pytest.approx(4380.486, abs=1e-3)
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.5
    sums, pa = clib.cn_counts(zif4.positions, zif4.cell, sp, S, rcm, [[zn, n], [n, zn]], per_atom=True)
    assert set(pa[0, 0][pa[0, 0] >= 0]) == {4} and set(pa[0, 1][pa[0, 1] >= 0]) == {1}
    ang = clib.angles(zif4.positions, zif4.cell, sp, S, rcm, zn, n)
    assert len(ang) == 96
    assert ang.min() == pytest.approx(103.3765, abs=1e-4) and ang.max() == pytest.approx(113.1895, abs=1e-4)
    assert ang.mean() == pytest.approx(109.4372, abs=1e-4)
    assert len(clib.angles(zif4.positions, zif4.cell, sp, S, rcm, n, zn)) == 0   # 'Zn-N-Zn' column absent


def test_rdf_cn_integer_identity(zif4):
    # sum of H_ab over bins with upper edge <= rc == sum of CN counts (same strict-< semantics)
    kinds, sp = H.species_of(zif4.numbers)
    S = len(kinds)
    rmax, nb = 7.5, 750
    h, _ = clib.rdf_hist(zif4.positions, zif4.cell, sp, S, rmax, nb)
    for a in range(S):
        for b in range(S):
            rc = 2.5
            rcm = np.zeros((S, S)); rcm[a, b] = rcm[b, a] = rc
            sums = clib.cn_counts(zif4.positions, zif4.cell, sp, S, rcm, [[a, b]])
            nbin = int(round(rc / (rmax / nb)))
            lo, hi = h[a, b][:nbin - 1].sum(), h[a, b][:nbin + 1].sum()
            assert lo <= sums[0, 0] <= hi


@pytest.mark.parametrize("seed", range(6))
def test_c_oracle_equals_bruteforce_numpy_random_cells(seed):
    rng = np.random.default_rng(seed)
    cell = np.diag(rng.uniform(4, 7, 3)) + (rng.uniform(-1.5, 1.5, (3, 3)) if seed % 2 else 0) * np.tri(3, k=-1)
    N, S = 60, 3
    pos = rng.uniform(-0.5, 1.5, (N, 3)) @ cell
    sp = rng.integers(0, S, N).astype(np.int32)
    pbc = (True, True, seed != 4)
    rmax = [2.0, 3.5, 5.0][seed % 3]
    h_c, _ = clib.rdf_hist(pos, cell, sp, S, rmax, 173, pbc=pbc)
    h_cl, _ = clib.rdf_hist(pos, cell, sp, S, rmax, 173, pbc=pbc, cell_list=True)
    h_np = no.rdf_hist(pos, cell, sp, S, rmax, 173, pbc=pbc)
    assert np.array_equal(h_c, h_cl)
    assert np.array_equal(h_c, h_np)
    assert np.array_equal(h_c, h_c.transpose(1, 0, 2))
    rcm = rng.uniform(1.0, 2.4, (S, S)); rcm = (rcm + rcm.T) / 2; rcm[0, 0] = 0
    sets = [(a, b) for a in range(S) for b in range(S)]
    sums = clib.cn_counts(pos, cell, sp, S, rcm, sets, pbc=pbc)
    ref = no.cn_sums(pos, cell, sp, S, rcm, sets, pbc=pbc)
    assert [int(x) for x in sums[0]] == [int(r.sum()) for r in ref]


def test_cell_list_equals_brute_2k_atoms(zif4):
    base = H.replicate(zif4, (2, 2, 1))
    packed = H.random_walk(base, 2, 0.05, 3)
    kinds, sp = H.species_of(packed.numbers)
    for rmax, nb in [(7.7, 770), (3.0, 300)]:
        a, va = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb)
        b, vb = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
        assert np.array_equal(a, b) and va == vb


def test_ideal_gas_rdf_is_one():
    from amof_amd.rdf import normalize_rdf
    N, F, L = 400, 6, 12.0
    packed = H.random_gas(N, [L, L, L], np.ones(N, int), 9, F=F)
    rmax, nb = 6.0, 30
    h, vol = clib.rdf_hist(packed.pos, packed.cell, np.zeros(N, np.int32), 1, rmax, nb)
    g = normalize_rdf(h[0, 0], F * N, N, vol / F, rmax, nb)
    # N-1 partners instead of N: expect (N-1)/N, with sqrt(counts) noise
    assert abs(g[5:].mean() - (N - 1) / N) < 0.01


def test_ideal_gas_discriminates_the_shell_volume(monkeypatch):
    """A1 without asap3: an ideal gas has g(r) = 1 in EVERY bin -- asap3 documents that its g(r) tends to
    1 for uncorrelated atoms -- which at a coarse dr only the exact shell volume (4 pi / 3)(r_hi^3 - r_lo^3) delivers:
    the midpoint shell 4 pi r_b^2 dr underestimates the volume of bin b by dr^2 / (12 r_b^2) = 1 / (3 (2b + 1)^2), i.e.
    33 % in bin 0, 3.7 % in bin 1, 1.3 % in bin 2.  The default of `normalize_rdf` must be the one that gives 1."""
    from amof_amd import rdf as amrdf
    monkeypatch.delenv("AMOF_RDF_SHELL", raising=False)
    N, F, L = 4000, 8, 10.0
    packed = H.random_gas(N, [L, L, L], np.ones(N, int), 20261004, F=F)
    rmax, nb = 5.0, 10                      # dr = 0.5 A
    h, vol = clib.rdf_hist(packed.pos, packed.cell, np.zeros(N, np.int32), 1, rmax, nb)
    counts = h[0, 0].astype(float)
    assert counts[0] > 5e4                  # ordered pairs in the first bin: 0.6 % noise on the unordered count
    sigma = np.sqrt(2.0 / counts)           # every unordered pair is counted twice
    expect = (N - 1) / N                    # N - 1 partners per centre, normalised with N
    args = (h[0, 0], F * N, N, vol / F, rmax, nb)
    g_default = amrdf.normalize_rdf(*args)
    g_exact = amrdf.normalize_rdf_shell(*args, "exact")
    g_mid = amrdf.normalize_rdf_shell(*args, "midpoint")
    assert amrdf.DEFAULT_SHELL == "exact" and np.array_equal(g_default, g_exact)
    assert np.all(np.abs(g_exact / expect - 1.0) < 4 * sigma)
    b = np.arange(nb)
    np.testing.assert_allclose(g_mid / g_exact, 1.0 + 1.0 / (3.0 * (2 * b + 1) ** 2), rtol=1e-13)
    # the midpoint shell is off by 33 % / 3.7 % in the first two bins: tens of sigma
    assert abs(g_mid[0] / expect - 1.0) > 0.3 and abs(g_mid[1] / expect - 1.0) > 10 * sigma[1]


def test_arccos_of_the_angle_code_is_the_published_algorithm_within_one_ulp():
    """The oracle (and the GPU kernels) evaluate arccos by ONE fixed algorithm (fdlibm's rational approximation) instead
    of whatever library is at hand: numpy's float64 arccos is itself CPU-dependent and differs from the correctly rounded
    value in the last place for a few percent of the arguments.  Pinned here: within one unit in the last place of the
    true value (mpmath, 200 bits) over the whole domain, exact at the ends, and within one ulp of numpy's."""
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.uniform(-1, 1, 4000), 1 - 10.0 ** rng.uniform(-16, 0, 1500), -1 + 10.0 ** rng.uniform(-16, 0, 1500),
                        [0.0, 0.5, -0.5, 1.0, -1.0, 0.7071067811865476, 0.7071067811865475, -0.7071067811865476, 1e-20, -1e-20]])
    got = clib.acos(x)
    # (the part that needs no multiprecision package runs everywhere: the ends, and one ulp of numpy's arccos)
    assert clib.acos(1.0) == 0.0 and clib.acos(-1.0) == np.pi and clib.acos(0.0) == np.pi / 2
    ref = np.arccos(x)
    assert np.all(np.abs(got - ref) <= np.spacing(np.maximum(ref, 1e-300)))
    try:
        import mpmath
    except ImportError:
        return
    mpmath.mp.prec = 200
    worst = 0.0
    for xv, yv in zip(x, got):
        t = mpmath.acos(mpmath.mpf(float(xv)))
        ulp = np.spacing(float(t)) if float(t) != 0.0 else 5e-324
        worst = max(worst, abs(float((mpmath.mpf(float(yv)) - t) / ulp)))
    assert worst < 1.0, worst
