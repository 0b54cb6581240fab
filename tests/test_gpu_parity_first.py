"""First GPU parity checks: HIP kernels (through the C ABI) vs the CPU oracle."""

import numpy as np
import pytest

from oracle import clib, numpy_oracle as no
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _rdf_both(ctx, packed, rmax, nbins):
    kinds, sp = H.species_of(packed.numbers)
    h_gpu, vol_gpu, k2 = ctx.rdf_accumulate(packed, rmax, nbins)
    assert k2 == kinds
    h_cpu, vol_cpu = clib.rdf_hist(packed.pos_host(), packed.cell, sp, len(kinds), rmax, nbins, pbc=packed.pbc)
    return h_gpu, h_cpu, vol_gpu, vol_cpu


@pytest.mark.parametrize("case", ["fixture_tri", "fixture_ortho", "fixture_images", "walk_jitter"])
def test_rdf_bit_exact(hip_ctx, case):
    z = H.zif4_frame()
    if case == "fixture_tri":
        packed, rmax, nb = H.random_walk(z, 3, 0.05, 1), 7.7021, 770
    elif case == "fixture_ortho":
        packed, rmax, nb = H.random_walk(z, 3, 0.05, 2, ortho=True), 7.7021, 770
    elif case == "fixture_images":
        packed, rmax, nb = H.random_walk(z, 2, 0.05, 3), 12.0, 300
    else:
        packed, rmax, nb = H.random_walk(z, 5, 0.05, 4, cell_jitter=0.01), 7.5, 749
    h_gpu, h_cpu, vg, vc = _rdf_both(hip_ctx, packed, rmax, nb)
    assert h_gpu.sum() > 0
    assert np.array_equal(h_gpu, h_cpu)
    assert vg == pytest.approx(vc, rel=1e-15)


def test_rdf_2k_atoms(hip_ctx):
    base = H.replicate(H.zif4_frame(), (2, 2, 2))
    packed = H.random_walk(base, 2, 0.05, 5, ortho=True)
    rmax = float(np.min(packed.cell_lengths()) / 2)
    h_gpu, h_cpu, _, _ = _rdf_both(hip_ctx, packed, rmax, int(rmax // 0.01))
    assert np.array_equal(h_gpu, h_cpu)


def test_cn_fixture(hip_ctx):
    packed = H.random_walk(H.zif4_frame(), 4, 0.03, 6)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    rcm = np.zeros((S, S))
    zn, n, c = kinds.index(30), kinds.index(7), kinds.index(6)
    rcm[zn, n] = rcm[n, zn] = 2.5
    rcm[c, n] = rcm[n, c] = 1.6
    sets = [(zn, n), (n, zn), (c, n), (n, c)]
    s_gpu, pa_gpu = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    s_cpu, pa_cpu = clib.cn_counts(packed.pos_host(), packed.cell, sp, S, rcm, sets, per_atom=True)
    assert np.array_equal(s_gpu, s_cpu)
    assert np.array_equal(pa_gpu, pa_cpu)
    assert (s_gpu[:, 0] == 64).all()


def test_bad_fixture(hip_ctx):
    packed = H.random_walk(H.zif4_frame(), 4, 0.03, 7)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    rcm = np.zeros((S, S))
    zn, n, c = kinds.index(30), kinds.index(7), kinds.index(6)
    rcm[zn, n] = rcm[n, zn] = 2.5
    rcm[c, n] = rcm[n, c] = 1.6
    triples = [(zn, n), (n, zn), (c, n), (n, c), (-1, -1), (n, -1)]
    edges = np.arange(3601) * 0.05
    h_gpu, a_gpu = hip_ctx.bad_hist(packed, rcm, triples, edges)
    h_cpu, a_cpu = clib.bad_hist(packed.pos_host(), packed.cell, sp, S, rcm, triples, edges)
    assert np.array_equal(a_gpu, a_cpu)
    assert np.array_equal(h_gpu, h_cpu)
    assert a_gpu[0] == 4 * 96


@pytest.mark.parametrize("unwrap", [False, True])
@pytest.mark.parametrize("tri", [False, True])
def test_msd_vs_reference_loops(hip_ctx, unwrap, tri):
    z = H.zif4_frame()
    packed = H.random_walk(z, 40, 0.3, 8, ortho=not tri)
    window, _ = no.msd_window_setup(40, delta_time=3, timestep=1)
    sumsq, kinds = hip_ctx.msd_window(packed, window, unwrap=unwrap)
    elements, ref = no.window_msd(packed.pos_host(), packed.cell, packed.numbers, packed.masses, window,
                                  unwrap=unwrap)
    for e, r in zip(elements, ref):
        n_e = (packed.numbers == e).sum()
        got = sumsq[kinds.index(int(e))] / n_e / (40 - window)
        np.testing.assert_allclose(got, r, rtol=1e-9, atol=1e-12)
