"""Property tests (hypothesis) of the C oracle against the independent brute-force numpy oracle:
random triclinic / partially periodic cells down to sizes where the cutoff exceeds the cell (true
periodic images, self images), random species, cutoffs and bins."""

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import clib, numpy_oracle as no


@st.composite
def systems(draw):
    seed = draw(st.integers(0, 2**31 - 1))
    rng = np.random.default_rng(seed)
    n = draw(st.integers(1, 14))
    S = draw(st.integers(1, 3))
    scale = draw(st.sampled_from([1.2, 2.5, 5.0]))           # tiny cells: the cutoff covers several images
    cell = np.diag(rng.uniform(0.8, 1.4, 3) * scale)
    if draw(st.booleans()):
        cell = cell + np.tril(rng.uniform(-0.4, 0.4, (3, 3)) * scale, k=-1)
    pbc = tuple(draw(st.lists(st.booleans(), min_size=3, max_size=3)))
    pos = rng.uniform(-0.6, 1.6, (n, 3)) @ cell
    sp = rng.integers(0, S, n).astype(np.int32)
    return pos, cell, pbc, sp, S, rng


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(systems(), st.sampled_from([0.7, 1.9, 3.1]), st.sampled_from([1, 7, 64]))
def test_rdf_equals_bruteforce(sysm, rmax, nb):
    pos, cell, pbc, sp, S, _ = sysm
    h_c, _ = clib.rdf_hist(pos, cell, sp, S, rmax, nb, pbc=pbc)
    h_cl, _ = clib.rdf_hist(pos, cell, sp, S, rmax, nb, pbc=pbc, cell_list=True)
    h_np = no.rdf_hist(pos, cell, sp, S, rmax, nb, pbc=pbc)
    assert np.array_equal(h_c, h_np) and np.array_equal(h_cl, h_np)
    assert np.array_equal(h_c, h_c.transpose(1, 0, 2))        # ordered-pair histograms are symmetric in (a, b)


@settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck))
@given(systems())
def test_cn_and_angles_equal_bruteforce(sysm):
    pos, cell, pbc, sp, S, rng = sysm
    rcm = rng.uniform(0.5, 2.2, (S, S))
    rcm = np.maximum(rcm, rcm.T) * (rng.uniform(0, 1, (S, S)) < 0.8)
    rcm = np.maximum(rcm, rcm.T)
    sets = [(a, b) for a in range(S) for b in range(S)]
    sums = clib.cn_counts(pos, cell, sp, S, rcm, sets, pbc=pbc)
    ref = no.cn_sums(pos, cell, sp, S, rcm, sets, pbc=pbc)
    assert [int(x) for x in sums[0]] == [int(r.sum()) for r in ref]
    # angles: ASE's get_angles(mic=True) works on atom INDICES (minimum-image vectors), so the two oracles only
    # describe the same thing when no neighbour is a farther periodic image
    if len(clib.images(cell, float(rcm.max()) if rcm.size else 0.0, pbc)) > 0 or rcm.max() <= 0:
        return
    for A in range(S):
        for B in range(S):
            try:
                a_np = np.sort(np.asarray(no.angles(pos, cell, sp, S, rcm, A, B, pbc=pbc), dtype=float))
            except ZeroDivisionError:
                with pytest.raises(ZeroDivisionError):
                    clib.angles(pos, cell, sp, S, rcm, A, B, pbc=pbc)
                continue
            a_c = np.sort(np.asarray(clib.angles(pos, cell, sp, S, rcm, A, B, pbc=pbc), dtype=float))
            assert a_c.shape == a_np.shape
            np.testing.assert_allclose(a_c, a_np, rtol=0, atol=1e-5)   # (acos near 0 / 180 deg amplifies rounding)
