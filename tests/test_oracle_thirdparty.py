"""The C oracle against an independent third-party implementation that IS installed here:
scipy's periodic cKDTree (orthorhombic boxes).  The reference's own neighbour engines (asap3,
ase.neighborlist) are absent, so this is the closest available outside check of the pair search:
same pairs, same counts, same histogram up to pairs that sit within rounding of a bin edge."""

import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import clib
from tests import helpers as H


def _case(seed, reps=(2, 2, 2)):
    packed = H.random_walk(H.replicate(H.zif4_frame(), reps), 2, 0.08, seed, ortho=True)
    L = np.diag(packed.cell[0]).copy()
    pos = packed.pos - np.floor(packed.pos / L) * L          # cKDTree wants [0, L)
    pos = np.where(pos >= L, pos - L, pos)
    kinds, sp = H.species_of(packed.numbers)
    return packed, pos, L, kinds, sp


@pytest.mark.parametrize("seed", [41, 42])
def test_cn_counts_equal_ckdtree(seed):
    packed, pos, L, kinds, sp = _case(seed)
    S = len(kinds)
    zn, n, c = kinds.index(30), kinds.index(7), kinds.index(6)
    rcm = np.zeros((S, S))
    rcm[zn, n] = rcm[n, zn] = 2.5
    rcm[c, n] = rcm[n, c] = 1.6
    rcm[c, c] = 1.7
    sets = [(zn, n), (n, zn), (c, n), (n, c), (c, c)]
    sums, per_atom = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)
    for f in range(packed.n_frames):
        for k, (a, b) in enumerate(sets):
            ia, ib = np.nonzero(sp == a)[0], np.nonzero(sp == b)[0]
            tree = cKDTree(pos[f][ib], boxsize=L)
            nb = tree.query_ball_point(pos[f][ia], r=rcm[a, b], return_length=True)
            if a == b:
                nb = nb - 1                                   # the centre itself
            assert int(nb.sum()) == int(sums[f, k])
            assert np.array_equal(nb, per_atom[f, k, ia])


@pytest.mark.parametrize("seed,rmax,nbins", [(43, 9.0, 900), (44, 15.0, 1499)])
def test_rdf_histogram_equals_ckdtree(seed, rmax, nbins):
    packed, pos, L, kinds, sp = _case(seed)
    S = len(kinds)
    assert rmax < 0.5 * L.min()
    ref, _ = clib.rdf_hist(packed.pos, packed.cell, sp, S, rmax, nbins)
    dr = rmax / nbins
    total = np.zeros(nbins, dtype=np.int64)
    near_edge = 0
    for f in range(packed.n_frames):
        tree = cKDTree(pos[f], boxsize=L)
        pairs = tree.sparse_distance_matrix(tree, rmax * (1 + 1e-9), output_type="ndarray")
        d = pairs["v"][pairs["i"] != pairs["j"]]              # ordered pairs, both directions
        q = d / dr
        near_edge += int((np.abs(q - np.rint(q)) < 1e-7).sum())
        b = q.astype(np.int64)
        total += np.bincount(b[(d < rmax) & (b < nbins)], minlength=nbins)[:nbins]
    mine = ref.reshape(S * S, nbins).sum(axis=0).astype(np.int64)
    assert mine.sum() > 5e5
    assert np.abs(mine - total).sum() <= 2 * near_edge        # identical unless a pair sits on an edge
    if near_edge == 0:
        assert np.array_equal(mine, total)
