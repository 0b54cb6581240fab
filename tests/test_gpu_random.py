"""Seeded randomised parity sweep: random cells (orthorhombic / triclinic, cubic / elongated /
flat), sizes, species mixes, cutoffs and bin counts -- HIP vs the CPU oracle, integers bit-exact."""

import numpy as np
import pytest

from amof_amd.frames import PackedTrajectory
from oracle import clib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(1, 5))
    N = int(rng.choice([2, 3, 17, 64, 129, 300, 700]))
    L = rng.uniform(6.0, 14.0, 3) * rng.choice([1.0, 1.0, 3.0], 3)          # some axes elongated
    cell = np.diag(L)
    if seed % 2:
        cell = cell + np.tril(rng.uniform(-0.25, 0.25, (3, 3)) * L[:, None], k=-1)
    F = int(rng.integers(1, 5))
    kinds = list(rng.choice([1, 6, 7, 8, 14, 30], size=S, replace=False))
    numbers = rng.choice(kinds, size=N)
    numbers[:min(S, N)] = kinds[:min(S, N)]                                  # (nearly) every species present
    pos = (rng.uniform(0, 1, (F, N, 3)) + rng.integers(-2, 3, (F, N, 3))) @ cell
    if seed % 3 == 0 and F > 1:                                              # changing cell
        cells = np.array([cell * (1 + 0.02 * rng.normal()) for _ in range(F)])
    else:
        cells = cell
    return rng, PackedTrajectory(pos, cells, numbers)


@pytest.mark.parametrize("seed", range(24))
def test_random_rdf(hip_ctx, seed):
    rng, packed = _case(seed)
    kinds, sp = H.species_of(packed.numbers)
    hmin = min(1.0 / np.linalg.norm(np.linalg.inv(c), axis=0).max() for c in packed.cell)
    rmax = float(rng.uniform(0.2, 1.3) * hmin / 2 if seed % 4 else np.min(packed.cell_lengths()) / 2)
    nb = int(rng.choice([1, 7, 100, 999, 2310]))
    h, vol, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
    ref, vref = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rmax, nb)
    assert np.array_equal(h, ref), (seed, packed.n_atoms, rmax, nb)
    assert vol == pytest.approx(vref, rel=1e-14)


@pytest.mark.parametrize("seed", range(16))
def test_random_cn_bad(hip_ctx, seed):
    rng, packed = _case(100 + seed)
    kinds, sp = H.species_of(packed.numbers)
    S = len(kinds)
    rcm = rng.uniform(0.0, 2.2, (S, S)) * (rng.uniform(0, 1, (S, S)) < 0.7)
    rcm = np.maximum(rcm, rcm.T)
    sets = [(a, b) for a in range(S) for b in range(S)]
    triples = [(a, b) for a in range(-1, S) for b in range(-1, S)][:8]
    edges = np.arange(int(180 // 2.5) + 2) * 2.5
    s_gpu, pa_gpu = hip_ctx.cn_count(packed, rcm, sets, per_atom=True)
    s_ref, pa_ref = clib.cn_counts(packed.pos, packed.cell, sp, S, rcm, sets, per_atom=True)
    assert np.array_equal(s_gpu, s_ref) and np.array_equal(pa_gpu, pa_ref), seed
    try:
        h_ref, a_ref = clib.bad_hist(packed.pos, packed.cell, sp, S, rcm, triples, edges)
    except ZeroDivisionError:
        with pytest.raises(ZeroDivisionError):
            hip_ctx.bad_hist(packed, rcm, triples, edges)
        return
    h_gpu, a_gpu = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert np.array_equal(a_gpu, a_ref) and np.array_equal(h_gpu, h_ref), seed
