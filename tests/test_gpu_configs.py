"""BASELINE.json configs[0] and configs[1] as parity cases (configs[2] is the bench line and has
its own full-width test in test_gpu_edges.py; configs[3] is covered by test_gpu_dist.py and
bench.py --with-bad; configs[4] by test_gpu_large.py)."""

import numpy as np
import pytest

from amof_amd.frames import PackedTrajectory
from amof_amd.rdf import Rdf
from amof_amd.cn import CoordinationNumber
from oracle import clib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_cfg1_216_atom_cubic_and_272_atom_fixture_50_frames(hip_ctx):
    # 6x6x6 simple-cubic cell (a = 2.5 A, rock-salt colouring), 50 frames of a wrapped Gaussian walk
    a, n = 2.5, 6
    g = np.arange(n) * a
    pos = np.array([[x, y, z] for x in g for y in g for z in g], dtype=float)
    idx = np.array([[i, j, k] for i in range(n) for j in range(n) for k in range(n)])
    from amof_amd.frames import Frame
    sc = Frame(np.where(idx.sum(axis=1) % 2 == 0, 11, 17), pos, np.diag([n * a] * 3))
    for base, bins in [(sc, 749), (H.zif4_frame(), 770)]:
        packed = H.random_walk(base, 50, 0.05, 20261003)
        rdf = Rdf.from_trajectory(packed)                       # defaults: dr = 0.01, rmax = half cell
        assert len(rdf.data) == bins and rdf.n_frames == 50
        kinds, sp = H.species_of(packed.numbers)
        ref, vol = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), rdf.rmax, bins, cell_list=True)
        assert np.array_equal(rdf.hist, ref)
        # normalisation against an independent evaluation of the asap3 formula
        d = rdf.rmax / bins
        r = (np.arange(bins) + 0.5) * d
        shell = 4 * np.pi / 3 * ((r + d / 2) ** 3 - (r - d / 2) ** 3)      # exact shell volume
        g_tot = ref.sum(axis=(0, 1)) * (vol / 50) / (shell * len(base) * 50 * len(base))
        np.testing.assert_allclose(rdf.data["X-X"].values, g_tot, rtol=1e-12)


def test_cfg2_2k_atoms_1000_frames_partial_rdf_and_cn(hip_ctx):
    base = H.replicate(H.zif4_frame(), (2, 2, 2))
    assert len(base) == 2176
    packed = H.random_walk(base, 1000, 0.02, 20261003).to_device(0)
    rdf = Rdf.from_trajectory(packed, dr=0.01, rmax=10.0)          # below half cell: no clamp
    assert rdf.rmax == 10.0 and len(rdf.data) == 999                # int(10 // 0.01) == 999
    kinds, sp = H.species_of(packed.numbers)
    zn, n = kinds.index(30), kinds.index(7)
    # oracle on the first 40 frames; linearity over frame blocks for the rest
    host = packed.pos_host()
    ref, _ = clib.rdf_hist(host[:40], packed.cell, sp, 4, 10.0, 999, cell_list=True)
    first, _, _ = hip_ctx.rdf_accumulate(packed, 10.0, 999, frame_range=(0, 40))
    assert np.array_equal(first, ref)
    blocks = sum(hip_ctx.rdf_accumulate(packed, 10.0, 999, frame_range=(k, min(k + 250, 1000)))[0]
                 for k in range(0, 1000, 250))
    assert np.array_equal(blocks, rdf.hist)
    # coordination number: integer counts bit-exact with the oracle
    cn = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5})
    assert cn.data.shape == (1000, 2) and (cn.data["Zn-N"].values[:5] == 4.0).all()      # (the random walk melts it later)
    rcm = np.zeros((4, 4)); rcm[zn, n] = rcm[n, zn] = 2.5
    ref_cn = clib.cn_counts(host[:40], packed.cell, sp, 4, rcm, [(zn, n)])
    assert np.array_equal(ref_cn[:, 0], (cn.data["Zn-N"].values[:40] * 128).astype(np.int64))
    # RDF <-> CN integer identity: pairs below the cutoff, counted through the histogram
    below = rdf.hist[zn, n][:249].sum(), rdf.hist[zn, n][:251].sum()   # bins ending <= 2.49 / 2.51 A
    total = int(cn.data["Zn-N"].values.sum() * 128)
    assert below[0] <= total <= below[1]
