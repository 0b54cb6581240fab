"""BASELINE configs[4]-shaped case: ~100k atoms, sheared (triclinic) cell, rmax = 10 A."""

import os
import time

import numpy as np
import pytest

from amof_amd.frames import Frame, PackedTrajectory
from amof_amd.rdf import Rdf
from oracle import clib
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_cfg5_100k_triclinic_rdf(hip_ctx):
    base = H.replicate(H.zif4_frame(), (7, 7, 8))
    assert len(base) == 106624
    shear = np.eye(3) + np.array([[0, 0.15, 0.10], [0, 0, 0.20], [0, 0, 0]])     # upper-triangular affine shear
    cell = base.cell @ shear
    pos = base.positions @ shear
    sheared = Frame(base.numbers, pos, cell)
    packed = H.random_walk(sheared, 3, 0.05, 51)
    rdf = Rdf.from_trajectory(packed, dr=0.01, rmax=10.0)
    assert len(rdf.data) == 999 and rdf.rmax == 10.0                 # int(10 // 0.01) == 999
    kinds, sp = H.species_of(packed.numbers)
    ref, vol = clib.rdf_hist(packed.pos[:1], packed.cell, sp, 4, 10.0, 999, cell_list=True)
    one, _, _ = hip_ctx.rdf_accumulate(packed, 10.0, 999, frame_range=(0, 1))
    assert np.array_equal(one, ref)
    assert hip_ctx.last_path() == "rdf_cell"                        # the 3-D cell-list kernel serves this config
    t0 = time.perf_counter()
    hip_ctx.rdf_accumulate(packed, 10.0, 999)
    dt = time.perf_counter() - t0
    print("cfg5-shaped RDF: %.1f ms per frame (3 frames, incl. H2D of %d MB)" % (1e3 * dt / 3, packed.pos.nbytes >> 20))
    # g(r) -> N_b / N at large r for partials normalised with the total density
    tail = rdf.data["X-X"].values[-50:].mean()
    assert 0.9 < tail < 1.1


def test_cfg5_device_resident_launch_across_frame_batches(hip_ctx):
    """configs[4] as the bench runs it: 106 624 atoms in a sheared cell, device resident, the 3-D cell-list kernel over
    SEVERAL frame batches (the bench's 2000 frames = 5.1 GB take five 1-GiB-of-scratch batches; here AMOF_RDF_BATCH
    cuts 20 frames into 8 + 8 + 4 so that boundaries are crossed without gigabytes).  The frames either side of the
    first boundary are tied to the oracle by leave-one-out on the whole launch (H[0,k) + oracle(k) + H[k+1,F) ==
    H[0,F)), the blocks add up, and the batched launch equals the unbatched one."""
    import torch
    base = H.replicate(H.zif4_frame(), (7, 7, 8))
    shear = np.eye(3) + np.array([[0, 0.15, 0.10], [0, 0, 0.20], [0, 0, 0]])
    sheared = Frame(base.numbers, base.positions @ shear, base.cell @ shear)
    F = 20
    host = H.random_walk(sheared, F, 0.05, 52)
    packed = PackedTrajectory(torch.as_tensor(host.pos).to("cuda:0"), host.cell, host.numbers)
    kinds, sp = H.species_of(packed.numbers)
    whole, _, _ = hip_ctx.rdf_accumulate(packed, 10.0, 999)            # one batch
    assert hip_ctx.last_path() == "rdf_cell"
    old = os.environ.get("AMOF_RDF_BATCH")
    os.environ["AMOF_RDF_BATCH"] = "8"
    try:
        batched, _, _ = hip_ctx.rdf_accumulate(packed, 10.0, 999)
        assert hip_ctx.last_path() == "rdf_cell" and hip_ctx.last_kernel_launches() == 3
        assert np.array_equal(batched, whole)
        for k in (7, 8):
            ref, _ = clib.rdf_hist(host.pos[k:k + 1], host.cell, sp, 4, 10.0, 999, cell_list=True)
            acc = ref + hip_ctx.rdf_accumulate(packed, 10.0, 999, frame_range=(0, k))[0] + \
                hip_ctx.rdf_accumulate(packed, 10.0, 999, frame_range=(k + 1, F))[0]
            assert np.array_equal(acc, batched), k
        parts = hip_ctx.rdf_accumulate(packed, 10.0, 999, frame_range=(0, 8))[0] + \
            hip_ctx.rdf_accumulate(packed, 10.0, 999, frame_range=(8, F))[0]
        assert np.array_equal(parts, batched)
    finally:
        if old is None:
            os.environ.pop("AMOF_RDF_BATCH", None)
        else:
            os.environ["AMOF_RDF_BATCH"] = old


def test_cell_sort_counter_overflow_falls_back(hip_ctx):
    """round 5: `cell_sort_frame_kernel` keeps the key counters as 16-bit halves in LDS.  A frame with more than 65 535 atoms in
    one cell cannot be sorted that way: the kernel must notice (the scan's total differs from N), leave an EMPTY table behind
    (nothing the pair kernel could walk out of bounds with), raise the flag -- and the call must still return the right
    histogram through the exact kernels.  68 000 atoms in a 0.3 A cube of a 40 A box, against the same call with the cell list
    switched off and (a 3 000-atom slice, same geometry) against the oracle."""
    rng = np.random.default_rng(5)
    N = 70000
    cell = np.diag([40.0, 40.0, 40.0])
    pos = 20.1 + rng.uniform(0.0, 0.3, (1, N, 3))                    # (inside ONE cell of the 32 x 33 x 33 grid the call picks)
    pos[0, :2000] = rng.uniform(0.0, 40.0, (2000, 3))                # a thin gas around the clump
    packed = PackedTrajectory(pos, cell, np.full(N, 8))
    old = {k: os.environ.get(k) for k in ("AMOF_RDF_FORCE_CELL", "AMOF_RDF_NOCELL")}
    try:
        os.environ["AMOF_RDF_FORCE_CELL"] = "1"
        got, _, _ = hip_ctx.rdf_accumulate(packed, 2.0, 20)
        path = hip_ctx.last_path()
        del os.environ["AMOF_RDF_FORCE_CELL"]
        os.environ["AMOF_RDF_NOCELL"] = "1"
        other, _, _ = hip_ctx.rdf_accumulate(packed, 2.0, 20)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert path != "rdf_cell", path                                  # the sorted table was refused
    assert np.array_equal(got, other) and int(got.sum()) > N * 60000
    small = PackedTrajectory(pos[:, 1000:4000], cell, np.full(3000, 8))
    os.environ["AMOF_RDF_FORCE_CELL"] = "1"
    try:
        g2, _, _ = hip_ctx.rdf_accumulate(small, 2.0, 20)
    finally:
        os.environ.pop("AMOF_RDF_FORCE_CELL", None)
        if old["AMOF_RDF_FORCE_CELL"] is not None:
            os.environ["AMOF_RDF_FORCE_CELL"] = old["AMOF_RDF_FORCE_CELL"]
    ref, _ = clib.rdf_hist(small.pos, small.cell, np.zeros(3000, dtype=np.int32), 1, 2.0, 20, cell_list=True)
    assert np.array_equal(g2, ref)
