"""The headline workload at its own launch geometry (BASELINE.json configs[2] and configs[3]).

The bench's numbers come from 9792-atom frames in launches of hundreds to thousands of frames: the XCD-aware chunk
mapping, heavy-first tile-pair order and equal XCD shares of ``rdf_tile_kernel_fast`` are only live there, and so is
the ``msd_comb_kernel<28>`` bucket at F = 5000, W = 25.  These tests tie exactly those launches to the oracle:

* RDF: a 544-frame device-resident launch (2 frames per chunk, 272 chunks, xcd_map on; the bench's 5000 frames run 16 per chunk) against the C oracle on frames
  sampled from different XCD shares -- by leave-one-out (H[0,F) - H[0,k) - H[k+1,F) is frame k's histogram as the
  BIG launches saw it) and as the sum of 16-frame blocks (which take the unmapped geometry);
* MSD: F = 5000, delta_time = 100 on all 9792 atoms against the numpy restatement on a 272-atom slice;
* configs[3]: Bad({'Zn-N': 2.5}, dtheta=0.05) + CoordinationNumber at 9792 atoms against the oracle on sampled
  frames, and the frame-sharded 2-rank run (both ranks on the one GPU of the box) against the single process.
"""

import os
import sys

import numpy as np
import pytest

from amof_amd.frames import PackedTrajectory
from tests import helpers as H
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

REPS = (3, 3, 4)


@pytest.fixture(scope="module")
def traj544():
    import torch
    packed = H.device_walk(torch.device("cuda", 0), REPS, 544, 0.05, 20261003)
    torch.cuda.synchronize()
    return packed


def _host_frames(packed, idx):
    import torch
    pick = torch.as_tensor(np.asarray(idx), device=packed.pos.device)
    return packed.pos[pick].cpu().numpy()


def test_rdf_at_bench_geometry_vs_oracle(hip_ctx, traj544):
    from oracle import clib
    packed = traj544
    F, N = packed.n_frames, packed.n_atoms
    assert (F, N) == (544, 9792)
    rmax = float(np.min(packed.cell_lengths()) / 2)
    nb = int(rmax // 0.01)
    assert nb == 2310
    kinds, sp = H.species_of(packed.numbers)
    full, vol, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
    assert hip_ctx.last_path() == "rdf_tile_zf"     # diagonal cell, slab culling: f32 slab coordinates, always-add histogram
    assert hip_ctx.last_kernel_launches() == 1
    # ordered-pair histograms are symmetric; every frame contributes the same number of pairs only statistically,
    # but the volume sum is exact
    assert np.array_equal(full, full.transpose(1, 0, 2))
    np.testing.assert_allclose(vol, F * abs(np.linalg.det(packed.cell[0])), rtol=1e-14)

    # frames from different XCD shares (share x owns frames [68x, 68x+68)), first and last frames of chunks
    sample = [0, 67, 68, 150, 271, 272, 407, 543]
    pos_s = _host_frames(packed, sample)
    for q, k in enumerate(sample):
        h_cpu, _ = clib.rdf_hist(pos_s[q:q + 1], packed.cell, sp, len(kinds), rmax, nb, cell_list=True)
        # (a) the frame on its own (small launch)
        h_one, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(k, k + 1))
        assert np.array_equal(h_one, h_cpu), "single-frame launch differs from the oracle at frame %d" % k
        # (b) the frame as the big launches saw it
        rest = np.zeros_like(full)
        if k > 0:
            rest += hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(0, k))[0]
        if k + 1 < F:
            rest += hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(k + 1, F))[0]
        assert np.array_equal(full - rest, h_cpu), "leave-one-out differs from the oracle at frame %d" % k

    # the bench's 5000-frame launches run 16 frames per chunk (34 -> 40 chunks here): same histogram
    os.environ["AMOF_RDF_FPC"] = "16"
    try:
        full16, _, _ = hip_ctx.rdf_accumulate(packed, rmax, nb)
    finally:
        del os.environ["AMOF_RDF_FPC"]
    assert np.array_equal(full16, full)
    # checksum of checksums: blocks of 16 frames (34 launches without the XCD mapping) add up to the big launch
    acc = np.zeros_like(full)
    for f0 in range(0, F, 16):
        acc += hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(f0, min(f0 + 16, F)))[0]
    assert np.array_equal(acc, full)


def test_rdf_device_output_equals_host_output(hip_ctx, traj544):
    """amof_rdf_accumulate_dev (the RCCL merge consumes its output in place) == amof_rdf_accumulate"""
    import torch
    packed = traj544
    rmax = float(np.min(packed.cell_lengths()) / 2)
    nb = int(rmax // 0.01)
    host, vol_h, _ = hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(100, 228))
    out = torch.zeros((4, 4, nb), dtype=torch.int64, device="cuda:0")
    got, vol_d, _ = hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(100, 228), out=out)
    torch.cuda.synchronize()
    assert got is out
    assert np.array_equal(out.cpu().numpy().view(np.uint64), host) and vol_d == vol_h
    # accumulates (+=): a second call doubles it
    hip_ctx.rdf_accumulate(packed, rmax, nb, frame_range=(100, 228), out=out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64), 2 * host)


def test_input_produced_by_pending_torch_kernels(hip_ctx):
    """The library runs on its own non-blocking stream: a device-resident trajectory still being written by queued
    torch kernels must be waited for (amof_ctx_wait_stream), not read half-finished."""
    import torch
    base = H.device_walk(torch.device("cuda", 0), (2, 2, 2), 300, 0.05, 5)
    torch.cuda.synchronize()
    rmax, nb = 7.0, 700
    ref, _, _ = hip_ctx.rdf_accumulate(base, rmax, nb)
    a = torch.randn((6144, 6144), dtype=torch.float64, device="cuda:0")
    for _ in range(3):
        buf = torch.zeros_like(base.pos)           # every atom at the origin until the copy below has run
        torch.cuda.synchronize()
        b = a @ a                                  # tens of milliseconds of queued work on torch's stream ...
        b = b @ a
        buf.copy_(base.pos)                        # ... and only then the positions are written
        got, _, _ = hip_ctx.rdf_accumulate(PackedTrajectory(buf, base.cell, base.numbers), rmax, nb)
        assert np.array_equal(got, ref)
        del b


def test_rdf_at_the_bench_launch_shape_through_the_class(hip_ctx):
    """The bench's own launch shape -- 2560 frames x 9792 atoms, 16 frames per chunk (the default: no AMOF_RDF_FPC), one
    tile-kernel launch, through Rdf.from_trajectory -- tied to the oracle by leave-one-out on seven frames spread over the
    launch (first / last frame of a chunk, chunk interiors, first and last frame of the trajectory):
    H[0,F) - H[0,k) - H[k+1,F) is frame k's histogram as the big launch counted it."""
    import torch
    from oracle import clib
    from amof_amd.rdf import Rdf
    assert "AMOF_RDF_FPC" not in os.environ
    F = 2560
    packed = H.device_walk(torch.device("cuda", 0), REPS, F, 0.05, 20261003)
    torch.cuda.synchronize()
    rdf = Rdf.from_trajectory(packed)
    nb = len(rdf.data)
    assert (nb, packed.n_atoms) == (2310, 9792)
    assert rdf._stats["path"] == "rdf_tile_zf" and rdf._stats["kernel_launches"] == 1
    kinds, sp = H.species_of(packed.numbers)
    full = np.asarray(rdf.hist)
    sample = [0, 15, 16, 1277, 1296, 2047, F - 1]
    pos_s = _host_frames(packed, sample)
    for q, k in enumerate(sample):
        h_cpu, _ = clib.rdf_hist(pos_s[q:q + 1], packed.cell, sp, len(kinds), rdf.rmax, nb, cell_list=True)
        rest = np.zeros_like(full)
        for a, b in ((0, k), (k + 1, F)):
            if b > a:
                rest += hip_ctx.rdf_accumulate(packed, rdf.rmax, nb, frame_range=(a, b))[0]
        assert np.array_equal(full - rest, h_cpu), k
    # and the DataFrame is the normalisation of exactly these counts
    from amof_amd.rdf import normalize_rdf
    want = normalize_rdf(full.sum(axis=(0, 1)), F * packed.n_atoms, packed.n_atoms, abs(np.linalg.det(packed.cell[0])), rdf.rmax, nb)
    np.testing.assert_allclose(rdf.data["X-X"].values, want, rtol=1e-13)
    del packed


def test_context_refuses_positions_of_another_device(hip_ctx):
    class FakeCtx(object):
        device = 3
    import torch
    from amof_amd import _hip
    packed = PackedTrajectory(torch.zeros((1, 4, 3), dtype=torch.float64, device="cuda:0"), np.eye(3) * 5, [1, 1, 8, 8])
    with pytest.raises(ValueError, match="cuda:0"):
        _hip.Context._traj(FakeCtx(), packed)


def test_msd_headline_shape_vs_numpy_on_atom_slice(hip_ctx):
    """F = 5000, delta_time = 100 (W = 25: template bucket 28), all 9792 atoms; the oracle on 272 of them."""
    import torch
    from oracle import numpy_oracle as no
    from amof_amd.msd import WindowMsd
    from amof_amd import data as eldata
    packed = H.device_walk(torch.device("cuda", 0), REPS, 5000, 0.05, 20261003)
    torch.cuda.synchronize()
    F, N = 5000, 9792
    window, time = no.msd_window_setup(F, 100, "half", 1)
    assert len(window) == 25
    kinds, sp = H.species_of(packed.numbers)
    # one contiguous atom range holding every species (the first replica of the 272-atom cell)
    a0, a1 = 0, 272
    sumsq, k2 = hip_ctx.msd_window(packed, window, atom_range=(a0, a1))
    assert hip_ctx.last_path() == "msd_stream"      # window spacing 100: one thread per residue class
    pos_h = packed.pos.cpu().numpy()
    mask = np.zeros(N, dtype=bool)
    mask[a0:a1] = True
    elements, ref = no.window_msd_fast(pos_h, packed.cell, packed.numbers, packed.masses, window, atom_subset=mask)
    for e, r in zip(elements, ref):
        n_e = int((packed.numbers[a0:a1] == e).sum())
        got = sumsq[k2.index(int(e))] / n_e / (F - window)
        np.testing.assert_allclose(got, r, rtol=1e-9, atol=1e-12)
    # the whole system: atom ranges add up to the full call, and the class agrees with both
    whole, _ = hip_ctx.msd_window(packed, window)
    parts = sum(hip_ctx.msd_window(packed, window, atom_range=(b, min(b + 2448, N)))[0] for b in range(0, N, 2448))
    np.testing.assert_allclose(parts, whole, rtol=1e-12)
    msd = WindowMsd.from_trajectory(packed, delta_time=100, timestep=1)
    for s, z in enumerate(k2):
        n_z = int((packed.numbers == z).sum())
        np.testing.assert_allclose(msd.data[eldata.chemical_symbols[z]].values, whole[s] / n_z / (F - window),
                                   rtol=1e-12)
    # a random walk with sigma = 0.05 per axis: MSD(m) ~ 3 sigma^2 m (the reference's skipped origin and F-m
    # divisor cost (F-m-1)/(F-m)); a loose physical sanity bound on top of the parity checks
    x = msd.data["X"].values[1:]
    np.testing.assert_allclose(x, 3 * 0.05 ** 2 * window[1:], rtol=0.05)
    del pos_h


def _cfg4_inputs(packed):
    from amof_amd import atom as amatom
    from amof_amd import _hip
    kinds, sp = H.species_of(packed.numbers)
    rcm = amatom.cutoff_matrix(amatom.format_cutoff({'Zn-N': 2.5}), kinds)
    zn, n = kinds.index(30), kinds.index(7)
    return kinds, sp, rcm, zn, n


def test_config3_bad_and_cn_at_9792_atoms(hip_ctx, traj544):
    """BASELINE configs[3]'s extra analyses at full width: Bad({'Zn-N': 2.5}, dtheta=0.05) and
    CoordinationNumber({'Zn-N': 2.5}) on 9792 atoms; oracle on sampled frames + block additivity."""
    from oracle import clib
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    packed = traj544
    F = packed.n_frames
    kinds, sp, rcm, zn, n = _cfg4_inputs(packed)
    bins = int(180 // 0.05)
    edges = np.arange(bins + 2) * 0.05
    triples = [(n, zn), (zn, n)]          # Zn-N-Zn (centre N), N-Zn-N (centre Zn)
    full, nang = hip_ctx.bad_hist(packed, rcm, triples, edges)
    assert hip_ctx.last_path() == "bad_frame"      # Zn + N of a frame fit in LDS: one workgroup sorts and searches it there
    sample = [0, 67, 68, 271, 272, 543]
    pos_s = _host_frames(packed, sample)
    for q, k in enumerate(sample):
        h_cpu, na_cpu = clib.bad_hist(pos_s[q:q + 1], packed.cell, sp, len(kinds), rcm, triples, edges)
        h_one, na_one = hip_ctx.bad_hist(packed, rcm, triples, edges, frame_range=(k, k + 1))
        assert np.array_equal(h_one, h_cpu) and np.array_equal(na_one, na_cpu), k
        rest = np.zeros_like(full)
        rest_n = np.zeros_like(nang)
        for a, b in ((0, k), (k + 1, F)):
            if b > a:
                h, m = hip_ctx.bad_hist(packed, rcm, triples, edges, frame_range=(a, b))
                rest += h
                rest_n += m
        assert np.array_equal(full - rest, h_cpu) and np.array_equal(nang - rest_n, na_cpu), k
    # CN: per-frame sums of the big launch vs the oracle on the sampled frames
    sets = [(zn, n), (n, zn)]
    sums = hip_ctx.cn_count(packed, rcm, sets)
    assert hip_ctx.last_path() in ("cn_cell", "cn_frame")
    s_cpu = clib.cn_counts(pos_s, packed.cell, sp, len(kinds), rcm, sets)
    assert np.array_equal(sums[sample], s_cpu)
    # classes on top: columns and normalisation
    bad = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05)
    assert "N-Zn-N" in bad.data.columns and len(bad.data) == bins + 1
    n_arr = full[1].astype(np.int64)
    np.testing.assert_allclose(bad.data["N-Zn-N"].values, n_arr / np.diff(edges) / n_arr.sum(), rtol=1e-14)
    cn = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5})
    n_zn = int((packed.numbers == 30).sum())
    assert np.array_equal(cn.data['Zn-N'].values, sums[:, 0] / n_zn)
    # a 0.05 A/frame walk dissolves the framework over 544 frames: CN starts at exactly 4
    assert cn.data['Zn-N'].values[0] == 4.0


def _worker_cfg4(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _run_cfg4(None)
    for k, arr in res.items():
        np.save(os.path.join(out_dir, "%s_rank%d.npy" % (k, rank)), arr)
    dist.barrier()
    dist.destroy_process_group()


def _run_cfg4(distributed):
    import torch
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    packed = H.device_walk(torch.device("cuda", 0), REPS, 96, 0.05, 99)       # same seed on every rank
    rdf = Rdf.from_trajectory(packed, device=0, distributed=distributed)
    bad = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05, device=0, distributed=distributed)
    msd = WindowMsd.from_trajectory(packed, delta_time=4, timestep=1, device=0, distributed=distributed)
    cn = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5}, device=0, distributed=distributed)
    return {"rdf_hist": np.asarray(rdf.hist).view(np.uint64), "rdf": rdf.data.values, "bad_hist": np.asarray(bad.hist),
            "bad": bad.data.values, "msd": msd.data.values, "cn": cn.data.values}


def test_config3_two_ranks_equal_single_process(tmp_path):
    """configs[3] sharding at 9792 atoms: frames of RDF / BAD / CN and atoms of MSD split over two ranks (both on
    cuda:0), merged by the collective; integers identical to the single process, MSD to float order."""
    import torch.multiprocessing as mp
    port = 30600 + os.getpid() % 2000
    mp.spawn(_worker_cfg4, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = _run_cfg4(False)
    for k, ref in single.items():
        for rank in (0, 1):
            got = np.load(os.path.join(str(tmp_path), "%s_rank%d.npy" % (k, rank)))
            if k == "msd":
                np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-15)
            elif k == "rdf":
                # g(r) = integer counts (compared exactly as "rdf_hist") x mean volume; the volume sum over the
                # frames is a float sum whose order depends on the sharding (1 ulp)
                np.testing.assert_allclose(got, ref, rtol=1e-14, atol=0)
            else:
                assert np.array_equal(got, ref), k
