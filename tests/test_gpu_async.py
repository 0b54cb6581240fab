"""Asynchronous constructors on the GPU: the four classes requested back to back on one device-resident trajectory,
nothing looked at in between, against the strictly sequential run (AMOF_ASYNC=0) -- bit for bit -- and against the
oracle.  RDF runs on the device's first context, MSD / BAD / CN on the second (a stream of the highest priority)."""
import pickle

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _requests(traj):
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    return (Rdf.from_trajectory(traj),
            WindowMsd.from_trajectory(traj, delta_time=10, timestep=1),
            Bad.from_trajectory(traj, {'Zn-N': 2.5, 'C-N': 1.6}, dtheta=0.05),
            CoordinationNumber.from_trajectory(traj, {'Zn-N': 2.5, 'C-N': 1.6}))


def test_four_classes_unsynchronised_equal_the_sequential_run(hip_ctx, monkeypatch):
    import torch
    from amof_amd import _hip
    from oracle import clib
    traj = H.device_walk(torch.device("cuda", 0), (2, 2, 2), 400, 0.05, 77)
    torch.cuda.synchronize()
    monkeypatch.setenv("AMOF_ASYNC", "0")
    seq = _requests(traj)
    assert all(o.__dict__.get("_pending") is None and o._ctx is hip_ctx for o in seq)
    monkeypatch.setenv("AMOF_ASYNC", "1")
    lane1 = _hip.get_context(0, lane=1)
    assert lane1 is not hip_ctx and lane1.high_priority and not hip_ctx.high_priority
    for rep in range(3):
        objs = _requests(traj)                    # four constructors, nothing looked at
        assert objs[0]._ctx is hip_ctx and all(o._ctx is lane1 for o in objs[1:])
        order = (2, 0, 3, 1) if rep == 1 else (0, 1, 2, 3)
        for k in order:
            assert objs[k].data.equals(seq[k].data), k
        assert np.array_equal(objs[0].hist, seq[0].hist) and objs[0].rmax == seq[0].rmax
        assert np.array_equal(objs[1].sumsq, seq[1].sumsq)
        assert np.array_equal(objs[2].hist, seq[2].hist) and np.array_equal(objs[2].n_angles, seq[2].n_angles)
        assert objs[0]._stats["path"].startswith("rdf_tile") and objs[1]._stats["path"].startswith("msd_")
        assert objs[0]._stats["kernel_s_dominant"] > 0 and objs[2]._stats["path"].startswith("bad_")
    # and the sequential run is the oracle's: RDF counts of two frames by leave-one-out
    kinds, sp = H.species_of(traj.numbers)
    nb = len(seq[0].data)
    pos = traj.pos[:2].cpu().numpy()
    h01, _ = clib.rdf_hist(pos, traj.cell, sp, len(kinds), seq[0].rmax, nb, cell_list=True)
    rest = hip_ctx.rdf_accumulate(traj, seq[0].rmax, nb, frame_range=(2, 400))[0]
    assert np.array_equal(h01 + rest, seq[0].hist)
    # pickling a pending result waits for it
    again = pickle.loads(pickle.dumps(_requests(traj)[1]))
    assert again.data.equals(seq[1].data)


def test_library_errors_surface_at_access(hip_ctx):
    from amof_amd import _hip
    from amof_amd.frames import PackedTrajectory
    from amof_amd.rdf import Rdf
    from amof_amd.bad import Bad
    z = H.random_walk(H.zif4_frame(), 2, 0.01, 0)
    flat = PackedTrajectory(z.pos, np.array([[10.0, 0, 0], [0, 10.0, 0], [10.0, 10.0, 0]]), z.numbers)     # singular cell
    rdf = Rdf.from_trajectory(flat, dr=0.05, rmax=2.0)
    for _ in range(2):
        with pytest.raises(_hip.AmofError) as e:
            rdf.data
        assert e.value.code == _hip.AMOF_ESINGULAR
    pos = np.array([[[1.0, 1, 1], [2.0, 1, 1], [1.0, 1, 1]]])
    twin = PackedTrajectory(pos, np.diag([20.0, 20, 20]), [30, 7, 7])      # two N on top of each other... of the same Zn
    bad = Bad.from_trajectory(twin, {'Zn-N': 1.5}, dtheta=1.0)
    with pytest.raises(ZeroDivisionError):                  # what ASE raises for an undefined angle (amof/bad.py:100)
        bad.data
    # the contexts are in working order afterwards
    ok = Rdf.from_trajectory(z, dr=0.05, rmax=2.0)
    assert len(ok.data) == int(2.0 // 0.05)


def test_second_lane_queues_its_kernels_behind_the_rdf_launch(hip_ctx, monkeypatch):
    """amof_ctx_follow (include/amof_hip.h): lane 1 waits until lane 0's pending job has queued its dominant kernel and
    orders its stream behind it -- the MSD kernels run after the RDF tile kernel, not between its workgroups"""
    import time
    import torch
    from amof_amd import _hip
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    traj = H.device_walk(torch.device("cuda", 0), (2, 2, 2), 4000, 0.05, 78)       # an RDF of ~3 ms
    torch.cuda.synchronize()
    lane1 = _hip.get_context(0, lane=1)
    assert lane1._follows is hip_ctx and hip_ctx._follows is None
    assert lane1.device_calls() == int(lane1._lib.amof_ctx_calls(lane1._h))
    monkeypatch.setenv("AMOF_LANE_ORDER", "1")                                     # (opt-in: profiles/r05/lane_order.txt)
    alone = WindowMsd.from_trajectory(traj, delta_time=50, timestep=1)             # leader idle: nothing to follow
    alone.data
    t_alone = alone._stats["kernel_s_all"]
    for rep in range(3):
        c0 = hip_ctx.device_calls()
        rdf = Rdf.from_trajectory(traj)
        msd = WindowMsd.from_trajectory(traj, delta_time=50, timestep=1)
        t0 = time.perf_counter()
        assert np.array_equal(msd.sumsq, alone.sumsq)                              # (waits for the MSD: it ran behind the RDF)
        waited = time.perf_counter() - t0
        t_rdf = rdf._stats["kernel_s_dominant"]
        assert hip_ctx.device_calls() == c0 + 1
        assert waited > 0.5 * t_rdf, (waited, t_rdf)
        # HIP events around the MSD's kernels: queued behind the stream order, they do not contain the tile kernel
        assert msd._stats["kernel_s_all"] < max(4 * t_alone, 0.25 * t_rdf), (msd._stats, t_alone, t_rdf)
    # the default, unordered lanes: results unchanged
    monkeypatch.delenv("AMOF_LANE_ORDER")
    rdf2 = Rdf.from_trajectory(traj)
    msd2 = WindowMsd.from_trajectory(traj, delta_time=50, timestep=1)
    assert np.array_equal(msd2.sumsq, alone.sumsq) and rdf2.data.equals(rdf.data)
    # the C entry point: no host wait asked for -> ordered at once; a call number that never comes -> 0 after the timeout
    assert lane1._lib.amof_ctx_follow(lane1._h, hip_ctx._h, 0, 0.0) == 1
    assert lane1._lib.amof_ctx_follow(lane1._h, hip_ctx._h, hip_ctx.device_calls() + 5, 0.002) == 0
    assert lane1._lib.amof_ctx_follow(lane1._h, lane1._h, 0, 0.0) == 1
