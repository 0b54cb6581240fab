"""Host input reaches the GPU once (amof_amd/frames.py: pack_trajectory + ResidentCopy): a list of frames is packed by the
native copy into page-locked memory and uploaded while it is packed, a host PackedTrajectory is uploaded by a background
thread, the first analysis walks the frames that have arrived, every later analysis of the same object reads the kept
copy -- results identical to the device-resident trajectory's, whatever the route."""
import numpy as np
import pytest

from amof_amd import frames as fr
from tests import helpers as H

pytestmark = pytest.mark.gpu

CUT = {'Zn-N': 2.5, 'C-N': 1.6}


def _all_four(traj):
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    return (Rdf.from_trajectory(traj, dr=0.02), WindowMsd.from_trajectory(traj, delta_time=20, timestep=1),
            Bad.from_trajectory(traj, CUT, dtheta=0.5), CoordinationNumber.from_trajectory(traj, CUT))


def _same(got, ref):
    assert got[0].data.equals(ref[0].data) and np.array_equal(got[0].hist, ref[0].hist)
    np.testing.assert_allclose(got[1].sumsq, ref[1].sumsq, rtol=1e-13, atol=0)
    assert list(got[1].data.columns) == list(ref[1].data.columns)
    assert got[2].data.equals(ref[2].data) and np.array_equal(got[2].hist, ref[2].hist)
    assert got[3].data.equals(ref[3].data)


@pytest.fixture(scope="module")
def walk():
    base = H.replicate(H.zif4_frame(), (2, 2, 2))
    return H.random_walk(base, 720, 0.05, 33)            # 720 x 2176 atoms: 37.6 MB of positions


def test_list_of_frames_is_packed_and_uploaded_once(hip_ctx, walk):
    fr.forget_packed_lists()
    ref = _all_four(walk.to_device(0))
    frames = [fr.Frame(walk.numbers, walk.pos[k], walk.cell_of(k)) for k in range(walk.n_frames)]
    got = _all_four(frames)                              # four constructors on the SAME list
    assert len(fr._PACKED_LISTS) == 1
    packed = fr._PACKED_LISTS[0]["packed"]
    assert packed._resident.device == 0
    _same(got, ref)
    # the copy is kept, the packed host array is read-only while it exists
    assert packed._resident.complete and packed._dev_pos is not None and packed._dev_pos.is_cuda
    assert not packed.pos.flags.writeable
    assert np.array_equal(packed.pos, walk.pos)
    # the same, unchanged list again: no new pack
    again = _all_four(frames)
    assert len(fr._PACKED_LISTS) == 1 and fr._PACKED_LISTS[0]["packed"] is packed
    _same(again, ref)
    # one coordinate of one frame edited in place: the checksum notices, the list is packed afresh
    frames[301].positions[17, 2] += 0.25
    edited = _all_four(frames)
    assert fr._PACKED_LISTS[0]["packed"] is not packed
    changed = walk.pos.copy()
    changed[301, 17, 2] += 0.25
    _same(edited, _all_four(fr.PackedTrajectory(changed, walk.cell, walk.numbers).to_device(0)))
    fr.forget_packed_lists()


def test_host_packed_trajectory_gets_one_resident_copy(hip_ctx, walk, monkeypatch):
    ref = _all_four(walk.to_device(0))
    host = fr.PackedTrajectory(walk.pos.copy(), walk.cell, walk.numbers)
    got = _all_four(host)
    assert host.__dict__.get("_resident") is not None and host._resident.device == 0
    _same(got, ref)
    assert host._resident.complete and host._dev_pos is not None and not host.pos.flags.writeable
    with pytest.raises(ValueError):
        host.pos[0, 0, 0] = 1.0                          # (the stale-copy trap is closed, as with keep_on_device)
    host.release_device()
    assert host.pos.flags.writeable and host.__dict__.get("_resident") is None and host._dev_pos is None
    host.pos[5, 3, 1] += 0.1
    changed = fr.PackedTrajectory(host.pos.copy(), walk.cell, walk.numbers).to_device(0)
    _same(_all_four(host), _all_four(changed))           # a fresh copy of the edited array
    host.release_device()
    # switched off: staged per call, as before -- same numbers
    monkeypatch.setenv("AMOF_KEEP_ON_DEVICE", "0")
    plain = fr.PackedTrajectory(walk.pos.copy(), walk.cell, walk.numbers)
    _same(_all_four(plain), ref)
    assert plain.__dict__.get("_resident") is None and getattr(plain, "_dev_pos", None) is None and plain.pos.flags.writeable
