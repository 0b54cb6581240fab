"""Packing a list of frames (amof_amd/frames.py pack_trajectory -> amof_pack_frames / amof_frames_checksum, host only):
the native copy equals the Python loop, the checksum sees every byte, an unchanged list is recognised and a changed one
is not (no GPU needed: without one the packed array is ordinary memory and nothing is uploaded)."""
import numpy as np

from amof_amd import frames as fr
from tests import helpers as H


def test_unchanged_list_is_packed_once_and_a_changed_one_again():
    fr.forget_packed_lists()
    z = H.replicate(H.zif4_frame(), (2, 2, 2))
    rng = np.random.default_rng(1)
    frames = [fr.Frame(z.numbers, z.positions + rng.normal(scale=0.05, size=z.positions.shape), z.cell) for _ in range(200)]
    a = fr.pack_trajectory(frames)                     # 200 x 2176 x 24 B = 10 MB: the native path
    assert a.pos.shape == (200, 2176, 3) and all(np.array_equal(a.pos[k], frames[k].positions) for k in (0, 57, 199))
    assert a.cell.shape == (1, 3, 3) and np.array_equal(a.numbers, z.numbers)
    assert fr.pack_trajectory(frames) is a
    frames[57].positions[2000, 1] = np.nextafter(frames[57].positions[2000, 1], 9.0)      # one ulp of one coordinate
    b = fr.pack_trajectory(frames)
    assert b is not a and np.array_equal(b.pos[57], frames[57].positions) and not np.array_equal(b.pos[57], a.pos[57])
    frames[3] = fr.Frame(z.numbers, z.positions, z.cell)                                     # a frame replaced
    c = fr.pack_trajectory(frames)
    assert c is not b and np.array_equal(c.pos[3], z.positions)
    other = list(frames)                                                                     # another list object
    assert fr.pack_trajectory(other) is not c
    frames[9].cell = z.cell * 1.01                                                           # a cell changed, positions not
    d = fr.pack_trajectory(frames)
    assert d.cell.shape == (200, 3, 3)
    import pytest
    with pytest.raises(ValueError):
        fr.pack_trajectory(frames[:5] + [fr.Frame(z.numbers[:10], z.positions[:10], z.cell)])
    with pytest.raises(ValueError):
        fr.pack_trajectory([])
    fr.forget_packed_lists()


def test_native_pack_equals_the_python_loop_and_checksums_see_every_byte():
    import ctypes
    from amof_amd import _hip
    lib = _hip.load_library()
    rng = np.random.default_rng(3)
    F, n = 40, 1000
    arrays = [rng.normal(size=(n, 3)) for _ in range(F)]
    ptrs = np.array([a.__array_interface__["data"][0] for a in arrays], dtype=np.uint64)
    dst = np.zeros((F, n, 3))
    sums = np.zeros(F, dtype=np.uint64)
    assert lib.amof_pack_frames(ctypes.c_void_p(ptrs.ctypes.data), F, n, ctypes.c_void_p(dst.ctypes.data),
                                ctypes.c_void_p(sums.ctypes.data), 5) == 0
    assert np.array_equal(dst, np.stack(arrays))
    again = np.zeros(F, dtype=np.uint64)
    assert lib.amof_frames_checksum(ctypes.c_void_p(ptrs.ctypes.data), F, n, ctypes.c_void_p(again.ctypes.data), 3) == 0
    assert np.array_equal(again, sums) and len(set(sums.tolist())) == F
    for word in (0, 1, 2, 3, 1499, 2998, 2999):          # every lane of the hash, the tail words
        arrays[7].reshape(-1)[word] = np.nextafter(arrays[7].reshape(-1)[word], np.inf)
        lib.amof_frames_checksum(ctypes.c_void_p(ptrs.ctypes.data), F, n, ctypes.c_void_p(again.ctypes.data), 2)
        assert again[7] != sums[7] and np.array_equal(np.delete(again, 7), np.delete(sums, 7)), word
        sums[7] = again[7]
