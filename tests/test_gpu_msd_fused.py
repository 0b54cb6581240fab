"""The fused window-MSD form (csrc/msd.hip msd_seg_kernel / msd_fused_kernel: no transposed copy, pos read twice) and
its atom-sharded halves amof_msd_shard_begin / _finish, through the C ABI, against the numpy restatement of the
reference's loops (amof/msd.py:185-205, amof/trajectory.py:285-303; 1e-9) and against the transposed forms."""
import os

import numpy as np
import pytest

from amof_amd.frames import Frame, PackedTrajectory
from oracle import numpy_oracle as no
from tests import helpers as H

pytestmark = pytest.mark.gpu


class _env(object):
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _walk(F, n, seed, cell, sigma=0.05, drift=None, cells=None):
    rng = np.random.default_rng(seed)
    pos = np.cumsum(rng.normal(scale=sigma, size=(F, n, 3)), axis=0) + 4.0
    if drift is not None:
        pos += np.arange(F)[:, None, None] * np.asarray(drift)[None, None, :]
    c = np.asarray(cells if cells is not None else cell)
    diag = np.diagonal(c, axis1=-2, axis2=-1)
    d = diag[:, None, :] if diag.ndim == 2 else diag
    s = pos / d
    numbers = ([1, 1, 8, 30, 1, 8, 8, 30, 30, 1, 6, 7] * ((n + 11) // 12))[:n]
    return PackedTrajectory((s - np.floor(s)) * d, c, numbers)


def _check_oracle(packed, window, got, kinds, rtol=1e-9):
    F = packed.n_frames
    elements, ref = no.window_msd_fast(packed.pos, packed.cell, packed.numbers, packed.masses, window)
    for e, r in zip(elements, ref):
        g = got[kinds.index(int(e))] / (packed.numbers == e).sum() / (F - window)
        np.testing.assert_allclose(g, r, rtol=rtol, atol=1e-12)


@pytest.mark.parametrize("F,d,W,n", [(5000, 100, 25, 40), (1000, 100, 5, 9), (999, 77, 13, 30), (1530, 100, 15, 25),
                                     (4999, 100, 32, 12), (650, 65, 10, 100), (3333, 90, 24, 17), (64, 16, 4, 5),
                                     (700, 16, 32, 64), (1680, 16, 8, 3), (1675, 16, 8, 5),     # nq = 44, 105: up to 21 threads per column
                                     (2000, 128, 12, 65), (4100, 256, 8, 24), (330, 64, 3, 40), (2600, 129, 9, 8)])
def test_fused_form_equals_oracle_and_transposed_forms(hip_ctx, F, d, W, n):
    # (steps of 0.05 A: with a dozen atoms the centre of mass of the WRAPPED positions jumps by angstroms whenever a heavy atom
    #  crosses the box -- the reference subtracts exactly that, amof/msd.py:235-237 -- and a larger raw step could then
    #  wrap again under it: such calls are answered by the transposed forms, see the gas test below)
    cell = np.diag([9.0, 10.0, 11.0])
    packed = _walk(F, n, F * 131 + d, cell)
    window = np.array([w * d for w in range(W) if w * d < F], dtype=np.int32)
    got, kinds = hip_ctx.msd_window(packed, window)
    assert hip_ctx.last_path() == "msd_fused", (F, d, W)
    with _env(AMOF_MSD_NOFUSED="1"):
        old, _ = hip_ctx.msd_window(packed, window)
        assert hip_ctx.last_path() != "msd_fused"
    np.testing.assert_allclose(got, old, rtol=1e-11, atol=1e-9)
    _check_oracle(packed, window, got, kinds)
    # device-resident positions: the same bits
    dev, _ = hip_ctx.msd_window(packed.to_device(0), window)
    assert np.array_equal(dev, got)


def test_fused_form_cell_per_frame_drift_and_open_axis(hip_ctx):
    rng = np.random.default_rng(3)
    F, n, d = 1200, 36, 50
    base = np.diag([9.0, 10.0, 11.0])
    cells = np.array([base * (1 + 0.004 * rng.normal()) for _ in range(F)])
    window = (np.arange(12) * d).astype(np.int32)
    # a breathing diagonal cell (NPT): the wrap of frame k uses the cell of frame k - 1 (amof/trajectory.py:302)
    npt = _walk(F, n, 11, base, cells=cells)
    got, kinds = hip_ctx.msd_window(npt, window)
    assert hip_ctx.last_path() == "msd_fused"
    _check_oracle(npt, window, got, kinds)
    # a drifting system (positions NOT folded into the cell: the reference only wraps differences): the centre of mass moves
    # 0.02 A per frame and the answer must not see it (amof/msd.py:235-237)
    rng2 = np.random.default_rng(12)
    walk = np.cumsum(rng2.normal(scale=0.05, size=(F, n, 3)), axis=0) + 4.0
    still = PackedTrajectory(walk, base, npt.numbers)
    drift = PackedTrajectory(walk + np.arange(F)[:, None, None] * np.array([0.02, -0.013, 0.007]), base, npt.numbers)
    a, kinds = hip_ctx.msd_window(still, window)
    b, _ = hip_ctx.msd_window(drift, window)
    assert hip_ctx.last_path() == "msd_fused"
    np.testing.assert_allclose(b, a, rtol=1e-7, atol=1e-9)
    _check_oracle(drift, window, b, kinds)
    still = _walk(F, n, 12, base, sigma=0.05)
    # one axis not periodic: never wrapped
    open_z = PackedTrajectory(still.pos + np.arange(F)[:, None, None] * np.array([0, 0, 0.3]), base, still.numbers, pbc=(True, True, False))
    c, kinds = hip_ctx.msd_window(open_z, window)
    assert hip_ctx.last_path() == "msd_fused"
    elements, ref = no.window_msd_fast(open_z.pos, open_z.cell, open_z.numbers, open_z.masses, window, pbc=(True, True, False))
    for e, r in zip(elements, ref):
        np.testing.assert_allclose(c[kinds.index(int(e))] / (open_z.numbers == e).sum() / (F - window), r, rtol=1e-9, atol=1e-12)


def test_gas_raises_the_flag_and_the_transposed_forms_answer(hip_ctx):
    """atoms that jump anywhere in the box from frame to frame: raw differences within the centre-of-mass step of half the
    cell exist in every column -- wrap(raw - dc) != raw - dc -- so the fused form must hand the call over"""
    rng = np.random.default_rng(8)
    F, n = 400, 30
    cell = np.diag([9.0, 11.0, 13.0])
    gas = PackedTrajectory(rng.uniform(0, 1, (F, n, 3)) @ cell, cell, [1, 8] * (n // 2))
    window = (np.arange(6) * 32).astype(np.int32)
    got, kinds = hip_ctx.msd_window(gas, window)
    assert hip_ctx.last_path() != "msd_fused"
    _check_oracle(gas, window, got, kinds)
    # exactly one such entry: a quiet walk in which one atom is moved by half the box in one frame
    quiet = _walk(F, n, 5, cell, sigma=0.03)
    pos = quiet.pos.copy()
    pos[200:, 7, 0] = (pos[200:, 7, 0] + 4.5) % 9.0
    one = PackedTrajectory(pos, cell, quiet.numbers)
    got, kinds = hip_ctx.msd_window(one, window)
    _check_oracle(one, window, got, kinds)


def test_general_cells_and_other_windows_keep_their_kernels(hip_ctx):
    tri = H.random_walk(H.zif4_frame(), 300, 0.05, 4)                 # the fixture's lattice has off-diagonal terms
    hip_ctx.msd_window(tri, (np.arange(4) * 64).astype(np.int32))
    assert hip_ctx.last_path() == "msd_stream"
    cell = np.diag([9.0, 10.0, 11.0])
    packed = _walk(300, 12, 1, cell)
    for window, want in (((np.arange(30) * 8), "msd_comb"), (np.array([0, 3, 50, 161]), "msd_group"),
                         ((np.arange(40) * 5), "msd_comb")):          # spacing < 16, irregular, more than 32 windows
        hip_ctx.msd_window(packed, window.astype(np.int32))
        assert hip_ctx.last_path() == want
    long = _walk(1700, 6, 2, cell)
    hip_ctx.msd_window(long, (np.arange(8) * 16).astype(np.int32))
    assert hip_ctx.last_path() == "msd_comb"                          # more than 105 segments per column
    sub, _ = hip_ctx.msd_window(packed, (np.arange(4) * 64).astype(np.int32), atom_range=(2, 9))
    assert hip_ctx.last_path() != "msd_fused"                        # (an atom range of one call: the centre of mass needs all atoms)


def test_sharded_halves_add_up_to_the_whole(hip_ctx):
    """amof_msd_shard_begin / _finish as the ranks of an atom-sharded run call them (here: one context after the other,
    the tables summed with torch): shares of any size, an empty one included, add up to the single call; a finish without
    its begin, or after another call, is refused; where the fused form does not apply begin says so."""
    import torch
    from amof_amd import _hip
    F, n, d = 1500, 50, 100
    cell = np.diag([9.0, 10.0, 11.0])
    packed = _walk(F, n, 21, cell).to_device(0)
    window = (np.arange(8) * d).astype(np.int32)
    whole, kinds = hip_ctx.msd_window(packed, window)
    assert hip_ctx.last_path() == "msd_fused"
    lane1 = _hip.get_context(0, lane=1)
    for shares in (((0, n),), ((0, 17), (17, 17), (17, 44), (44, n)), tuple((a, a + 1) for a in range(n))):
        # "ranks": every share on a context (scratch of its own); the tables are summed with torch, as the all-reduce would
        ctxs = [hip_ctx if k % 2 == 0 else lane1 for k in range(len(shares))]
        tables = []
        for c, r in zip(ctxs, shares):
            tb = torch.empty((F, 3), dtype=torch.float64, device="cuda:0")
            c.msd_shard_begin(packed, window, r, tb)
            tables.append(tb)
        assert hip_ctx.last_path() == "msd_fused"
        total = torch.stack(tables).sum(dim=0)
        out = torch.zeros((len(kinds), len(window)), dtype=torch.float64, device="cuda:0")
        for c, r in zip(ctxs, shares):
            # (a context keeps ONE begin: the share's first half again, then its second half with the complete table)
            c.msd_shard_begin(packed, window, r, torch.empty((F, 3), dtype=torch.float64, device="cuda:0"))
            c.msd_shard_finish(packed, window, r, total, out)
        np.testing.assert_allclose(out.cpu().numpy(), whole, rtol=1e-12, atol=1e-12)
    host = _walk(F, n, 21, cell)
    m = host.masses
    np.testing.assert_allclose(total.cpu().numpy(), (host.pos * m[None, :, None]).sum(axis=1), rtol=1e-13)
    # protocol errors
    tb = torch.empty((F, 3), dtype=torch.float64, device="cuda:0")
    out = torch.zeros((len(kinds), len(window)), dtype=torch.float64, device="cuda:0")
    with pytest.raises(ValueError, match="must follow"):
        hip_ctx.msd_shard_finish(packed, window, (0, n), total, out)          # (the begin above was finished already)
    hip_ctx.msd_shard_begin(packed, window, (0, n), tb)
    hip_ctx.msd_window(packed, window, atom_range=(0, 3))                      # another call in between: scratch is gone
    with pytest.raises(ValueError, match="must follow"):
        hip_ctx.msd_shard_finish(packed, window, (0, n), total, out)
    hip_ctx.msd_shard_begin(packed, window, (0, n), tb)
    with pytest.raises(ValueError, match="must follow"):
        hip_ctx.msd_shard_finish(packed, window, (0, n - 1), total, out)      # other arguments
    assert float(out.abs().sum()) == 0.0
    # not applicable: general cell / irregular windows / host positions
    tri = H.random_walk(H.zif4_frame(), 200, 0.05, 4).to_device(0)
    t2 = torch.empty((200, 3), dtype=torch.float64, device="cuda:0")
    with pytest.raises(_hip.Unsupported):
        hip_ctx.msd_shard_begin(tri, (np.arange(3) * 64).astype(np.int32), (0, 10), t2)
    with pytest.raises(_hip.Unsupported):
        hip_ctx.msd_shard_begin(packed, np.array([0, 3, 50], dtype=np.int32), (0, 10), tb)
    with pytest.raises(_hip.Unsupported):
        hip_ctx.msd_shard_begin(host, window, (0, 10), tb)
    # a gas: finish answers through the transposed forms with the completed centre of mass
    rng = np.random.default_rng(8)
    gas = PackedTrajectory(rng.uniform(0, 1, (400, 30, 3)) @ cell, cell, [1, 8] * 15).to_device(0)
    w2 = (np.arange(6) * 32).astype(np.int32)
    ref, k2 = hip_ctx.msd_window(gas, w2)
    tg = torch.empty((400, 3), dtype=torch.float64, device="cuda:0")
    og = torch.zeros((len(k2), len(w2)), dtype=torch.float64, device="cuda:0")
    halves = []
    for r in ((0, 11), (11, 30)):
        t = torch.empty((400, 3), dtype=torch.float64, device="cuda:0")
        hip_ctx.msd_shard_begin(gas, w2, r, t)
        halves.append(t)
    tg = halves[0] + halves[1]
    for r in ((0, 11), (11, 30)):
        t = torch.empty((400, 3), dtype=torch.float64, device="cuda:0")
        hip_ctx.msd_shard_begin(gas, w2, r, t)
        hip_ctx.msd_shard_finish(gas, w2, r, tg, og)
    np.testing.assert_allclose(og.cpu().numpy(), ref, rtol=1e-11, atol=1e-9)
