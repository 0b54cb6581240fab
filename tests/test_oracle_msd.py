"""Known-answer tests of the MSD restatement (oracle/numpy_oracle.py)."""

import numpy as np
import pytest

from oracle import numpy_oracle as no


def _masses(n):
    return np.linspace(1.0, 3.0, n)


def test_ballistic_motion():
    # r = r0 + v t  ->  MSD(m) = |v|^2 m^2 (F-m-1)/(F-m) averaged over atoms
    rng = np.random.default_rng(1)
    F, n = 30, 7
    v = rng.normal(size=(n, 3)) * 0.01
    v -= (v * _masses(n)[:, None]).sum(0) / _masses(n).sum()        # no centre-of-mass drift
    pos = rng.uniform(2, 3, (n, 3))[None] + v[None] * np.arange(F)[:, None, None]
    window = np.arange(0, 15, 2)
    _, out = no.window_msd(pos, np.diag([50.0] * 3), np.ones(n, int), _masses(n), window)
    want = (v ** 2).sum(axis=1).mean() * window ** 2 * (F - window - 1) / (F - window)
    np.testing.assert_allclose(out[0], want, rtol=1e-9, atol=1e-15)


def test_uniform_drift_is_removed():
    rng = np.random.default_rng(2)
    F, n = 25, 9
    steps = rng.normal(scale=0.05, size=(F, n, 3))
    pos = np.cumsum(steps, axis=0) + 5.0
    drift = np.arange(F)[:, None, None] * np.array([0.07, -0.02, 0.04])
    window = np.arange(0, 12, 3)
    cell = np.diag([40.0] * 3)
    _, a = no.window_msd(pos, cell, np.ones(n, int), _masses(n), window)
    _, b = no.window_msd(pos + drift, cell, np.ones(n, int), _masses(n), window)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("tri", [False, True])
def test_boundary_crossing_equals_unwrapped_truth(tri):
    rng = np.random.default_rng(3)
    F, n = 40, 12
    cell = np.array([[6.0, 0, 0], [1.0, 7.0, 0], [-0.5, 1.5, 8.0]]) if tri else np.diag([6.0, 7.0, 8.0])
    masses = _masses(n)
    steps = rng.normal(scale=0.25, size=(F, n, 3))
    steps -= (steps * masses[None, :, None]).sum(1, keepdims=True) / masses.sum()   # COM fixed
    true = np.cumsum(steps, axis=0) + rng.uniform(0, 1, (n, 3)) @ cell
    s = np.linalg.solve(cell.T, true.reshape(-1, 3).T).T
    wrapped = ((s - np.floor(s)) @ cell).reshape(F, n, 3)
    window = np.arange(0, 20, 4)
    numbers = np.array([1] * 6 + [8] * 6)
    el_t, t = no.window_msd(true, np.diag([1e3] * 3), numbers, masses, window)
    el_w, w = no.window_msd(wrapped, cell, numbers, masses, window, unwrap=True)
    for a, b in zip(t, w):
        np.testing.assert_allclose(a, b, rtol=1e-8, atol=1e-12)


def test_fast_equals_loops_with_changing_cell():
    rng = np.random.default_rng(4)
    F, n = 18, 10
    cells = np.array([np.diag([5.0, 6.0, 7.0]) * (1 + 0.01 * rng.normal()) for _ in range(F)])
    pos = np.cumsum(rng.normal(scale=0.4, size=(F, n, 3)), axis=0) + 2.5
    numbers = np.array([30, 30, 7, 7, 7, 7, 1, 1, 1, 6])
    window = np.arange(0, 9)
    for unwrap in (False, True):
        e1, a = no.window_msd(pos, cells, numbers, _masses(n), window, unwrap=unwrap)
        e2, b = no.window_msd_fast(pos, cells, numbers, _masses(n), window, unwrap=unwrap)
        assert [int(x) for x in e1] == [int(x) for x in e2]
        for x, y in zip(a, b):
            np.testing.assert_allclose(x, y, rtol=1e-10, atol=1e-13)


def test_wrap_positions_semantics():
    cell = np.diag([2.0, 3.0, 4.0])
    d = np.array([[1.2, -1.6, 0.1], [-1.0, 1.5, -2.0], [0.99999999, 0.0, 1.999999]])
    w = no.wrap_positions(d, cell, center=(0., 0., 0.))
    # into [-1/2 - eps, 1/2 - eps) in fractional coordinates
    s = w / np.diag(cell)
    assert (s >= -0.5 - 2e-7).all() and (s < 0.5).all()
    np.testing.assert_allclose((w - d) / np.diag(cell), np.round((w - d) / np.diag(cell)), atol=1e-12)
    # +1/2 goes to -1/2 (eps shift), pbc=False axis is left alone
    w2 = no.wrap_positions(np.array([[1.0, 1.5, 2.0]]), cell, pbc=(True, True, False), center=(0., 0., 0.))
    np.testing.assert_allclose(w2, [[-1.0, -1.5, 2.0]], atol=1e-12)
