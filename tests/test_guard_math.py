"""Host-side arithmetic behind the RDF fast paths' error bounds (amof_amd/csrc/guard_math.h, no HIP dependency),
compiled with g++ and compared with numpy: the lower-triangular factor the general-cell kernels use as their scale
matrix (|f C| = |f L| for every f, whatever the cell's orientation), kappa of the error model, and the bound of the
variant with f32 slab coordinates."""

import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("gm") / "guard_math_driver")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off",
                        os.path.join(ROOT, "tests", "native", "guard_math_driver.cpp"), "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _ask(driver, lines):
    r = subprocess.run([driver], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return [np.array(l.split(), dtype=float) for l in r.stdout.strip().splitlines()]


def test_lower_factor_preserves_every_distance(driver):
    rng = np.random.default_rng(11)
    cells = [np.diag([3.0, 4.0, 5.0]),
             np.array([[10.0, 0, 0], [5.0, 8.66, 0], [0, 0, 12.0]]),                  # hexagonal
             np.array([[16.0, 0, 0], [8.0, 16.0, 0], [8.0, 8.0, 16.0]])]              # strongly sheared
    for _ in range(20):                                                                # arbitrary orientations
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        cells.append((np.diag(rng.uniform(5, 40, 3)) + np.tril(rng.uniform(-6, 6, (3, 3)), -1)) @ q)
    for perm in ([0, 1, 2], [1, 2, 0], [2, 0, 1], [0, 2, 1]):                          # stored axis orders
        rows = [c[perm] * 2.0 ** -32 / 0.01 for c in cells]
        out = _ask(driver, ["L " + " ".join("%.17g" % v for v in r.ravel()) for r in rows])
        for r, o in zip(rows, out):
            L, kappa = o[:9].reshape(3, 3), o[9]
            assert np.all(L[np.triu_indices(3, 1)] == 0.0) and np.all(np.diag(L) > 0)
            np.testing.assert_allclose(L @ L.T, r @ r.T, rtol=1e-13, atol=0)
            np.testing.assert_allclose(L, np.linalg.cholesky(r @ r.T), rtol=1e-12, atol=1e-300)
            f = rng.integers(-2 ** 31, 2 ** 31, (200, 3)).astype(float)              # fixed-point differences
            np.testing.assert_allclose(np.linalg.norm(f @ L, axis=1), np.linalg.norm(f @ r, axis=1), rtol=1e-13)
            P = np.abs(np.linalg.inv(L)) @ np.abs(L)
            np.testing.assert_allclose(kappa, np.sqrt(np.abs(P).sum(axis=0).max() * np.abs(P).sum(axis=1).max()), rtol=1e-12)
            assert kappa >= 1.0 - 1e-12


def test_zf_guard_formula(driver):
    u = 2.0 ** -24
    cases = [(2310, 7375.0, 0.3133), (2310, 4620.0, 0.5), (50, 400.0, 0.2), (12000, 3.0e5, 0.5), (999, 3680.0, 0.27)]
    out = _ask(driver, ["Z %d %.17g %.17g" % c for c in cases])
    for (nb, hb, gf), o in zip(cases, out):
        qmax = nb + 1.0
        A = u * hb * (gf + 1 / 16 + 1 / 128)                       # the two converted coordinates
        # the bound is the maximum over zeta in [0, 1] of  u q (5.06 - 2 zeta) + A sqrt(zeta)  at q = qmax
        z = np.linspace(0.0, 1.0, 200001)
        brute = 1.1 * np.max(u * qmax * (5.06 - 2.0 * z) + A * np.sqrt(z))
        assert o[0] >= brute * (1 - 1e-9) and o[0] <= brute * (1 + 1e-6)
        assert o[0] >= 1.1 * u * qmax * 5.06                       # never below the plain chain's bound
