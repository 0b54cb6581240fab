"""The oracle against vectors produced by the REFERENCE's own code
(tests/golden/make_reference_goldens.py).  These pin the oracle."""

import json
import os

import numpy as np
import pytest

from oracle import numpy_oracle as no
from amof_amd import trajectory as product_trajectory
from tests.conftest import GOLDEN


def test_compute_msd_of_m_matches_reference():
    g = np.load(os.path.join(GOLDEN, "reference_msd_of_m.npz"))
    for k in range(int(g["n_cases"])):
        base, ms, want = g["delta_%d" % k], g["m_%d" % k], g["msd_%d" % k]
        for m, w in zip(ms, want):
            delta = [b.copy() for b in base]
            got = no.compute_msd_of_m(delta, int(m))
            assert got == pytest.approx(w, rel=1e-14, abs=1e-300), (k, m)


def test_compute_msd_of_m_shared_delta_list():
    # the reference reuses one delta list across successive m and mutates delta[0]
    g = np.load(os.path.join(GOLDEN, "reference_msd_of_m.npz"))
    shared = [b.copy() for b in g["delta_shared"]]
    for m, w in zip(g["m_shared"], g["msd_shared"]):
        assert no.compute_msd_of_m(shared, int(m)) == pytest.approx(w, rel=1e-13)


def test_msd_quirk_origin_zero_skipped():
    # MSD(m) = 1/(F-m) * sum_{k=1}^{F-m-1} |r(k+m)-r(k)|^2 / n  (not the docstring formula)
    g = np.load(os.path.join(GOLDEN, "reference_msd_of_m.npz"))
    base = g["delta_1"]
    F, n = base.shape[0], base.shape[1]
    r = np.cumsum(base, axis=0)
    for m, w in zip(g["m_1"], g["msd_1"]):
        m = int(m)
        s = sum(((r[k + m] - r[k]) ** 2).sum() for k in range(1, F - m)) / n / (F - m)
        assert s == pytest.approx(w, rel=1e-12, abs=1e-300)


@pytest.mark.parametrize("impl", [no.construct_step, product_trajectory.construct_step])
def test_construct_step_matches_reference(impl):
    with open(os.path.join(GOLDEN, "reference_construct_step.json")) as fh:
        cases = json.load(fh)
    for c in cases:
        kw = dict(c["kwargs"])
        if isinstance(kw.get("step"), list) and kw["step"] and kw["step"][0] == "slice":
            _, a, b, s = kw["step"]
            kw["step"] = slice(a, b, s)
        got = impl(**kw)
        if c["result"] is None:
            assert got is None
        else:
            assert np.array_equal(np.asarray(got), np.asarray(c["result"]))
            assert str(np.asarray(got).dtype) == c["dtype"]


@pytest.mark.parametrize("name", ["ortho_raw", "ortho_unwrap", "tri_raw", "tri_unwrap", "zif4_rattle"])
def test_window_msd_pipeline_matches_reference_e2e(name):
    g = np.load(os.path.join(GOLDEN, "reference_e2e_msd_%s.npz" % name))
    F = len(g["pos"])
    window, time = no.msd_window_setup(F, int(g["delta_time"]), "half", int(g["timestep"]))
    from amof_amd import data as eldata
    masses = np.array([eldata.atomic_masses[z] for z in g["numbers"]])
    for fn in (no.window_msd, no.window_msd_fast):
        elements, out = fn(g["pos"], g["cell"], g["numbers"], masses, window, unwrap=bool(g["unwrap"]))
        cols = list(g["columns"])
        assert np.array_equal(g["values"][:, 0], time)
        for e, col in zip(elements, out):
            want = g["values"][:, cols.index(eldata.chemical_symbols[int(e)])]
            np.testing.assert_allclose(col, want, rtol=1e-10, atol=1e-13)


def test_direct_msd_matches_reference_e2e():
    # DirectMsd is pure reference numpy on get_cell()/get_positions(): this golden carries no stub
    g = np.load(os.path.join(GOLDEN, "reference_e2e_directmsd_ortho.npz"))
    from amof_amd import data as eldata
    elements, out = no.direct_msd(g["pos"], g["cell"], g["numbers"])
    cols = [str(c) for c in g["columns"]]
    np.testing.assert_allclose(out[None], g["values"][:, cols.index("X")], rtol=1e-12, atol=1e-14)
    for e in elements:
        np.testing.assert_allclose(out[e], g["values"][:, cols.index(eldata.chemical_symbols[int(e)])],
                                   rtol=1e-12, atol=1e-14)
