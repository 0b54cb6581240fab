"""BadByCn's xarray leg (amof_amd/bad.py, mirror of amof/bad.py:294-309) executed with a stand-in xarray module
(tests/fake_xarray.py) on oracle-backed contexts (tests/oracle_context.py): dims, coords and values of the Dataset against
the oracle's per-cn histograms, 'total' and 'partial' normalisation, the netCDF round trip through write_to_file / from_file."""
import sys

import numpy as np
import pytest

from tests import helpers as H
from tests import fake_xarray, oracle_context


@pytest.fixture()
def ctxs(monkeypatch):
    monkeypatch.setitem(sys.modules, "xarray", fake_xarray)
    ls = oracle_context.install(monkeypatch)
    yield ls
    for c in ls.values():
        c.close_lane()


@pytest.mark.parametrize("normalisation", ["total", "partial"])
def test_dataset_dims_coords_and_values(ctxs, tmp_path, normalisation):
    from amof_amd.bad import Bad, BadByCn
    from oracle import clib
    # a rattled ZIF-4 cell with a generous cutoff: centres with 2, 3, 4 ... neighbours
    packed = H.random_walk(H.zif4_frame(), 4, 0.15, 9)
    cut = {'Zn-N': 2.6, 'C-N': 1.55}
    by = BadByCn.from_trajectory(packed, cut, dtheta=2.0, normalization=normalisation, distributed=False)
    ds = by.data
    assert isinstance(ds, fake_xarray.Dataset) and list(ds.data_vars) == ["bad"]
    xa = ds["bad"]
    assert xa.dims == ("atom_triple", "cn", "theta")
    assert list(xa.coords["atom_triple"]) == list(by.bad.keys()) and len(by.bad) >= 2
    bins = int(180 // 2.0)
    np.testing.assert_array_equal(xa.coords["theta"], np.arange(bins + 1) * 2.0 + 1.0)
    all_cn = sorted(set(c for v in by.bad.values() for c in v))
    assert list(xa.coords["cn"]) == all_cn and len(all_cn) >= 2
    # values: the oracle's integer counts, normalised as the reference does (numpy.histogram density, weighted by the share of
    # the angles with normalisation='partial', amof/bad.py:287-293)
    kinds, sp = H.species_of(packed.numbers)
    from amof_amd import atom as amatom
    rcm = amatom.cutoff_matrix(amatom.format_cutoff(cut), kinds)
    edges = np.arange(bins + 2) * 2.0
    triples = [(kinds.index(by_sym(n.split('-')[1])), kinds.index(by_sym(n.split('-')[0]))) for n in by.columns]
    hist, nang = clib.bad_hist_by_cn(packed.pos, packed.cell, sp, len(kinds), rcm, triples, edges, by.hist.shape[1] - 1)
    assert np.array_equal(hist, by.hist) and np.array_equal(nang, by.n_angles)
    for q, name in enumerate(xa.coords["atom_triple"]):
        k = by.columns.index(name)
        for c in all_cn:
            got = xa.sel(atom_triple=name, cn=c).values
            if nang[k, c] == 0:
                assert np.isnan(got).all()                 # (the outer join of Dataset.to_array: this triple never has c neighbours)
                continue
            n = hist[k, c].astype(np.int64)
            want = n / np.diff(edges) / n.sum()
            if normalisation == "partial":
                want = want * (float(nang[k, c]) / float(nang[k].sum()))
            np.testing.assert_allclose(got, want, rtol=1e-15)
    if normalisation == "partial":
        # the partial distributions of a triple add up to the plain Bad of that triple
        bad = Bad.from_trajectory(packed, cut, dtheta=2.0, distributed=False)
        for name in xa.coords["atom_triple"]:
            total = np.nansum(xa.sel(atom_triple=name).values, axis=0)
            np.testing.assert_allclose(total, bad.data[name].values, rtol=1e-12, atol=1e-15)
    path = str(tmp_path / "by_cn")
    by.write_to_file(path)
    again = BadByCn.from_file(path)
    np.testing.assert_array_equal(again.data["bad"].values, xa.values)


def by_sym(sym):
    from amof_amd import data
    return data.atomic_numbers[sym]
