"""A 60-line stand-in for the part of xarray that BadByCn touches (amof/bad.py:294-309: DataArray(list, coords) ->
Dataset(dict).to_array("atom_triple") -> Dataset({'bad': ...}), to_netcdf / open_dataset).  xarray is absent from the build
and test machines; with this module registered as ``xarray`` the product's Dataset leg (amof_amd/bad.py) executes and its
dims / coords / values can be checked.  TEST INFRASTRUCTURE ONLY: semantics restated from xarray's documented behaviour --
``Dataset.to_array`` stacks the variables along a new leading dimension after an OUTER join of their coordinates (sorted
union, NaN where a variable has no entry)."""
import pickle

import numpy as np


class DataArray(object):
    def __init__(self, data, coords=None, dims=None, name=None):
        self.values = np.asarray(data, dtype=np.float64)
        coords = dict(coords or {})
        self.dims = tuple(dims) if dims is not None else tuple(coords.keys())
        assert len(self.dims) == self.values.ndim, (self.dims, self.values.shape)
        self.coords = {k: np.asarray(coords[k]) for k in self.dims}
        for k, n in zip(self.dims, self.values.shape):
            assert len(self.coords[k]) == n, k
        self.name = name

    def sel(self, **kw):
        idx = [slice(None)] * self.values.ndim
        dims = list(self.dims)
        for k, v in kw.items():
            ax = self.dims.index(k)
            idx[ax] = int(np.nonzero(self.coords[k] == v)[0][0])
            dims.remove(k)
        return DataArray(self.values[tuple(idx)], {d: self.coords[d] for d in dims}, dims)


class Dataset(object):
    def __init__(self, data_vars=None):
        self.data_vars = dict(data_vars or {})

    def __getitem__(self, key):
        return self.data_vars[key]

    def to_array(self, dim="variable"):
        names = list(self.data_vars)
        first = self.data_vars[names[0]]
        dims = first.dims
        union = {d: np.array(sorted(set(np.concatenate([np.asarray(self.data_vars[n].coords[d]) for n in names]).tolist()))) for d in dims}
        out = np.full((len(names),) + tuple(len(union[d]) for d in dims), np.nan)
        for q, n in enumerate(names):
            v = self.data_vars[n]
            assert v.dims == dims
            where = np.ix_(*[np.searchsorted(union[d], v.coords[d]) for d in dims])
            out[q][where] = v.values
        coords = {dim: np.array(names)}
        coords.update(union)
        return DataArray(out, coords, (dim,) + dims)

    def to_netcdf(self, filename):
        with open(filename, "wb") as fh:
            pickle.dump(self, fh)


def open_dataset(filename):
    with open(filename, "rb") as fh:
        return pickle.load(fh)
