"""ctypes bindings of the C oracle (``oracle/libamof_oracle.so``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Parity status: RDF / CN /
BAD "parity unpinned" (the reference holds no vectors for them); see the header
of oracle/amof_oracle.c.
"""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libamof_oracle.so")
    src = os.path.join(_HERE, "amof_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libamof_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        if not _LIB.amof_oracle_has_fma():
            raise RuntimeError("oracle was built with -mfma but this CPU has no FMA")
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def _prep(pos, cell, pbc, species):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    if pos.ndim == 2:
        pos = pos[None]
    F, N = pos.shape[0], pos.shape[1]
    cell = np.ascontiguousarray(cell, dtype=np.float64).reshape(-1, 3, 3)
    if cell.shape[0] not in (1, F):
        raise ValueError("cell must be [3][3] or [F][3][3]")
    pbc = np.ascontiguousarray(np.broadcast_to(np.asarray(pbc, dtype=bool), (3,)), dtype=np.uint8)
    species = np.ascontiguousarray(species, dtype=np.int32)
    return pos, cell, pbc, species, F, N


def _check(rc, what):
    if rc == -3:
        raise ZeroDivisionError("Undefined angle")
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (what, rc))


def geom(cell, pbc=(True, True, True)):
    cell = np.ascontiguousarray(cell, dtype=np.float64).reshape(3, 3)
    pbc = np.ascontiguousarray(np.broadcast_to(np.asarray(pbc, dtype=bool), (3,)), dtype=np.uint8)
    out = np.zeros(22)
    _check(lib().amof_oracle_geom(_p(cell, ctypes.c_double), _p(pbc, ctypes.c_ubyte), _p(out, ctypes.c_double)), "geom")
    return {"cell": out[:9].reshape(3, 3), "inv": out[9:18].reshape(3, 3), "h": out[18:21], "vol": out[21]}


def images(cell, R, pbc=(True, True, True)):
    cell = np.ascontiguousarray(cell, dtype=np.float64).reshape(3, 3)
    pbc = np.ascontiguousarray(np.broadcast_to(np.asarray(pbc, dtype=bool), (3,)), dtype=np.uint8)
    E = np.zeros((4096, 3))
    n = ctypes.c_int(0)
    _check(lib().amof_oracle_images(_p(cell, ctypes.c_double), _p(pbc, ctypes.c_ubyte), ctypes.c_double(R),
                                    _p(E, ctypes.c_double), 4096, ctypes.byref(n)), "images")
    return E[:n.value].copy()


def rdf_hist(pos, cell, species, S, rmax, nbins, pbc=(True, True, True), cell_list=False):
    """``(hist u64 [S][S][nbins], volume_sum)`` -- ordered-pair counts."""
    pos, cell, pbc, species, F, N = _prep(pos, cell, pbc, species)
    hist = np.zeros((S, S, nbins), dtype=np.uint64)
    vol = ctypes.c_double(0.0)
    rc = lib().amof_oracle_rdf(_p(pos, ctypes.c_double), _p(cell, ctypes.c_double), ctypes.c_int64(cell.shape[0]),
                               _p(pbc, ctypes.c_ubyte), ctypes.c_int64(F), ctypes.c_int64(N),
                               _p(species, ctypes.c_int), int(S), ctypes.c_double(rmax), int(nbins),
                               _p(hist, ctypes.c_uint64), ctypes.byref(vol), 1 if cell_list else 0)
    _check(rc, "rdf")
    return hist, vol.value


def cn_counts(pos, cell, species, S, rcm, sets, pbc=(True, True, True), per_atom=False):
    """``sums i64 [F][n_sets]`` (and per-atom counts ``[F][n_sets][N]``)."""
    pos, cell, pbc, species, F, N = _prep(pos, cell, pbc, species)
    rcm = np.ascontiguousarray(rcm, dtype=np.float64).reshape(S, S)
    sets = np.ascontiguousarray(sets, dtype=np.int32).reshape(-1, 2)
    sums = np.zeros((F, len(sets)), dtype=np.int64)
    pa = np.zeros((F, len(sets), N), dtype=np.int32) if per_atom else None
    rc = lib().amof_oracle_cn(_p(pos, ctypes.c_double), _p(cell, ctypes.c_double), ctypes.c_int64(cell.shape[0]),
                              _p(pbc, ctypes.c_ubyte), ctypes.c_int64(F), ctypes.c_int64(N),
                              _p(species, ctypes.c_int), int(S), _p(rcm, ctypes.c_double),
                              _p(sets, ctypes.c_int), len(sets), _p(sums, ctypes.c_int64),
                              _p(pa, ctypes.c_int32) if per_atom else None)
    _check(rc, "cn")
    return (sums, pa) if per_atom else sums


def acos(x):
    """the fixed-algorithm arccos the angle code uses (amof_oracle_acos: fdlibm's rational approximation), elementwise"""
    f = lib().amof_oracle_acos
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_double]
    return np.array([f(float(v)) for v in np.asarray(x, dtype=np.float64).ravel()]).reshape(np.shape(x))


def bad_hist(pos, cell, species, S, rcm, triples, edges, pbc=(True, True, True)):
    """``(hist u64 [T][nb], n_angles u64 [T])``."""
    pos, cell, pbc, species, F, N = _prep(pos, cell, pbc, species)
    rcm = np.ascontiguousarray(rcm, dtype=np.float64).reshape(S, S)
    triples = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 2)
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    nb = len(edges) - 1
    hist = np.zeros((len(triples), nb), dtype=np.uint64)
    nang = np.zeros(len(triples), dtype=np.uint64)
    rc = lib().amof_oracle_bad(_p(pos, ctypes.c_double), _p(cell, ctypes.c_double), ctypes.c_int64(cell.shape[0]),
                               _p(pbc, ctypes.c_ubyte), ctypes.c_int64(F), ctypes.c_int64(N),
                               _p(species, ctypes.c_int), int(S), _p(rcm, ctypes.c_double),
                               _p(triples, ctypes.c_int), len(triples), _p(edges, ctypes.c_double), nb,
                               _p(hist, ctypes.c_uint64), _p(nang, ctypes.c_uint64))
    _check(rc, "bad")
    return hist, nang


def bad_hist_by_cn(pos, cell, species, S, rcm, triples, edges, cn_max, pbc=(True, True, True)):
    """``(hist u64 [T][cn_max+1][nb], n_angles u64 [T][cn_max+1])``."""
    pos, cell, pbc, species, F, N = _prep(pos, cell, pbc, species)
    rcm = np.ascontiguousarray(rcm, dtype=np.float64).reshape(S, S)
    triples = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 2)
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    nb = len(edges) - 1
    hist = np.zeros((len(triples), cn_max + 1, nb), dtype=np.uint64)
    nang = np.zeros((len(triples), cn_max + 1), dtype=np.uint64)
    rc = lib().amof_oracle_bad_by_cn(_p(pos, ctypes.c_double), _p(cell, ctypes.c_double), ctypes.c_int64(cell.shape[0]),
                                     _p(pbc, ctypes.c_ubyte), ctypes.c_int64(F), ctypes.c_int64(N),
                                     _p(species, ctypes.c_int), int(S), _p(rcm, ctypes.c_double),
                                     _p(triples, ctypes.c_int), len(triples), _p(edges, ctypes.c_double), nb,
                                     int(cn_max), _p(hist, ctypes.c_uint64), _p(nang, ctypes.c_uint64))
    _check(rc, "bad_by_cn")
    return hist, nang


def angles(pos, cell, species, S, rcm, A, B, pbc=(True, True, True)):
    """All B-A-B angles (degrees) of one frame."""
    pos, cell, pbc, species, F, N = _prep(pos, cell, pbc, species)
    rcm = np.ascontiguousarray(rcm, dtype=np.float64).reshape(S, S)
    cap = 1 << 16
    while True:
        out = np.zeros(cap)
        n = ctypes.c_int64(0)
        rc = lib().amof_oracle_angles(_p(pos, ctypes.c_double), _p(cell, ctypes.c_double), _p(pbc, ctypes.c_ubyte),
                                      ctypes.c_int64(N), _p(species, ctypes.c_int), int(S),
                                      _p(rcm, ctypes.c_double), int(A), int(B), _p(out, ctypes.c_double),
                                      ctypes.c_int64(cap), ctypes.byref(n))
        _check(rc, "angles")
        if n.value <= cap:
            return out[:n.value].copy()
        cap = int(n.value)
