"""Trajectory helpers on the hot path (mirror of reference amof/trajectory.py).

``construct_step`` (amof/trajectory.py:244-283) is host logic;
``get_delta_pos`` (amof/trajectory.py:285-303) runs inside the MSD kernels.

Ingest: ``read_lammps_traj`` / ``read_cp2k_traj`` keep the
reference's signatures (amof/trajectory.py:193-228) but return a
:class:`amof_amd.frames.PackedTrajectory` -- the arrays the kernels consume --
parsed by the native reader in ``libamofhip.so`` instead of a Python list of
``ase.Atoms`` built by ``ase.io.read``.  ``.to_frames()`` gives ``Frame`` objects
when a list is wanted."""

import ctypes
import logging

import numpy as np

from . import data as _data
from .frames import PackedTrajectory

logger = logging.getLogger(__name__)


def string2index(string):
    """ase.io.formats.string2index: 'a:b:c' -> slice, '3' -> 3"""
    if ':' not in string:
        return int(string)
    i = []
    for s in string.split(':'):
        i.append(None if s == '' else int(s))
    i += (3 - len(i)) * [None]
    return slice(*i)


def _ingest_error(lib, rc):
    msg = lib.amof_ingest_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    raise RuntimeError("libamofhip ingest error %d: %s" % (rc, msg))


def default_parser_threads():
    """threads for the native text reader: the CPUs this process may USE (affinity mask, cgroup quota), not the
    machine's -- a GPU box hands a job a share of its 256 hardware threads, and 64 parser threads on a 16-CPU share
    run at a third of the speed of 16"""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                # container CPU quota (cgroup v2), e.g. "1600000 100000"
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return int(max(1, min(n, 64)))


def read_xyz(path, index=None, n_threads=0):
    """Read an XYZ / extended-XYZ trajectory into packed arrays.

    Args:
        index: like ``ase.io.read``: None or int -> that single frame (None = the
            last one, ASE's default), slice or 'first:last:step' -> those frames
    Returns:
        (pos[F][N][3] f64, numbers[N] int64, lattice[F][3][3] or None)
    """
    from . import _hip
    lib = _hip.load_library()
    bpath = str(path).encode()
    nf, na = ctypes.c_int64(0), ctypes.c_int64(0)
    rc = lib.amof_xyz_scan(bpath, ctypes.byref(nf), ctypes.byref(na))
    if rc:
        _ingest_error(lib, rc)
    F, N = nf.value, na.value
    if isinstance(index, str):
        index = string2index(index)
    if index is None:
        index = -1
    if isinstance(index, slice):
        first, stop, step = index.indices(F)
        count = len(range(first, stop, step))
    else:
        first = index + F if index < 0 else index
        if not 0 <= first < F:
            raise IndexError("frame %d out of range (%d frames)" % (index, F))
        count, step = 1, 1
    pos = np.empty((count, N, 3), dtype=np.float64)
    symbols = np.zeros((N, 4), dtype=np.uint8)
    lattice = np.zeros((count, 9), dtype=np.float64)
    has = ctypes.c_int32(0)
    if count:
        rc = lib.amof_xyz_read(bpath, first, count, step, N, ctypes.c_void_p(pos.ctypes.data),
                               ctypes.c_void_p(symbols.ctypes.data), ctypes.c_void_p(lattice.ctypes.data),
                               ctypes.byref(has), int(n_threads) if int(n_threads) > 0 else default_parser_threads())
        if rc:
            _ingest_error(lib, rc)
    names = [bytes(row).split(b"\0")[0].decode() for row in symbols]
    numbers = np.array([_data.atomic_numbers[s.capitalize()] for s in names], dtype=np.int64)
    return pos, numbers, (lattice.reshape(count, 3, 3) if has.value else None)


def read_cp2k_cell(path_to_cell):
    """CP2K cell log -> cell[rows][3][3] (columns [2:-1], amof/trajectory.py:217-224)"""
    from . import _hip
    lib = _hip.load_library()
    bpath = str(path_to_cell).encode()
    n = ctypes.c_int64(0)
    rc = lib.amof_cp2k_cell_read(bpath, 0, None, ctypes.byref(n))
    if rc:
        _ingest_error(lib, rc)
    cell = np.empty((n.value, 9), dtype=np.float64)
    rc = lib.amof_cp2k_cell_read(bpath, n.value, ctypes.c_void_p(cell.ctypes.data), ctypes.byref(n))
    if rc:
        _ingest_error(lib, rc)
    return cell.reshape(-1, 3, 3)


def _fit(pos, cell):
    """Trajectory.set_cell(fit_size=True): trim to the shorter of the two (amof/trajectory.py:104-113)"""
    if len(pos) != len(cell):
        logger.warning("Mismatch in file sizes; traj: %s vs cell: %s", len(pos), len(cell))
        n = min(len(pos), len(cell))
        pos, cell = pos[:n], cell[:n]
    return pos, cell


class Trajectory(object):
    """Mirror of the reference's ``amof.trajectory.Trajectory`` for the formats the native reader
    covers (reference amof/trajectory.py:27-109).  ``self.traj`` is a :class:`PackedTrajectory`
    (the reference holds a list of ``ase.Atoms``); every analysis class accepts it as is, and
    ``.to_frames()`` materialises the list."""

    def __init__(self):
        self.traj = None

    @classmethod
    def from_traj(cls, filename, index=None, format=None, unzip=False):
        """Read an XYZ / extended-XYZ file (amof/trajectory.py:37-60).

        Args:
            index: 'first_frame:last_frame:step' or slice(first_frame, last_frame, step); None = last frame
            format: None, 'xyz' or 'extxyz' -- other ASE formats are not read natively
            unzip: gunzip into a temporary file first
        """
        if format not in (None, "xyz", "extxyz"):
            raise NotImplementedError("format %r is not read natively; convert to (ext)xyz" % (format,))
        logger.info("Read trajectory %s", filename)
        out = cls()
        pos, numbers, lattice = _read_xyz_maybe_gz(filename, index, unzip)
        if lattice is not None:
            cell, pbc = lattice, (True, True, True)
            if len(cell) > 1 and (cell == cell[0]).all():
                cell = cell[:1]
        else:                                   # plain xyz: no cell yet (ASE: zero cell, pbc False)
            cell, pbc = np.zeros((1, 3, 3)), (False, False, False)
        out.traj = PackedTrajectory(pos, cell, numbers, pbc=pbc)
        return out

    @staticmethod
    def get_index_closest(myList, myNumber):
        """Index of the entry of the sorted ``myList`` closest to ``myNumber``; the smaller index on a
        tie (amof/trajectory.py:76-94; like the reference, the first/last VALUE is returned at the ends)."""
        import bisect
        pos = bisect.bisect_left(myList, myNumber)
        if pos == 0:
            return myList[0]
        if pos == len(myList):
            return myList[-1]
        before = myList[pos - 1]
        after = myList[pos]
        if after - myNumber < myNumber - before:
            return pos
        return pos - 1

    def set_cell(self, cell, set_pbc=True, fit_size=True):
        """One cell per frame (amof/trajectory.py:96-114); sizes are fitted to the shorter of the two."""
        cell = np.asarray(cell, dtype=np.float64).reshape(-1, 3, 3)
        pos = self.traj.pos_host()
        if len(pos) != len(cell):
            if not fit_size:
                raise ValueError("traj has %d frames, cell has %d" % (len(pos), len(cell)))
            pos, cell = _fit(pos, cell)
        pbc = (True, True, True) if set_pbc else tuple(self.traj.pbc)
        self.traj = PackedTrajectory(pos, cell, self.traj.numbers, self.traj.masses, pbc)

    def get_traj(self):
        return self.traj


def _read_xyz_maybe_gz(path, index, unzip):
    """``unzip=True``: gunzip into a temporary file first, as the reference does
    (amof/trajectory.py:50-56), then map that file with the native reader."""
    if not unzip:
        return read_xyz(path, index)
    import gzip
    import shutil
    import tempfile
    logger.info("Unzip trajectory file")
    with tempfile.NamedTemporaryFile(suffix=".xyz") as tmp:
        with gzip.open(path, "rb") as f_in:
            shutil.copyfileobj(f_in, tmp)
        tmp.flush()
        return read_xyz(tmp.name, index)


def read_lammps_traj(path_to_xyz, index=None, cell=None, unzip_xyz=False):
    """
    Args:
        index: 'first_frame:last_frame:step' or slice(first_frame, last_frame, step)
        cell: cell vectors, one per frame read (or a single 3x3); if None the extended-XYZ
            Lattice is used when present
    Returns:
        PackedTrajectory (pbc = True, as Trajectory.set_cell does)
    """
    pos, numbers, lattice = _read_xyz_maybe_gz(path_to_xyz, index, unzip_xyz)
    if cell is None:
        if lattice is None:
            raise ValueError("no cell given and the file carries no Lattice")
        cell = lattice
    cell = np.asarray(cell, dtype=np.float64)
    if cell.ndim == 3:
        pos, cell = _fit(pos, cell)
        if len(cell) > 1 and (cell == cell[0]).all():
            cell = cell[:1]                     # constant cell: one record serves every frame
    return PackedTrajectory(pos, cell, numbers, pbc=(True, True, True))


def read_cp2k_traj(path_to_xyz, path_to_cell, index=None, unzip_xyz=False):
    """
    Args:
        index: slice(first_frame, last_frame, step) (or an int / None like the reference)
    Returns:
        PackedTrajectory with one cell per frame (pbc = True)
    """
    pos, numbers, _ = _read_xyz_maybe_gz(path_to_xyz, index, unzip_xyz)
    cell = read_cp2k_cell(path_to_cell)
    if isinstance(index, str):
        index = string2index(index)
    if isinstance(index, slice):
        cell = cell[index]
    elif index is None:
        cell = cell[-1:] if len(pos) == 1 else cell
    else:
        cell = cell[index:index + 1] if index != -1 else cell[-1:]
    pos, cell = _fit(pos, cell)
    return PackedTrajectory(pos, cell, numbers, pbc=(True, True, True))


def write_xyz(path, packed, comment_lattice=True, fmt="%.10f"):
    """Write a PackedTrajectory as (extended) XYZ -- test / example helper."""
    pos = packed.pos_host()
    syms = [_data.chemical_symbols[int(z)] for z in packed.numbers]
    with open(path, "w") as fh:
        for k in range(packed.n_frames):
            fh.write("%d\n" % packed.n_atoms)
            if comment_lattice:
                c = packed.cell_of(k).reshape(-1)
                fh.write('Lattice="%s" Properties=species:S:1:pos:R:3\n' % " ".join(repr(float(x)) for x in c))
            else:
                fh.write("frame %d\n" % k)
            for s, p in zip(syms, pos[k]):
                fh.write(("%s " + fmt + " " + fmt + " " + fmt + "\n") % (s, p[0], p[1], p[2]))


def _steps_from_slice(step):
    return np.array(list(range(step.start or 0, step.stop, step.step or 1)))


def construct_step(step=None, delta_Step=None, first_frame=None, last_frame=None, number_of_frames=None, **_ignored):
    """The ``Step`` column of per-frame results (behaviour of reference amof/trajectory.py:244-283, pinned by
    tests/golden/reference_construct_step.json; unknown keywords are ignored there too).

    Rules, first match wins:
      1. ``step`` given: a slice is expanded with range(), anything else becomes an array as is;
      2. ``delta_Step`` with both ends: arange(first_frame, last_frame, delta_Step);
         with ``number_of_frames`` and one end: that many steps of delta_Step (counted back from
         ``last_frame`` when only the end is known);
      3. no spacing but both ends and ``number_of_frames``: linspace.
    Anything else yields None, as in the reference; an arithmetic failure is logged and raised as ValueError.
    """
    have_first, have_last = first_frame is not None, last_frame is not None
    try:
        if step is not None:
            return _steps_from_slice(step) if isinstance(step, slice) else np.array(step)
        if delta_Step is not None:
            if have_first and have_last:
                return np.arange(first_frame, last_frame, delta_Step)
            if number_of_frames is None:
                return None
            span = number_of_frames * delta_Step
            if have_last:                              # (only the end is known here)
                first_frame = last_frame - span
            elif not have_first:
                return None
            return np.arange(first_frame, first_frame + span, delta_Step)
        if number_of_frames is not None and have_first and have_last:
            return np.linspace(first_frame, last_frame, number_of_frames)
        return None
    except Exception:
        logger.exception("Cannot construct step from provided args")
        raise ValueError
