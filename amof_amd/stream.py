"""Streamed ingest: file -> pinned frame batches -> HBM -> analysis, the parse of batch k+1 hidden behind batch k.

The reference reads a whole trajectory into a Python list and only then analyses it, serially
(``amof/trajectory.py:37-60,193-228`` -> ``amof/rdf.py:88-93``).  On an MI355X the analysis of a 9792-atom frame takes
15 us while parsing its 0.5 MB of text takes 80 us on 16 host threads, so for a trajectory that lives in
a text file the parse is the run time -- unless it overlaps.  ``XyzStream`` cuts an (extended-)XYZ file into batches of
frames; a background thread parses the next batch with the native multithreaded reader (``amof_xyz_read``, which
releases the GIL) straight into page-locked host memory while the caller's analysis of the previous batch runs; the
library stages a pinned batch to the GPU asynchronously and in sub-batches that overlap its kernels.

Frames are independent for RDF, CN and BAD, whose classes accept a stream wherever they accept a trajectory
(``Rdf.from_trajectory(XyzStream(path, cell=cell))``) and accumulate their integer counts batch by batch -- the
result is the one the whole trajectory gives.  The MSD couples frames half a trajectory apart: ``WindowMsd`` reads
a stream completely first (``read_all``).
"""

import ctypes
import threading

import numpy as np

from . import data as _data
from .frames import PackedTrajectory
from .trajectory import _ingest_error, default_parser_threads, string2index


class XyzStream(object):
    """A big XYZ / extended-XYZ trajectory, read in batches of frames.

    Args:
        path: the file
        cell: ``[3][3]`` (constant cell) or ``[F][3][3]`` (one per frame of the file, e.g. ``read_cp2k_cell``); None:
            the extended-XYZ ``Lattice`` of every frame (read ahead of the frames; a plain XYZ file without cells
            raises when a cell is first needed)
        batch_frames: frames per batch (default: about 256 MiB of positions)
        index: slice or 'first:last:step' selecting frames, in ``ase.io.read``'s slice syntax.  NOTE the default:
            None means ALL frames here (a stream exists to walk a trajectory), whereas ``ase.io.read`` and
            ``amof_amd.trajectory.read_xyz`` return the LAST frame for index=None.  A selection without frames (an empty
            file, '5:5') raises ValueError.
        n_threads: parser threads (0 = the CPUs this process may use)
        pinned: parse into page-locked memory (needs torch; falls back to ordinary memory without a GPU)
    """

    is_stream = True
    on_device = False

    def __init__(self, path, cell=None, batch_frames=None, index=None, n_threads=0, pbc=(True, True, True), pinned=True):
        from . import _hip
        self._lib = _hip.load_library()
        self.path = str(path)
        nf, na = ctypes.c_int64(0), ctypes.c_int64(0)
        # the file stays open: its mapping and frame index serve every batch (amof_xyz_open)
        self._file = ctypes.c_void_p(None)
        rc = self._lib.amof_xyz_open(self.path.encode(), ctypes.byref(self._file), ctypes.byref(nf), ctypes.byref(na))
        if rc:
            _ingest_error(self._lib, rc)
        if isinstance(index, str):
            index = string2index(index)
        if index is None:
            index = slice(None)
        if not isinstance(index, slice):
            raise ValueError("index must select a range of frames")
        self._first, stop, self._step = index.indices(nf.value)
        if self._step < 1:
            raise ValueError("frames must be read forwards")
        self.n_frames = len(range(self._first, stop, self._step))
        self.n_atoms = na.value
        if self.n_frames == 0:
            self.close()
            raise ValueError("empty selection: %s holds %d frames, index %r selects none" % (self.path, nf.value, index))
        if int(n_threads) <= 0:
            n_threads = default_parser_threads()
        self.n_threads = int(max(1, min(int(n_threads), 64)))
        self.pinned = bool(pinned)
        if batch_frames is None:
            batch_frames = max(1, (256 << 20) // max(1, 24 * self.n_atoms))
        self.batch_frames = int(max(1, min(batch_frames, max(1, self.n_frames))))
        # species of the first frame read (the reference assumes them constant: amof/rdf.py:71, amof/msd.py:215)
        _, symbols, _ = self._read(0, min(1, self.n_frames), None)
        names = [bytes(row).split(b"\0")[0].decode() for row in symbols]
        self.numbers = np.array([_data.atomic_numbers[s.capitalize()] for s in names], dtype=np.int64)
        self.masses = np.array([_data.atomic_masses[z] for z in self.numbers], dtype=np.float64)
        self.pbc = np.array(pbc, dtype=bool)
        if cell is not None:
            cell = np.ascontiguousarray(cell, dtype=np.float64)
            if cell.shape == (3, 3):
                cell = cell.reshape(1, 3, 3)
            elif cell.ndim == 3 and cell.shape[1:] == (3, 3) and len(cell) > 1:
                cell = np.ascontiguousarray(cell[index])        # one per frame OF THE FILE -> the frames selected
                if len(cell) < self.n_frames:
                    raise ValueError("%d cells for %d frames" % (len(cell), self.n_frames))
                cell = cell[:self.n_frames]
            if cell.ndim != 3 or cell.shape[1:] != (3, 3):
                raise ValueError("cell must be [3][3] or [F][3][3]")
        if cell is None and self.n_frames:
            # extended XYZ: the Lattice of every selected frame, read ahead of the frames (comment lines only), so that
            # the analyses know all cells up front exactly as they do for a trajectory held in memory
            lat = np.zeros((self.n_frames, 9), dtype=np.float64)
            has = ctypes.c_int32(0)
            rc = self._lib.amof_xyz_read_frames(self._file, self._first, self.n_frames, self._step, self.n_atoms, None, None,
                                                ctypes.c_void_p(lat.ctypes.data), ctypes.byref(has), self.n_threads)
            if rc:
                _ingest_error(self._lib, rc)
            if has.value:
                cell = lat.reshape(-1, 3, 3)
                if len(cell) > 1 and (cell == cell[0]).all():
                    cell = cell[:1].copy()
        self.cell = cell
        self._const = PackedTrajectory(np.zeros((0, self.n_atoms, 3)), np.eye(3), self.numbers, self.masses, self.pbc)

    def __len__(self):
        return self.n_frames

    def close(self):
        if getattr(self, "_file", None) is not None and self._file.value:
            self._lib.amof_xyz_close(self._file)
            self._file = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the per-trajectory constants the analysis classes ask a PackedTrajectory for ---------------------------------
    def unique_numbers(self):
        return self._const.unique_numbers()

    def species_counts(self):
        return self._const.species_counts()

    def formula_count(self):
        return self._const.formula_count()

    def _cells_known(self):
        if self.cell is None:
            raise ValueError("%s carries no Lattice and no cell was given: build the stream with cell=..." % self.path)
        return self.cell

    def cell_lengths(self):
        return np.sqrt((self._cells_known() ** 2).sum(axis=2))

    def volume_sum(self):
        c = self._cells_known()
        return PackedTrajectory(np.zeros((self.n_frames, 0, 3)), c, np.zeros(0, dtype=np.int64),
                                masses=np.zeros(0)).volume_sum()

    # -- reading -------------------------------------------------------------------------------------------------------
    def _buffer(self, frames):
        if self.pinned:
            try:
                import torch
                if torch.cuda.is_available():
                    t = torch.empty((frames, self.n_atoms, 3), dtype=torch.float64, pin_memory=True)
                    return t.numpy(), t          # (the tensor keeps the pinned allocation alive)
            except Exception:
                pass
        a = np.empty((frames, self.n_atoms, 3), dtype=np.float64)
        return a, a

    def _read(self, k0, count, pos):
        """frames k0 .. k0 + count of the selection into pos (or a fresh array); returns (pos, symbols, lattice|None)"""
        if pos is None:
            pos = np.empty((count, self.n_atoms, 3), dtype=np.float64)
        symbols = np.zeros((self.n_atoms, 4), dtype=np.uint8)
        lattice = np.zeros((max(count, 1), 9), dtype=np.float64)
        has = ctypes.c_int32(0)
        if count:
            rc = self._lib.amof_xyz_read_frames(self._file, self._first + k0 * self._step, count, self._step, self.n_atoms,
                                                ctypes.c_void_p(pos.ctypes.data), ctypes.c_void_p(symbols.ctypes.data),
                                                ctypes.c_void_p(lattice.ctypes.data), ctypes.byref(has), self.n_threads)
            if rc:
                _ingest_error(self._lib, rc)
        return pos, symbols, (lattice[:count].reshape(count, 3, 3) if has.value else None)

    def _batch(self, k0, count, buf):
        pos, _, lattice = self._read(k0, count, buf[:count])
        if self.cell is not None:
            cell = self.cell if len(self.cell) == 1 else self.cell[k0:k0 + count]
        elif lattice is not None:
            cell = lattice
        else:
            raise ValueError("%s carries no Lattice and no cell was given" % self.path)
        return PackedTrajectory(pos, cell, self.numbers, self.masses, self.pbc)

    def batches(self):
        """PackedTrajectory batches in frame order; batch k+1 is parsed in the background while the caller works on
        batch k (two buffers: a batch is valid until the next one is requested)."""
        starts = list(range(0, self.n_frames, self.batch_frames))
        if not starts:
            return
        bufs = [self._buffer(min(self.batch_frames, self.n_frames)) for _ in range(min(2, len(starts)))]
        box = {}

        def work(i):
            try:
                k0 = starts[i]
                box[i] = self._batch(k0, min(self.batch_frames, self.n_frames - k0), bufs[i % len(bufs)][0])
            except BaseException as exc:       # handed to the consumer
                box[i] = exc

        th = threading.Thread(target=work, args=(0,))
        th.start()
        for i in range(len(starts)):
            th.join()
            got = box.pop(i)
            if i + 1 < len(starts):
                th = threading.Thread(target=work, args=(i + 1,))
                th.start()
            if isinstance(got, BaseException):
                if i + 1 < len(starts):
                    th.join()
                raise got
            yield got

    def read_all(self):
        """the whole selection as one PackedTrajectory (analyses that couple distant frames: WindowMsd)"""
        pos, _, lattice = self._read(0, self.n_frames, None)
        cell = self.cell if self.cell is not None else lattice
        if cell is None:
            raise ValueError("%s carries no Lattice and no cell was given" % self.path)
        return PackedTrajectory(pos, cell, self.numbers, self.masses, self.pbc)
