"""Frame carrier for the pair-distance hot path.

The reference's frame type is ``ase.Atoms`` and a "trajectory" is a Python list
of them (reference amof/trajectory.py:27-35,56-59).  ASE is not available where
this package is built and tested, so two carriers are defined here:

* :class:`Frame` -- a small ``ase.Atoms`` look-alike exposing exactly the
  methods the reference path calls (amof/rdf.py:71,74; amof/msd.py:218-237,263;
  amof/atom.py:22,40-46,82-83; amof/bad.py:85,100; amof/cn.py:66,70).  Real
  ``ase.Atoms`` objects are accepted everywhere a ``Frame`` is (duck typing).
* :class:`PackedTrajectory` -- the packed form handed to the C ABI:
  ``pos[F][N][3]`` f64, ``cell[F][3][3]`` f64 (or one cell for all frames),
  ``numbers[N]``, ``masses[N]``, ``pbc[3]``.  ``pos`` may be a numpy array
  (host) or a torch CUDA tensor (already resident in HBM).
"""

import numpy as np

from . import data as _data


class _Formula(object):
    """Stand-in for ``ase.formula.Formula``: only ``_count`` is used
    (reference amof/msd.py:263)."""

    def __init__(self, symbols):
        count = {}
        for s in symbols:
            count[s] = count.get(s, 0) + 1
        self._count = count

    def count(self):
        return dict(self._count)


class _Symbols(object):
    """Stand-in for ``ase.symbols.Symbols`` (``atoms.symbols.formula._count``)."""

    def __init__(self, numbers):
        self._numbers = numbers

    def __iter__(self):
        return (_data.chemical_symbols[z] for z in self._numbers)

    def __len__(self):
        return len(self._numbers)

    @property
    def formula(self):
        return _Formula(list(self))


class Frame(object):
    """Minimal ``ase.Atoms`` look-alike (one configuration of N atoms)."""

    def __init__(self, numbers=None, positions=None, cell=None, pbc=True,
                 masses=None, symbols=None):
        if numbers is None:
            numbers = [_data.atomic_numbers[s] for s in symbols]
        self.numbers = np.array(numbers, dtype=np.int64)
        self.positions = np.array(positions, dtype=np.float64).reshape(-1, 3)
        if len(self.numbers) != len(self.positions):
            raise ValueError("numbers and positions have different lengths")
        cell = np.array(cell, dtype=np.float64)
        if cell.shape == (3,):
            cell = np.diag(cell)
        if cell.shape != (3, 3):
            raise ValueError("cell must be 3 lengths or a 3x3 matrix")
        self.cell = cell
        if isinstance(pbc, (bool, np.bool_)):
            pbc = (pbc,) * 3
        self.pbc = np.array(pbc, dtype=bool)
        self._masses = None if masses is None else np.array(masses, dtype=np.float64)

    # -- ase.Atoms API subset -------------------------------------------------
    def __len__(self):
        return len(self.numbers)

    def get_global_number_of_atoms(self):
        return len(self.numbers)

    get_number_of_atoms = get_global_number_of_atoms

    def get_positions(self):
        return self.positions.copy()

    def set_positions(self, newpositions):
        self.positions[:] = newpositions

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_chemical_symbols(self):
        return [_data.chemical_symbols[z] for z in self.numbers]

    def get_cell(self):
        return self.cell.copy()

    def get_pbc(self):
        return self.pbc.copy()

    def get_cell_lengths_and_angles(self):
        lengths = np.sqrt((self.cell ** 2).sum(axis=1))
        angles = []
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            ll = lengths[j] * lengths[k]
            if ll > 1e-16:
                x = np.dot(self.cell[j], self.cell[k]) / ll
                angles.append(180.0 / np.pi * np.arccos(x))
            else:
                angles.append(90.0)
        return np.array(list(lengths) + angles)

    def get_volume(self):
        return abs(np.linalg.det(self.cell))

    def get_masses(self):
        if self._masses is None:
            return np.array([_data.atomic_masses[z] for z in self.numbers])
        return self._masses.copy()

    def get_center_of_mass(self):
        m = self.get_masses()
        return np.dot(m, self.positions) / m.sum()

    def translate(self, displacement):
        self.positions += np.array(displacement)

    @property
    def symbols(self):
        return _Symbols(self.numbers)

    def copy(self):
        return Frame(self.numbers, self.positions, self.cell, self.pbc, self._masses)

    def __repr__(self):
        return "Frame(N=%d, cell=%s)" % (len(self), np.array2string(self.cell, precision=4))


def _is_torch_tensor(x):
    return type(x).__module__.startswith("torch")


class PackedTrajectory(object):
    """Packed trajectory: the buffers the C ABI consumes.

    Args:
        pos: f64 ``[F][N][3]``, C-contiguous; numpy array or torch tensor
            (CPU or CUDA).  A CUDA tensor is used in place (no copy).
        cell: f64 ``[F][3][3]`` or ``[3][3]`` (constant cell).
        numbers: int ``[N]`` atomic numbers (same for every frame, as the
            reference assumes: amof/rdf.py:71, amof/msd.py:215).
        masses: f64 ``[N]`` or None (standard atomic weights).
        pbc: 3 bools.
    """

    def __init__(self, pos, cell, numbers, masses=None, pbc=(True, True, True)):
        if _is_torch_tensor(pos):
            import torch
            if pos.dtype != torch.float64 or pos.dim() != 3 or pos.shape[2] != 3:
                raise ValueError("pos must be float64 [F][N][3]")
            if not pos.is_contiguous():
                pos = pos.contiguous()
            self.pos = pos
        else:
            pos = np.ascontiguousarray(pos, dtype=np.float64)
            if pos.ndim != 3 or pos.shape[2] != 3:
                raise ValueError("pos must be float64 [F][N][3]")
            self.pos = pos
        self.n_frames = int(self.pos.shape[0])
        self.n_atoms = int(self.pos.shape[1])
        cell = np.ascontiguousarray(cell, dtype=np.float64)
        if cell.shape == (3, 3):
            cell = cell.reshape(1, 3, 3)
        if cell.ndim != 3 or cell.shape[1:] != (3, 3) or cell.shape[0] not in (1, self.n_frames):
            raise ValueError("cell must be [3][3] or [F][3][3]")
        self.cell = cell
        self.numbers = np.ascontiguousarray(numbers, dtype=np.int64)
        if self.numbers.shape != (self.n_atoms,):
            raise ValueError("numbers must have N entries")
        if masses is None:
            masses = np.array([_data.atomic_masses[z] for z in self.numbers], dtype=np.float64)
        self.masses = np.ascontiguousarray(masses, dtype=np.float64)
        if isinstance(pbc, (bool, np.bool_)):
            pbc = (pbc,) * 3
        self.pbc = np.array(pbc, dtype=bool)

    def __len__(self):
        return self.n_frames

    @property
    def on_device(self):
        return _is_torch_tensor(self.pos) and self.pos.is_cuda

    def cell_of(self, k):
        return self.cell[k if self.cell.shape[0] > 1 else 0]

    def cells_full(self):
        """``[F][3][3]`` view of the cells (broadcast when constant)."""
        if self.cell.shape[0] == self.n_frames:
            return self.cell
        return np.broadcast_to(self.cell, (self.n_frames, 3, 3))

    def cell_lengths(self):
        return np.sqrt((self.cell ** 2).sum(axis=2))

    def volumes(self):
        return np.abs(np.linalg.det(self.cell))

    def volume_sum(self, frame_range=None):
        """Sum of the cell volumes over the frames, in the library's own operation order (``csrc/ctx.hip geom_one`` /
        ``build_geometry``: cofactor expansion along the first row, ``|det|``, added frame by frame -- a constant cell
        is ADDED once per frame, not multiplied), so that the value equals the ``volume_sum`` of an
        ``amof_rdf_accumulate`` call over the same frames bit for bit.  A rank of a frame-sharded run computes the
        whole trajectory's sum here instead of all-reducing partial sums (one collective less, and the mean volume no
        longer depends on the sharding)."""
        f0, f1 = (0, self.n_frames) if frame_range is None else frame_range
        c = self.cell.reshape(self.cell.shape[0], 9)
        if c.shape[0] > 1:
            c = c[f0:f1]
        m00 = c[:, 4] * c[:, 8] - c[:, 5] * c[:, 7]
        m01 = c[:, 3] * c[:, 8] - c[:, 5] * c[:, 6]
        m02 = c[:, 3] * c[:, 7] - c[:, 4] * c[:, 6]
        vol = np.abs(c[:, 0] * m00 - c[:, 1] * m01 + c[:, 2] * m02)
        if vol.shape[0] == 1:
            vol = np.full(f1 - f0, vol[0])
        # np.cumsum adds strictly left to right (np.sum does not: pairwise blocks)
        return float(np.cumsum(vol)[-1]) if vol.size else 0.0

    def pos_host(self):
        if _is_torch_tensor(self.pos):
            return self.pos.detach().cpu().numpy()
        return self.pos

    @property
    def device_index(self):
        """GPU that holds the positions (a CUDA ``pos`` tensor, or the copy made by ``keep_on_device``); else None"""
        kept = getattr(self, "_dev_pos", None)
        if kept is not None:
            return kept.device.index
        return self.pos.device.index if self.on_device else None

    def keep_on_device(self, device=0):
        """Host trajectory with a resident device copy: the positions are uploaded ONCE, every later analysis call on
        this object reads the copy instead of staging 24 N F bytes over PCIe again (a host-resident 9792 x 5000
        trajectory: 22 ms per pass for 1.4 ms of MSD kernels).  Explicit, and safe against the stale-copy trap: while
        the copy exists the host array is read-only (numpy raises on ``packed.pos[...] = ...``); ``release_device()``
        drops the copy and makes the array writable again.  (A view taken BEFORE this call can still be written
        through -- do not keep one.)"""
        kept = getattr(self, "_dev_pos", None)
        if self.on_device or kept is not None:
            have = self.pos.device.index if self.on_device else kept.device.index
            if have != int(device):
                raise ValueError("the trajectory already has a copy on cuda:%d; release_device() first to move it to "
                                 "cuda:%d" % (have, int(device)))
            return self
        import torch
        was_writeable = bool(self.pos.flags.writeable)
        src = self.pos
        if not was_writeable:           # (torch.from_numpy warns on read-only arrays; the upload only reads)
            src = self.pos.view()
            try:
                src.flags.writeable = True
            except ValueError:          # a truly read-only base (np.memmap mode='r', np.frombuffer): stage through a copy
                src = np.array(self.pos)
        self._dev_pos = torch.from_numpy(src).to(torch.device("cuda", int(device)))
        self._was_writeable = was_writeable
        self.pos.flags.writeable = False
        return self

    def release_device(self):
        if getattr(self, "_dev_pos", None) is not None:
            self._dev_pos = None
            if getattr(self, "_was_writeable", True):      # (an array that was read-only before stays read-only)
                self.pos.flags.writeable = True
        return self

    def to_device(self, device=0):
        """Copy the positions to HBM once (torch CUDA tensor); every later analysis call on the
        returned trajectory reads them in place instead of staging 24*N*F bytes over PCIe."""
        import torch
        if self.on_device:
            return self
        pos = torch.as_tensor(self.pos_host()).to(torch.device("cuda", device))
        return PackedTrajectory(pos, self.cell, self.numbers, self.masses, self.pbc)

    # -- per-trajectory constants, computed once (``numbers`` is never modified after construction) --------------
    def _const(self, key, make):
        cache = self.__dict__.setdefault("_const_cache", {})
        if key not in cache:
            cache[key] = make()
        return cache[key]

    def unique_numbers(self):
        """``list(set(atoms.get_atomic_numbers()))`` -- the reference's species order (amof/rdf.py:71,
        amof/atom.py:44-46), which fixes DataFrame column order."""
        return list(self._const("unique", lambda: list(set(self.numbers))))

    def species_counts(self):
        """``{atomic number: number of atoms}``"""
        def make():
            zs, counts = np.unique(self.numbers, return_counts=True)
            return {int(z): int(c) for z, c in zip(zs, counts)}
        return self._const("counts", make)

    def formula_count(self):
        """``{symbol: count}`` in order of first appearance (``atoms.symbols.formula._count``)."""
        def make():
            zs, first, counts = np.unique(self.numbers, return_index=True, return_counts=True)
            order = np.argsort(first)
            return {_data.chemical_symbols[int(zs[k])]: int(counts[k]) for k in order}
        return dict(self._const("formula", make))

    def frame(self, k):
        """Materialise frame ``k`` as a :class:`Frame` (host copy)."""
        if _is_torch_tensor(self.pos):
            p = self.pos[k].detach().cpu().numpy()
        else:
            p = self.pos[k]
        return Frame(self.numbers, p, self.cell_of(k), self.pbc, self.masses)

    def to_frames(self):
        return [self.frame(k) for k in range(self.n_frames)]


def _usable_cpus():
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                # container CPU quota (cgroup v2)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pack_trajectory(trajectory):
    """Pack a list of ``ase.Atoms``-like frames into a :class:`PackedTrajectory`.

    Accepts a :class:`PackedTrajectory` unchanged.  Every frame must hold the
    same atoms in the same order (the reference assumes it: species are read
    from frame 0 only, amof/rdf.py:71, amof/cn.py:52, amof/msd.py:215).
    """
    if isinstance(trajectory, PackedTrajectory) or getattr(trajectory, "is_stream", False):
        return trajectory                   # (an amof_amd.stream.XyzStream: the classes walk it batch by batch)
    frames = list(trajectory)
    if len(frames) == 0:
        raise ValueError("empty trajectory")
    first = frames[0]
    numbers = np.array(first.get_atomic_numbers(), dtype=np.int64)
    n = len(numbers)
    pos = np.empty((len(frames), n, 3), dtype=np.float64)
    cell = np.empty((len(frames), 3, 3), dtype=np.float64)

    def copy_range(k0, k1):
        for k in range(k0, k1):
            atoms = frames[k]
            # ``atoms.positions`` is a view in ASE (and here): one copy, straight into the packed array
            p = getattr(atoms, "positions", None)
            if p is None:
                p = atoms.get_positions()
            if len(p) != n:
                raise ValueError("frame %d has %d atoms, frame 0 has %d" % (k, len(p), n))
            pos[k] = p
            c = getattr(atoms, "cell", None)
            cell[k] = np.asarray(c if c is not None else atoms.get_cell(), dtype=np.float64).reshape(3, 3)

    # big trajectories: the frame copies (numpy releases the GIL for them) on a few threads -- a 9792-atom x 5000-frame
    # list is 1.2 GB of memcpy, which one thread moves in ~0.13 s, more than the analysis of the whole trajectory takes
    nbytes = pos.nbytes
    workers = min(8, _usable_cpus(), len(frames) // 64) if nbytes >= (64 << 20) else 1
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        step = (len(frames) + workers - 1) // workers
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(lambda w: copy_range(w * step, min((w + 1) * step, len(frames))), range(workers)))
    else:
        copy_range(0, len(frames))
    if (cell == cell[0]).all():
        cell = cell[:1].copy()
    pbc = np.array(getattr(first, "pbc", (True, True, True)), dtype=bool)
    if pbc.shape == ():
        pbc = np.array([bool(pbc)] * 3)
    return PackedTrajectory(pos, cell, numbers, np.array(first.get_masses(), dtype=np.float64), pbc)
