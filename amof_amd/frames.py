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

import os
import threading

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
        self._protect()
        return self

    def _protect(self):
        """while a device copy exists ``self.pos`` is a READ-ONLY view of the host array (an array that cannot be made
        writeable again later -- one backed by a torch tensor, say -- keeps its flag: the view carries the protection)"""
        if "_pos_rw" not in self.__dict__:
            self._pos_rw = self.pos
            view = self.pos.view()
            view.flags.writeable = False
            self.pos = view

    def _unprotect(self):
        rw = self.__dict__.pop("_pos_rw", None)
        if rw is not None:
            self.pos = rw

    def release_device(self):
        rc = self.__dict__.pop("_resident", None)
        if rc is not None and not rc.complete:
            rc.finish()                 # (an upload in flight is completed first: its thread writes into the tensor)
        if getattr(self, "_dev_pos", None) is not None:
            self._dev_pos = None
            self._unprotect()
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


def _gpu_for_uploads(device):
    """torch, when a host trajectory can get a resident copy on ``device``; None otherwise (no GPU, torch missing, switched
    off with AMOF_KEEP_ON_DEVICE=0)"""
    import os
    if device is None or isinstance(device, (list, tuple)) or os.environ.get("AMOF_KEEP_ON_DEVICE", "1") == "0":
        return None
    try:
        import torch
    except ImportError:
        return None
    if not torch.cuda.is_available() or int(device) >= torch.cuda.device_count():
        return None
    return torch


# Two page-locked staging buffers, kept for the process: locking 2 x 48 MB of pages costs ~10 ms -- per upload, when every
# ResidentCopy allocated its own.  One upload at a time borrows them; a concurrent one allocates its own pair.
_STAGING = []
_STAGING_LOCK = threading.Lock()


def _staging_slots(torch, n_doubles):
    """(two pinned float64 buffers of >= n_doubles elements, borrowed?) -- borrowed ones are returned by releasing
    _STAGING_LOCK when the upload is through"""
    if _STAGING_LOCK.acquire(False):
        try:
            if not _STAGING or _STAGING[0].numel() < n_doubles:
                _STAGING[:] = [torch.empty(n_doubles, dtype=torch.float64, pin_memory=True) for _ in range(2)]
            return list(_STAGING), True
        except BaseException:
            _STAGING_LOCK.release()
            raise
    return [torch.empty(n_doubles, dtype=torch.float64, pin_memory=True) for _ in range(2)], False


class ResidentCopy(object):
    """The device copy of a HOST trajectory, filled in frame order while its first analyses already run.

    A list of frames or a host ``PackedTrajectory`` used to be staged over PCIe by every analysis call again (1.2 GB per
    call at the headline size: 139 ms for RDF + MSD where the kernels need 75).  Now the first analysis that meets a host
    trajectory starts ONE upload -- chunks of frames through page-locked memory on a stream of its own, every chunk marked
    with an event -- and walks the chunks that have arrived like the batches of a stream (RDF / BAD / CN add their integer
    counts up batch by batch: amof_amd/stream.py); when the upload is complete the trajectory keeps the copy
    (``keep_on_device``: the host array is read-only while it exists; ``release_device()`` drops it) and every later
    analysis of the same object reads it in place.  Offers what the classes ask of a stream: ``batches()``, ``read_all()``.
    """

    is_stream = True
    on_device = False
    CHUNK_BYTES = 48 << 20
    PIECE_FRAMES = 512          # a batch handed to an analysis: at least this many frames (an RDF launch per batch)

    def __init__(self, packed, device, torch):
        import threading
        self.packed = packed
        self.device = int(device)
        self._torch = torch
        F, N = packed.n_frames, packed.n_atoms
        self.n_frames, self.n_atoms = F, N
        self.numbers, self.masses, self.pbc, self.cell = packed.numbers, packed.masses, packed.pbc, packed.cell
        self.dev = torch.empty((F, N, 3), dtype=torch.float64, device=torch.device("cuda", self.device))
        self.stream = torch.cuda.Stream(device=self.device)
        self.chunk = max(1, min(F, self.CHUNK_BYTES // max(1, 24 * N)))
        self._marks = []                # (frames uploaded so far, event)
        self._cond = threading.Condition()
        self._error = None
        self.complete = F == 0
        if F:
            self._thread = threading.Thread(target=self._run, name="amof-upload", daemon=True)
            self._thread.start()

    # -- producer -------------------------------------------------------------------------------------------------------
    def _run(self):
        lent = False
        try:
            torch = self._torch
            F = self.n_frames
            # through two page-locked staging buffers (a pageable hipMemcpyAsync blocks the caller and
            # runs at a fraction of the link rate)
            slots, lent = _staging_slots(torch, self.chunk * self.n_atoms * 3)
            slots = [b[:self.chunk * self.n_atoms * 3].view(self.chunk, self.n_atoms, 3) for b in slots]
            busy = [None, None]
            src = self.packed.pos
            # the copy into the staging buffer on several cores (amof_pack_frames with the frames of the contiguous array
            # as its "list"): one core moves 48 MB in ~4 ms, 1.2 GB in ~100 -- longer than the RDF needs for the frames, which
            # then waited for the upload; the DMA itself takes 1 - 2 ms per chunk
            copy = None
            if src.flags.c_contiguous and src.dtype == np.float64 and self.n_atoms > 0:
                try:
                    import ctypes
                    from . import _hip
                    lib = _hip.load_library()
                    ptrs = (np.uint64(src.__array_interface__["data"][0]) +
                            np.arange(F, dtype=np.uint64) * np.uint64(24 * self.n_atoms))
                    threads = max(1, int(os.environ.get("AMOF_UPLOAD_THREADS", 0)) or min(8, _usable_cpus() // 2))

                    def copy(dst, f0, f1):
                        rc = lib.amof_pack_frames(ctypes.c_void_p(ptrs[f0:f1].ctypes.data), f1 - f0, self.n_atoms,
                                                  ctypes.c_void_p(dst.data_ptr()), None, threads)
                        if rc != 0:
                            raise RuntimeError("amof_pack_frames failed (%d)" % rc)
                except Exception:
                    copy = None
            for q, f0 in enumerate(range(0, F, self.chunk)):
                f1 = min(F, f0 + self.chunk)
                j = q % 2
                if busy[j] is not None:
                    busy[j].synchronize()
                if copy is not None:
                    copy(slots[j], f0, f1)
                else:
                    np.copyto(slots[j].numpy()[:f1 - f0], src[f0:f1])
                with torch.cuda.stream(self.stream):
                    self.dev[f0:f1].copy_(slots[j][:f1 - f0], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                busy[j] = ev
                with self._cond:
                    self._marks.append((f1, ev))
                    self._cond.notify_all()
            for ev in busy:
                if ev is not None:
                    ev.synchronize()
        except BaseException as exc:       # handed to the consumers
            with self._cond:
                self._error = exc
                self._cond.notify_all()
        finally:
            if lent:                        # (nothing of this upload may still read the borrowed buffers)
                try:
                    self.stream.synchronize()
                finally:
                    _STAGING_LOCK.release()

    # -- consumers ------------------------------------------------------------------------------------------------------
    def _wait_frames(self, f1):
        """block until frames [0, f1) are on the device"""
        with self._cond:
            while True:
                if self._error is not None:
                    raise self._error
                hit = next((ev for upto, ev in self._marks if upto >= f1), None)
                if hit is not None:
                    break
                self._cond.wait(0.5)
        hit.synchronize()

    def __len__(self):
        return self.n_frames

    def unique_numbers(self):
        return self.packed.unique_numbers()

    def species_counts(self):
        return self.packed.species_counts()

    def formula_count(self):
        return self.packed.formula_count()

    def cell_lengths(self):
        return self.packed.cell_lengths()

    def volume_sum(self):
        return self.packed.volume_sum()

    def _piece(self, f0, f1):
        cell = self.cell if self.cell.shape[0] == 1 else self.cell[f0:f1]
        piece = PackedTrajectory(self.dev[f0:f1], cell, self.numbers, self.masses, self.pbc)
        # (what depends on the atoms only is computed once, for the trajectory, not per batch)
        if "_abi_species" not in self.packed.__dict__:
            from . import _hip
            _hip.packed_species(self.packed)
        self.packed.__dict__.setdefault("_const_cache", {})
        for key in ("_const_cache", "_abi_species"):
            if key in self.packed.__dict__:
                piece.__dict__[key] = self.packed.__dict__[key]
        return piece

    def batches(self):
        """device-resident pieces in frame order, each as soon as it has arrived"""
        F = self.n_frames
        f0 = 0
        while f0 < F:
            # (the first batch: one staging chunk -- the analysis starts ~3 ms after the upload instead of ~8)
            f1 = min(F, f0 + (self.PIECE_FRAMES if f0 else min(self.PIECE_FRAMES, self.chunk)))
            # more has ARRIVED meanwhile (its copy is complete, not merely queued: waiting for the chunk in flight cost ~1 ms of
            # idle GPU between two batches): take it all in one batch
            with self._cond:
                marks = list(self._marks)
            have = 0
            for upto, ev in marks:
                if upto > have and ev.query():
                    have = upto
            if have > f1:
                f1 = have
            self._wait_frames(f1)
            yield self._piece(f0, f1)
            f0 = f1
        self.finish()

    def finish(self):
        """wait for the whole copy and hand it to the trajectory (``keep_on_device`` semantics)"""
        self._wait_frames(self.n_frames)
        p = self.packed
        if getattr(p, "_dev_pos", None) is None and not self.complete:
            p._dev_pos = self.dev
            p._protect()
        self.complete = True
        return p

    def read_all(self):
        return self.finish()


def resident_source(packed, device, allow=True):
    """What an analysis walks: the trajectory itself (resident already, a stream, no GPU for a copy, a multi-rank run) or its
    ``ResidentCopy`` in the making (a host trajectory's first analyses)."""
    if not allow or getattr(packed, "is_stream", False) or packed.on_device or getattr(packed, "_dev_pos", None) is not None:
        return packed
    rc = getattr(packed, "_resident", None)
    if rc is not None:
        if rc.complete:
            return packed
        return rc if rc.device == int(device) else packed
    torch = _gpu_for_uploads(device)
    if torch is None or packed.n_frames == 0 or packed.n_atoms == 0:
        return packed
    need = packed.n_frames * packed.n_atoms * 24
    try:
        free, _total = torch.cuda.mem_get_info(int(device))
    except Exception:
        return packed
    if need < (32 << 20) or 2 * need + (4 << 30) > free:       # (small: not worth a thread; huge: staged per call as before)
        return packed
    packed._resident = ResidentCopy(packed, device, torch)
    return packed._resident


# the lists packed last: a caller that hands the SAME list of frames to several analyses (the reference's documented use,
# examples/Compute structural properties.py:58-118) packs and uploads it once.  A list is recognised by identity, length and
# the identity of its frames, and is trusted only when a checksum of EVERY frame's bytes (amof_frames_checksum: memory
# speed, all cores) equals the one taken when it was packed -- positions edited in place give a new pack.
_PACKED_LISTS = []
_PACKED_KEEP = 2


def _frame_pointers(frames, n):
    """(uint64 pointer array, keep-alive list): every frame's positions as a C-contiguous float64 [n][3] block"""
    ptrs = np.empty(len(frames), dtype=np.uint64)
    keep = []
    for k, atoms in enumerate(frames):
        p = getattr(atoms, "positions", None)
        if p is None:
            p = atoms.get_positions()
        if not (isinstance(p, np.ndarray) and p.dtype == np.float64 and p.flags.c_contiguous):
            p = np.ascontiguousarray(p, dtype=np.float64)
        if p.shape != (n, 3):
            raise ValueError("frame %d has %d atoms, frame 0 has %d" % (k, len(p), n))
        keep.append(p)
        ptrs[k] = p.__array_interface__["data"][0]
    return ptrs, keep


def pack_trajectory(trajectory, device=None):
    """Pack a list of ``ase.Atoms``-like frames into a :class:`PackedTrajectory`.

    Accepts a :class:`PackedTrajectory` unchanged.  Every frame must hold the
    same atoms in the same order (the reference assumes it: species are read
    from frame 0 only, amof/rdf.py:71, amof/cn.py:52, amof/msd.py:215).

    The frames are copied natively on all cores (``amof_pack_frames``: 13 ms for 5000 x 9792 atoms).  ``device`` (what the
    analysis classes pass): the GPU the frames are headed for -- their ONE upload starts at once (``ResidentCopy``).  The
    list is remembered, so the next analysis of the same, unchanged list reuses pack and device copy.
    """
    if isinstance(trajectory, PackedTrajectory) or getattr(trajectory, "is_stream", False):
        return trajectory                   # (an amof_amd.stream.XyzStream: the classes walk it batch by batch)
    frames = trajectory if isinstance(trajectory, list) else list(trajectory)
    if len(frames) == 0:
        raise ValueError("empty trajectory")
    first = frames[0]
    numbers = np.array(first.get_atomic_numbers(), dtype=np.int64)
    n = len(numbers)
    F = len(frames)
    lib = None
    if F * n * 24 >= (8 << 20):
        try:
            from . import _hip
            lib = _hip.load_library()
        except Exception:
            lib = None
    if lib is None:
        # small trajectories (or no library): the plain loop
        pos = np.empty((F, n, 3), dtype=np.float64)
        for k, atoms in enumerate(frames):
            p = getattr(atoms, "positions", None)
            if p is None:
                p = atoms.get_positions()
            if len(p) != n:
                raise ValueError("frame %d has %d atoms, frame 0 has %d" % (k, len(p), n))
            pos[k] = p
        return PackedTrajectory(pos, _cells_of(frames), numbers, np.array(first.get_masses(), dtype=np.float64), _pbc_of(first))
    import ctypes
    threads = _usable_cpus()
    # the same list, unchanged?  (cheap identities first: the list, its frames, their position arrays; then the bytes)
    ids = np.fromiter(map(id, frames), dtype=np.int64, count=F)
    for entry in _PACKED_LISTS:
        if entry["list_id"] == id(trajectory) and entry["n"] == n and np.array_equal(entry["ids"], ids):
            pids = np.fromiter((id(getattr(f, "positions", None)) for f in frames), dtype=np.int64, count=F)
            if not np.array_equal(pids, entry["pids"]):
                continue
            sums = np.empty(F, dtype=np.uint64)
            rc = lib.amof_frames_checksum(ctypes.c_void_p(entry["ptrs"].ctypes.data), F, n, ctypes.c_void_p(sums.ctypes.data), threads)
            if rc == 0 and np.array_equal(sums, entry["sums"]) and np.array_equal(_cells_of(frames), entry["packed"].cell):
                return entry["packed"]
    ptrs, keep = _frame_pointers(frames, n)
    pids = np.fromiter((id(getattr(f, "positions", None)) for f in frames), dtype=np.int64, count=F)
    pos = np.empty((F, n, 3), dtype=np.float64)
    packed = PackedTrajectory(pos, _cells_of(frames), numbers, np.array(first.get_masses(), dtype=np.float64), _pbc_of(first))
    sums = np.empty(F, dtype=np.uint64)
    rc = lib.amof_pack_frames(ctypes.c_void_p(ptrs.ctypes.data), F, n, ctypes.c_void_p(pos.ctypes.data), ctypes.c_void_p(sums.ctypes.data),
                              threads)
    if rc != 0:
        raise RuntimeError("amof_pack_frames failed (%d)" % rc)
    # (measured and rejected: packing straight into a page-locked array and queueing every batch's DMA from it -- locking
    #  1.2 GB of pages costs 150 ms per list, ten times the copy it saves.  The device copy is started by the first analysis:
    #  resident_source -> ResidentCopy, a background thread through two page-locked 48 MB buffers.)
    if device is not None:
        resident_source(packed, device)
    # (`keep` stays with the entry: the pointers remain valid while the frames keep their position arrays, which `pids` checks)
    _PACKED_LISTS.insert(0, {"list_id": id(trajectory), "n": n, "ids": ids, "pids": pids, "ptrs": ptrs, "keep": keep, "sums": sums,
                             "packed": packed})
    del _PACKED_LISTS[_PACKED_KEEP:]
    return packed


def forget_packed_lists():
    """drop the remembered packs (and with them their page-locked arrays and device copies)"""
    del _PACKED_LISTS[:]


def _cells_of(frames):
    def rows(atoms):
        c = getattr(atoms, "cell", None)
        if c is None:
            c = atoms.get_cell()
        return getattr(c, "array", c)          # (ase.cell.Cell keeps its 3 x 3 array in .array)
    cell = np.array([rows(a) for a in frames], dtype=np.float64).reshape(len(frames), 3, 3)
    if (cell == cell[0]).all():
        cell = cell[:1].copy()
    return cell


def _pbc_of(first):
    pbc = np.array(getattr(first, "pbc", (True, True, True)), dtype=bool)
    if pbc.shape == ():
        pbc = np.array([bool(pbc)] * 3)
    return pbc
