"""ctypes binding of ``libamofhip.so`` (the C ABI declared in include/amof_hip.h).

This is the only compute back end of the package: there is no CPU fallback.
If the shared library is missing, or no GPU is visible, the analysis classes
raise -- they never silently compute somewhere else.
"""

import ctypes
import os
import threading
import time

import numpy as np

from .frames import PackedTrajectory, _is_torch_tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMOF_HIP_LIB") or os.path.join(_HERE, "csrc", "libamofhip.so")

AMOF_OK = 0
AMOF_EINVAL = -1
AMOF_ESINGULAR = -2
AMOF_EANGLE = -3
AMOF_ENOMEM = -4
AMOF_EHIP = -5
AMOF_ECAPACITY = -6
AMOF_ENODEVICE = -7
AMOF_EUNSUPPORTED = -8
ABI_VERSION = 4

EXPORTS = [
    "amof_abi_version", "amof_device_count", "amof_ctx_create", "amof_ctx_create2", "amof_ctx_destroy", "amof_last_error",
    "amof_ctx_set_stream", "amof_ctx_synchronize", "amof_ctx_wait_stream", "amof_ctx_follow", "amof_ctx_calls", "amof_ctx_debug_poison", "amof_last_kernel_seconds", "amof_last_kernel_launches",
    "amof_last_path",
    "amof_rdf_accumulate", "amof_rdf_accumulate_dev", "amof_cn_count", "amof_bad_hist", "amof_bad_hist_dev",
    "amof_bad_hist_by_cn",
    "amof_msd_window", "amof_msd_window_dev", "amof_msd_com_dev", "amof_msd_shard_begin", "amof_msd_shard_finish", "amof_msd_direct",
    "amof_xyz_scan", "amof_xyz_read", "amof_xyz_open", "amof_xyz_read_frames", "amof_xyz_close", "amof_cp2k_cell_read", "amof_ingest_last_error",
    "amof_pack_frames", "amof_frames_checksum",
]


class AmofError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libamofhip error %d: %s" % (code, message))
        self.code = code


class Unsupported(AmofError):
    """AMOF_EUNSUPPORTED: a specialised entry point does not take the arguments; the caller uses the general one"""


class AmofTraj(ctypes.Structure):
    """``struct amof_traj`` (include/amof_hip.h)."""
    _fields_ = [
        ("pos", ctypes.c_void_p),
        ("pos_on_device", ctypes.c_int32),
        ("n_species", ctypes.c_int32),
        ("cell", ctypes.c_void_p),
        ("n_cells", ctypes.c_int64),
        ("n_frames", ctypes.c_int64),
        ("n_atoms", ctypes.c_int64),
        ("species", ctypes.c_void_p),
        ("masses", ctypes.c_void_p),
        ("pbc", ctypes.c_uint8 * 3),
        ("_pad", ctypes.c_uint8 * 5),
    ]


_lib = None
_lib_lock = threading.Lock()
_ctx_lock = threading.Lock()


def load_library():
    """Load ``libamofhip.so`` (in-tree build) and declare the prototypes."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "amof_amd: %s not found. Build it with `make -C amof_amd/csrc` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If torch is
        # importable, let it load first so that this library binds to the same runtime (loading the
        # system runtime first makes torch.cuda report "No HIP GPUs are available" later on).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        P = ctypes.c_void_p
        lib.amof_abi_version.restype = ctypes.c_int
        lib.amof_device_count.restype = ctypes.c_int
        lib.amof_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(P)]
        lib.amof_ctx_create2.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(P)]
        lib.amof_ctx_destroy.argtypes = [P]
        lib.amof_ctx_destroy.restype = None
        lib.amof_last_error.argtypes = [P]
        lib.amof_last_error.restype = ctypes.c_char_p
        lib.amof_ctx_set_stream.argtypes = [P, P]
        lib.amof_ctx_synchronize.argtypes = [P]
        lib.amof_ctx_wait_stream.argtypes = [P, P]
        lib.amof_ctx_follow.argtypes = [P, P, ctypes.c_int64, ctypes.c_double]
        lib.amof_ctx_calls.argtypes = [P]
        lib.amof_ctx_calls.restype = ctypes.c_int64
        lib.amof_ctx_debug_poison.argtypes = [P, ctypes.c_int]
        lib.amof_last_kernel_seconds.argtypes = [P, ctypes.c_int]
        lib.amof_last_kernel_seconds.restype = ctypes.c_double
        lib.amof_last_kernel_launches.argtypes = [P]
        lib.amof_last_kernel_launches.restype = ctypes.c_int64
        lib.amof_last_path.argtypes = [P]
        lib.amof_last_path.restype = ctypes.c_char_p
        TP = ctypes.POINTER(AmofTraj)
        lib.amof_rdf_accumulate.argtypes = [P, TP, ctypes.c_double, ctypes.c_int32, P, ctypes.POINTER(ctypes.c_double)]
        lib.amof_rdf_accumulate_dev.argtypes = lib.amof_rdf_accumulate.argtypes
        lib.amof_cn_count.argtypes = [P, TP, P, P, ctypes.c_int32, P, P]
        lib.amof_bad_hist.argtypes = [P, TP, P, P, ctypes.c_int32, P, ctypes.c_int32, P, P]
        lib.amof_bad_hist_dev.argtypes = lib.amof_bad_hist.argtypes
        lib.amof_bad_hist_by_cn.argtypes = [P, TP, P, P, ctypes.c_int32, P, ctypes.c_int32, ctypes.c_int32, P, P]
        lib.amof_msd_window.argtypes = [P, TP, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_int64, ctypes.c_int64, P]
        lib.amof_msd_window_dev.argtypes = [P, TP, P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int64, ctypes.c_int64, P, P]
        lib.amof_msd_com_dev.argtypes = [P, TP, ctypes.c_int64, ctypes.c_int64, P]
        lib.amof_msd_shard_begin.argtypes = [P, TP, P, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, P]
        lib.amof_msd_shard_finish.argtypes = [P, TP, P, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, P, P]
        lib.amof_msd_direct.argtypes = [P, TP, P]
        lib.amof_xyz_scan.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
        lib.amof_xyz_read.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                      P, P, P, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
        lib.amof_xyz_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(P), ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_int64)]
        lib.amof_xyz_read_frames.argtypes = [P, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                             P, P, P, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32]
        lib.amof_xyz_close.argtypes = [P]
        lib.amof_xyz_close.restype = None
        lib.amof_cp2k_cell_read.argtypes = [ctypes.c_char_p, ctypes.c_int64, P, ctypes.POINTER(ctypes.c_int64)]
        lib.amof_ingest_last_error.restype = ctypes.c_char_p
        lib.amof_pack_frames.argtypes = [P, ctypes.c_int64, ctypes.c_int64, P, P, ctypes.c_int32]
        lib.amof_frames_checksum.argtypes = [P, ctypes.c_int64, ctypes.c_int64, P, ctypes.c_int32]
        if lib.amof_abi_version() != ABI_VERSION:
            raise RuntimeError("libamofhip.so ABI version %d, expected %d" % (lib.amof_abi_version(), ABI_VERSION))
        _lib = lib
        return lib


def device_count():
    return load_library().amof_device_count()


def species_index(numbers):
    """Map atomic numbers to the C ABI's species indices.

    Returns ``(kinds, species)``: ``kinds`` = sorted unique atomic numbers,
    ``species[i]`` = index of atom i's number in ``kinds`` (int32)."""
    numbers = np.asarray(numbers)
    uniq, inverse = np.unique(numbers, return_inverse=True)
    return [int(z) for z in uniq], inverse.astype(np.int32).reshape(-1)


def packed_species(packed):
    """:func:`species_index` of a PackedTrajectory, computed once per trajectory (numbers never change)."""
    cached = getattr(packed, "_abi_species", None)
    if cached is None:
        cached = species_index(packed.numbers)
        packed._abi_species = cached
    return cached


class _TrajHandle(object):
    """Keeps the numpy buffers behind an ``AmofTraj`` alive."""

    def __init__(self, packed, frame_range=None):
        if getattr(packed, "is_stream", False):
            raise TypeError("this analysis does not walk a stream batch by batch: pass stream.read_all() (Rdf, "
                            "cn.CoordinationNumber and Bad accept the stream itself)")
        if not isinstance(packed, PackedTrajectory):
            raise TypeError("expected a PackedTrajectory, got %s" % type(packed).__name__)
        f0, f1 = (0, packed.n_frames) if frame_range is None else frame_range
        self.kinds, self.species = packed_species(packed)
        pos = getattr(packed, "_dev_pos", None)        # (PackedTrajectory.keep_on_device: resident copy of a host array)
        if pos is None:
            pos = packed.pos
        cell = packed.cell if packed.cell.shape[0] == 1 else packed.cell[f0:f1]
        self.cell = np.ascontiguousarray(cell, dtype=np.float64)
        self.masses = np.ascontiguousarray(packed.masses, dtype=np.float64)
        t = AmofTraj()
        if _is_torch_tensor(pos):
            sub = pos[f0:f1]
            if not sub.is_contiguous():
                sub = sub.contiguous()
            self._pos = sub
            if sub.is_cuda:
                t.pos = sub.data_ptr()
                t.pos_on_device = 1
                self.device_index = sub.device.index
            else:
                t.pos = sub.data_ptr()
                t.pos_on_device = 0
                self.device_index = None
        else:
            sub = np.ascontiguousarray(pos[f0:f1])
            self._pos = sub
            t.pos = sub.ctypes.data
            t.pos_on_device = 0
            self.device_index = None
        t.n_species = len(self.kinds)
        t.cell = self.cell.ctypes.data
        t.n_cells = self.cell.shape[0]
        t.n_frames = f1 - f0
        t.n_atoms = packed.n_atoms
        t.species = self.species.ctypes.data
        t.masses = self.masses.ctypes.data
        for k in range(3):
            t.pbc[k] = 1 if packed.pbc[k] else 0
        self.c = t
        self.n_frames = f1 - f0
        self.n_atoms = packed.n_atoms
        self.S = len(self.kinds)


def _locked(method):
    """An ``amof_ctx`` is not thread-safe (include/amof_hip.h): serialise the calls of one Context.
    ctypes drops the GIL during a call, so two Python threads could otherwise be inside the same
    context at once; distinct Context objects still run concurrently."""
    def wrapper(self, *args, **kwargs):
        with self._lock:
            return method(self, *args, **kwargs)
    wrapper.__name__ = method.__name__
    wrapper.__doc__ = method.__doc__
    return wrapper


AMOF_CTX_HIGH_PRIORITY = 1
AMOF_CTX_LOW_PRIORITY = 2

_tls = threading.local()      # .producer_stream: the stream a lane job orders itself after (see Lane.submit)


class Lane(object):
    """One worker thread that runs jobs in submission order (the analysis classes enqueue their computation here and
    return: amof_amd/_lazy.py).  Mixed into :class:`Context`; ``device`` is the GPU whose current torch stream a job is
    ordered after (None: no GPU involved -- the CPU stand-ins of the test-suite)."""

    device = None
    _lane = None
    _lane_thread = None
    _last_job = None
    _lane_name = "amof-lane"
    _follows = None         # the lane whose kernels this lane's jobs are queued behind (Context.follow_leader)
    _jobs_submitted = 0     # (plain counters, written under the GIL: submit by the callers, the other two by the worker)
    _jobs_started = 0
    _jobs_finished = 0
    _calls_at_job_start = 0

    def submit(self, fn):
        """Run ``fn()`` on the worker thread, after everything submitted before; returns the
        ``concurrent.futures.Future``.  The job orders the context's stream after the work queued so far on the
        CALLER's current torch stream of this device (captured here: what produced a device-resident trajectory)."""
        from concurrent.futures import ThreadPoolExecutor
        if self._lane is None:
            with _ctx_lock:
                if self._lane is None:
                    self._lane = ThreadPoolExecutor(1, thread_name_prefix=self._lane_name)
        producer = None
        if self.device is not None:
            try:
                import torch
                if torch.cuda.is_available():
                    producer = torch.cuda.current_stream(self.device).cuda_stream
            except ImportError:
                pass

        def job():
            self._lane_thread = threading.get_ident()
            _tls.producer_stream = producer
            self._calls_at_job_start = self.device_calls()
            self._jobs_started += 1
            try:
                if self._follows is not None:
                    self.follow_leader()
                return fn()
            finally:
                _tls.producer_stream = None
                self._jobs_finished += 1
        self._jobs_submitted += 1
        fut = self._lane.submit(job)
        self._last_job = fut
        return fut

    def device_calls(self):
        """device calls begun on this lane's context so far (0: not a GPU context)"""
        return 0

    def follow_leader(self):
        """(GPU contexts) queue this lane's next kernels behind the leader lane's"""

    def drain(self):
        """wait for every job submitted to the lane (their errors stay with their futures)"""
        fut = self._last_job
        if fut is None or threading.get_ident() == self._lane_thread:
            return
        fut.exception()         # (jobs run in submission order: the last one done = all done)

    def close_lane(self):
        self.drain()
        if self._lane is not None:
            self._lane.shutdown(wait=True)
            self._lane = None


class Context(Lane):
    """One ``amof_ctx``: a device, a stream and its scratch memory -- and, for the analysis classes, a LANE: one worker
    thread that runs the (synchronous) entry points of this context in submission order, so that a class constructor
    can enqueue its analysis and return (``submit``; amof_amd/_lazy.py).  A device has two cached contexts
    (``get_context``): lane 0 for the pair-evaluation-bound RDF, lane 1 -- a stream of the highest priority -- for the
    memory-bound MSD / BAD / CN, whose kernels and host work then run beside an RDF launch instead of behind it."""

    def __init__(self, device=0, high_priority=False, priority=None):
        self._lib = load_library()
        n = self._lib.amof_device_count()
        if n <= 0:
            raise RuntimeError("amof_amd: no GPU visible to HIP; the MI355X kernels cannot run "
                               "(there is no CPU fallback)")
        h = ctypes.c_void_p()
        if priority is None:
            priority = "high" if high_priority else "normal"
        flags = {"high": AMOF_CTX_HIGH_PRIORITY, "low": AMOF_CTX_LOW_PRIORITY, "normal": 0}[priority]
        rc = self._lib.amof_ctx_create2(int(device), flags, ctypes.byref(h))
        if rc != AMOF_OK:
            raise AmofError(rc, "amof_ctx_create2(device=%d) failed" % device)
        self._h = h
        self.device = int(device)
        self.priority = priority
        self.high_priority = priority == "high"
        self._lane_name = "amof-lane-%d%s" % (self.device, "h" if high_priority else "")
        self._lock = threading.RLock()

    def close(self):
        self.close_lane()
        if getattr(self, "_h", None):
            self._lib.amof_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == AMOF_OK:
            return
        msg = self._lib.amof_last_error(self._h).decode("utf-8", "replace")
        if rc == AMOF_EANGLE:
            raise ZeroDivisionError("Undefined angle")  # what ASE raises (reference amof/bad.py:100)
        if rc == AMOF_EINVAL:
            raise ValueError(msg)
        if rc == AMOF_EUNSUPPORTED:
            raise Unsupported(rc, msg)
        raise AmofError(rc, msg)

    def device_calls(self):
        """calls of this context that have started device work (``amof_ctx_calls``; any thread may ask)"""
        return int(self._lib.amof_ctx_calls(self._h)) if getattr(self, "_h", None) else 0

    def follow_leader(self):
        """Queue this context's next kernels BEHIND the leader lane's (``amof_ctx_follow``; lane 1 follows lane 0 of its
        device).  If the leader has a job pending, wait -- on the host, a fraction of a millisecond -- until that job has
        queued its dominant kernel, then order this context's stream after the leader's: the kernels of the memory-bound
        analyses then run when the RDF tile kernel has finished, beside the RDF's read-back and DataFrame assembly, and
        this lane's host work beside the tile kernel.  Side by side the two lose: the tile kernel fills LDS and registers,
        a second stream's workgroups trickle in between (profiles/r05/stops.txt A).  Nothing to follow (leader idle):
        returns at once.  OPT-IN, ``AMOF_LANE_ORDER=1``: measured against the unordered lanes it changes nothing at N = 1
        (74.2 / 74.6 against 74.8 / 74.6 ms per step, sequential 74.6 / 74.1) and costs 0.1 - 0.2 ms for one rank of eight,
        where the followers' kernels and result assembly end up behind the RDF's instead of before it
        (profiles/r05/lane_order.txt)."""
        lead = self._follows
        if lead is None or lead is self or os.environ.get("AMOF_LANE_ORDER", "0") != "1":
            return
        target = lead._jobs_submitted
        if lead._jobs_finished >= target:
            return                                  # the leader is idle: nothing queued that this lane could be stuck behind
        while lead._jobs_finished < target:
            if lead._jobs_started >= target:
                # the job we follow is running: its first device call is number _calls_at_job_start + 1 of its context
                with self._lock:
                    rc = self._lib.amof_ctx_follow(self._h, lead._h, lead._calls_at_job_start + 1, 0.0005)
                if rc < 0:
                    self._check(rc)
                if rc == 1:
                    return
            else:
                time.sleep(2e-5)                    # (still queued behind an earlier job of its lane)
        # the leader's job ended without a dominant kernel (an empty trajectory, an error): plain stream order
        with self._lock:
            self._check(min(0, self._lib.amof_ctx_follow(self._h, lead._h, 0, 0.0)))

    @_locked
    def set_stream(self, stream_ptr):
        self._check(self._lib.amof_ctx_set_stream(self._h, ctypes.c_void_p(stream_ptr or None)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        self.drain()
        with self._lock:
            self._check(self._lib.amof_ctx_synchronize(self._h))

    def debug_poison(self, byte=0xA5):
        """Test hook: overwrite every scratch buffer the context owns (no call may rely on a previous call's scratch)."""
        self.drain()
        with self._lock:
            self._check(self._lib.amof_ctx_debug_poison(self._h, int(byte)))

    @_locked
    def wait_stream(self, stream_ptr):
        """Order this context's stream after the work queued so far on another HIP stream."""
        self._check(self._lib.amof_ctx_wait_stream(self._h, ctypes.c_void_p(stream_ptr or None)))

    def _torch_stream(self):
        """the torch stream whose queued work this call must run after: the current stream of the calling thread -- in a
        lane job, the current stream of the thread that SUBMITTED the job (and the lane thread's own, on which the job
        may have zeroed its output tensors)"""
        import torch
        own = torch.cuda.current_stream(self.device).cuda_stream
        producer = getattr(_tls, "producer_stream", None)
        if producer is not None and producer != own:
            self.wait_stream(producer)
        return own

    def _order_after_torch(self):
        """device outputs (``out=`` tensors) were zeroed / last written by torch: run after that.  The "_dev" entry
        points are synchronous like all others, so torch work queued afterwards needs no further ordering."""
        self.wait_stream(self._torch_stream())

    def _traj(self, packed, frame_range=None):
        """``_TrajHandle`` of a trajectory this context may read.  A device-resident ``pos`` must live on this
        context's GPU, and the context's (non-blocking) stream is ordered after torch's current stream on that
        device, i.e. after whatever produced the tensor (generation kernels, a peer copy, ``.to()``)."""
        th = _TrajHandle(packed, frame_range)
        if th.device_index is not None:
            if th.device_index != self.device:
                raise ValueError("trajectory positions live on cuda:%d but this context drives cuda:%d; use "
                                 "device=%d or copy the trajectory" % (th.device_index, self.device, th.device_index))
            self.wait_stream(self._torch_stream())
        return th

    def job_stats(self):
        """the library's record of the call that just returned on this thread"""
        with self._lock:
            return {"kernel_s_all": self._lib.amof_last_kernel_seconds(self._h, 0),
                    "kernel_s_dominant": self._lib.amof_last_kernel_seconds(self._h, 1),
                    "kernel_launches": self._lib.amof_last_kernel_launches(self._h),
                    "path": self._lib.amof_last_path(self._h).decode()}

    # (diagnostics describe the last COMPLETED call: they wait for the lane first)
    def last_kernel_seconds(self, dominant=True):
        self.drain()
        with self._lock:
            return self._lib.amof_last_kernel_seconds(self._h, 1 if dominant else 0)

    def last_path(self):
        """Kernel family that produced the last result ("rdf_tile", "rdf_cell", "cn_fast", "msd_comb", ...)."""
        self.drain()
        with self._lock:
            return self._lib.amof_last_path(self._h).decode()

    def last_kernel_launches(self):
        self.drain()
        with self._lock:
            return self._lib.amof_last_kernel_launches(self._h)

    # ------------------------------------------------------------ analyses --
    @_locked
    def rdf_accumulate(self, packed, rmax, nbins, frame_range=None, out=None):
        """ordered-pair histograms ``[S][S][nbins]`` (u64) and the volume sum.

        ``out``: optional torch CUDA int64 tensor ``[S][S][nbins]`` to
        accumulate into on the device (stays resident for an RCCL merge)."""
        th = self._traj(packed, frame_range)
        vol = ctypes.c_double(0.0)
        if out is not None:
            assert out.is_cuda and out.is_contiguous() and out.numel() == th.S * th.S * nbins
            assert out.device.index == self.device and out.element_size() == 8
            self._order_after_torch()
            rc = self._lib.amof_rdf_accumulate_dev(self._h, ctypes.byref(th.c), float(rmax), int(nbins),
                                                   ctypes.c_void_p(out.data_ptr()), ctypes.byref(vol))
            self._check(rc)
            return out, vol.value, th.kinds
        hist = np.zeros((th.S, th.S, nbins), dtype=np.uint64)
        rc = self._lib.amof_rdf_accumulate(self._h, ctypes.byref(th.c), float(rmax), int(nbins),
                                           ctypes.c_void_p(hist.ctypes.data), ctypes.byref(vol))
        self._check(rc)
        return hist, vol.value, th.kinds

    @_locked
    def cn_count(self, packed, cutoff, sets, frame_range=None, per_atom=False):
        th = self._traj(packed, frame_range)
        cutoff = np.ascontiguousarray(cutoff, dtype=np.float64).reshape(th.S, th.S)
        sets = np.ascontiguousarray(sets, dtype=np.int32).reshape(-1, 2)
        sums = np.zeros((th.n_frames, len(sets)), dtype=np.int64)
        pa = np.zeros((th.n_frames, len(sets), th.n_atoms), dtype=np.int32) if per_atom else None
        rc = self._lib.amof_cn_count(self._h, ctypes.byref(th.c), ctypes.c_void_p(cutoff.ctypes.data),
                                     ctypes.c_void_p(sets.ctypes.data), len(sets),
                                     ctypes.c_void_p(sums.ctypes.data),
                                     ctypes.c_void_p(pa.ctypes.data) if per_atom else None)
        self._check(rc)
        return (sums, pa) if per_atom else sums

    @_locked
    def bad_hist(self, packed, cutoff, triples, edges, frame_range=None, out=None):
        th = self._traj(packed, frame_range)
        cutoff = np.ascontiguousarray(cutoff, dtype=np.float64).reshape(th.S, th.S)
        triples = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 2)
        edges = np.ascontiguousarray(edges, dtype=np.float64)
        nb = len(edges) - 1
        if out is not None:
            hist_t, nang_t = out
            for x, n in ((hist_t, len(triples) * nb), (nang_t, len(triples))):
                assert x.is_cuda and x.is_contiguous() and x.numel() == n and x.element_size() == 8
                assert x.device.index == self.device
            self._order_after_torch()
            rc = self._lib.amof_bad_hist_dev(self._h, ctypes.byref(th.c), ctypes.c_void_p(cutoff.ctypes.data),
                                             ctypes.c_void_p(triples.ctypes.data), len(triples),
                                             ctypes.c_void_p(edges.ctypes.data), nb,
                                             ctypes.c_void_p(hist_t.data_ptr()), ctypes.c_void_p(nang_t.data_ptr()))
            self._check(rc)
            return hist_t, nang_t
        hist = np.zeros((len(triples), nb), dtype=np.uint64)
        nang = np.zeros(len(triples), dtype=np.uint64)
        rc = self._lib.amof_bad_hist(self._h, ctypes.byref(th.c), ctypes.c_void_p(cutoff.ctypes.data),
                                     ctypes.c_void_p(triples.ctypes.data), len(triples),
                                     ctypes.c_void_p(edges.ctypes.data), nb,
                                     ctypes.c_void_p(hist.ctypes.data), ctypes.c_void_p(nang.ctypes.data))
        self._check(rc)
        return hist, nang

    @_locked
    def bad_hist_by_cn(self, packed, cutoff, triples, edges, cn_max=16, frame_range=None):
        """``(hist u64 [T][cn_max+1][nb], n_angles u64 [T][cn_max+1])`` keyed by neighbour count."""
        th = self._traj(packed, frame_range)
        cutoff = np.ascontiguousarray(cutoff, dtype=np.float64).reshape(th.S, th.S)
        triples = np.ascontiguousarray(triples, dtype=np.int32).reshape(-1, 2)
        edges = np.ascontiguousarray(edges, dtype=np.float64)
        nb = len(edges) - 1
        hist = np.zeros((len(triples), cn_max + 1, nb), dtype=np.uint64)
        nang = np.zeros((len(triples), cn_max + 1), dtype=np.uint64)
        rc = self._lib.amof_bad_hist_by_cn(self._h, ctypes.byref(th.c), ctypes.c_void_p(cutoff.ctypes.data),
                                           ctypes.c_void_p(triples.ctypes.data), len(triples),
                                           ctypes.c_void_p(edges.ctypes.data), nb, int(cn_max),
                                           ctypes.c_void_p(hist.ctypes.data), ctypes.c_void_p(nang.ctypes.data))
        self._check(rc)
        return hist, nang

    @_locked
    def msd_com(self, packed, frame_range, out):
        """centre of mass of frames ``[f0, f1)`` into rows ``f0 .. f1`` of ``out`` (torch CUDA f64 ``[F][3]``); the other
        rows are left alone (an atom-sharded run: every rank its frame share, then one sum of the zero-filled tables)."""
        th = self._traj(packed)
        assert out.is_cuda and out.is_contiguous() and out.numel() == 3 * th.n_frames and out.element_size() == 8
        assert out.device.index == self.device
        self._order_after_torch()
        self._check(self._lib.amof_msd_com_dev(self._h, ctypes.byref(th.c), int(frame_range[0]), int(frame_range[1]),
                                               ctypes.c_void_p(out.data_ptr())))
        return out

    @_locked
    def msd_window(self, packed, windows, unwrap=False, remove_com=True, atom_range=None, com=None, out=None):
        """``(sumsq [S][W] f64, kinds)``: raw sums of squared displacements.

        ``out``: optional torch CUDA f64 tensor ``[S][W]`` the sums are ADDED into on the device (stays resident for
        the ranks' all-reduce); ``com``: optional torch CUDA f64 ``[F][3]`` precomputed centre of mass (``msd_com``)."""
        th = self._traj(packed)
        windows = np.ascontiguousarray(windows, dtype=np.int32)
        a0, a1 = (0, th.n_atoms) if atom_range is None else atom_range
        if out is not None or com is not None:
            import torch
            if out is None:
                out = torch.zeros((th.S, len(windows)), dtype=torch.float64, device=torch.device("cuda", self.device))
            assert out.is_cuda and out.is_contiguous() and out.numel() == th.S * len(windows) and out.element_size() == 8
            assert out.device.index == self.device
            if com is not None:
                assert com.is_cuda and com.is_contiguous() and com.numel() == 3 * th.n_frames and com.element_size() == 8
                assert com.device.index == self.device
            self._order_after_torch()
            rc = self._lib.amof_msd_window_dev(self._h, ctypes.byref(th.c), ctypes.c_void_p(windows.ctypes.data),
                                               len(windows), 1 if unwrap else 0, 1 if remove_com else 0, int(a0), int(a1),
                                               ctypes.c_void_p(com.data_ptr()) if com is not None else None,
                                               ctypes.c_void_p(out.data_ptr()))
            self._check(rc)
            return out, th.kinds
        out = np.zeros((th.S, len(windows)), dtype=np.float64)
        rc = self._lib.amof_msd_window(self._h, ctypes.byref(th.c), ctypes.c_void_p(windows.ctypes.data),
                                       len(windows), 1 if unwrap else 0, 1 if remove_com else 0,
                                       int(a0), int(a1), ctypes.c_void_p(out.ctypes.data))
        self._check(rc)
        return out, th.kinds


    @_locked
    def msd_shard_begin(self, packed, windows, atom_range, csum):
        """first half of an atom-sharded window MSD (``amof_msd_shard_begin``): ``csum`` (torch CUDA f64 ``[F][3]``) receives
        the mass-weighted coordinate sums of the atoms ``[a0, a1)`` per frame -- the caller all-reduces it over the ranks.
        Raises :class:`Unsupported` where the fused form does not apply (the caller then takes ``msd_com`` + ``msd_window``)."""
        th = self._traj(packed)
        windows = np.ascontiguousarray(windows, dtype=np.int32)
        assert csum.is_cuda and csum.is_contiguous() and csum.numel() == 3 * th.n_frames and csum.element_size() == 8
        assert csum.device.index == self.device
        self._order_after_torch()
        self._check(self._lib.amof_msd_shard_begin(self._h, ctypes.byref(th.c), ctypes.c_void_p(windows.ctypes.data), len(windows),
                                                   int(atom_range[0]), int(atom_range[1]), ctypes.c_void_p(csum.data_ptr())))
        return csum

    @_locked
    def msd_shard_finish(self, packed, windows, atom_range, csum, out):
        """second half (``amof_msd_shard_finish``, the next call on this context after its begin): adds the sums of the
        atoms ``[a0, a1)`` into ``out`` (torch CUDA f64 ``[S][W]``); ``csum`` = the table summed over the ranks"""
        th = self._traj(packed)
        windows = np.ascontiguousarray(windows, dtype=np.int32)
        assert csum.is_cuda and csum.is_contiguous() and csum.numel() == 3 * th.n_frames and csum.element_size() == 8
        assert out.is_cuda and out.is_contiguous() and out.numel() == th.S * len(windows) and out.element_size() == 8
        assert csum.device.index == self.device and out.device.index == self.device
        self._order_after_torch()
        self._check(self._lib.amof_msd_shard_finish(self._h, ctypes.byref(th.c), ctypes.c_void_p(windows.ctypes.data), len(windows),
                                                    int(atom_range[0]), int(atom_range[1]), ctypes.c_void_p(csum.data_ptr()),
                                                    ctypes.c_void_p(out.data_ptr())))
        return out, th.kinds

    @_locked
    def msd_direct(self, packed):
        """``(msd [F][S+1] f64, kinds)``: column 0 = all atoms, then one per species."""
        th = self._traj(packed)
        out = np.zeros((th.n_frames, th.S + 1), dtype=np.float64)
        self._check(self._lib.amof_msd_direct(self._h, ctypes.byref(th.c), ctypes.c_void_p(out.ctypes.data)))
        return out, th.kinds


class MultiContext(object):
    """Several GPUs from ONE process: a :class:`Context` per device, a thread per context (ctypes
    drops the GIL during the calls).  Frames are sharded for RDF / CN / BAD, atoms for MSD; integer
    results are summed on the host, so they equal the single-GPU ones bit for bit.  The usual
    multi-GPU route stays one process per GPU with torch.distributed (amof_amd.dist); this is the
    convenience for scripts that are not launched with torchrun.  A device may be listed twice
    (two contexts with their own streams on one GPU)."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("empty device list")
        self.ctxs = [Context(d) for d in self.devices]
        self.device = self.devices[0]

    def close(self):
        for c in self.ctxs:
            c.close()

    def _shards(self, lo, hi):
        from . import dist as _d
        n = len(self.ctxs)
        return [tuple(x + lo for x in _d.shard_range(hi - lo, k, n)) for k in range(n)]

    @staticmethod
    def _for_device(packed, ctx, frame_range=None):
        """The trajectory as device ``ctx.device`` can read it: host arrays as they are, a CUDA tensor on
        another device copied over (the needed frames only)."""
        if not packed.on_device or packed.pos.device.index == ctx.device:
            return packed, frame_range
        import torch
        f0, f1 = (0, packed.n_frames) if frame_range is None else frame_range
        pos = packed.pos[f0:f1].to(torch.device("cuda", ctx.device))
        cell = packed.cell if packed.cell.shape[0] == 1 else packed.cell[f0:f1]
        return PackedTrajectory(pos, cell, packed.numbers, packed.masses, packed.pbc), (0, f1 - f0)

    def _run(self, jobs):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(jobs)) as ex:
            return [f.result() for f in [ex.submit(j) for j in jobs]]

    def rdf_accumulate(self, packed, rmax, nbins, frame_range=None, out=None):
        assert out is None, "device-resident accumulation is a single-context feature"
        lo, hi = (0, packed.n_frames) if frame_range is None else frame_range
        jobs = []
        for ctx, (a, b) in zip(self.ctxs, self._shards(lo, hi)):
            def job(ctx=ctx, a=a, b=b):
                tr, fr = self._for_device(packed, ctx, (a, b))
                return ctx.rdf_accumulate(tr, rmax, nbins, frame_range=fr)
            jobs.append(job)
        res = self._run(jobs)
        return sum(r[0] for r in res), float(sum(r[1] for r in res)), res[0][2]

    def cn_count(self, packed, cutoff, sets, frame_range=None, per_atom=False):
        lo, hi = (0, packed.n_frames) if frame_range is None else frame_range
        jobs = []
        for ctx, (a, b) in zip(self.ctxs, self._shards(lo, hi)):
            def job(ctx=ctx, a=a, b=b):
                tr, fr = self._for_device(packed, ctx, (a, b))
                return ctx.cn_count(tr, cutoff, sets, frame_range=fr, per_atom=per_atom)
            jobs.append(job)
        res = self._run(jobs)
        if per_atom:
            return np.concatenate([r[0] for r in res], axis=0), np.concatenate([r[1] for r in res], axis=0)
        return np.concatenate(res, axis=0)

    def bad_hist(self, packed, cutoff, triples, edges, frame_range=None, out=None):
        assert out is None, "device-resident accumulation is a single-context feature"
        lo, hi = (0, packed.n_frames) if frame_range is None else frame_range
        jobs = []
        for ctx, (a, b) in zip(self.ctxs, self._shards(lo, hi)):
            def job(ctx=ctx, a=a, b=b):
                tr, fr = self._for_device(packed, ctx, (a, b))
                return ctx.bad_hist(tr, cutoff, triples, edges, frame_range=fr)
            jobs.append(job)
        res = self._run(jobs)
        return sum(r[0] for r in res), sum(r[1] for r in res)

    def bad_hist_by_cn(self, packed, cutoff, triples, edges, cn_max=16, frame_range=None):
        lo, hi = (0, packed.n_frames) if frame_range is None else frame_range
        jobs = []
        for ctx, (a, b) in zip(self.ctxs, self._shards(lo, hi)):
            def job(ctx=ctx, a=a, b=b):
                tr, fr = self._for_device(packed, ctx, (a, b))
                return ctx.bad_hist_by_cn(tr, cutoff, triples, edges, cn_max=cn_max, frame_range=fr)
            jobs.append(job)
        res = self._run(jobs)
        return sum(r[0] for r in res), sum(r[1] for r in res)

    def msd_window(self, packed, windows, unwrap=False, remove_com=True, atom_range=None):
        lo, hi = (0, packed.n_atoms) if atom_range is None else atom_range
        jobs = []
        for ctx, (a, b) in zip(self.ctxs, self._shards(lo, hi)):
            def job(ctx=ctx, a=a, b=b):
                tr, _ = self._for_device(packed, ctx)
                return ctx.msd_window(tr, windows, unwrap=unwrap, remove_com=remove_com, atom_range=(a, b))
            jobs.append(job)
        res = self._run(jobs)
        return sum(r[0] for r in res), res[0][1]

    def msd_direct(self, packed):
        return self.ctxs[0].msd_direct(self._for_device(packed, self.ctxs[0])[0])

    def last_kernel_seconds(self, dominant=False):
        return max(c.last_kernel_seconds(dominant) for c in self.ctxs)

    def last_kernel_launches(self):
        return sum(c.last_kernel_launches() for c in self.ctxs)

    def last_path(self):
        return self.ctxs[0].last_path()

    def synchronize(self):
        for c in self.ctxs:
            c.synchronize()


_contexts = {}


def get_context(device=None, lane=0):
    """Cached :class:`Context` of a device (default: LOCAL_RANK or 0); a list / tuple of devices gives
    a cached :class:`MultiContext` (several GPUs driven from this process).  ``lane=1``: the device's second context,
    on a stream of the highest priority (the memory-bound analyses of the classes run there)."""
    if isinstance(device, (list, tuple)):
        key = tuple(int(d) for d in device)
        with _ctx_lock:
            ctx = _contexts.get(key)
            if ctx is None:
                ctx = MultiContext(key)
                _contexts[key] = ctx
            return ctx
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
        n = device_count()
        if n > 0 and device >= n:
            raise RuntimeError("LOCAL_RANK=%d but only %d GPU(s) are visible: refusing to pile every rank onto "
                               "cuda:0 (pass device= explicitly to share a GPU on purpose)" % (device, n))
    key = device if not lane else (device, "lane", int(lane))
    with _ctx_lock:
        ctx = _contexts.get(key)
        if ctx is None:
            ctx = Context(device, priority=os.environ.get("AMOF_LANE1_PRIORITY", "high") if lane else "normal")
            _contexts[key] = ctx
            if lane:
                lead = _contexts.get(device)
                if lead is None:
                    lead = _contexts[device] = Context(device, priority="normal")
                ctx._follows = lead
        return ctx


def default_device():
    """the GPU an analysis goes to when the caller names none: LOCAL_RANK (one process per GPU) or 0"""
    return int(os.environ.get("LOCAL_RANK", "0"))


def lane_context(device, lane):
    """the context an analysis class runs on: ``lane`` 1 (memory-bound analyses) only while the classes run
    asynchronously (amof_amd/_lazy.py); a device list stays one MultiContext"""
    from . import _lazy
    if isinstance(device, (list, tuple)) or not lane or not _lazy.async_enabled():
        return get_context(device)
    return get_context(device, lane=lane)
