"""The `data` attribute of the analysis classes, and results that arrive later.

1. The reference's constructors start every object with an empty DataFrame holding the first column (`amof/rdf.py:33-35,144-146`,
   `amof/msd.py:64-66,152-154`, `amof/bad.py:66-68`, `amof/cn.py:30-32`).  Building it costs ~0.1 ms of pandas per object --
   4 % of one rank's 10 ms step in an 8-GPU run (profiles/r04/shards.txt) -- for a frame that `compute_*` replaces at once,
   so it is built when somebody looks at it before a result has been stored: same object for every reader afterwards.

2. Round 5: `from_trajectory` of `Rdf`, `WindowMsd`, `Bad` and `cn.CoordinationNumber` ENQUEUES the analysis on a lane of
   its device (`amof_amd._hip.Context.submit`: one worker thread per context; lane 0 = RDF, lane 1 = the memory-bound
   analyses on a stream of the highest priority) and returns; the object is what the reference returns -- its `.data` is the
   result -- and the first look at `.data` (or at any attribute the computation sets: `.hist`, `.rmax`, `.sumsq`, ...)
   waits for it.  The reference runs its analyses as independent calls (`examples/Compute structural properties.py:58-118`)
   and parallelises inside them with joblib processes (`amof/msd.py:252-256`, `amof/bad.py:151-152`, `amof/cn.py:79-80`);
   here the 1.2 ms of MSD, the 0.8 ms of BAD and the pandas assembly of every class run beside the 72 ms RDF launch.
   What a call raises from argument checks it still raises at once; what the GPU library reports (a singular cell, an
   undefined angle: the SAME exception types as before) is raised by the first access, and by every later one.
   `AMOF_ASYNC=0` restores strictly synchronous constructors.  Collectives of a multi-rank run are never issued from a
   lane: the lane runs the rank's local kernels, the first access finishes -- all-reduce, DataFrame -- in the calling
   thread, so every rank issues its collectives in program order.  Looking at a result of the device's FIRST lane (the
   RDF: the long one) first finishes the pending results of its other lanes, in the order they were requested: their
   kernels are done long before the RDF's, so their merges and DataFrames are assembled while the RDF launch still runs
   instead of one after the other behind it (one rank of eight: RDF + MSD + BAD 10.3 -> 9.9 ms); the order is a function
   of the program alone, the same on every rank.
"""
import os
import threading

import numpy as np
import pandas as pd

import weakref

_tls = threading.local()        # .depth > 0: inside a lane job or a finishing step (attribute access must not wait on itself)
_merging = []                   # weak references to the results whose finishing step runs in the calling thread, oldest first
_merging_lock = threading.Lock()


def async_enabled():
    return os.environ.get("AMOF_ASYNC", "1") != "0"


class _Failed(object):
    """a finishing step that raised: every later access raises the same exception (the step is not run twice -- it may
    have issued collectives)"""

    def __init__(self, exc):
        self.exc = exc

    def result(self):
        raise self.exc


class Deferred(object):
    """Mixin of the analysis classes: `_defer(ctx, local, finish, collective)` runs `finish(local())` -- inline
    (`AMOF_ASYNC=0`, a MultiContext), wholly on the context's lane, or (`collective`: finish issues torch.distributed
    calls) `local` on the lane and `finish` in the thread that first looks at a result."""

    def _defer(self, ctx, local, finish, collective=False):
        d = self.__dict__
        d["_ctx"] = ctx         # (the context that computes this result)
        inner = local

        def local():
            # the library's own record of the call (kernel seconds from HIP events on its stream, kernel family), taken
            # before the lane's next job overwrites it: `_stats` of the result object
            raw = inner()
            stats = getattr(ctx, "job_stats", None)
            if stats is not None:
                d["_stats"] = stats()
            return raw
        if not async_enabled() or not hasattr(ctx, "submit"):
            finish(local())
            return
        d["_wait_lock"] = threading.Lock()

        def scoped(fn):
            def run():
                with in_job():
                    return fn()
            return run
        if collective:
            d["_pending"] = (ctx.submit(scoped(local)), finish)
            with _merging_lock:
                _merging[:] = [r for r in _merging if r() is not None and r().__dict__.get("_pending") is not None]
                _merging.append(weakref.ref(self))
        else:
            d["_pending"] = (ctx.submit(scoped(lambda: finish(local()))), None)

    def _wait(self):
        d = self.__dict__
        if d.get("_pending") is None or getattr(_tls, "depth", 0):
            return
        if d["_pending"][1] is not None and getattr(d.get("_ctx"), "_follows", None) is None:
            self._finish_followers_first()
        with d["_wait_lock"]:
            p = d.get("_pending")
            if p is None:
                return
            fut, finish = p
            raw = fut.result()          # (raises what the job raised -- at every access: _pending stays)
            if finish is not None:
                _tls.depth = getattr(_tls, "depth", 0) + 1
                try:
                    finish(raw)
                except BaseException as exc:
                    d["_pending"] = (_Failed(exc), None)
                    raise
                finally:
                    _tls.depth -= 1
            d["_pending"] = None

    def _finish_followers_first(self):
        """this result belongs to a device's first lane and is finished (merged) in the calling thread: the pending results
        of the lanes that follow it come first, oldest first -- what they raise stays with them, for their own readers"""
        lead = self.__dict__.get("_ctx")
        with _merging_lock:
            others = [r() for r in _merging]
        for other in others:
            if other is None or other is self or getattr(other.__dict__.get("_ctx"), "_follows", None) is not lead:
                continue
            try:
                other._wait()
            except BaseException:       # noqa: B036  (kept by the object: its next reader gets it)
                pass

    def result(self):
        """wait for the analysis (raising what it raised) and return the object"""
        self._wait()
        return self

    def __getattr__(self, name):
        # only reached when normal lookup fails: an attribute the pending computation has yet to set
        d = self.__dict__
        if name.startswith("__") or d.get("_pending") is None or getattr(_tls, "depth", 0):
            raise AttributeError("%r object has no attribute %r" % (type(self).__name__, name))
        self._wait()
        return object.__getattribute__(self, name)

    def __getstate__(self):
        self._wait()
        return {k: v for k, v in self.__dict__.items() if k not in ("_pending", "_wait_lock", "_ctx")}

    def __setstate__(self, state):
        self.__dict__.update(state)


def in_job():
    """context manager used by the lanes' jobs: attribute access of the object under construction does not wait"""
    class _Scope(object):
        def __enter__(self_inner):
            _tls.depth = getattr(_tls, "depth", 0) + 1

        def __exit__(self_inner, *exc):
            _tls.depth -= 1
    return _Scope()


class EmptyUntilComputed:
    def __init__(self, column):
        self.column = column

    def __set_name__(self, owner, name):
        self.slot = "_" + name
        self.name = name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        d = obj.__dict__
        if d.get("_pending") is not None:
            obj._wait()
        val = d.get(self.slot)
        if val is None:
            val = d.get(self.name)      # (objects pickled before the descriptor existed carry the plain key)
        if val is None:
            val = pd.DataFrame({self.column: np.empty([0])})
            d[self.slot] = val
        return val

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = value
