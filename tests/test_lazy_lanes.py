"""The asynchronous constructors (amof_amd/_lazy.py, the lanes of amof_amd/_hip.py) on a machine without a GPU:
the classes run on two oracle-backed stand-in lanes (tests/oracle_context.py -- test infrastructure), so what is
checked here is the host machinery: enqueue-and-return, wait at first access, the same exceptions at access (again
at every later access), pickling, AMOF_ASYNC=0, and results identical to the synchronous path."""
import pickle
import threading
import time

import numpy as np
import pandas as pd
import pytest

from amof_amd import _hip, _lazy
from amof_amd.bad import Bad
from amof_amd.cn import CoordinationNumber
from amof_amd.msd import WindowMsd
from amof_amd.rdf import Rdf
from tests import helpers as H
from tests import oracle_context


@pytest.fixture()
def lanes(monkeypatch):
    ls = oracle_context.install(monkeypatch)
    yield ls
    for ctx in ls.values():
        ctx.close_lane()


@pytest.fixture(scope="module")
def traj():
    return H.random_walk(H.zif4_frame(), 9, 0.05, 5)


def _four(traj):
    return (Rdf.from_trajectory(traj, dr=0.05, rmax=6.0, distributed=False),
            WindowMsd.from_trajectory(traj, delta_time=2, timestep=1, distributed=False),
            Bad.from_trajectory(traj, {'Zn-N': 2.5}, dtheta=0.5, distributed=False),
            CoordinationNumber.from_trajectory(traj, {'Zn-N': 2.5}, distributed=False))


def test_async_results_equal_the_synchronous_ones(lanes, traj, monkeypatch):
    monkeypatch.setenv("AMOF_ASYNC", "0")
    sync = _four(traj)
    assert all(o.__dict__.get("_pending") is None for o in sync)
    assert lanes[1].calls == [] and lanes[0].calls == ["rdf", "msd", "bad", "cn"]     # one context, in call order
    monkeypatch.setenv("AMOF_ASYNC", "1")
    lanes[0].calls.clear()
    # hold lane 0 back: the constructors must return with the RDF still queued
    gate = threading.Event()
    lanes[0].submit(gate.wait)
    t0 = time.perf_counter()
    objs = _four(traj)
    assert time.perf_counter() - t0 < 5.0
    assert objs[0].__dict__["_pending"] is not None and not objs[0].__dict__["_pending"][0].done()
    assert objs[0]._ctx is lanes[0] and all(o._ctx is lanes[1] for o in objs[1:])
    gate.set()
    for a, b in zip(objs, sync):
        assert a.data.equals(b.data)                      # bit for bit
        assert a.__dict__["_pending"] is None
    assert np.array_equal(objs[0].hist, sync[0].hist) and objs[0].rmax == sync[0].rmax
    assert np.array_equal(objs[1].sumsq, sync[1].sumsq)
    assert np.array_equal(objs[2].hist, sync[2].hist) and np.array_equal(objs[2].n_angles, sync[2].n_angles)
    assert lanes[0].calls == ["rdf"] and lanes[1].calls == ["msd", "bad", "cn"]
    assert objs[1]._stats["path"] == "msd"


def test_attributes_the_computation_sets_wait_for_it(lanes, traj):
    gate = threading.Event()
    lanes[0].submit(gate.wait)
    rdf = Rdf.from_trajectory(traj, dr=0.05, rmax=6.0, distributed=False)
    assert "hist" not in rdf.__dict__
    threading.Timer(0.2, gate.set).start()
    assert rdf.hist.shape == (4, 4, int(6.0 // 0.05))                  # (waited for the gate and the job)
    with pytest.raises(AttributeError):
        rdf.no_such_attribute
    # an object read back from a file has nothing pending: plain attribute errors, the stored frame
    blank = Rdf()
    with pytest.raises(AttributeError):
        blank.hist
    assert list(blank.data.columns) == ["r"] and len(blank.data) == 0


def test_errors_of_the_library_surface_at_access_every_time(lanes, traj, monkeypatch):
    def boom(*a, **k):
        raise _hip.AmofError(_hip.AMOF_ESINGULAR, "cell of frame 0 is singular")
    monkeypatch.setattr(lanes[0], "rdf_accumulate", boom)
    rdf = Rdf.from_trajectory(traj, dr=0.05, rmax=6.0, distributed=False)       # returns: nothing has been looked at
    for _ in range(2):
        with pytest.raises(_hip.AmofError) as e:
            rdf.data
        assert e.value.code == _hip.AMOF_ESINGULAR
    with pytest.raises(_hip.AmofError):
        rdf.hist
    with pytest.raises(_hip.AmofError):
        rdf.result()

    def zero(*a, **k):
        raise ZeroDivisionError("Undefined angle")
    monkeypatch.setattr(lanes[1], "bad_hist", zero)
    bad = Bad.from_trajectory(traj, {'Zn-N': 2.5}, dtheta=0.5, distributed=False)
    with pytest.raises(ZeroDivisionError):
        bad.data
    # argument errors are raised by the call itself, as in the reference
    with pytest.raises(ValueError):
        Rdf.from_trajectory(traj, dr=10.0, rmax=6.0, distributed=False)
    with pytest.raises(ValueError):
        Rdf.from_trajectory([], distributed=False)


def test_pickle_waits_and_old_pickles_keep_their_frame(lanes, traj):
    msd = WindowMsd.from_trajectory(traj, delta_time=2, timestep=1, distributed=False)
    again = pickle.loads(pickle.dumps(msd))               # (waits; futures and locks are not pickled)
    assert again.data.equals(msd.data) and np.array_equal(again.sumsq, msd.sumsq)
    assert "_pending" not in again.__dict__ and "_ctx" not in again.__dict__
    # an object pickled before `data` became a descriptor carries the plain key
    old = Rdf.__new__(Rdf)
    old.__dict__["data"] = pd.DataFrame({"r": [0.0, 0.1], "X-X": [0.0, 1.0]})
    back = pickle.loads(pickle.dumps(old))
    assert list(back.data.columns) == ["r", "X-X"] and len(back.data) == 2


def test_a_finishing_step_that_fails_is_not_run_twice():
    class Obj(_lazy.Deferred):
        pass
    lane = oracle_context.OracleContext("t")
    runs = []

    def finish(raw):
        runs.append(raw)
        raise RuntimeError("merge failed")
    o = Obj()
    o._defer(lane, lambda: 7, finish, collective=True)
    for _ in range(3):
        with pytest.raises(RuntimeError, match="merge failed"):
            o.result()
    assert runs == [7]
    lane.close_lane()


def test_lane_runs_jobs_in_order_and_drain_waits():
    lane = oracle_context.OracleContext("t2")
    seen = []
    futs = [lane.submit(lambda k=k: (time.sleep(0.01), seen.append(k))) for k in range(5)]
    lane.drain()
    assert seen == [0, 1, 2, 3, 4] and all(f.done() for f in futs)
    lane.submit(lambda: lane.drain())      # draining from the lane's own thread must not wait on itself
    lane.drain()
    lane.close_lane()


def test_a_following_lane_without_a_gpu_context_runs_at_once():
    """Lane.follow_leader is a GPU matter (amof_amd/_hip.py Context.follow_leader): the stand-in lanes of the CPU suite have
    no leader, count their jobs all the same, and run a job as soon as it is submitted"""
    from tests.oracle_context import OracleContext
    lead, second = OracleContext("lead"), OracleContext("second")
    assert second._follows is None and second.device_calls() == 0
    import threading
    gate = threading.Event()
    f0 = lead.submit(lambda: gate.wait(5) and "lead done")
    f1 = second.submit(lambda: "second done")
    assert f1.result(timeout=5) == "second done" and not f0.done()
    gate.set()
    assert f0.result(timeout=5) == "lead done"
    assert (lead._jobs_submitted, lead._jobs_started, lead._jobs_finished) == (1, 1, 1)
    lead.close_lane()
    second.close_lane()


def test_reading_the_first_lanes_result_finishes_the_other_lanes_first(monkeypatch):
    """results merged in the calling thread (multi-rank runs): a look at the first lane's result first finishes the pending
    results of the lanes that follow it, oldest first; what one of them raises stays with it"""
    from amof_amd._lazy import Deferred
    from tests.oracle_context import OracleContext
    monkeypatch.setenv("AMOF_ASYNC", "1")
    lead, second = OracleContext("lead"), OracleContext("second")
    second._follows = lead

    class Result(Deferred):
        pass
    order = []
    a, b, c, d = Result(), Result(), Result(), Result()
    b._defer(second, lambda: "b", lambda raw: order.append(raw), collective=True)
    a._defer(lead, lambda: "a", lambda raw: order.append(raw), collective=True)

    def boom(raw):
        order.append(raw)
        raise ValueError("c failed")
    c._defer(second, lambda: "c", boom, collective=True)
    d._defer(second, lambda: order.append("d (wholly on its lane)"), lambda raw: None, collective=False)
    a.result()
    assert [x for x in order if len(x) == 1] == ["b", "c", "a"]
    assert b.__dict__["_pending"] is None and a.__dict__["_pending"] is None
    for _ in range(2):
        with pytest.raises(ValueError):
            c.result()
    assert [x for x in order if len(x) == 1] == ["b", "c", "a"]        # (the failed step is not run again)
    # a follower's own reader finishes nothing else
    e, f = Result(), Result()
    e._defer(lead, lambda: "e", lambda raw: order.append(raw), collective=True)
    f._defer(second, lambda: "f", lambda raw: order.append(raw), collective=True)
    f.result()
    assert e.__dict__["_pending"] is not None and order[-1] == "f"
    e.result()
    lead.close_lane()
    second.close_lane()
