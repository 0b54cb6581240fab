"""N > 1 path on CPU: world_size-2 gloo.  Each rank histograms its shard of
frames with the oracle (stand-in for the GPU kernel, which needs a GPU), the
product's dist helpers merge; the result must equal the single-process one."""

import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.conftest import ROOT


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from amof_amd import dist as adist
    from oracle import clib
    from tests import helpers as H
    packed = H.random_walk(H.zif4_frame(), 7, 0.05, 21, cell_jitter=0.01)
    kinds, sp = H.species_of(packed.numbers)
    assert adist.world() == (rank, world)
    assert adist.merging(world) and not adist.device_collectives()      # gloo: results travel through the host
    lo, hi = adist.shard_range(packed.n_frames, rank, world)
    rmax = adist.all_reduce_min(7.0 + rank)          # MIN over ranks -> 7.0
    hist, vol = clib.rdf_hist(packed.pos[lo:hi], packed.cell[lo:hi], sp, len(kinds), rmax, 700)
    merged = adist.all_reduce_sum(hist)
    tot = adist.all_reduce_sum(np.array([vol, float(hi - lo)]))
    rows = adist.all_gather_rows(np.arange(lo, hi, dtype=np.int64)[:, None])
    msd_part = adist.all_reduce_sum(np.full((2, 3), float(rank + 1)))
    if rank == 0:
        np.savez(os.path.join(out_dir, "merged.npz"), hist=merged, tot=tot, rows=rows, rmax=rmax, msd=msd_part)
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_histogram_merge_world2(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "merged.npz"))
    from oracle import clib
    from tests import helpers as H
    packed = H.random_walk(H.zif4_frame(), 7, 0.05, 21, cell_jitter=0.01)
    kinds, sp = H.species_of(packed.numbers)
    full, vol = clib.rdf_hist(packed.pos, packed.cell, sp, len(kinds), 7.0, 700)
    assert got["rmax"] == 7.0
    assert got["hist"].dtype == np.uint64 and np.array_equal(got["hist"], full)      # bit-exact 1 vs 2 ranks
    assert got["tot"][1] == 7 and got["tot"][0] == pytest.approx(vol, rel=1e-14)
    assert np.array_equal(got["rows"][:, 0], np.arange(7))
    assert (got["msd"] == 3.0).all()


def _worker_one(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from amof_amd import dist as adist
    assert not adist.merging(1)                                          # no group yet: nothing to merge
    os.environ["AMOF_DIST_FORCE_MERGE"] = "1"
    assert not adist.merging(1)                                          # the switch needs an initialised group
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert adist.merging(1)                                              # one rank, but every collective runs
    x = np.arange(6, dtype=np.uint64).reshape(2, 3)
    assert np.array_equal(adist.all_reduce_sum(x), x) and adist.all_reduce_sum(x).dtype == np.uint64
    assert np.array_equal(adist.all_gather_rows(np.arange(4, dtype=np.int64)[:, None])[:, 0], np.arange(4))
    assert adist.shard_range(7, 0, 1) == (0, 7)
    dist.destroy_process_group()
    with open(os.path.join(out_dir, "ok"), "w") as fh:
        fh.write("1")


def test_forced_merge_in_a_one_rank_group(tmp_path):
    """AMOF_DIST_FORCE_MERGE=1: how the RCCL path is exercised on a single-GPU box (tests/test_gpu_dist.py, bench.py)"""
    port = 27500 + os.getpid() % 2000
    mp.spawn(_worker_one, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(os.path.join(str(tmp_path), "ok"))


def _worker_classes(rank, world, port, out_dir):
    """the four CLASSES as rank `rank` of `world`: frames (RDF / BAD / CN) and atoms (MSD) sharded, asynchronous
    constructors (the rank's local work on the lanes, the collectives at first access in the calling thread), merged
    over gloo.  The GPU entry points are answered by the CPU oracle (tests/oracle_context.py: test infrastructure)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import helpers as H
    from tests import oracle_context
    lanes = oracle_context.install()
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    packed = H.random_walk(H.zif4_frame(), 9, 0.05, 21, cell_jitter=0.01)
    # every constructor returns before anything is merged; the results are looked at in ANOTHER order than they were
    # requested -- the same on every rank, which is all the collectives need
    rdf = Rdf.from_trajectory(packed, dr=0.05, rmax=6.0)
    msd = WindowMsd.from_trajectory(packed, delta_time=2, timestep=1)
    bad = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.5)
    cn = CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5})
    assert all(o.__dict__.get("_pending") is not None for o in (rdf, msd, bad, cn))
    frames = {"cn": cn.data, "msd": msd.data, "rdf": rdf.data, "bad": bad.data}
    assert lanes[0].calls == ["rdf"] and lanes[1].calls == ["msd", "bad", "cn"]
    # looking at the first lane's result first: the other lanes' pending results are merged before it, oldest first
    # (amof_amd/_lazy.py Deferred._finish_followers_first) -- on every rank alike, or the collectives below would hang
    rdf2 = Rdf.from_trajectory(packed, dr=0.05, rmax=6.0)
    msd2 = WindowMsd.from_trajectory(packed, delta_time=2, timestep=1)
    bad2 = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.5)
    assert rdf2.data.equals(frames["rdf"])
    assert msd2.__dict__["_pending"] is None and bad2.__dict__["_pending"] is None
    assert msd2.data.equals(frames["msd"]) and bad2.data.equals(frames["bad"])
    if rank == 0:
        for k, v in frames.items():
            v.to_pickle(os.path.join(out_dir, k + ".pkl"))
        np.save(os.path.join(out_dir, "hist.npy"), np.asarray(rdf.hist))
    dist.barrier()
    dist.destroy_process_group()


def test_classes_sharded_over_two_ranks_equal_one_process(tmp_path, monkeypatch):
    import pandas as pd
    port = 25500 + os.getpid() % 2000
    mp.spawn(_worker_classes, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from tests import helpers as H
    from tests import oracle_context
    lanes = oracle_context.install(monkeypatch)
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    packed = H.random_walk(H.zif4_frame(), 9, 0.05, 21, cell_jitter=0.01)
    one = {"rdf": Rdf.from_trajectory(packed, dr=0.05, rmax=6.0, distributed=False),
           "msd": WindowMsd.from_trajectory(packed, delta_time=2, timestep=1, distributed=False),
           "bad": Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.5, distributed=False),
           "cn": CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5}, distributed=False)}
    for k in ("rdf", "bad", "cn"):          # integer counts underneath: bit-identical 1 vs 2 ranks
        assert pd.read_pickle(os.path.join(str(tmp_path), k + ".pkl")).equals(one[k].data), k
    assert np.array_equal(np.load(os.path.join(str(tmp_path), "hist.npy")), one["rdf"].hist)
    two = pd.read_pickle(os.path.join(str(tmp_path), "msd.pkl"))
    assert list(two.columns) == list(one["msd"].data.columns)
    np.testing.assert_allclose(two.values, one["msd"].data.values, rtol=1e-12, atol=1e-15)      # float64 sums, another order
    for ctx in lanes.values():
        ctx.close_lane()
