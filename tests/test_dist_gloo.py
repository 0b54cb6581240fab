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
