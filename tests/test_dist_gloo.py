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
