"""The classes' N > 1 path on a real GPU: two ranks (gloo rendezvous, both on cuda:0 -- the box
has one GPU) shard frames (RDF / BAD / CN) or atoms (MSD) of a replicated trajectory and merge;
every rank must end with exactly the single-process DataFrames (integers: bit-identical)."""

import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _build():
    from tests import helpers as H
    return H.random_walk(H.zif4_frame(), 9, 0.1, 77, cell_jitter=0.005)


def _run_all(packed, distributed):
    from amof_amd.rdf import Rdf
    from amof_amd.msd import WindowMsd
    from amof_amd.bad import Bad
    from amof_amd.cn import CoordinationNumber
    cut = {'Zn-N': 2.5, 'C-N': 1.6}
    return {
        "rdf": Rdf.from_trajectory(packed, dr=0.02, device=0, distributed=distributed).data,
        "msd": WindowMsd.from_trajectory(packed, delta_time=1, timestep=1, device=0, distributed=distributed).data,
        "bad": Bad.from_trajectory(packed, cut, dtheta=0.5, device=0, distributed=distributed).data,
        "cn": CoordinationNumber.from_trajectory(packed, cut, device=0, distributed=distributed).data,
    }


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _run_all(_build(), None)          # None: shard over the initialised group
    for k, df in res.items():
        df.to_pickle(os.path.join(out_dir, "%s_rank%d.pkl" % (k, rank)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process(tmp_path):
    import pandas as pd
    port = 29600 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = _run_all(_build(), False)
    for k, df in single.items():
        for rank in (0, 1):
            got = pd.read_pickle(os.path.join(str(tmp_path), "%s_rank%d.pkl" % (k, rank)))
            assert list(got.columns) == list(df.columns), k
            if k == "msd":      # float sums in a different (but fixed) order
                np.testing.assert_allclose(got.values, df.values, rtol=1e-12, atol=1e-15)
            else:               # built from merged integer counts: identical
                assert np.array_equal(got.values, df.values), k


def _worker_rccl(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", AMOF_DIST_FORCE_MERGE="1")     # one rank, but every collective really runs
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from amof_amd import dist as adist
    assert adist.merging(1) and adist.device_collectives()
    for mode in (None, "local"):
        res = _run_all(_build(), mode)
        for k, df in res.items():
            df.to_pickle(os.path.join(out_dir, "%s_%s.pkl" % (k, mode)))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_single_rank(tmp_path):
    # the code path the 8-GPU bench takes (backend "nccl" = RCCL: histograms accumulated by the "_dev" entry points
    # into CUDA tensors and all-reduced in place, tensor all-gather of the CN rows), with the one rank a
    # single-GPU box allows (AMOF_DIST_FORCE_MERGE=1 runs the merge path although world_size is 1)
    import pandas as pd
    port = 31600 + os.getpid() % 2000
    mp.spawn(_worker_rccl, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    single = _run_all(_build(), False)
    for k, df in single.items():
        for mode in ("None", "local"):
            got = pd.read_pickle(os.path.join(str(tmp_path), "%s_%s.pkl" % (k, mode)))
            assert list(got.columns) == list(df.columns), k
            np.testing.assert_allclose(got.values, df.values, rtol=1e-12, atol=1e-15)


def test_bench_starts_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's multi-GPU command): the parent
    must start the two ranks itself (child process, 127.0.0.1), relay rank 0's JSON line and exit 0.  Both ranks share
    cuda:0 here (gloo rendezvous -- RCCL refuses two ranks per device); the merged result is verified against the
    oracle by the bench itself."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--all-on-device", "0", "--backend", "gloo",
           "--frames", "96", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-bad"]
    run = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    rows = [json.loads(l) for l in run.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(rows) == 1
    d = rows[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["verified"] is True
    assert [r["rdf_frames"] for r in d["per_rank"]] == [[0, 48], [48, 96]]
