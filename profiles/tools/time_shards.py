"""What one rank of an N-way strong-scaled configs[3] step costs, measured on ONE GPU.

For N in 1, 2, 4, 8: the first rank's shard of the 5000-frame, 9792-atom trajectory -- frames [0, F/N) for RDF and
BAD, atoms [0, 9792/N) for MSD -- through the C ABI, kernel seconds (HIP events inside the library) beside wall
seconds of the call.  The per-call host cost (wall - kernels) is what decides whether 8 GPUs reach >= 7x.

The second table runs the PUBLIC CLASSES as rank 0 of an N-way job: a one-rank RCCL group is initialised
(AMOF_DIST_FORCE_MERGE=1, so every collective of the N > 1 path is really issued -- in a one-rank group, i.e. without
the wire time of 7 peers) and ``amof_amd.dist.world`` is patched to answer (0, N), so the classes shard exactly as the
first rank of N would: host work, device-side merge buffers, collectives' launch cost and DataFrame assembly included.

    python profiles/tools/time_shards.py [frames]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import torch                                                # noqa: E402
from amof_amd import _hip, atom as amatom                   # noqa: E402
from tests import helpers as H                              # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dev = torch.device("cuda", 0)
packed = H.device_walk(dev, (3, 3, 4), F, 0.05, 20261003)
torch.cuda.synchronize()
ctx = _hip.get_context(0)
N = packed.n_atoms
rmax = float(np.min(packed.cell_lengths()) / 2)
nb = int(rmax // 0.01)
kinds, sp = H.species_of(packed.numbers)
rcm = amatom.cutoff_matrix(amatom.format_cutoff({'Zn-N': 2.5}), kinds)
triples = [(kinds.index(7), kinds.index(30)), (kinds.index(30), kinds.index(7))]
edges = np.arange(int(180 // 0.05) + 2) * 0.05
window = np.arange(0, F // 2, 100)


def timed(fn, reps=3):
    fn()
    best = (1e9, 0.0)
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        w = time.perf_counter() - t0
        best = min(best, (w, ctx.last_kernel_seconds(False)))
    return best


print("N_gpus  rdf_wall  rdf_kern | bad_wall  bad_kern | msd_wall  msd_kern | sum_wall   speedup_vs_1 (ms)")
base = None
for n in (1, 2, 4, 8):
    fr = (0, F // n)
    ar = (0, N // n)
    r = timed(lambda: ctx.rdf_accumulate(packed, rmax, nb, frame_range=fr))
    b = timed(lambda: ctx.bad_hist(packed, rcm, triples, edges, frame_range=fr))
    m = timed(lambda: ctx.msd_window(packed, window, atom_range=ar))
    tot = r[0] + b[0] + m[0]
    base = base or tot
    print("%6d  %8.3f  %8.3f | %8.3f  %8.3f | %8.3f  %8.3f | %8.3f   %.2fx" %
          (n, 1e3 * r[0], 1e3 * r[1], 1e3 * b[0], 1e3 * b[1], 1e3 * m[0], 1e3 * m[1], 1e3 * tot, base / tot))

# the public classes as rank 0 of N (see the header): RDF + MSD per step is the bench's `value`, + BAD is configs[3]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
os.environ["AMOF_DIST_FORCE_MERGE"] = "1"
import torch.distributed as tdist                           # noqa: E402
from amof_amd import dist as adist                          # noqa: E402
from amof_amd.rdf import Rdf                                # noqa: E402
from amof_amd.msd import WindowMsd                          # noqa: E402
from amof_amd.bad import Bad                                # noqa: E402
torch.cuda.set_device(0)
tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
real_world = adist.world
def best_wall(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def step(with_bad):
    # what bench.py times: the constructors enqueue (RDF on the first lane, MSD / BAD on the second, a high-priority
    # stream), every result is looked at before the clock stops
    a = Rdf.from_trajectory(packed, device=0, distributed=None)
    b = WindowMsd.from_trajectory(packed, delta_time=100, timestep=1, device=0, distributed=None)
    c = Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05, device=0, distributed=None) if with_bad else None
    n = len(a.data) + len(b.data) + (len(c.data) if with_bad else 0)
    return n


print("classes as rank 0 of N (one-rank RCCL group): wall ms, best of 5.  'each' = the class alone (constructor + .data);")
print("'step' = the constructors back to back, then every .data (AMOF_ASYNC=%s)" % os.environ.get("AMOF_ASYNC", "1"))
print("N_gpus  Rdf_each  Msd_each  Bad_each | step rdf+msd  speedup_vs_1 | step rdf+msd+bad  speedup_vs_1")
base2 = base3 = None
for n in (1, 2, 4, 8):
    adist.world = (lambda group=None, n=n: (0, n)) if n > 1 else real_world
    r = best_wall(lambda: Rdf.from_trajectory(packed, device=0, distributed=None).result())
    m = best_wall(lambda: WindowMsd.from_trajectory(packed, delta_time=100, timestep=1, device=0, distributed=None).result())
    b = best_wall(lambda: Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05, device=0, distributed=None).result())
    t2, t3 = best_wall(lambda: step(False)), best_wall(lambda: step(True))
    base2, base3 = base2 or t2, base3 or t3
    print("%6d  %8.3f  %8.3f  %8.3f | %12.3f   %.2fx        | %16.3f     %.2fx" %
          (n, 1e3 * r, 1e3 * m, 1e3 * b, 1e3 * t2, base2 / t2, 1e3 * t3, base3 / t3))
adist.world = real_world
tdist.destroy_process_group()
os.environ.pop("AMOF_DIST_FORCE_MERGE")

# single-process classes (no merge path): kernels vs host share
from amof_amd.rdf import Rdf                                # noqa: E402,F811
from amof_amd.msd import WindowMsd                          # noqa: E402
from amof_amd.bad import Bad                                # noqa: E402
for name, fn in (("Rdf", lambda: Rdf.from_trajectory(packed, device=0, distributed=False)),
                 ("Bad", lambda: Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05, device=0, distributed=False)),
                 ("WindowMsd", lambda: WindowMsd.from_trajectory(packed, delta_time=100, timestep=1, device=0,
                                                                 distributed=False))):
    best = (1e9, 0.0)
    for _ in range(4):
        t0 = time.perf_counter()
        o = fn().result()
        best = min(best, (time.perf_counter() - t0, o._stats["kernel_s_all"]))
    w, k = best
    print("class %-10s wall %.3f ms, kernels %.3f ms, host %.3f ms" % (name, 1e3 * w, 1e3 * k, 1e3 * (w - k)))
