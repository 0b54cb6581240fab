"""MSD pipeline timings on the headline shape (9792 atoms x 5000 frames, W = 25, spacing 100): the fused form (round 5:
no transposed copy), the 2-pass and 3-pass transposed forms, and the two halves of an atom-sharded call for 1 / N of the
atoms.  Kernel seconds from the library's HIP events (whole pipeline / dominant kernel); results must agree.

    python profiles/tools/time_msd.py [frames]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
import torch                                                # noqa: E402
from amof_amd import _hip                                   # noqa: E402
from tests import helpers as H                              # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
dev = torch.device("cuda", 0)
packed = H.device_walk(dev, (3, 3, 4), F, 0.05, 20261003)
torch.cuda.synchronize()
ctx = _hip.get_context(0)
N = packed.n_atoms
window = np.arange(0, F // 2, 100).astype(np.int32)
alg = F * (24 * N + 72)


def run(env, reps=6):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        best = (1e9, 0, 0)
        for _ in range(reps):
            t0 = time.perf_counter()
            out, _ = ctx.msd_window(packed, window)
            w = time.perf_counter() - t0
            best = min(best, (ctx.last_kernel_seconds(False), ctx.last_kernel_seconds(True), w))
        return best, out, ctx.last_path()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


ref = None
print("form                      pipeline  dominant   rest    wall (ms)   frac of 8 TB/s on 24 N F bytes   max rel dev   path")
for name, env in (("3-pass transposed", {"AMOF_MSD_NOFUSED": "1", "AMOF_MSD_NOFOLD": "1"}),
                  ("2-pass transposed", {"AMOF_MSD_NOFUSED": "1"}),
                  ("fused (default)", {})):
    best, out, path = run(env)
    ref = out if ref is None else ref
    dev_rel = float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-300)))
    print("%-24s %8.3f  %8.3f %8.3f %8.3f      %.3f                          %.1e   %s"
          % (name, 1e3 * best[0], 1e3 * best[1], 1e3 * (best[0] - best[1]), 1e3 * best[2], alg / best[0] / 8e12, dev_rel, path))

# the halves of an atom-sharded call (rank 0 of n): kernel seconds of begin and finish, wall of both
S = len(_hip.packed_species(packed)[0])
print("atom-sharded halves (amof_msd_shard_begin / _finish), rank 0 of n:")
for n in (1, 2, 4, 8):
    ar = (0, N // n)
    best = (1e9, 0, 0, 0)
    for _ in range(6):
        csum = torch.empty((F, 3), dtype=torch.float64, device=dev)
        out = torch.zeros((S, len(window)), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.msd_shard_begin(packed, window, ar, csum)
        kb = ctx.last_kernel_seconds(False)
        ctx.msd_shard_finish(packed, window, ar, csum, out)
        kf = ctx.last_kernel_seconds(False)
        w = time.perf_counter() - t0
        best = min(best, (kb + kf, kb, kf, w))
    old = 1e9
    for _ in range(4):
        ctx.msd_window(packed, window, atom_range=ar)
        old = min(old, ctx.last_kernel_seconds(False))
    print("  1/%d of the atoms: begin %.3f + finish %.3f = %.3f ms of kernels, wall %.3f ms   (general kernels on the range: %.3f ms)"
          % (n, 1e3 * best[1], 1e3 * best[2], 1e3 * best[0], 1e3 * best[3], 1e3 * old))
