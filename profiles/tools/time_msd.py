"""MSD pipeline timings on the headline shape (9792 atoms x 5000 frames, W = 25) for the single- / double-buffered
comb kernel (AMOF_MSD_NODB) and for atom-sharded calls; results must stay identical.

    python profiles/tools/time_msd.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch                                                # noqa: E402
from amof_amd import _hip                                   # noqa: E402
from tests import helpers as H                              # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
packed = H.device_walk(torch.device("cuda", 0), (3, 3, 4), F, 0.05, 20261003)
torch.cuda.synchronize()
ctx = _hip.get_context(0)
window = np.arange(0, F // 2, 100).astype(np.int32)
os.environ["AMOF_MSD_NODB"] = "1"
os.environ["AMOF_MSD_NOSTREAM"] = "1"
ref, _ = ctx.msd_window(packed, window)
print("variant                     pipeline   comb    rest (ms)   max rel dev vs single buffer")
for nodb in ("1", ""):
    for tr in (0,):
        if nodb:
            os.environ["AMOF_MSD_NODB"] = "1"
        else:
            os.environ.pop("AMOF_MSD_NODB", None)
        best = (1e9, 0)
        for _ in range(4):
            out, _ = ctx.msd_window(packed, window)
            best = min(best, (ctx.last_kernel_seconds(False), ctx.last_kernel_seconds(True)))
        dev = float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-300)))
        print("%-25s %8.3f %8.3f %8.3f   %.2e  %s" % ("single" if nodb else "double-buffered", 1e3 * best[0],
                                                            1e3 * best[1], 1e3 * (best[0] - best[1]), dev, ctx.last_path()))
os.environ.pop("AMOF_MSD_NOSTREAM", None)
best = (1e9, 0)
for _ in range(4):
    out, _ = ctx.msd_window(packed, window)
    best = min(best, (ctx.last_kernel_seconds(False), ctx.last_kernel_seconds(True)))
dev = float(np.max(np.abs(out - ref) / np.maximum(np.abs(ref), 1e-300)))
print("%-25s %8.3f %8.3f %8.3f   %.2e  %s" % ("streaming", 1e3 * best[0], 1e3 * best[1], 1e3 * (best[0] - best[1]), dev,
                                            ctx.last_path()))
# atom-sharded: one eighth of the atoms
for n in (1, 2, 4, 8):
    best = 1e9
    for _ in range(4):
        ctx.msd_window(packed, window, atom_range=(0, packed.n_atoms // n))
        best = min(best, ctx.last_kernel_seconds(False))
    print("1/%d of the atoms: pipeline %.3f ms" % (n, 1e3 * best))
