"""A few calls of one analysis on the headline workload, for rocprofv3 (kernel trace or --pmc passes).

    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES ... -d gpurun_out/pmc -- python3 profiles/tools/run_once.py msd|rdf|bad|cn|bad3|cn3|cfg4 [frames]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch                                                # noqa: E402
from amof_amd import _hip, atom as amatom                   # noqa: E402
from tests import helpers as H                              # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "msd"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
reps = int(os.environ.get("RUN_ONCE_REPS", "3"))
dev = torch.device("cuda", 0)
ctx = _hip.get_context(0)
if what == "cfg4":
    from amof_amd.frames import Frame, PackedTrajectory
    base = H.replicate(H.zif4_frame(), (7, 7, 8))
    shear = np.eye(3) + np.array([[0, 0.15, 0.10], [0, 0, 0.20], [0, 0, 0]])
    sheared = Frame(base.numbers, base.positions @ shear, base.cell @ shear)
    host = H.random_walk(sheared, min(F, int(os.environ.get("RUN_ONCE_CFG4_FRAMES", "64"))), 0.05, 51)
    packed = PackedTrajectory(torch.tensor(host.pos, device=dev), host.cell, host.numbers)
    for _ in range(reps):
        ctx.rdf_accumulate(packed, 10.0, 999)
    print(ctx.last_path(), ctx.last_kernel_seconds(True))
    sys.exit(0)
packed = H.device_walk(dev, (3, 3, 4), F, float(os.environ.get("RUN_ONCE_SIGMA", "0.05")), 20261003)   # 0.002: the framework stays intact
torch.cuda.synchronize()
kinds, sp = H.species_of(packed.numbers)
if what == "msd":
    window = np.arange(0, F // 2, 100).astype(np.int32)
    for _ in range(reps):
        ctx.msd_window(packed, window)
elif what == "rdf":
    rmax = float(np.min(packed.cell_lengths()) / 2)
    for _ in range(reps):
        ctx.rdf_accumulate(packed, rmax, int(rmax // 0.01))
elif what in ("bad3", "cn3"):       # the three-cutoff case: 17 triples / 3 + 3 sets over Zn+N, C+N, C+H
    rcm = amatom.cutoff_matrix(amatom.format_cutoff({'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}), kinds)
    S = len(kinds)
    if what == "bad3":
        from amof_amd.bad import Bad
        for _ in range(reps):
            Bad.from_trajectory(packed, {'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}, dtheta=0.05)
    else:
        sets = [(a, b) for a in range(S) for b in range(S) if rcm[a, b] > 0]
        for _ in range(reps):
            ctx.cn_count(packed, rcm, sets)
else:
    rcm = amatom.cutoff_matrix(amatom.format_cutoff({'Zn-N': 2.5}), kinds)
    zn, n = kinds.index(30), kinds.index(7)
    if what == "bad":
        edges = np.arange(int(180 // 0.05) + 2) * 0.05
        for _ in range(reps):
            ctx.bad_hist(packed, rcm, [(n, zn), (zn, n)], edges)
    else:
        for _ in range(reps):
            ctx.cn_count(packed, rcm, [(zn, n), (n, zn)])
print(ctx.last_path(), ctx.last_kernel_seconds(True), ctx.last_kernel_seconds(False))
