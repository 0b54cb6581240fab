import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from amof_amd import _hip
from amof_amd.frames import Frame, PackedTrajectory
from tests import helpers as H
base = H.replicate(H.zif4_frame(), (7, 7, 8))
shear = np.eye(3) + np.array([[0, 0.15, 0.10], [0, 0, 0.20], [0, 0, 0]])
sheared = Frame(base.numbers, base.positions @ shear, base.cell @ shear)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
packed = H.random_walk(sheared, F, 0.05, 51)
dev = PackedTrajectory(torch.tensor(packed.pos, device="cuda:0"), packed.cell, packed.numbers)
ctx = _hip.get_context(0)
for rep in range(2):
    t0 = time.perf_counter(); h, _, _ = ctx.rdf_accumulate(dev, 10.0, 999); dt = time.perf_counter() - t0
    print("F=%d wall %.2f ms/frame, dominant kernel %.4f ms/frame, all kernels %.4f ms/frame, pairs in range %d" % (F, 1e3*dt/F, 1e3*ctx.last_kernel_seconds(True)/F, 1e3*ctx.last_kernel_seconds(False)/F, int(h.sum())//F))
