"""Kernel time of the tile kernel's variants on the 9792-atom cell: diagonal with slab culling (the headline: rdf_tile_zf),
diagonal without culling, sheared with the cutoff clear of the half heights (plain general-cell variant), sheared at
the default cutoff (image-aware variant).  `python3 profiles/tools/time_tile_variants.py [frames]`"""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
import torch                                                # noqa: E402
from amof_amd import _hip                                   # noqa: E402
from amof_amd.frames import Frame, PackedTrajectory         # noqa: E402
from tests import helpers as H                              # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
ctx = _hip.get_context(0)
base = H.replicate(H.zif4_frame(), (3, 3, 4))
base = Frame(base.numbers, base.positions, np.diag(np.diag(base.cell)))
os.environ["AMOF_RDF_NOCELL"] = "1"
os.environ["AMOF_RDF_NORANGE"] = "1"


def run(name, frame, rmax, env=None):
    host = H.random_walk(frame, F, 0.05, 7)
    packed = PackedTrajectory(torch.tensor(host.pos, device=dev), host.cell, host.numbers)
    for k, v in (env or {}).items():
        os.environ[k] = v
    best = 1e9
    for _ in range(3):
        ctx.rdf_accumulate(packed, rmax, int(rmax // 0.01))
        best = min(best, ctx.last_kernel_seconds(True))
    for k in (env or {}):
        os.environ.pop(k)
    print("%-44s %-13s %8.4f ms/frame" % (name, ctx.last_path(), 1e3 * best / F), flush=True)


half = float(np.min(np.sqrt((base.cell ** 2).sum(axis=1))) / 2)
run("diagonal, slab culling (headline)", base, half)
run("diagonal, culling off", base, half, {"AMOF_RDF_NOCULL": "1"})
if os.environ.get("AMOF_TILE_DEBUG"):
    run("  diagonal, rmax = 0.9 half (as the sheared rows)", base, 0.9 * half)
for eps in (0.02, 0.10):
    shear = np.eye(3) + np.array([[0, eps, 0.5 * eps], [0, 0, eps], [0, 0, 0]])
    fr = Frame(base.numbers, base.positions @ shear, base.cell @ shear)
    hmin = 1.0 / np.linalg.norm(np.linalg.inv(fr.cell), axis=0).max()
    run("sheared %2.0f %%, rmax = 0.45 min height" % (100 * eps), fr, 0.9 * hmin / 2)
    if os.environ.get("AMOF_TILE_DEBUG"):
        run("  same, old plain general variant", fr, 0.9 * hmin / 2, {"AMOF_RDF_NOTRI": "1"})
    run("sheared %2.0f %%, default rmax (images)" % (100 * eps), fr, float(np.min(np.sqrt((fr.cell ** 2).sum(axis=1))) / 2))
# the typical aMOF input: a near-cubic NPT cell, slightly sheared, a different cell every frame, default cutoff
cub = H.replicate(H.zif4_frame(), (4, 4, 3))                       # 13 056 atoms, 61.6 x 61.2 x 55.3 A
cub = Frame(cub.numbers, cub.positions, np.diag(np.diag(cub.cell)))
run("near-cubic 13 056 atoms, diagonal", cub, float(np.min(np.diag(cub.cell)) / 2))
shear = np.eye(3) + np.array([[0, 0.02, 0.01], [0, 0, 0.02], [0, 0, 0]])
fr = Frame(cub.numbers, cub.positions @ shear, cub.cell @ shear)
host = H.random_walk(fr, F, 0.05, 9, cell_jitter=0.005)
packed = PackedTrajectory(torch.tensor(host.pos, device=dev), host.cell, host.numbers)
rm = float(np.min(packed.cell_lengths()) / 2)
best = 1e9
for _ in range(3):
    ctx.rdf_accumulate(packed, rm, int(rm // 0.01))
    best = min(best, ctx.last_kernel_seconds(True))
print("%-44s %-13s %8.4f ms/frame" % ("near-cubic 13 056 atoms, 2 % shear, NPT cells", ctx.last_path(), 1e3 * best / F), flush=True)
