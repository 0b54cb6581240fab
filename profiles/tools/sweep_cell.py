import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from amof_amd import _hip
ctx = _hip.get_context(0)
dev = torch.device("cuda", 0)
for reps, F in [((3, 3, 4), 500), ((2, 2, 2), 1000), ((4, 4, 4), 200)]:
    tr = bench.make_trajectory(dev, reps, F, 0.05, 1)
    for rmax in [3.0, 5.0, 7.0, 9.0, 11.0, 13.0]:
        nb = int(rmax // 0.01)
        res = {}
        for mode in ["NOCELL", "FORCE_CELL", "default"]:
            for k in ("AMOF_RDF_NOCELL", "AMOF_RDF_FORCE_CELL"): os.environ.pop(k, None)
            if mode != "default": os.environ["AMOF_RDF_" + mode] = "1"
            try:
                for rep in range(2):
                    h, _, _ = ctx.rdf_accumulate(tr, rmax, nb)
                res[mode] = 1e6 * ctx.last_kernel_seconds(False) / F
            except Exception as e:
                res[mode] = float("nan")
        pick = "cell" if abs(res["default"] - res["FORCE_CELL"]) < abs(res["default"] - res["NOCELL"]) else "other"
        best = "cell" if res["FORCE_CELL"] < res["NOCELL"] else "other"
        print("N=%6d rmax=%5.1f  other %8.2f us/frame  cell %8.2f us/frame  default %8.2f  picked=%s best=%s %s" % (
            tr.n_atoms, rmax, res["NOCELL"], res["FORCE_CELL"], res["default"], pick, best, "" if pick == best else "<-- MISPICK"), flush=True)
