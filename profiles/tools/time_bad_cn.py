import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from amof_amd import _hip
from amof_amd.bad import Bad
from amof_amd.cn import CoordinationNumber
dev = torch.device("cuda", 0)
packed = bench.make_trajectory(dev, (3, 3, 4), 5000, 0.05, 1)
ctx = _hip.get_context(0)
for name, fn in [("BAD Zn-N 2.5", lambda: Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05)),
                 ("CN Zn-N 2.5", lambda: CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5})),
                 ("BAD Zn-N,C-N,C-H", lambda: Bad.from_trajectory(packed, {'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}, dtheta=0.05)),
                 ("CN Zn-N,C-N,C-H", lambda: CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}))]:
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-20s wall %.1f ms  kernel %.2f ms  cols=%s" % (name, 1e3 * (time.perf_counter() - t0), 1e3 * ctx.last_kernel_seconds(True), list(r.data.columns)[:6]))
