import sys, time, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
from amof_amd import _hip
from amof_amd.bad import Bad
from amof_amd.cn import CoordinationNumber
dev = torch.device("cuda", 0)
# sigma = 0.05 A / frame (the bench's walk: the framework dissolves, few angles survive) or e.g. 0.002 (the framework
# stays intact: 4 N around every Zn, 3456 N-Zn-N angles per frame)
sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
packed = bench.make_trajectory(dev, (3, 3, 4), 5000, sigma, 1)
torch.cuda.synchronize()
ctx = _hip.get_context(0)
print("sigma = %g A per frame and axis" % sigma)
for name, fn in [("BAD Zn-N 2.5", lambda: Bad.from_trajectory(packed, {'Zn-N': 2.5}, dtheta=0.05)),
                 ("CN Zn-N 2.5", lambda: CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5})),
                 ("BAD Zn-N,C-N,C-H", lambda: Bad.from_trajectory(packed, {'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}, dtheta=0.05)),
                 ("CN Zn-N,C-N,C-H", lambda: CoordinationNumber.from_trajectory(packed, {'Zn-N': 2.5, 'C-N': 1.6, 'C-H': 1.3}))]:
    fn()
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    extra = " angles/frame=%s" % (np.asarray(r.n_angles) // 5000).tolist() if hasattr(r, "n_angles") else ""
    print("%-20s wall %.1f ms  kernels %.2f ms (dominant %.2f, %s)  cols=%s%s" % (
        name, 1e3 * (time.perf_counter() - t0), 1e3 * ctx.last_kernel_seconds(False), 1e3 * ctx.last_kernel_seconds(True),
        ctx.last_path(), list(r.data.columns)[:6], extra))
