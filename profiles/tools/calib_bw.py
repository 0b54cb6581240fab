"""Calibration: what this MI355X sustains for the access patterns of the MSD pipeline (torch kernels as yardstick).

    python profiles/tools/calib_bw.py
"""
import torch

F, N = 5000, 9792
dev = torch.device("cuda", 0)
pos = torch.randn((F, 3 * N), dtype=torch.float64, device=dev)
dst = torch.empty_like(pos)
dstT = torch.empty((3 * N, F), dtype=torch.float64, device=dev)
gb = pos.numel() * 8 / 1e9


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


t = timed(lambda: pos.sum())
print("read-only reduction      %.3f ms  %.2f TB/s (%.3f GB read)" % (1e3 * t, gb / t / 1e3, gb))
t = timed(lambda: dst.copy_(pos))
print("flat copy                %.3f ms  %.2f TB/s (read + write)" % (1e3 * t, 2 * gb / t / 1e3))
t = timed(lambda: dstT.copy_(pos.t()))
print("transposing copy (torch) %.3f ms  %.2f TB/s (read + write)" % (1e3 * t, 2 * gb / t / 1e3))
t = timed(lambda: dst.zero_())
print("write-only fill          %.3f ms  %.2f TB/s" % (1e3 * t, gb / t / 1e3))
t = timed(lambda: torch.sub(pos[1:], pos[:-1], out=dst[1:]))
print("frame difference (flat)  %.3f ms  %.2f TB/s (2 reads of which 1 cached + write: %.3f GB min)" % (1e3 * t, 2 * gb / t / 1e3, 2 * gb))
