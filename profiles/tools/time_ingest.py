import time, os, sys, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
from amof_amd import _hip
rng = np.random.default_rng(0)
N, F = 9792, 200
path = "/tmp/big.xyz"
with open(path, "w") as fh:
    for f in range(F):
        p = rng.uniform(0, 70, (N, 3))
        fh.write("%d\n" % N)
        fh.write('Lattice="46.27 0.0 0.0 0.0 46.21 0.0 0.0 0.0 73.75" Properties=species:S:1:pos:R:3\n')
        fh.write("".join(["%-2s %16.8f %16.8f %16.8f\n" % ("H", a, b, c) for a, b, c in p]))
lib = _hip.load_library()
bp = path.encode()
sz = os.path.getsize(path) / 1e6
nf, na = ctypes.c_int64(0), ctypes.c_int64(0)
for _ in range(2):
    t0 = time.perf_counter(); lib.amof_xyz_scan(bp, ctypes.byref(nf), ctypes.byref(na)); t1 = time.perf_counter()
    print("scan %.1f ms (%.0f MB/s)" % ((t1 - t0) * 1e3, sz / (t1 - t0)))
F, N = nf.value, na.value
pos = np.zeros((F, N, 3)); sym = np.zeros((N, 4), np.uint8); lat = np.zeros((F, 9)); has = ctypes.c_int32(0)
lib.amof_xyz_read.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
for th in (1, 4, 8, 16, 16, 24, 32, 48, 64, 64, 16):
    t0 = time.perf_counter()
    rc = lib.amof_xyz_read(bp, 0, F, 1, N, pos.ctypes.data, sym.ctypes.data, lat.ctypes.data, ctypes.addressof(has), th)
    t1 = time.perf_counter()
    print("read threads=%d rc=%d %.1f ms (%.0f MB/s, %.0f frames/s)" % (th, rc, (t1 - t0) * 1e3, sz / (t1 - t0), F / (t1 - t0)))
