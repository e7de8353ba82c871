"""Per-call times of the first host-buffer calls of a process (the staging lanes warm up: threads, pinned buffers, link)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
lib = cpa.load()
n = 1 << 22
rng = np.random.default_rng(5)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= (1 << 60) - 1
b = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); b[:, 3] &= (1 << 60) - 1
o = np.zeros_like(a)
small = a[:1024].copy(); so = np.zeros_like(small)
t0 = time.perf_counter(); lib.fpgpu_mul(f._h, small.ctypes.data, small.ctypes.data, so.ctypes.data, 1024); print("first small call (runtime init) ms", round((time.perf_counter() - t0) * 1e3, 1))
ts = []
for i in range(12):
    t0 = time.perf_counter(); lib.fpgpu_mul(f._h, a.ctypes.data, b.ctypes.data, o.ctypes.data, n); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    if i == 5: time.sleep(1.0)
print("mul 2^22, 12 calls (1 s pause after the 6th):", ts)
