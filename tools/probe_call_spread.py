"""Spread of the steady-state time of one host-buffer call (40 calls of fpgpu_mul over 2^22 BN254 elements)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4); lib = cpa.load()
n = 1 << 22
rng = np.random.default_rng(5)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= (1 << 60) - 1
b = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); b[:, 3] &= (1 << 60) - 1
o = np.zeros_like(a)
for _ in range(8): lib.fpgpu_mul(f._h, a.ctypes.data, b.ctypes.data, o.ctypes.data, n)
ts = []
for i in range(40):
    t0 = time.perf_counter(); lib.fpgpu_mul(f._h, a.ctypes.data, b.ctypes.data, o.ctypes.data, n); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
print(os.environ.get("EXP_ONE_DOWN"), os.environ.get("FPGPU_COPY_THREADS"), "median", sorted(ts)[20], "mean", round(sum(ts) / 40, 2), ts)
