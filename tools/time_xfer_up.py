import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 22
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
v = cpa.DeviceVector.alloc(f, n)
for i in range(3):
    t0 = time.perf_counter(); v.upload(a); dt = time.perf_counter() - t0
    print("upload", i, round(a.nbytes / dt / 1e9, 1), "GB/s", file=sys.stderr, flush=True)
