"""PCIe-inclusive rate of the host-buffer entry points (fpgpu_mul: H2D, kernel, D2H on pageable
numpy memory): the boundary a caller with host-resident vectors sees.  Never the bench's `value`."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # torch first: one HIP runtime per process
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
rng = np.random.default_rng(0)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
b = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
f.mul(a[:1024], b[:1024])
for op, fn in (("mul", lambda: f.mul(a, b)), ("inv", lambda: f.inv(a)), ("sum", lambda: f.sum(a))):
    fn()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps
    print({"op": op, "n": n, "ms": round(dt * 1e3, 2), "Gelem_per_s": round(n / dt / 1e9, 3), "host_GBps": round(n * 32 * (3 if op == "mul" else 2 if op == "inv" else 1) / dt / 1e9, 1)}, flush=True)
