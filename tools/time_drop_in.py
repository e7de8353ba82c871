"""Drop-in chain (bench.py's bench_drop_in_chain) and single host-buffer calls, for a given FPGPU_COPY_THREADS."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crypto_primitives_amd as cpa
import bench
f = cpa.MontyField(cpa.BN254_FR, 4)
out = bench.bench_drop_in_chain(f)
n = 1 << 22
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= (1 << 60) - 1
b = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); b[:, 3] &= (1 << 60) - 1
for name, fn in (("mul", lambda: f.mul(a, b)), ("inv", lambda: f.inv(a)), ("sum", lambda: f.sum(a))):
    fn()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    dt = (time.perf_counter() - t0) / 3
    out["host_" + name] = {"ms": round(dt * 1e3, 2), "Gelem_per_s": round(n / dt / 1e9, 3)}
# the same call straight through the C ABI into a result array that is already mapped (what a Rust
# caller's pre-sized Vec is): no page faults inside the timed region
lib = cpa.load()
o = np.zeros_like(a)
for name, call in (("mul", lambda: lib.fpgpu_mul(f._h, a.ctypes.data, b.ctypes.data, o.ctypes.data, n)),
                   ("sqr", lambda: lib.fpgpu_sqr(f._h, a.ctypes.data, o.ctypes.data, n))):
    assert call() == 0
    t0 = time.perf_counter()
    for _ in range(5): call()
    dt = (time.perf_counter() - t0) / 5
    out["abi_" + name + "_mapped_out"] = {"ms": round(dt * 1e3, 2), "Gelem_per_s": round(n / dt / 1e9, 3)}
print(json.dumps({"copy_threads": os.environ.get("FPGPU_COPY_THREADS", "default"),
                  **{k: (round(v["Gelem_per_s"], 3) if isinstance(v, dict) else v) for k, v in out.items()}}), flush=True)
