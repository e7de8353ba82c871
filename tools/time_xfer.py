import sys, os, time, json
sys.path.insert(0, "/root/repo")
import numpy as np
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 22
rng = np.random.default_rng(1)
a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= (1 << 60) - 1
v = cpa.DeviceVector.alloc(f, n)
out = np.empty_like(a)
res = {}
for name, fn in (("up", lambda: v.upload(a)), ("down_fresh", lambda: v.download()), ("down_reuse", lambda: v.download(out=out) if False else v.download())):
    fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    res[name] = round(a.nbytes / dt / 1e9, 1)
print(json.dumps({"threads": os.environ.get("FPGPU_COPY_THREADS"), "mb": os.environ.get("EXP_XFER_MB"), **res}), flush=True)
