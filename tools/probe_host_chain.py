"""Why the host-buffer chain runs slower inside bench.py than alone: time each call, with fresh and with reused result arrays,
before and after the process has allocated and used 24 GiB of device memory through torch."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
lib = cpa.load()
n = 1 << 22
rng = np.random.default_rng(5)
ha = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ha[:, 3] &= (1 << 60) - 1
hb = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); hb[:, 3] &= (1 << 60) - 1
o1, o2, o3 = np.zeros_like(ha), np.zeros_like(ha), np.zeros_like(ha)
def t(fn, reps=3):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return round((time.perf_counter() - t0) / reps * 1e3, 2)
def report(tag):
    r = {"tag": tag}
    r["mul_fresh"] = t(lambda: f.mul(ha, hb))
    r["add_fresh"] = t(lambda: f.add(ha, hb))
    r["sqr_fresh"] = t(lambda: f.sqr(ha))
    r["chain_fresh"] = t(lambda: f.sqr(f.add(f.mul(ha, hb), ha)))
    p = lambda x: x.ctypes.data
    r["mul_reused"] = t(lambda: lib.fpgpu_mul(f._h, p(ha), p(hb), p(o1), n))
    def chain():
        lib.fpgpu_mul(f._h, p(ha), p(hb), p(o1), n); lib.fpgpu_add(f._h, p(o1), p(ha), p(o2), n); lib.fpgpu_sqr(f._h, p(o2), p(o3), n)
    r["chain_reused"] = t(chain)
    print(json.dumps(r), flush=True)
report("fresh process")
import torch
big = [torch.empty(1 << 33, dtype=torch.uint8, device="cuda") for _ in range(3)]
for b in big: b.zero_()
torch.cuda.synchronize()
report("after 24 GiB of torch tensors")
x = f.random_dev(1 << 26, 3); y = f.alloc(1 << 26)
for _ in range(20): f.mul_dev(x, x, y)
torch.cuda.synchronize()
report("after kernels on torch's stream")
