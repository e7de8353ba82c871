"""Launch durations of the seven 256-bit elementwise kernels over 2^28 BN254 elements (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 28
a = f.random_dev(n, 1); b = f.random_dev(n, 2); out = f.alloc(n)
ops = (("mul", lambda: f.mul_dev(a, b, out), 96), ("sqr", lambda: f.sqr_dev(a, out), 64), ("to_mont", lambda: f.to_mont_dev(a, out), 64),
       ("add", lambda: f.add_dev(a, b, out), 96), ("sub", lambda: f.sub_dev(a, b, out), 96), ("neg", lambda: f.neg_dev(a, out), 64),
       ("from_mont", lambda: f.from_mont_dev(a, out), 64))
for rep in range(2):
    for name, fn, nb in ops:
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print({"op": name, "ms": round(ms, 3), "TBps": round(nb * n / ms / 1e9, 3), "frac_of_8TBps": round(nb * n / ms / 1e9 / 8, 3)}, flush=True)
