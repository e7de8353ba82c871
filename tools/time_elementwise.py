"""HBM rate of the elementwise kernels at other widths (BASELINE's headline is the 256-bit case)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
for name, p, n64, lg in (("goldilocks", cpa.GOLDILOCKS, 1, 28), ("p192", 2**192 - 2**64 - 1, 3, 27), ("bls12_381_fq", cpa.BLS12_381_FQ, 6, 26), ("p512", 2**512 - 569, 8, 26)):
    f = cpa.MontyField(p, n64)
    n = 1 << lg
    a = f.random_dev(n, 1); b = f.random_dev(n, 2); out = f.alloc(n)
    for op, fn, nb in (("mul", lambda: f.mul_dev(a, b, out), 3), ("add", lambda: f.add_dev(a, b, out), 3), ("sqr", lambda: f.sqr_dev(a, out), 2)):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print({"field": name, "n64": n64, "op": op, "log2n": lg, "ms": round(ms, 3), "Gelem_per_s": round(n / ms / 1e6, 1), "TBps": round(nb * 8 * n64 * n / ms / 1e9, 2)}, flush=True)
