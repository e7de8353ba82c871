"""Why does bench.py see sqr 6 % slower than a bare loop?  Replays bench.py's sequence piece by piece."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 28
variant = sys.argv[1] if len(sys.argv) > 1 else "bench_seeds"
if variant == "bench_seeds":
    a = f.random_dev(n, 0x9E3779B97F4A7C15 ^ 2); b = f.random_dev(n, 0x9E3779B97F4A7C15 ^ 3)
else:
    a = f.random_dev(n, 1); b = f.random_dev(n, 2)
out = f.alloc(n)
def time_steps(fn, steps, warmup):
    for _ in range(warmup): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e0, e1 in evs:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return round(sum(e0.elapsed_time(e1) for e0, e1 in evs) / steps, 3)
print(variant, "mul", time_steps(lambda: f.mul_dev(a, b, out), 10, 2))
if variant != "no_sample":
    idx = torch.arange(0, n, max(1, n // 4096), device="cuda")
    s = (a[idx].cpu().numpy(), b[idx].cpu().numpy(), out[idx].cpu().numpy())
print(variant, "sqr", time_steps(lambda: f.sqr_dev(a, out), 10, 1))
print(variant, "sqr again", time_steps(lambda: f.sqr_dev(a, out), 10, 1))
print(variant, "add", time_steps(lambda: f.add_dev(a, b, out), 10, 1))
print(variant, "sqr after add", time_steps(lambda: f.sqr_dev(a, out), 10, 1))
