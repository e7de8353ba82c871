import sys, os
sys.path.insert(0, "/root/repo")
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 28
a = f.random_dev(n, 1); b = f.random_dev(n, 2); out = f.alloc(n)
def per_step(fn, steps):
    fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return [round(e0.elapsed_time(e1), 3) for e0, e1 in evs]
def bulk(fn, steps):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / steps, 3)
print("mul per-step", per_step(lambda: f.mul_dev(a, b, out), 20))
print("sqr per-step", per_step(lambda: f.sqr_dev(a, out), 10))
print("sqr bulk", bulk(lambda: f.sqr_dev(a, out), 10))
print("sqr per-step", per_step(lambda: f.sqr_dev(a, out), 10))
print("neg per-step", per_step(lambda: f.neg_dev(a, out), 10))
print("sqr bulk", bulk(lambda: f.sqr_dev(a, out), 10))
c = f.alloc(n)
print("sqr into fresh buffer bulk", bulk(lambda: f.sqr_dev(a, c), 10))
print("sqr b->out bulk", bulk(lambda: f.sqr_dev(b, out), 10))
