"""Experiment: does the relative placement of a, b and out in HBM matter for the 2-reads-1-write
stream?  (Three 8 GiB arrays at identical offsets from 2 MiB-aligned bases touch the same channels
and banks at the same time.)  Carves the three operands out of one allocation with a byte skew."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 28
pad = 1 << 22                      # elements of slack between the operands (128 MiB)
big = torch.empty((3 * (n + pad), 4), dtype=torch.int64, device="cuda")
def timed(a, b, out, reps=10):
    f.mul_dev(a, b, out); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f.mul_dev(a, b, out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for skew_bytes in (0, 256, 4096, 65536, 1 << 20, 3 << 20, (1 << 20) + 4096 * 37, 33 << 20):
    se = skew_bytes // 32
    a = big[0:n]
    b = big[n + pad + se: n + pad + se + n]
    out = big[2 * (n + pad) + 2 * se: 2 * (n + pad) + 2 * se + n]
    check = f.random_dev(16, 1)   # keep the context warm; operands need not be canonical for timing
    ms = timed(a, b, out)
    print({"skew_bytes": skew_bytes, "ms": round(ms, 4), "TBps": round(96 * n / ms / 1e9, 3)}, flush=True)
