"""Batch inversion at the per-rank sizes of configs[2] over 1 / 2 / 4 / 8 GPUs (2^24 / N BLS12-381 Fq elements per rank):
what does the latency-bound top of the tree cost a rank whose slice is an eighth of the batch?"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
total = 1 << 24
x = f.random_dev(total, 3)
for n_gpus in (1, 2, 4, 8, 16):
    n = total // n_gpus
    xs = x[:n]
    out = f.alloc(n)
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ws = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
    for _ in range(10): f.inv_dev(xs, out, flags, ws)
    torch.cuda.synchronize()
    reps = 10 * n_gpus
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f.inv_dev(xs, out, flags, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"ranks": n_gpus, "elements": n, "ms": round(ms, 4), "x_ranks_ms": round(ms * n_gpus, 3)}), flush=True)
