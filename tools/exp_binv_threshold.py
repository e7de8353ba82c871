"""Batch inversion just above the lengths where the plan adds a tree level (more than 65 536 chain totals): is a slightly
longer level-0 chain (FPGPU_INV_C0) that keeps the totals at or below 65 536 cheaper than the extra level?"""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
cases = {6_000_000: (0, 92), 1 << 23: (0, 128), 12_000_000: (0, 184), 1 << 24: (0, 256), 7_000_000: (0, 107), 10_000_000: (0, 153)}
if len(sys.argv) > 1 and sys.argv[1] == 'small':
    cases = {300_000: (0, 8, 5), 600_000: (0, 8, 10), 70_000: (0, 8, 2), 131_073: (0, 8, 3), 524_289: (0, 8, 9), 1 << 19: (0, 8)}
for n, c0s in cases.items():
    x = f.random_dev(n, 3); ref = f.alloc(n); out = f.alloc(n)
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    row = {"n": n}
    for c0 in c0s:
        _lib.tuning_set("FPGPU_INV_C0", c0) if c0 else _lib.tuning_set("FPGPU_INV_C0")
        ws = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
        dst = ref if c0 == 0 else out
        for _ in range(10): f.inv_dev(x, dst, flags, ws)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): f.inv_dev(x, dst, flags, ws)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30)
        if c0:
            assert torch.equal(out, ref), (n, c0)
        row["plan" if c0 == 0 else "c0=%d" % c0] = round(best * 1e3, 1)
    _lib.tuning_set("FPGPU_INV_C0")
    print(json.dumps(row), flush=True)
