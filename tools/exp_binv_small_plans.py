"""Batch inversion of a rank's slice (2^20 .. 2^22 BLS12-381 Fq elements) against the level-0 chain length (FPGPU_INV_C0):
the plan's default keeps 65536 lanes (one wave per SIMD) in the sweeps; are shorter chains (more waves) faster there?"""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
lib = _lib.load()
DEFAULT = -0x7fffffff - 1
f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
for lg in (20, 21, 22, 23):
    n = 1 << lg
    x = f.random_dev(n, 3)
    out = f.alloc(n); ref = f.alloc(n)
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    row = {"log2_n": lg}
    for c0 in (DEFAULT, 8, 16, 32, 64):
        assert lib.fpgpu_tuning_set(b"FPGPU_INV_C0", ctypes.c_long(c0)) == 0
        ws = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
        dst = ref if c0 == DEFAULT else out
        for _ in range(10): f.inv_dev(x, dst, flags, ws)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40): f.inv_dev(x, dst, flags, ws)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40)
        if c0 != DEFAULT:
            assert torch.equal(out, ref), (lg, c0)
        row["default" if c0 == DEFAULT else "c0=%d" % c0] = round(best * 1e3, 1)
    lib.fpgpu_tuning_set(b"FPGPU_INV_C0", ctypes.c_long(DEFAULT))
    print(json.dumps(row), flush=True)
