"""Round-3 batch-inversion experiments (BLS12-381, 2^24 elements unless LOG2N says otherwise): chain length of
level 0, the fused top kernel on / off, level-0 prefixes as limbs in a workspace array on / off.  Every variant
is checked (x * inv(x) == 1 everywhere, a strided sample against the oracle) before it is timed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
from oracle.fp_oracle import Oracle

f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
o = Oracle(cpa.BLS12_381_FQ, 6)
n = 1 << int(os.environ.get("LOG2N", "24"))
x = f.random_dev(n, 7)
x[12345] = 0
x[0] = 0
out, chk = f.alloc(n), f.alloc(n)
flags = torch.empty(n, dtype=torch.uint8, device="cuda")
small = f.inv_workspace_bytes(n)
ws_small = torch.empty(small, dtype=torch.uint8, device="cuda")
ws_big = torch.empty(small + n * 64 + 4096, dtype=torch.uint8, device="cuda")
one = torch.from_numpy(f.one_words.view(np.int64)).cuda()
idx = torch.arange(0, n, max(1, n // 512), device="cuda")
want, want_flags = o.batch_inv(x[idx].cpu().numpy().view(np.uint64))


def run(label, ws, **hooks):
    for k, v in hooks.items():
        _lib.tuning_set(k, v)
    try:
        ws = torch.empty(f.inv_workspace_bytes(n, fast=ws is ws_big), dtype=torch.uint8, device="cuda")   # sized under this variant's hooks
        out.zero_()
        f.inv_dev(x, out, flags, ws)
        torch.cuda.synchronize()
        f.mul_dev(x, out, chk)
        nz = flags == 0
        ok = bool(torch.equal(chk[nz], one.expand(int(nz.sum()), 6))) and int((~nz).sum()) == 2
        ok = ok and np.array_equal(out[idx].cpu().numpy().view(np.uint64), want) and np.array_equal(flags[idx].cpu().numpy(), want_flags)
        best = None
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f.inv_dev(x, out, flags, ws)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            best = ms if best is None else min(best, ms)
        print({"variant": label, "ms": round(best, 4), "Ginv_per_s": round(n / best / 1e6, 2), "ok": ok, **hooks}, flush=True)
    finally:
        for k in hooks:
            _lib.tuning_set(k)


only = os.environ.get("ONLY")
exact = os.environ.get("EXACT")
variants = [
    ("default (fused top, words)", ws_small, {}),
    ("unfused top", ws_small, {"FPGPU_INV_FUSE_TOP": 0}),
    ("limb prefixes", ws_big, {}),
    ("limb prefixes, unfused top", ws_big, {"FPGPU_INV_FUSE_TOP": 0}),
    ("C0 = 64", ws_small, {"FPGPU_INV_C0": 64}),
    ("C0 = 64, C = 4", ws_small, {"FPGPU_INV_C0": 64, "FPGPU_INV_C": 4}),
    ("C0 = 128, C = 4", ws_small, {"FPGPU_INV_C0": 128, "FPGPU_INV_C": 4}),
    ("C0 = 256", ws_small, {"FPGPU_INV_C0": 256}),
]
for label, ws, hooks in variants:
    if (only and only not in label) or (exact and exact != label):
        continue
    run(label, ws, **hooks)
