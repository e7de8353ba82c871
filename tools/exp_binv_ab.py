"""Same-box A/B of the batch inversion: FPGPU_LIB selects the library (build/libfpgpu_r2.so = round 2's kernels);
MODULUS = bls | bn254 | secp (tight) | m512; minimum workspace and, where the library has it, the fast one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import crypto_primitives_amd as cpa

MODS = {"bls": (cpa.BLS12_381_FQ, 6), "bn254": (cpa.BN254_FR, 4), "secp": ((1 << 256) - (1 << 32) - 977, 4),
        "m512": ((1 << 512) - 569, 8), "gold": (cpa.GOLDILOCKS, 1)}
p, n64 = MODS[os.environ.get("MODULUS", "bls")]
f = cpa.MontyField(p, n64)
n = 1 << int(os.environ.get("LOG2N", "24"))
x = f.random_dev(n, 7)
out, chk = f.alloc(n), f.alloc(n)
flags = torch.empty(n, dtype=torch.uint8, device="cuda")
one = torch.from_numpy(f.one_words.view(np.int64)).cuda()
sizes = {"min": f._lib.fpgpu_batch_inv_workspace_bytes(f._h, n)}
if hasattr(f._lib, "fpgpu_batch_inv_fast_workspace_bytes"):
    try:
        sizes["fast"] = f.inv_workspace_bytes(n, fast=True)
    except Exception:
        pass
for label, sz in sizes.items():
    ws = torch.empty(sz, dtype=torch.uint8, device="cuda")
    f.inv_dev(x, out, flags, ws)
    f.mul_dev(x, out, chk)
    ok = bool(torch.equal(chk[flags == 0], one.expand(int((flags == 0).sum()), n64)))
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f.inv_dev(x, out, flags, ws)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print({"lib": os.path.basename(os.environ.get("FPGPU_LIB", "tree")), "modulus": os.environ.get("MODULUS", "bls"), "workspace": label,
           "ms": round(best, 4), "Ginv_per_s": round(n / best / 1e6, 2), "ok": ok}, flush=True)
