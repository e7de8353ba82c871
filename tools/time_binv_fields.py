"""Batch inversion of 2^24 elements by field width: ms, ns per element, and the share of a 3 x (2 n32^2 + n32) multiply-add
roofline at 39.3 T lane-slots/s (the figure bench.py quotes for BLS12-381)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
fields = [("goldilocks", 2**64 - 2**32 + 1, 1), ("p128", 2**127 - 1, 2), ("p192", 2**192 - 2**64 - 1, 3), ("bn254_fr", cpa.BN254_FR, 4),
          ("secp256k1", 2**256 - 2**32 - 977, 4), ("p320", 2**320 - 197, 5), ("bls12_381_fq", cpa.BLS12_381_FQ, 6), ("p448", 2**448 - 2**224 - 1, 7),
          ("p512", 2**512 - 569, 8)]
n = 1 << 24
for name, p, n64 in fields:
    f = cpa.MontyField(p, n64)
    x = f.random_dev(n, 3); out = f.alloc(n); chk = f.alloc(n)
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ws = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
    for _ in range(5): f.inv_dev(x, out, flags, ws)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f.inv_dev(x, out, flags, ws)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    f.mul_dev(x, out, chk)
    one = torch.from_numpy(__import__("numpy").array([[(f.one_int >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n64)]], dtype="uint64").view("int64")).cuda() if hasattr(f, "one_int") else None
    n32 = 2 * n64
    mads = 3 * (2 * n32 * n32 + n32)
    print(json.dumps({"field": name, "n64": n64, "ms": round(best, 4), "ns_per_element": round(best * 1e6 / n, 4),
                      "valu_frac_of_3_mul_roofline": round(mads * n / (best * 1e-3) / 39.3e12, 3),
                      "hbm_TBps_in_out": round(2 * 8 * n64 * n / best / 1e9, 2), "flags_set": int(flags.sum())}), flush=True)
    del x, out, chk, flags, ws
    torch.cuda.empty_cache()
