"""Batch inversion with sparse level-0 prefixes (FPGPU_INV_SP) against the dense sweeps, by field width and length."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
fields = [("goldilocks", 2**64 - 2**32 + 1, 1), ("p128", 2**127 - 1, 2), ("p192", 2**192 - 2**64 - 1, 3), ("bn254_fr", cpa.BN254_FR, 4),
          ("secp256k1", 2**256 - 2**32 - 977, 4), ("p320", 2**320 - 197, 5), ("bls12_381_fq", cpa.BLS12_381_FQ, 6), ("p512", 2**512 - 569, 8)]
for name, p, n64 in fields:
    f = cpa.MontyField(p, n64)
    for n in (1 << 24, 3_000_017, 1 << 20):
        x = f.random_dev(n, 3); x[5] = 0; x[n - 1] = 0
        outs = [f.alloc(n), f.alloc(n)]
        flags = [torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
        ws = torch.empty(f.inv_workspace_bytes(n, fast=False), dtype=torch.uint8, device="cuda")   # minimum workspace: no limb prefixes
        row = {"field": name, "n": n}
        for sp in (0, 1):
            _lib.tuning_set("FPGPU_INV_SP", sp)
            for _ in range(5): f.inv_dev(x, outs[sp], flags[sp], ws)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f.inv_dev(x, outs[sp], flags[sp], ws)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            row["sparse_ms" if sp else "dense_ms"] = round(best, 4)
        _lib.tuning_set("FPGPU_INV_SP")
        row["equal"] = bool(torch.equal(outs[0], outs[1])) and bool(torch.equal(flags[0], flags[1])) and int(flags[0].sum()) == 2
        print(json.dumps(row), flush=True)
        del x, outs, flags, ws
        torch.cuda.empty_cache()
