"""Looking for shape cliffs outside the BASELINE configs: batch inversion by length (ns per element should be flat
above ~2^22) and the single-limb matrix-core mat-mul by shape (POP/s should not depend on how the tiles divide)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "inv"):
    for name, p, n64 in (("bls12_381_fq", cpa.BLS12_381_FQ, 6), ("bn254_fr", cpa.BN254_FR, 4)):
        f = cpa.MontyField(p, n64)
        for n in (1 << 24, 10_000_000, (1 << 23) + 1, 5_000_000, 3_000_017, (1 << 21) + 1, 1 << 21, 1_500_000, (1 << 20) + 1):
            x = f.random_dev(n, 3); out = f.alloc(n)
            flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
            ws = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
            ms = timed(lambda: f.inv_dev(x, out, flags, ws), 10)
            print(json.dumps({"op": "batch_inv", "field": name, "n": n, "ms": round(ms, 4), "ns_per_element": round(ms * 1e6 / n, 4)}), flush=True)
            del x, out, flags, ws
if what in ("all", "matmul"):
    f = cpa.MontyField(2**64 - 2**32 + 1, 1)
    for m, k, n in ((8192, 8192, 8192), (4096, 8192, 8192), (8192, 8192, 4096), (8192, 4096, 8192), (6000, 6000, 6000), (5000, 8192, 5000),
                    (8192, 1024, 8192), (1000, 16000, 1000), (2048, 2048, 2048), (8200, 8192, 8200), (1024, 8192, 8192), (8192, 8192, 1024)):
        A = f.random_dev(m * k, 1); B = f.random_dev(k * n, 2); C = f.alloc(m * n)
        ws = torch.empty(max(16, f.matmul_workspace_bytes(m, k, n)), dtype=torch.uint8, device="cuda")
        ms = timed(lambda: f.matmul_dev(A, B, C, m, k, n, workspace=ws), 3)
        print(json.dumps({"op": "matmul_goldilocks", "m": m, "k": k, "n": n, "ms": round(ms, 3), "int8_POPs": round(128.0 * m * k * n / ms / 1e12, 3),
                          "T_terms_per_s": round(m * k * n / ms / 1e9, 2)}), flush=True)
        del A, B, C, ws
        torch.cuda.empty_cache()
