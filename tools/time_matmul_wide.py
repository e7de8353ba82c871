"""Times the multi-limb mat-mul (k_matmul<N>, lazy reduction) on BN254 Fr and BLS12-381 Fq and
checks (A B) v == A (B v) through the library itself (parity with the oracle: tests/).  Usage: time_matmul_wide.py [dim]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, p, n64 in (("bn254_fr", cpa.BN254_FR, 4), ("bls12_381_fq", cpa.BLS12_381_FQ, 6)):
    f = cpa.MontyField(p, n64)
    A = f.random_dev(dim * dim, 1); B = f.random_dev(dim * dim, 2); C = f.alloc(dim * dim)
    f.matmul_dev(A, B, C, dim, dim, dim); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f.matmul_dev(A, B, C, dim, dim, dim)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    v = f.random_dev(dim, 3); bv = f.alloc(dim); abv = f.alloc(dim); cv = f.alloc(dim)
    f.matvec_dev(B, dim, dim, v, bv); f.matvec_dev(A, dim, dim, bv, abv); f.matvec_dev(C, dim, dim, v, cv)
    torch.cuda.synchronize()
    ok = bool(torch.equal(cv, abv))
    print({"field": name, "dim": dim, "ms": round(ms, 3), "Gterms_per_s": round(dim ** 3 / ms / 1e6, 1), "ok": ok}, flush=True)
