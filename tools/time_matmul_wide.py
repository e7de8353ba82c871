"""Times the multi-limb mat-mul (k_matmul<N>, lazy reduction) on BN254 Fr and BLS12-381 Fq and
spot-checks rows against the oracle.  Usage: time_matmul_wide.py [dim]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
from oracle.fp_oracle import Oracle
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name, p, n64 in (("bn254_fr", cpa.BN254_FR, 4), ("bls12_381_fq", cpa.BLS12_381_FQ, 6)):
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    A = f.random_dev(dim * dim, 1); B = f.random_dev(dim * dim, 2); C = f.alloc(dim * dim)
    f.matmul_dev(A, B, C, dim, dim, dim); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f.matmul_dev(A, B, C, dim, dim, dim)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    hB = B.cpu().numpy().view(np.uint64).reshape(dim, dim, n64)[:, :16].copy().reshape(-1, n64)
    ok = True
    for r in (0, dim // 2 + 3, dim - 1):
        hA = A[r * dim:(r + 1) * dim].cpu().numpy().view(np.uint64)
        exp = o.matmul(hA, hB, 1, dim, 16, threads=0)
        got = C[r * dim:r * dim + 16].cpu().numpy().view(np.uint64)
        ok &= bool(np.array_equal(got.reshape(-1), exp.reshape(-1)))
    print({"field": name, "dim": dim, "ms": round(ms, 3), "Gterms_per_s": round(dim ** 3 / ms / 1e6, 1), "ok": ok}, flush=True)
