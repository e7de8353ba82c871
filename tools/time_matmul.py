"""Times fpgpu_matmul_dev (BASELINE.json configs[4] shape) and checks (A B) v == A (B v)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
f = cpa.MontyField(cpa.GOLDILOCKS, 1)
# BALLAST=<GiB>: memory held (and touched) before the operands are allocated, as in bench.py's extras
ballast = [torch.zeros(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(int(os.environ.get("BALLAST", "0")))]
A = f.random_dev(dim * dim, 1); B = f.random_dev(dim * dim, 2); C = f.alloc(dim * dim)
ws = None
if os.environ.get("NO_WS") != "1":
    ws = torch.empty(f.matmul_workspace_bytes(dim, dim, dim), dtype=torch.uint8, device="cuda")
f.matmul_dev(A, B, C, dim, dim, dim, workspace=ws); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("REPS", "3"))
e0.record()
for _ in range(reps): f.matmul_dev(A, B, C, dim, dim, dim, workspace=ws)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
# consistency through the library itself, (A B) v == A (B v) for a random vector; parity with the
# oracle is the tests' job (tests/test_gpu_parity.py, tests/test_gpu_full_size.py)
v = f.random_dev(dim, 3); bv = f.alloc(dim); abv = f.alloc(dim); cv = f.alloc(dim)
f.matvec_dev(B, dim, dim, v, bv); f.matvec_dev(A, dim, dim, bv, abv); f.matvec_dev(C, dim, dim, v, cv)
torch.cuda.synchronize()
ok = bool(torch.equal(cv, abv))
print({"workspace": ws is not None, "dim": dim, "ms": ms, "Gterms_per_s": dim ** 3 / ms / 1e6, "mad_Tops": dim ** 3 * 10 / ms / 1e9, "ok": ok})
