"""Times fpgpu_matmul_dev (BASELINE.json configs[4] shape) and spot-checks it against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
from oracle.fp_oracle import Oracle
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
f = cpa.MontyField(cpa.GOLDILOCKS, 1)
o = Oracle(cpa.GOLDILOCKS, 1)
A = f.random_dev(dim * dim, 1); B = f.random_dev(dim * dim, 2); C = f.alloc(dim * dim)
ws = None
if os.environ.get("NO_WS") != "1":
    ws = torch.empty(f.matmul_workspace_bytes(dim, dim, dim), dtype=torch.uint8, device="cuda")
f.matmul_dev(A, B, C, dim, dim, dim, workspace=ws); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
reps = 3
e0.record()
for _ in range(reps): f.matmul_dev(A, B, C, dim, dim, dim, workspace=ws)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
# check a few full output rows against the oracle
rows = [0, 1, dim // 2 + 3, dim - 1]
hB = B.cpu().numpy().view(np.uint64)
ok = True
for r in rows:
    hA = A[r * dim:(r + 1) * dim].cpu().numpy().view(np.uint64)
    exp = o.matmul(hA, hB, 1, dim, dim, threads=0) if dim <= 2048 else o.matmul(hA, hB[:, :].reshape(dim, dim)[:, :64].copy().reshape(-1, 1), 1, dim, 64, threads=0)
    got = C[r * dim:(r + 1) * dim].cpu().numpy().view(np.uint64)
    got = got if dim <= 2048 else got[:64]
    ok &= bool(np.array_equal(got.reshape(-1), exp.reshape(-1)))
print({"workspace": ws is not None, "dim": dim, "ms": ms, "Gterms_per_s": dim ** 3 / ms / 1e6, "mad_Tops": dim ** 3 * 10 / ms / 1e9, "ok": ok})
