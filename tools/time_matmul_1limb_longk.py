"""Single-limb (Goldilocks) mat-mul with k beyond 16320: limb kernel on the matrix cores (workspace) against the VALU kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.GOLDILOCKS, 1)
def timed(fn, reps=4):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for m, k, n in ((1024, 32768, 1024), (2048, 20000, 2048), (4096, 16384, 4096), (4096, 32768, 4096), (256, 65536, 256), (8192, 8192, 8192)):
    A, B, C0, C1 = f.random_dev(m * k, 1), f.random_dev(k * n, 2), f.alloc(m * n), f.alloc(m * n)
    t0 = timed(lambda: f.matmul_dev(A, B, C0, m, k, n))
    nb = f.matmul_workspace_bytes(m, k, n)
    ws = torch.empty(max(256, nb), dtype=torch.uint8, device="cuda")
    t1 = timed(lambda: f.matmul_dev(A, B, C1, m, k, n, workspace=ws))
    print(json.dumps({"shape": [m, k, n], "valu_ms": round(t0, 3), "workspace_ms": round(t1, 3), "T_terms_per_s": round(m * k * n / t1 / 1e9, 2),
                      "same": bool(torch.equal(C0, C1)), "workspace_MiB": round(nb / 2**20, 1)}), flush=True)
    del A, B, C0, C1, ws
