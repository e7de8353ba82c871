"""Multi-limb mat-mul (BN254) over awkward shapes: G terms/s; looking for cliffs (16 x 16 output tiles, one per workgroup)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
for m, k, n in ((1024, 1024, 1024), (4096, 4096, 16), (16, 4096, 4096), (16, 1 << 20, 16), (64, 1 << 18, 64), (4096, 4096, 1), (1, 4096, 4096),
                (256, 65536, 256), (1000, 1000, 1000), (2048, 64, 2048), (8192, 8, 8192), (1, 1 << 10, 1 << 20), (1, 1 << 16, 1 << 14), (4, 4096, 1 << 16), (1, 1 << 20, 1024)):
    A = f.random_dev(m * k, 1); B = f.random_dev(k * n, 2); C = f.alloc(m * n)
    wb = f.matmul_workspace_bytes(m, k, n)
    ws = torch.empty(wb, dtype=torch.uint8, device="cuda") if wb else None
    for _ in range(2): f.matmul_dev(A, B, C, m, k, n, workspace=ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f.matmul_dev(A, B, C, m, k, n, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"m": m, "k": k, "n": n, "ms": round(ms, 3), "G_terms_per_s": round(m * k * n / ms / 1e6, 1),
                      "tiles": ((m + 15) // 16) * ((n + 15) // 16), "workspace_bytes": wb}), flush=True)
    del A, B, C
