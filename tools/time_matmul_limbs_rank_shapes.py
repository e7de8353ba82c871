"""What one rank of g sees of a row-sharded multi-limb mat-mul (BN254 2048^3 and 4096^3: A and C split by rows, B whole):
the limb kernel on rows / g, event-timed; efficiency = t_1 / (g t_g)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dim in (2048, 4096):
    B = f.random_dev(dim * dim, 2)
    t1 = None
    for g in (1, 2, 4, 8):
        m = dim // g
        A, C = f.random_dev(m * dim, 1), f.alloc(m * dim)
        ws = torch.empty(max(256, f.matmul_workspace_bytes(m, dim, dim)), dtype=torch.uint8, device="cuda")
        t = timed(lambda: f.matmul_dev(A, B, C, m, dim, dim, workspace=ws))
        t1 = t1 or t
        print(json.dumps({"dim": dim, "ranks": g, "rows_per_rank": m, "ms": round(t, 3), "efficiency": round(t1 / (g * t), 3)}), flush=True)
        del A, C, ws
