"""Mat-vec at the per-rank shapes of configs[3] over 1 / 2 / 4 / 8 GPUs (rows = 2^15 / N, cols = 2^15, BN254): does a
rank's row block still fill the chip?  Event-timed, workspace path (matrix cores)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
cols = 1 << 15
A = f.random_dev(cols * cols, 1)
x = f.random_dev(cols, 2)
for n_gpus in (1, 2, 4, 8, 16):
    rows = cols // n_gpus
    y = f.alloc(rows)
    ws = torch.empty(f.matvec_workspace_bytes(rows, cols), dtype=torch.uint8, device="cuda")
    Ar = A[: rows * cols]
    for _ in range(3): f.matvec_dev(Ar, rows, cols, x, y, workspace=ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 10 * n_gpus
    e0.record()
    for _ in range(reps): f.matvec_dev(Ar, rows, cols, x, y, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"ranks": n_gpus, "rows": rows, "ms": round(ms, 4), "TBps": round(32.0 * rows * cols / ms / 1e9, 3),
                      "x_ranks_ideal_ms": round(ms * n_gpus, 3)}), flush=True)
