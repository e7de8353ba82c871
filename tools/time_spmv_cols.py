"""Sparse mat-vec time against the size of x (how much of the gather cost is cache residency): 2^20 rows x 32 cells, BN254."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
rows, density = 1 << 20, 32
vals = f.random_dev(rows * density, 0x5A)
y = f.alloc(rows)
for lg in (12, 14, 16, 17, 18, 19, 20, 22):
    cols = 1 << lg
    x = f.random_dev(cols, 0x5B)
    g = torch.Generator(device="cuda"); g.manual_seed(1234)
    col = torch.randint(0, cols, (rows * density,), dtype=torch.int32, device="cuda", generator=g)
    for _ in range(3): f.spmv_dev(vals, col, rows, cols, density, x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f.spmv_dev(vals, col, rows, cols, density, x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"log2_cols": lg, "x_MiB": cols * 32 / 2**20, "ms": round(ms, 4), "stream_TBps": round(36 * rows * density / ms / 1e9, 2)}), flush=True)
