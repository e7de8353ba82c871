"""More shape sweeps (BN254): sparse mat-vec by density and row count, folds / dot by length, pow by length."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
def timed(fn, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g = torch.Generator(device="cuda"); g.manual_seed(7)
cells = 1 << 25
for density, cols in ((1, 1 << 20), (2, 1 << 20), (8, 1 << 20), (32, 1 << 20), (128, 1 << 20), (1024, 1 << 20), (32768, 1 << 20), (32, 1 << 10), (32, 1 << 26)):
    rows = cells // density
    vals = f.random_dev(rows * density, 1); x = f.random_dev(cols, 2); y = f.alloc(rows)
    col = torch.randint(0, cols, (rows * density,), dtype=torch.int32, device="cuda", generator=g)
    ms = timed(lambda: f.spmv_dev(vals, col, rows, cols, density, x, y))
    print(json.dumps({"op": "spmv", "rows": rows, "density": density, "cols": cols, "ms": round(ms, 3), "G_cells_per_s": round(cells / ms / 1e6, 1)}), flush=True)
    del vals, x, y, col
wsf = torch.empty(max(16, f.fold_workspace_bytes()), dtype=torch.uint8, device="cuda")
s = f.alloc(1)
for n in (1 << 28, 1 << 24, 1 << 20, 100_000, 10_000, 1000):
    a = f.random_dev(n, 1); b = f.random_dev(n, 2)
    out = {"op": "folds", "n": n}
    for label, fn, nb in (("sum", lambda: f.sum_dev(a, s, workspace=wsf), 32), ("product", lambda: f.product_dev(a, s, workspace=wsf), 32), ("dot", lambda: f.dot_dev(a, b, s, workspace=wsf), 64)):
        ms = timed(fn)
        out[label + "_us"] = round(ms * 1e3, 1); out[label + "_TBps"] = round(nb * n / ms / 1e9, 2)
    print(json.dumps(out), flush=True)
    del a, b
