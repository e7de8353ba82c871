"""Times the sparse mat-vec of bench.py (2^20 rows x 32 cells over 2^20 columns, BN254) for the library named by FPGPU_LIB."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
import bench
f = cpa.MontyField(cpa.BN254_FR, 4)
def time_steps(fn, steps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / steps / 1e3
    return t * steps, t
for cols_log2 in (20, 16):
    bench_spmv = bench.bench_spmv
    r = bench_spmv(f, 0, time_steps) if cols_log2 == 20 else None
    if r: print(json.dumps({"lib": os.environ.get("FPGPU_LIB", "default"), "ms": round(r["ms"], 4), "Gcells": round(r["cells_per_s"] / 1e9, 2)}), flush=True)
