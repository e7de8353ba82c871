"""Times fpgpu_matvec_dev for several (rows, cols) shapes with the same matrix size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
total = 1 << 30
A = f.random_dev(total, 1)
for lc in (15, 13, 11, 10, 17):
    cols = 1 << lc; rows = total // cols
    x = f.random_dev(cols, 2); y = f.alloc(rows)
    f.matvec_dev(A, rows, cols, x, y); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f.matvec_dev(A, rows, cols, x, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print({"rows": rows, "cols": cols, "ms": round(ms, 3), "GBps": round(32.0 * total / ms / 1e6)})
