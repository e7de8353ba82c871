"""configs[3] (2^15 x 2^15 BN254) against the column split, interleaved repeats: does a finer grid than one workgroup per CU pay?"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
f = cpa.MontyField(cpa.BN254_FR, 4)
rows = cols = 1 << 15
A = f.random_dev(rows * cols, 1); x = f.random_dev(cols, 2); y = f.alloc(rows)
ws = torch.empty(f.matvec_workspace_bytes(rows, cols), dtype=torch.uint8, device="cuda")
splits = (1, 2, 4, 8, 12, 16, 20, 24, 32, 48, 64)
res = {k: [] for k in splits}
for _ in range(10): f.matvec_dev(A, rows, cols, x, y, workspace=ws)
for rep in range(4):
    for ks in splits:
        _lib.tuning_set("FPGPU_MATVEC_KSPLIT", ks)
        for _ in range(3): f.matvec_dev(A, rows, cols, x, y, workspace=ws)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f.matvec_dev(A, rows, cols, x, y, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        res[ks].append(round(e0.elapsed_time(e1) / 10, 3))
_lib.tuning_set("FPGPU_MATVEC_KSPLIT")
for ks in splits:
    print(json.dumps({"ksplit": ks, "ms": res[ks], "best": min(res[ks])}), flush=True)
