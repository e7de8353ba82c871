"""Times the shapes one rank gets when BASELINE.json's configs 3-5 are split over 1, 2, 4, 8 GPUs (all on this GPU):
what strong scaling can reach before any communication."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa

def timed(fn, steps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps

f = cpa.MontyField(cpa.BN254_FR, 4)
dim = 1 << 15
A = f.random_dev(dim * dim, 1); x = f.random_dev(dim, 2)
base = None
for g in (1, 2, 4, 8):
    rows = dim // g
    y = f.alloc(rows)
    ws = torch.empty(f.matvec_workspace_bytes(rows, dim), dtype=torch.uint8, device="cuda")
    ms = timed(lambda: f.matvec_dev(A[: rows * dim], rows, dim, x, y, ws))
    base = base or ms
    print(json.dumps({"config": "matvec 2^15 x 2^15 BN254", "gpus": g, "rows_per_rank": rows, "ms": round(ms, 4), "TBps": round(rows * dim * 32 / ms / 1e9, 2),
                      "compute_only_efficiency": round(base / g / ms, 3)}), flush=True)
del A
torch.cuda.empty_cache()
fq = cpa.MontyField(cpa.BLS12_381_FQ, 6)
base = None
for g in (1, 2, 4, 8):
    n = (1 << 24) // g
    xx = fq.random_dev(n, 3); out = fq.alloc(n); flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    ws = torch.empty(fq.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    ms = timed(lambda: fq.inv_dev(xx, out, flags, ws), 10, 3)
    base = base or ms
    print(json.dumps({"config": "batch inversion 2^24 BLS12-381", "gpus": g, "elements_per_rank": n, "ms": round(ms, 4),
                      "compute_only_efficiency": round(base / g / ms, 3)}), flush=True)
gl = cpa.MontyField(cpa.GOLDILOCKS, 1)
d = 8192
Am = gl.random_dev(d * d, 4); Bm = gl.random_dev(d * d, 5)
base = None
for g in (1, 2, 4, 8):
    m = d // g
    C = gl.alloc(m * d)
    ws = torch.empty(gl.matmul_workspace_bytes(m, d, d), dtype=torch.uint8, device="cuda")
    ms = timed(lambda: gl.matmul_dev(Am[: m * d], Bm, C, m, d, d, ws), 3, 1)
    base = base or ms
    print(json.dumps({"config": "matmul 8192^3 Goldilocks", "gpus": g, "rows_per_rank": m, "ms": round(ms, 4),
                      "compute_only_efficiency": round(base / g / ms, 3)}), flush=True)
