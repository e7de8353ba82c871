"""Mat-vec over extreme aspect ratios (BN254 and BLS12-381, ~4 GiB matrices): TB/s of matrix bytes with the workspace
(matrix cores where the plan allows) and without (VALU kernel).  Looking for cliffs."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, p, n64 in (("bn254_fr", cpa.BN254_FR, 4), ("bls12_381_fq", cpa.BLS12_381_FQ, 6)):
    f = cpa.MontyField(p, n64)
    total = (4 << 30) // (8 * n64)
    for cols in (16, 64, 256, 1000, 4096, 65536, 200000, 262144, 1 << 20, 1 << 22, total // 16, total // 3, total):
        rows = max(1, total // cols)
        A = f.random_dev(rows * cols, 1); x = f.random_dev(cols, 2); y = f.alloc(rows); y2 = f.alloc(rows)
        wb = f.matvec_workspace_bytes(rows, cols)
        ws = torch.empty(max(16, wb), dtype=torch.uint8, device="cuda")
        t1 = timed(lambda: f.matvec_dev(A, rows, cols, x, y, workspace=ws))
        t2 = timed(lambda: f.matvec_dev(A, rows, cols, x, y2))
        b = 8.0 * n64 * rows * cols
        print(json.dumps({"field": name, "rows": rows, "cols": cols, "matrix_cores": wb > 0, "with_ws_TBps": round(b / t1 / 1e9, 2),
                          "no_ws_TBps": round(b / t2 / 1e9, 2), "equal": bool(torch.equal(y, y2))}), flush=True)
        del A, x, y, y2, ws
        torch.cuda.empty_cache()
