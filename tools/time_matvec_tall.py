"""Tall, narrow mat-vec (~4 GiB matrices): one row per lane (k_matvec_rows) against the matrix-core path (workspace) and
one row per wave (k_matvec_cols / k_matvec, no workspace).  TB/s of matrix bytes."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, p, n64 in (("goldilocks", 2**64 - 2**32 + 1, 1), ("bn254_fr", cpa.BN254_FR, 4), ("bls12_381_fq", cpa.BLS12_381_FQ, 6), ("p512", 2**512 - 569, 8)):
    f = cpa.MontyField(p, n64)
    total = (4 << 30) // (8 * n64)
    for cols in (1, 4, 16, 33, 64, 128, 256, 512, 1024):
        rows = total // cols
        A = f.random_dev(rows * cols, 1); x = f.random_dev(cols, 2); y = [f.alloc(rows) for _ in range(3)]
        ws = torch.empty(max(16, f.matvec_workspace_bytes(rows, cols)), dtype=torch.uint8, device="cuda")
        b = 8.0 * n64 * rows * cols
        out = {"field": name, "rows": rows, "cols": cols}
        _lib.tuning_set("FPGPU_MATVEC_ROWS", 1)
        out["row_per_lane"] = round(b / timed(lambda: f.matvec_dev(A, rows, cols, x, y[0])) / 1e9, 2)
        _lib.tuning_set("FPGPU_MATVEC_ROWS", 0)
        out["matrix_cores"] = round(b / timed(lambda: f.matvec_dev(A, rows, cols, x, y[1], workspace=ws)) / 1e9, 2)
        out["row_per_wave"] = round(b / timed(lambda: f.matvec_dev(A, rows, cols, x, y[2])) / 1e9, 2)
        _lib.tuning_set("FPGPU_MATVEC_ROWS")
        out["equal"] = bool(torch.equal(y[0], y[1])) and bool(torch.equal(y[0], y[2]))
        print(json.dumps(out), flush=True)
        del A, x, y, ws
        torch.cuda.empty_cache()
