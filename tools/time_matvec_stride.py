"""Is the matrix-core mat-vec's rate a matter of the ROW STRIDE (cols x element bytes) rather than of the field width?
Same kernel, ~8 GiB matrices, BN254 (32 B) and BLS12-381 (48 B) at several column counts."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
cases = [("bn254_fr", cpa.BN254_FR, 4, c) for c in (16384, 12288, 12000, 8192, 6144, 24576)] + \
        [("bls12_381_fq", cpa.BLS12_381_FQ, 6, c) for c in (8192, 8000, 10922, 10924, 12288, 16384, 5462, 4096)]
for name, p, n64, cols in cases:
    f = cpa.MontyField(p, n64)
    rows = ((8 << 30) // (8 * n64) // cols) // 128 * 128
    A = f.random_dev(rows * cols, 1); x = f.random_dev(cols, 2); y = f.alloc(rows)
    ws = torch.empty(max(16, f.matvec_workspace_bytes(rows, cols)), dtype=torch.uint8, device="cuda")
    for _ in range(2): f.matvec_dev(A, rows, cols, x, y, workspace=ws)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f.matvec_dev(A, rows, cols, x, y, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(json.dumps({"field": name, "rows": rows, "cols": cols, "row_stride_bytes": cols * 8 * n64, "ms": round(ms, 3),
                      "TBps": round(8.0 * n64 * rows * cols / ms / 1e9, 2)}), flush=True)
    del A, x, y, ws
    torch.cuda.empty_cache()
