"""Mat-vec rate by field width (8 GiB matrices, square-ish): matrix-core path (workspace), the two VALU kernels (k_matvec:
"valu", k_matvec_cols: "cols"), TB/s of matrix bytes.  configs[3] is BN254; this shows what the other widths get from the same kernels."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
fields = [("goldilocks", 2**64 - 2**32 + 1, 1), ("p128", 2**127 - 1, 2), ("p192", 2**192 - 2**64 - 1, 3), ("bn254_fr", cpa.BN254_FR, 4),
          ("secp256k1", 2**256 - 2**32 - 977, 4), ("p320", 2**320 - 197, 5), ("bls12_381_fq", cpa.BLS12_381_FQ, 6), ("p448", 2**448 - 2**224 - 1, 7),
          ("p512", 2**512 - 569, 8)]
only = sys.argv[1:]
for name, p, n64 in fields:
    if only and name not in only:
        continue
    f = cpa.MontyField(p, n64)
    elems = (8 << 30) // (8 * n64)
    cols = 1 << ((elems.bit_length() - 1) // 2)
    rows = elems // cols
    A = f.random_dev(rows * cols, 1); x = f.random_dev(cols, 2); y = f.alloc(rows); y2 = f.alloc(rows)
    ws = torch.empty(max(16, f.matvec_workspace_bytes(rows, cols)), dtype=torch.uint8, device="cuda")
    out = {"field": name, "n64": n64, "rows": rows, "cols": cols}
    y3 = f.alloc(rows)
    for label, w, dst in (("mfma", ws, y), ("valu", None, y2), ("cols", None, y3)):
        _lib.tuning_set("FPGPU_MATVEC_COLS", 1 if label == "cols" else 0)
        for _ in range(2): f.matvec_dev(A, rows, cols, x, dst, workspace=w)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f.matvec_dev(A, rows, cols, x, dst, workspace=w)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[label + "_ms"] = round(ms, 3); out[label + "_TBps"] = round(8.0 * n64 * rows * cols / ms / 1e9, 2)
    out["equal"] = bool(torch.equal(y, y2)) and bool(torch.equal(y, y3))
    print(json.dumps(out), flush=True)
    del A, x, y, y2, y3, ws
    torch.cuda.empty_cache()
