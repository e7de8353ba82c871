"""Matrix-core mat-vec against the number of column parts per row block (FPGPU_MATVEC_KSPLIT), for row counts that do and
do not fill 256 CUs evenly: the data behind make_mv_plan's choice."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib
cases = [("bls12_381_fq", cpa.BLS12_381_FQ, 6, 21760, 8192), ("bls12_381_fq", cpa.BLS12_381_FQ, 6, 21845, 8192),
         ("bn254_fr", cpa.BN254_FR, 4, 21760, 12288), ("bn254_fr", cpa.BN254_FR, 4, 32768, 32768), ("bn254_fr", cpa.BN254_FR, 4, 33000, 8192),
         ("bn254_fr", cpa.BN254_FR, 4, 4096, 32768)]
for name, p, n64, rows, cols in cases:
    f = cpa.MontyField(p, n64)
    A = f.random_dev(rows * cols, 1); x = f.random_dev(cols, 2); y = f.alloc(rows)
    ws = torch.empty(max(16, f.matvec_workspace_bytes(rows, cols)), dtype=torch.uint8, device="cuda")
    out = {"field": name, "rows": rows, "cols": cols, "rowblocks": (rows + 127) // 128}
    for ks in (0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32):
        _lib.tuning_set("FPGPU_MATVEC_KSPLIT", ks) if ks else _lib.tuning_set("FPGPU_MATVEC_KSPLIT")
        for _ in range(2): f.matvec_dev(A, rows, cols, x, y, workspace=ws)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f.matvec_dev(A, rows, cols, x, y, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out["plan" if ks == 0 else "ks%d" % ks] = round(8.0 * n64 * rows * cols / ms / 1e9, 2)
    _lib.tuning_set("FPGPU_MATVEC_KSPLIT")
    print(json.dumps(out), flush=True)
    del A, x, y, ws
    torch.cuda.empty_cache()
