"""Times fpgpu_matvec_dev (matrix-core path with a workspace, VALU path without) for several
(rows, cols) shapes of the same 2^30-element BN254 matrix, and checks the two paths agree."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
total = 1 << 30
A = f.random_dev(total, 1)
shapes = [int(a) for a in sys.argv[1:]] or [15, 12, 17]
for lc in shapes:
    cols = 1 << lc; rows = total // cols
    x = f.random_dev(cols, 2); y = f.alloc(rows); y2 = f.alloc(rows)
    ws = torch.empty(f.matvec_workspace_bytes(rows, cols), dtype=torch.uint8, device="cuda")
    out = {"rows": rows, "cols": cols}
    for name, w, dst in (("mfma", ws, y), ("valu", None, y2)):
        f.matvec_dev(A, rows, cols, x, dst, workspace=w); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f.matvec_dev(A, rows, cols, x, dst, workspace=w)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out[name + "_ms"] = round(ms, 3); out[name + "_GBps"] = round(32.0 * total / ms / 1e6)
    out["equal"] = bool(torch.equal(y, y2))
    print(out, flush=True)
