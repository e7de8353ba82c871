"""Same-box A/B of two builds of the library on the multi-limb matrix-core mat-mul (BN254): each build in its own process
(FPGPU_LIB), alternating, event-timed.  usage: python tools/ab_limb_staging.py build/libfpgpu_<variant>.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
out = {}
for m, k, n in ((2048, 2048, 2048), (4096, 1024, 4096), (3000, 500, 3000)):
    A, B, C = f.random_dev(m * k, 1), f.random_dev(k * n, 2), f.alloc(m * n)
    ws = torch.empty(max(256, f.matmul_workspace_bytes(m, k, n)), dtype=torch.uint8, device="cuda")
    for _ in range(3): f.matmul_dev(A, B, C, m, k, n, workspace=ws)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f.matmul_dev(A, B, C, m, k, n, workspace=ws); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    out["%%dx%%dx%%d" %% (m, k, n)] = [round(ts[0], 3), round(ts[len(ts) // 2], 3)]
print(json.dumps(out))
''' % ROOT
variant = sys.argv[1]
for rep in range(3):
    for label, lib in (("variant", variant), ("shipped", None)):
        env = dict(os.environ)
        if lib:
            env["FPGPU_LIB"] = os.path.join(ROOT, lib)
        else:
            env.pop("FPGPU_LIB", None)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=200)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(json.dumps({"rep": rep, "build": label, "best_median_ms": json.loads(line[-1]) if line else r.stderr[-300:]}), flush=True)
