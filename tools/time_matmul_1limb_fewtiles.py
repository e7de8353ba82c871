"""Single-limb (Goldilocks) products of few 64 x 64 tiles, k <= 16320: FPGPU_MATMUL_MFMA unset (the plan's choice: the limb
kernel with k split over work items where that wins) against the VALU kernel; same bits."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.GOLDILOCKS, 1)
def timed(fn, reps=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for m, k, n in ((256, 8192, 256), (512, 4096, 512), (128, 16000, 128), (640, 2048, 640), (300, 1000, 300), (1024, 1024, 1024), (800, 800, 800)):
    A, B, C0, C1 = f.random_dev(m * k, 1), f.random_dev(k * n, 2), f.alloc(m * n), f.alloc(m * n)
    t0 = timed(lambda: f.matmul_dev(A, B, C0, m, k, n))
    nb = f.matmul_workspace_bytes(m, k, n)
    ws = torch.empty(max(256, nb), dtype=torch.uint8, device="cuda")
    t1 = timed(lambda: f.matmul_dev(A, B, C1, m, k, n, workspace=ws))
    planes = 256 + 8 * ((k + 63) // 64 * 64) * (((m + 63) // 64 + (n + 63) // 64) * 64)
    ws2 = torch.empty(planes, dtype=torch.uint8, device="cuda")      # only the single-limb kernel's planes: that kernel runs
    t2 = timed(lambda: f.matmul_dev(A, B, C1, m, k, n, workspace=ws2))
    print(json.dumps({"shape": [m, k, n], "valu_ms": round(t0, 3), "plan_ms": round(t1, 3), "single_limb_kernel_ms": round(t2, 3),
                      "same": bool(torch.equal(C0, C1)), "workspace_MiB": round(nb / 2**20, 1), "limb_kernel": nb > planes}), flush=True)
