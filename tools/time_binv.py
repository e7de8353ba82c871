"""Times the BLS12-381 batch inversion over 2^24 elements (HIP events, 20 launches) and checks x * inv(x) == 1."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
n = 1 << int(os.environ.get("LOG2N", "24"))
x = f.random_dev(n, 7); out = f.alloc(n); chk = f.alloc(n)
flags = torch.empty(n, dtype=torch.uint8, device="cuda")
ws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
for rep in range(2):
    f.inv_dev(x, out, flags, ws); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f.inv_dev(x, out, flags, ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
f.mul_dev(x, out, chk)
one = torch.from_numpy(f.one_words.view(np.int64)).cuda()
ok = bool(torch.equal(chk[flags == 0], one.expand(int((flags == 0).sum()), 6)))
print({"lib": os.environ.get("FPGPU_LIB", "default"), "ms": round(ms, 4), "Ginv_per_s": round(n / ms / 1e6, 2), "ok": ok}, flush=True)
