"""Batch inversion of one rank's slice (argv: log2 n, default 21), BLS12-381 Fq: for `rocprofv3 --kernel-trace --stats`."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 21
f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
n = 1 << lg
x = f.random_dev(n, 3)
out = f.alloc(n)
flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
ws = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
for _ in range(10): f.inv_dev(x, out, flags, ws)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): f.inv_dev(x, out, flags, ws)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"log2_n": lg, "ms": round(e0.elapsed_time(e1) / 50, 4)}), flush=True)
