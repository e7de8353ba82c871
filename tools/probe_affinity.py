import os, sys, time, json
sys.path.insert(0, "/root/repo")
print("before torch", len(os.sched_getaffinity(0)), flush=True)
import torch
x = torch.zeros(1 << 20, device="cuda"); y = (x + 1).sum().item()
print("after torch cuda op", len(os.sched_getaffinity(0)), flush=True)
z = torch.randn(4096, 4096) @ torch.randn(4096, 4096)   # CPU matmul: starts torch's OpenMP pool
print("after torch cpu matmul", len(os.sched_getaffinity(0)), os.environ.get("OMP_PROC_BIND"), os.environ.get("OMP_NUM_THREADS"), flush=True)
import numpy as np
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 22
rng = np.random.default_rng(5)
ha = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ha[:, 3] &= (1 << 60) - 1
hb = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); hb[:, 3] &= (1 << 60) - 1
def t(fn, reps=3):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return round((time.perf_counter() - t0) / reps * 1e3, 2)
print("chain ms", t(lambda: f.sqr(f.add(f.mul(ha, hb), ha))), "mul ms", t(lambda: f.mul(ha, hb)), flush=True)
