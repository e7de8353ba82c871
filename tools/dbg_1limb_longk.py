import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import crypto_primitives_amd as cpa
from oracle.fp_oracle import Oracle
from soak_inversion_plans import rand_elems
p = cpa.GOLDILOCKS
f, o = cpa.MontyField(p, 1), Oracle(p, 1)
rng = np.random.default_rng(5)
for m, k, n in ((200, 16400, 210), (128, 16400, 128), (128, 16384 + 64, 128), (128, 16320 + 64 * 3, 128)):
    A, B = rand_elems(rng, m * k, 1, p), rand_elems(rng, k * n, 1, p)
    want = o.matmul(A, B, m, k, n, threads=0).reshape(m, n)
    dA, dB = torch.from_numpy(A.view(np.int64)).cuda(), torch.from_numpy(B.view(np.int64)).cuda()
    C0, C1 = f.alloc(m * n), f.alloc(m * n)
    f.matmul_dev(dA, dB, C0, m, k, n)
    nb = f.matmul_workspace_bytes(m, k, n)
    ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda")
    f.matmul_dev(dA, dB, C1, m, k, n, workspace=ws)
    torch.cuda.synchronize()
    g0 = C0.cpu().numpy().view(np.uint64).reshape(m, n)
    g1 = C1.cpu().numpy().view(np.uint64).reshape(m, n)
    bad0, bad1 = np.argwhere(g0 != want), np.argwhere(g1 != want)
    if len(bad1):
        print("rows%64", sorted(set((bad1[:, 0] % 64).tolist())), "cols%64", sorted(set((bad1[:, 1] % 64).tolist())), "tiles", sorted(set(map(tuple, (bad1 // 64).tolist()))))
        r, c = bad1[0]
        print("first wrong", int(r), int(c), hex(int(g1[r, c])), "want", hex(int(want[r, c])), "diff mod p", hex((int(g1[r, c]) - int(want[r, c])) % p))
    print((m, k, n), "nb", nb, "valu wrong:", len(bad0), bad0[:5].tolist(), "mfma wrong:", len(bad1), bad1[:5].tolist(), flush=True)
