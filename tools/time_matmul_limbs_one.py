"""One multi-limb matrix-core product, a few launches (for rocprofv3 --kernel-trace): python tools/time_matmul_limbs_one.py [dim] [n64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import crypto_primitives_amd as cpa  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n64 = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = {4: cpa.BN254_FR, 6: 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 8: 2**512 - 569, 2: 2**127 - 1}[n64]
f = cpa.MontyField(p, n64)
A, B, C = f.random_dev(dim * dim, 1), f.random_dev(dim * dim, 2), f.alloc(dim * dim)
ws = torch.empty(max(256, f.matmul_workspace_bytes(dim, dim, dim)), dtype=torch.uint8, device="cuda")
for _ in range(6):
    f.matmul_dev(A, B, C, dim, dim, dim, workspace=ws)
torch.cuda.synchronize()
print("done")
