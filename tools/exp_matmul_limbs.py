"""Multi-limb mat-mul on the matrix cores (fp_matmul_mfma_limbs.cuh): parity on small shapes against the oracle with the
hook forcing the path, then timing against the VALU kernel at larger shapes (equal results checked)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import _lib  # noqa: E402
from oracle.fp_oracle import Oracle  # noqa: E402
from soak_inversion_plans import rand_elems  # noqa: E402

FIELDS = [("bn254", cpa.BN254_FR, 4), ("bls12_381", 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 6),
          ("p127", 2**127 - 1, 2), ("secp256k1", 2**256 - 2**32 - 977, 4), ("p192", 2**192 - 2**64 - 1, 3), ("p512", 2**512 - 569, 8),
          ("p320", 2**320 - 197, 5), ("p448", 2**448 - 2**224 - 1, 7)]


def dev(a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    rng = np.random.default_rng(11)
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode in ("all", "parity"):
        _lib.tuning_set("FPGPU_MATMUL_MFMA", 1)
        for name, p, n64 in FIELDS:
            f, o = cpa.MontyField(p, n64), Oracle(p, n64)
            for m, k, n in ((1, 1, 2), (64, 64, 64), (70, 130, 65), (17, 300, 129), (130, 64 * (255 // n64) + 70, 66)):
                A, B = rand_elems(rng, m * k, n64, p), rand_elems(rng, k * n, n64, p)
                # extreme digits: 0, p - 1, 0x7f / 0x80 byte patterns reduced below p
                for arr in (A, B):
                    for i, v in enumerate((0, p - 1, int.from_bytes(b"\x7f" * (8 * n64), "little") % p, int.from_bytes(b"\x80" * (8 * n64), "little") % p,
                                           (int.from_bytes(b"\x7f" * (8 * n64), "little") + 1) % p)):
                        if i < len(arr):
                            for j in range(n64):
                                arr[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
                want = o.matmul(A, B, m, k, n, threads=0)
                nb = f.matmul_workspace_bytes(m, k, n)
                ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda")
                C = f.alloc(m * n)
                f.matmul_dev(dev(A), dev(B), C, m, k, n, workspace=ws)
                torch.cuda.synchronize()
                ok = np.array_equal(C.cpu().numpy().view(np.uint64).reshape(m * n, n64), want)
                print(json.dumps({"field": name, "shape": [m, k, n], "workspace": nb, "ok": bool(ok)}), flush=True)
                if not ok:
                    sys.exit(1)
        _lib.tuning_set("FPGPU_MATMUL_MFMA")
    if mode in ("all", "time"):
        for name, p, n64 in FIELDS[:2] + FIELDS[5:6]:
            f = cpa.MontyField(p, n64)
            for m, k, n in ((1024, 1024, 1024), (2048, 2048, 2048), (4096, 1024, 4096), (512, 4096, 512), (256, 1024, 256), (3000, 500, 3000), (1024, 8192, 1024),
                            (128, 2048, 128), (300, 300, 300), (640, 640, 640), (100, 10000, 100), (64, 512, 64), (1000, 1000, 1000)):
                A, B = f.random_dev(m * k, 1), f.random_dev(k * n, 2)
                C0, C1 = f.alloc(m * n), f.alloc(m * n)
                _lib.tuning_set("FPGPU_MATMUL_MFMA", 0)
                t_valu = timed(lambda: f.matmul_dev(A, B, C0, m, k, n))
                _lib.tuning_set("FPGPU_MATMUL_MFMA", 1)
                nb = f.matmul_workspace_bytes(m, k, n)
                ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda")
                t_mfma = timed(lambda: f.matmul_dev(A, B, C1, m, k, n, workspace=ws))
                _lib.tuning_set("FPGPU_MATMUL_MFMA")
                auto = f.matmul_workspace_bytes(m, k, n)
                print(json.dumps({"field": name, "shape": [m, k, n], "valu_ms": round(t_valu, 3), "mfma_ms": round(t_mfma, 3),
                                  "G_terms_per_s_mfma": round(m * k * n / t_mfma / 1e6, 1), "same": bool(torch.equal(C0, C1)),
                                  "workspace_MiB": round(nb / 2**20, 1), "plan_takes_it": bool(auto > 0)}), flush=True)
                del A, B, C0, C1, ws


if __name__ == "__main__":
    main()
