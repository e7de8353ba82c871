"""Soak (not collected by pytest; run as `python tests/soak_matrix_shapes.py [seconds] [seed]` on a GPU box):
mat-vec and mat-mul at MID-SIZE random shapes -- large enough for the matrix-core kernels with their column splits,
k-step batches and int32 flushes, small enough for the oracle -- on random odd moduli of every width, elements with
extreme digits mixed in (0, 1, p - 1, all-ones and 0x80 / 0x7f byte patterns reduced below p).  Each product runs on
the matrix cores (workspace given) AND on the VALU kernel (no workspace) and both are compared with the oracle bit
for bit; the suite's fuzz test stops at 150 x 300.  One JSON line per case; exits non-zero at the first mismatch."""
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import _lib  # noqa: E402
from oracle.fp_oracle import Oracle  # noqa: E402
from soak_inversion_plans import rand_elems  # noqa: E402


def spiced(rng, rnd, n, n64, p):
    w = rand_elems(rng, n, n64, p)
    pats = [0, 1, p - 1, p - 2, p >> 1]
    for byte in (0xFF, 0x80, 0x7F, 0x81):
        v = int.from_bytes(bytes([byte]) * (8 * n64), "little") % p
        pats.append(v)
    k = max(1, n // 50)
    for i in rng.integers(0, n, size=k):
        v = rnd.choice(pats)
        for j in range(n64):
            w[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return w


def dev(a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rnd = random.Random(seed)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = 0
    print(json.dumps({"soak": "matrix shapes", "seed": seed, "seconds": budget}), flush=True)
    while time.time() < t_end:
        n64 = rnd.randint(1, 8)
        bits = rnd.randint(max(3, 64 * (n64 - 1) + 2), 64 * n64)
        p = ((1 << bits) - rnd.randrange(1, 1 << min(20, bits - 1), 2)) if rnd.random() < 0.3 else (rnd.randrange(1 << (bits - 1), 1 << bits) | 1)
        f, o = cpa.MontyField(p, n64), Oracle(p, n64)
        # ---- mat-vec: rows x cols up to 2^22 elements, shapes from tall to wide
        cols = rnd.choice([rnd.randint(1, 64), rnd.randint(64, 2048), rnd.randint(2048, 40000)])
        rows = rnd.randint(1, max(1, min(3000, (1 << 22) // cols)))
        kind = rnd.random()
        if kind < 0.15:        # tall and narrow: one row per lane
            cols = rnd.randint(1, 40)
            rows = rnd.randint(65536, 110000)
        elif kind < 0.3:       # a few rows longer than the matrix-core plan takes: column parts through the workspace
            cols = ((1 << 23) // (8 * n64)) + rnd.randint(1, 50000)
            rows = rnd.randint(1, 6)
        A, x = spiced(rng, rnd, rows * cols, n64, p), spiced(rng, rnd, cols, n64, p)
        want = o.matvec(A, rows, cols, x, threads=0)
        dA, dx = dev(A), dev(x)
        y_m, y_v = f.alloc(rows), f.alloc(rows)
        ws = torch.empty(max(16, f.matvec_workspace_bytes(rows, cols)), dtype=torch.uint8, device="cuda")
        f.matvec_dev(dA, rows, cols, dx, y_m, workspace=ws)
        f.matvec_dev(dA, rows, cols, dx, y_v)
        torch.cuda.synchronize()
        ok_mv = (np.array_equal(y_m.cpu().numpy().view(np.uint64).reshape(rows, n64), want)
                 and np.array_equal(y_v.cpu().numpy().view(np.uint64).reshape(rows, n64), want))
        # ---- mat-mul: m x k x n up to 2^25 terms (host call: matrix cores for single-limb fields, VALU kernels otherwise)
        m, k, n = rnd.randint(1, 400), rnd.choice([rnd.randint(1, 70), rnd.randint(70, 1200)]), rnd.randint(1, 400)
        if rnd.random() < 0.25:   # a small output over a long k (k parts through the workspace), now and then one column
            m, k, n = rnd.randint(1, 40), rnd.randint(4096, 30000), rnd.choice([1, rnd.randint(1, 40)])
        elif rnd.random() < 0.2:  # a few vectors times a matrix: a lane per output column, k parts through the workspace
            m, k, n = rnd.randint(1, 4), rnd.randint(1, 3000), rnd.randint(1024, 6000)
        while m * k * n > (1 << 25):
            k = max(1, k // 2)
        # multi-limb products: now and then on the matrix cores whatever the shape (the plan alone takes them from ~64 tiles on)
        force_mfma = n64 >= 2 and rnd.random() < 0.4
        if force_mfma:
            _lib.tuning_set("FPGPU_MATMUL_MFMA", 1)
        Am, Bm = spiced(rng, rnd, m * k, n64, p), spiced(rng, rnd, k * n, n64, p)
        wantm = o.matmul(Am, Bm, m, k, n, threads=0)
        C, C2 = f.alloc(m * n), f.alloc(m * n)
        f.matmul_dev(dev(Am), dev(Bm), C, m, k, n)                      # no workspace: VALU kernels, whole k
        wsm = torch.empty(max(256, f.matmul_workspace_bytes(m, k, n)), dtype=torch.uint8, device="cuda")
        f.matmul_dev(dev(Am), dev(Bm), C2, m, k, n, workspace=wsm)      # matrix cores / k parts / the mat-vec kernels, as the shape has it
        torch.cuda.synchronize()
        ok_mm = (np.array_equal(f.matmul(Am, Bm, m, k, n), wantm) and np.array_equal(C.cpu().numpy().view(np.uint64).reshape(m * n, n64), wantm)
                 and np.array_equal(C2.cpu().numpy().view(np.uint64).reshape(m * n, n64), wantm))
        _lib.tuning_set("FPGPU_MATMUL_MFMA")
        cases += 1
        print(json.dumps({"case": cases, "n64": n64, "bits": bits, "matvec": [rows, cols], "matmul": [m, k, n], "mfma_forced": force_mfma,
                          "ok": bool(ok_mv and ok_mm), "p": hex(p)}), flush=True)
        if not (ok_mv and ok_mm):
            print(json.dumps({"matvec_ok": bool(ok_mv), "matmul_ok": bool(ok_mm)}), flush=True)
            sys.exit(1)
    print(json.dumps({"soak": "matrix shapes", "cases": cases, "all_ok": True}), flush=True)


if __name__ == "__main__":
    main()
