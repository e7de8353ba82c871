"""Soak (not collected by pytest; run as `python tests/soak_inversion_plans.py [seconds] [seed]` on a GPU box):
batch inversion, pow and the multiplication chain on RANDOM odd moduli of every width, with random tree plans
(chain lengths per level, the fused top on / off, limb-form prefixes on / off, sparse prefixes on / off), random lengths up to 2^18 and
zeros / non-invertible elements sprinkled in, against the oracle bit for bit.  The suite's own tests pin seven
hand-picked moduli for these paths; this looks for the modulus or the length nobody picked.  Prints one JSON line
per case and a summary; exits non-zero at the first mismatch."""
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


def rand_elems(rng, n, n64, p):
    """n residues below p as an (n, n64) uint64 array: uniform words masked to p's bit length, the (fewer than half)
    rows that come out >= p get their top bit cleared (p has that bit set, so they end up below it)."""
    w = rng.integers(0, 1 << 64, size=(n, n64), dtype=np.uint64)
    top = (p.bit_length() - 1) // 64                    # the word that holds p's top bit (the container may be wider)
    top_bits = p.bit_length() - 64 * top
    w[:, top + 1:] = 0
    w[:, top] &= np.uint64((1 << top_bits) - 1)
    pw = [(p >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n64)]
    ge = np.ones(n, dtype=bool)      # lexicographic compare from the top word
    decided = np.zeros(n, dtype=bool)
    for i in reversed(range(top + 1)):
        gt = (w[:, i] > np.uint64(pw[i])) & ~decided
        lt = (w[:, i] < np.uint64(pw[i])) & ~decided
        ge[lt] = False
        decided |= gt | lt
    w[ge, top] &= np.uint64((1 << (top_bits - 1)) - 1)
    return w


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rnd = random.Random(seed)
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = 0
    hooks = ("FPGPU_INV_C", "FPGPU_INV_C0", "FPGPU_INV_FUSE_TOP", "FPGPU_INV_LP", "FPGPU_INV_SP")
    print(json.dumps({"soak": "inversion plans", "seed": seed, "seconds": budget}), flush=True)
    while time.time() < t_end:
        n64 = rnd.randint(1, 8)
        bits = rnd.randint(max(3, 64 * (n64 - 1) + 2), 64 * n64)
        shape = rnd.random()
        if shape < 0.25:
            p = (1 << bits) - rnd.randrange(1, 1 << min(20, bits - 1), 2)          # tight: just below a power of two
        elif shape < 0.4:
            p = (1 << (bits - 1)) + rnd.randrange(1, 1 << min(20, bits - 1), 2)    # just above one
        else:
            p = rnd.randrange(1 << (bits - 1), 1 << bits) | 1
        n = rnd.choice([1, 2, 63, 64, 65, 511, 4097, rnd.randint(1, 1 << 12), rnd.randint(1 << 12, 1 << 16), rnd.randint(1 << 16, 1 << 18)])
        plan = {h: None for h in hooks}
        if rnd.random() < 0.8:
            plan["FPGPU_INV_C"] = rnd.choice([2, 3, 4, 8, 16, 32])
            plan["FPGPU_INV_C0"] = rnd.choice([2, 4, 8, 16, 32, 64])
        plan["FPGPU_INV_FUSE_TOP"] = rnd.choice([None, 0, 1])
        plan["FPGPU_INV_LP"] = rnd.choice([None, 0, 1])
        plan["FPGPU_INV_SP"] = rnd.choice([None, 0, 1, 1])   # sparse level-0 prefixes wherever the limb form is not in use
        fast = rnd.random() < 0.6
        for h, v in plan.items():
            _lib.tuning_set(h) if v is None else _lib.tuning_set(h, v)
        f, o = cpa.MontyField(p, n64), Oracle(p, n64)
        x = rand_elems(rng, n, n64, p)
        for i in rng.integers(0, n, size=max(1, n // 997)):
            x[i] = 0
        if rnd.random() < 0.3:
            x[0] = 0
        if rnd.random() < 0.3:
            x[-1] = 0
        want, wflags = o.batch_inv(x, threads=8)
        dx = torch.from_numpy(x.view(np.int64)).cuda()
        out = f.alloc(n)
        flags = torch.empty(n, dtype=torch.uint8, device="cuda")
        ws = torch.empty(f.inv_workspace_bytes(n, fast=fast), dtype=torch.uint8, device="cuda")
        f.inv_dev(dx, out, flags, ws)
        torch.cuda.synchronize()
        ok = np.array_equal(flags.cpu().numpy(), wflags) and np.array_equal(out.cpu().numpy().view(np.uint64).reshape(n, n64), want)
        # the chain multiplier and the ladder on the same modulus (a short slice: the oracle's pow is a scalar loop)
        m = min(n, 257)
        a, b = x[:m].copy(), rand_elems(rng, m, n64, p)
        iters = rnd.randint(0, 9)
        wc = a
        for _ in range(iters):
            wc = o.mul(wc, b)
        gc = f.mul_chain(a, b, iters)
        e = rnd.getrandbits(rnd.choice([1, 13, 64, 64 * n64]))
        ok = ok and np.array_equal(gc, wc) and np.array_equal(f.pow(a, e), o.pow(a, e, exponent_bits=64 * n64))
        cases += 1
        print(json.dumps({"case": cases, "n64": n64, "bits": bits, "n": n, "plan": plan, "fast_ws": fast, "ok": bool(ok),
                          "none": int(wflags.sum()), "p": hex(p)}), flush=True)
        if not ok:
            sys.exit(1)
    for h in hooks:
        _lib.tuning_set(h)
    print(json.dumps({"soak": "inversion plans", "cases": cases, "all_ok": True}), flush=True)


if __name__ == "__main__":
    main()
