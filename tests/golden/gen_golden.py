#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ with plain Python `int`
arithmetic -- an oracle independent of both oracle/fp_oracle.c and the HIP path.

The reference (Rust, crypto-bigint backed) cannot be built in this image, so
these vectors are not outputs of the reference itself; they are the unique
canonical residues its operations are specified to return (SURVEY.md section 0
fact 4), for the layouts and Montgomery radix the reference uses
(R = 2^(64*n64)):

  mul(a,b)      = a*b*R^-1 mod p   (src/helpers/crypto_bigint.rs:74-107)
  add/sub/neg   = (a+-b) mod p, -a mod p (0 -> 0)
  to_mont(x)    = x*R mod p        (project, crypto_bigint_monty.rs:319-340; x may be >= p)
  from_mont(x)  = x*R^-1 mod p     (lift, crypto_bigint_monty.rs:369-385)
  inv(x)        = (x*R^-1)^-1 * R mod p, None for 0 (crypto_bigint_monty.rs:223-237)
  matvec/matmul = folds of mul then add (semiring.rs:213-226 over matrix.rs:162-164 rows)

Run:  python tests/golden/gen_golden.py      (rewrites tests/golden/*.json)
"""
import json
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))

FIELDS = {
    # name: (modulus, n64)   -- moduli named in BASELINE.json configs and the reference's tests/benches
    "bn254_fr": (0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4),
    "bls12_381_fq": (
        0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
    "goldilocks": (0xFFFFFFFF00000001, 1),
    "p64_ref_tests": (0x8BAC0006D9927ABB, 1),           # crypto_bigint_monty.rs:619
    "secp256k1_p": (2**256 - 2**32 - 977, 4),            # crypto_bigint_monty.rs:488-495
    "bench_120bit": (0x860995AE68FC80E1B1BD1E39D54B33, 4),  # benches/const_monty_field_benches.rs:9-13
    "odd_composite_128": ((2**107 - 1) * 3, 2),          # non-prime odd modulus: inv must report None for gcd != 1
}


def hx(v):
    return format(v, "x")


def core_values(p, n64):
    R = 1 << (64 * n64)
    return sorted({0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, R % p, (R * R) % p})


def edge_values(p, n64):
    R = 1 << (64 * n64)
    vals = set(core_values(p, n64))
    for k in (31, 32, 33, 63, 64, 65, 64 * n64 - 1):
        for d in (-1, 0, 1):
            v = (1 << k) + d
            if 0 <= v < p:
                vals.add(v)
    return sorted(vals)


def gen_field(name, p, n64, rng):
    assert p % 2 == 1 and p < (1 << (64 * n64)), name
    R = 1 << (64 * n64)
    Ri = pow(R, -1, p)
    inv64 = (-pow(p, -1, 1 << 64)) % (1 << 64)
    out = {
        "name": name, "modulus": hx(p), "n64": n64,
        "mod_neg_inv": hx(inv64), "one": hx(R % p), "r2": hx(R * R % p),
    }
    edges = edge_values(p, n64)
    pairs = [(a, b) for a in edges for b in core_values(p, n64)]
    nrand = 256
    pairs += [(rng.randrange(p), rng.randrange(p)) for _ in range(nrand)]
    out["binary"] = [
        {"a": hx(a), "b": hx(b), "mul": hx(a * b * Ri % p), "add": hx((a + b) % p), "sub": hx((a - b) % p)}
        for a, b in pairs
    ]
    unary_in = edges + [rng.randrange(p) for _ in range(nrand)]
    un = []
    import math
    for a in unary_in:
        plain = a * Ri % p
        invertible = math.gcd(plain, p) == 1
        un.append({
            "a": hx(a), "neg": hx((-a) % p), "sqr": hx(a * a * Ri % p),
            "to_mont": hx(a * R % p), "from_mont": hx(plain),
            "inv": hx(pow(plain, -1, p) * R % p) if invertible else None,
        })
    out["unary"] = un
    # project accepts any n64-limb integer, also >= p (crypto_bigint_monty.rs:501-509)
    wide_in = [p, p + 1, R - 1, R - 2] + [rng.randrange(R) for _ in range(64)]
    out["to_mont_wide"] = [{"a": hx(a), "to_mont": hx(a * R % p)} for a in wide_in]
    # mul with a < R arbitrary, b < p (monty_mul's documented domain, helpers:74-107)
    out["mul_wide_a"] = [
        {"a": hx(a), "b": hx(b), "mul": hx(a * b * Ri % p)}
        for a, b in [(rng.randrange(R), rng.randrange(p)) for _ in range(64)]
    ]
    # batch inversion cases: zeros first / last / adjacent, ragged lengths
    binv = []
    for length in (1, 2, 63, 64, 65, 257):
        xs = [rng.randrange(1, p) for _ in range(length)]
        if length >= 2:
            xs[0] = 0
            xs[-1] = 0
        if length >= 65:
            xs[31] = 0
            xs[32] = 0
        if length == 1:
            xs[0] = 0
        res = []
        for x in xs:
            plain = x * Ri % p
            res.append(hx(pow(plain, -1, p) * R % p) if math.gcd(plain, p) == 1 else None)
        binv.append({"x": [hx(x) for x in xs], "inv": res})
    out["batch_inv"] = binv
    # small dense matrices: mat-vec and mat-mul (Montgomery form in, Montgomery form out)
    mats = []
    for rows, cols in ((1, 1), (7, 5), (33, 70)):
        A = [[rng.randrange(p) for _ in range(cols)] for _ in range(rows)]
        x = [rng.randrange(p) for _ in range(cols)]
        y = [sum(A[i][j] * x[j] * Ri for j in range(cols)) % p for i in range(rows)]
        mats.append({"rows": rows, "cols": cols, "A": [hx(v) for r in A for v in r],
                     "x": [hx(v) for v in x], "y": [hx(v) for v in y]})
    out["matvec"] = mats
    mms = []
    for m, k, n in ((1, 1, 1), (5, 7, 3), (17, 40, 9)):
        A = [[rng.randrange(p) for _ in range(k)] for _ in range(m)]
        B = [[rng.randrange(p) for _ in range(n)] for _ in range(k)]
        C = [[sum(A[i][l] * B[l][j] * Ri for l in range(k)) % p for j in range(n)] for i in range(m)]
        mms.append({"m": m, "k": k, "n": n, "A": [hx(v) for r in A for v in r],
                    "B": [hx(v) for r in B for v in r], "C": [hx(v) for r in C for v in r]})
    out["matmul"] = mms
    return out


def main():
    for name, (p, n64) in FIELDS.items():
        rng = random.Random("golden-" + name)
        data = gen_field(name, p, n64, rng)
        path = os.path.join(HERE, name + ".json")
        with open(path, "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
