"""GPU: property-based differential test (the reference uses proptest for its field laws,
crypto_bigint_const_monty.rs:1486-1584): random moduli of every compiled width -- random odd numbers, not
only primes: the multiplication, addition and reduction rules do not need primality -- random ragged batch
lengths and random programs of elementwise operations, inversions, divisions, powers and multiplication chains, run on device-resident handles and compared with the
oracle applied step by step.  Derandomised (the same cases every run) and bounded."""
import os
import random

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import ints_to_array

pytestmark = pytest.mark.gpu

UNARY = ["sqr", "neg", "double", "div_by_2", "inv", "pow"]
BINARY = ["mul", "add", "sub", "mul_chain", "div"]


@st.composite
def cases(draw):
    n64 = draw(st.integers(1, 8))
    bits = draw(st.integers(max(2, 64 * (n64 - 1) + 2), 64 * n64))          # top word in use, headroom 0 .. 62 bits
    p = draw(st.integers(1 << (bits - 1), (1 << bits) - 1)) | 1
    n = draw(st.sampled_from([1, 2, 63, 64, 65, 255, 257, 1000, 4097]))
    prog = draw(st.lists(st.tuples(st.sampled_from(UNARY + BINARY), st.integers(0, 3), st.integers(0, 3)), min_size=1, max_size=8))
    seed = draw(st.integers(0, 2**32 - 1))
    return n64, p, n, prog, seed


# FUZZ_EXAMPLES=<n> runs a longer, freshly seeded soak (tools/ use; the suite itself stays deterministic)
_SOAK = int(os.environ.get("FUZZ_EXAMPLES", "0"))


@settings(max_examples=_SOAK or 150, deadline=None, derandomize=not _SOAK, suppress_health_check=list(HealthCheck))
@given(cases())
def test_random_programs_on_random_odd_moduli(case):
    import crypto_primitives_amd as cpa
    from oracle.fp_oracle import Oracle
    n64, p, n, prog, seed = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(seed)
    edge = [0, 1, p - 1, p - 2, p >> 1]
    regs_h = [ints_to_array([rnd.choice(edge) if rnd.random() < 0.1 else rnd.randrange(p) for _ in range(n)], n64) for _ in range(4)]
    V = cpa.DeviceVector
    regs_d = [V.from_host(f, h) for h in regs_h]
    tmp = V.alloc(f, n)
    for op, i, j in prog:
        if op == "inv":
            want, wflags = o.batch_inv(regs_h[i])
            assert np.array_equal(f.inv_vec(regs_d[i], tmp), wflags)
        elif op == "div":
            if i == j:
                continue
            inv, wflags = o.batch_inv(regs_h[j])
            want = o.mul(regs_h[i], inv)
            assert np.array_equal(f.div_vec(regs_d[i], regs_d[j], tmp), wflags)
        elif op == "pow":
            e = rnd.getrandbits(rnd.choice([1, 7, 64, 64 * n64]))
            want = o.pow(regs_h[i], e, exponent_bits=64 * n64)
            f.pow_vec(regs_d[i], e, tmp)
        elif op == "mul_chain":
            iters = rnd.randrange(0, 6)
            want = regs_h[i]
            for _ in range(iters):
                want = o.mul(want, regs_h[j])
            f.mul_chain_vec(regs_d[i], regs_d[j], tmp, iters)
        elif op in UNARY:
            want = getattr(o, op)(regs_h[i])
            getattr(f, op + "_vec")(regs_d[i], tmp)
        else:
            want = getattr(o, op)(regs_h[i], regs_h[j])
            getattr(f, op + "_vec")(regs_d[i], regs_d[j], tmp)
        regs_h[i] = want
        regs_d[i], tmp = tmp, regs_d[i]
        assert np.array_equal(regs_d[i].download(), want), (op, hex(p), n64, n)
    # the host-buffer entry points on the final state, and the folds
    assert np.array_equal(f.mul(regs_h[0], regs_h[1]), o.mul(regs_h[0], regs_h[1]))
    assert np.array_equal(f.sum_vec(regs_d[2]), o.sum(regs_h[2]))
    assert np.array_equal(f.dot_vec(regs_d[0], regs_d[3]), o.sum(o.mul(regs_h[0], regs_h[3])))


@st.composite
def matrix_cases(draw):
    n64 = draw(st.integers(1, 8))
    bits = draw(st.integers(max(2, 64 * (n64 - 1) + 2), 64 * n64))
    p = draw(st.integers(1 << (bits - 1), (1 << bits) - 1)) | 1
    rows = draw(st.integers(1, 150))
    cols = draw(st.integers(1, 300))
    inner = draw(st.integers(1, 70))
    density = draw(st.integers(0, 9))
    layout = draw(st.sampled_from([None, [0], [0, 0, 0]]))
    seed = draw(st.integers(0, 2**32 - 1))
    return n64, p, rows, cols, inner, density, layout, seed


@settings(max_examples=_SOAK or 50, deadline=None, derandomize=not _SOAK, suppress_health_check=list(HealthCheck))
@given(matrix_cases())
def test_random_matrix_products_on_random_odd_moduli(case):
    """Mat-vec (host call = matrix cores where the shape allows, VALU kernel otherwise; and on handles, row-sharded
    with the output gathered), mat-mul (single-limb matrix-core / VALU and multi-limb kernels) and sparse mat-vec
    on random shapes and random odd moduli of every width, extreme elements mixed in."""
    import crypto_primitives_amd as cpa
    from oracle.fp_oracle import Oracle
    n64, p, rows, cols, inner, density, layout, seed = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(seed)
    el = lambda k: ints_to_array([rnd.choice((0, 1, p - 1)) if rnd.random() < 0.15 else rnd.randrange(p) for _ in range(k)], n64)
    A, x = el(rows * cols), el(cols)
    want = o.matvec(A, rows, cols, x, threads=0)
    assert np.array_equal(f.matvec(A, rows, cols, x), want), ("matvec", hex(p), rows, cols)
    V = cpa.DeviceVector
    cpa.set_devices(layout or [])
    try:
        sharded, repl = (cpa.PLACE_CURRENT, cpa.PLACE_CURRENT) if layout is None else (cpa.PLACE_SHARDED, cpa.PLACE_REPLICATED)
        vA, vx, vy = V.from_host(f, A, sharded, row_len=cols), V.from_host(f, x, repl), V.alloc(f, rows, repl)
        f.matvec_vec(vA, rows, cols, vx, vy)
        assert np.array_equal(vy.download(), want), ("matvec handles", hex(p), rows, cols, layout)
        if density:
            vals = el(rows * density)
            idx = np.array([rnd.randrange(cols) for _ in range(rows * density)], dtype=np.uint32)
            swant = o.spmv(vals, idx, rows, cols, density, x)
            assert np.array_equal(f.spmv(vals, idx, rows, cols, density, x), swant), ("spmv", hex(p), rows, cols, density)
    finally:
        cpa.set_devices([])
    m, k, n = min(rows, 40), inner, min(cols, 45)
    Am, Bm = el(m * k), el(k * n)
    assert np.array_equal(f.matmul(Am, Bm, m, k, n), o.matmul(Am, Bm, m, k, n, threads=0)), ("matmul", hex(p), m, k, n)
