"""GPU parity: the HIP path (through the C ABI) against the committed Python-int
golden vectors and against the CPU oracle on seeded random inputs.  Bit-exact.

Mirrors the reference's own test shapes: known-answer residues
(crypto_bigint_monty.rs:564-592,614-687), algebraic laws (const_monty.rs:1486-1584),
inv(0) = None (crypto_bigint_monty.rs:778-780), Division by zero
(crypto_bigint_monty.rs:715-723), even modulus rejected (:1050-1055).
"""
import random

import numpy as np
import pytest

from conftest import FIELD_NAMES, array_to_ints, h2i, ints_to_array, load_golden, synth_elements

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as c
    assert c.load().fpgpu_device_count() >= 1, "no MI355X visible"
    return c


def field_of(cpa, g):
    return cpa.MontyField(h2i(g["modulus"]), g["n64"])


def oracle_of(g):
    from oracle.fp_oracle import Oracle
    return Oracle(h2i(g["modulus"]), g["n64"])


def test_ctx_params_match_golden(cpa, golden):
    f = field_of(cpa, golden)
    assert f.mod_neg_inv == h2i(golden["mod_neg_inv"])
    assert array_to_ints(f.one_words.reshape(1, -1))[0] == h2i(golden["one"])
    assert array_to_ints(f.r2_words.reshape(1, -1))[0] == h2i(golden["r2"])


def test_binary_ops_golden(cpa, golden):
    f = field_of(cpa, golden)
    n64 = golden["n64"]
    a = ints_to_array([h2i(v["a"]) for v in golden["binary"]], n64)
    b = ints_to_array([h2i(v["b"]) for v in golden["binary"]], n64)
    assert array_to_ints(f.mul(a, b)) == [h2i(v["mul"]) for v in golden["binary"]]
    assert array_to_ints(f.add(a, b)) == [h2i(v["add"]) for v in golden["binary"]]
    assert array_to_ints(f.sub(a, b)) == [h2i(v["sub"]) for v in golden["binary"]]


def test_unary_ops_golden(cpa, golden):
    f = field_of(cpa, golden)
    n64 = golden["n64"]
    a = ints_to_array([h2i(v["a"]) for v in golden["unary"]], n64)
    assert array_to_ints(f.neg(a)) == [h2i(v["neg"]) for v in golden["unary"]]
    assert array_to_ints(f.sqr(a)) == [h2i(v["sqr"]) for v in golden["unary"]]
    assert array_to_ints(f.project(a)) == [h2i(v["to_mont"]) for v in golden["unary"]]
    assert array_to_ints(f.lift(a)) == [h2i(v["from_mont"]) for v in golden["unary"]]
    out, flags = f.inv(a)
    exp = [v["inv"] for v in golden["unary"]]
    assert list(flags) == [1 if e is None else 0 for e in exp]
    assert array_to_ints(out) == [0 if e is None else h2i(e) for e in exp]


def test_wide_inputs_golden(cpa, golden):
    """project() of values >= p (crypto_bigint_monty.rs:501-509) and mul with a < R."""
    f = field_of(cpa, golden)
    n64 = golden["n64"]
    a = ints_to_array([h2i(v["a"]) for v in golden["to_mont_wide"]], n64)
    assert array_to_ints(f.project(a)) == [h2i(v["to_mont"]) for v in golden["to_mont_wide"]]
    a = ints_to_array([h2i(v["a"]) for v in golden["mul_wide_a"]], n64)
    b = ints_to_array([h2i(v["b"]) for v in golden["mul_wide_a"]], n64)
    assert array_to_ints(f.mul(a, b)) == [h2i(v["mul"]) for v in golden["mul_wide_a"]]


def test_batch_inv_golden(cpa, golden):
    f = field_of(cpa, golden)
    n64 = golden["n64"]
    for case in golden["batch_inv"]:
        x = ints_to_array([h2i(v) for v in case["x"]], n64)
        out, flags = f.inv(x)
        assert list(flags) == [1 if e is None else 0 for e in case["inv"]]
        assert array_to_ints(out) == [0 if e is None else h2i(e) for e in case["inv"]]


def test_matvec_matmul_golden(cpa, golden):
    f = field_of(cpa, golden)
    n64 = golden["n64"]
    for c in golden["matvec"]:
        A = ints_to_array([h2i(v) for v in c["A"]], n64)
        x = ints_to_array([h2i(v) for v in c["x"]], n64)
        assert array_to_ints(f.matvec(A, c["rows"], c["cols"], x)) == [h2i(v) for v in c["y"]]
    for c in golden["matmul"]:
        A = ints_to_array([h2i(v) for v in c["A"]], n64)
        B = ints_to_array([h2i(v) for v in c["B"]], n64)
        assert array_to_ints(f.matmul(A, B, c["m"], c["k"], c["n"])) == [h2i(v) for v in c["C"]]


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 257, 100003])
def test_elementwise_vs_oracle_ragged(cpa, n):
    """Empty, single and ragged lengths (not multiples of the 64-lane wave / 256 block)."""
    for name in ("bn254_fr", "bls12_381_fq", "goldilocks"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        a = ints_to_array(synth_elements(p, n64, n, 11), n64)
        b = ints_to_array(synth_elements(p, n64, n, 12), n64)
        if n == 0:
            assert f.mul(a, b).shape[0] == 0
            continue
        assert np.array_equal(f.mul(a, b), o.mul(a, b))
        assert np.array_equal(f.sqr(a), o.sqr(a))
        assert np.array_equal(f.add(a, b), o.add(a, b))
        assert np.array_equal(f.sub(a, b), o.sub(a, b))
        assert np.array_equal(f.neg(a), o.neg(a))
        assert np.array_equal(f.project(a), o.to_mont(a))
        assert np.array_equal(f.lift(a), o.from_mont(a))


def test_mul_out_of_range_inputs_match_reference_rule(cpa):
    """monty_mul subtracts p once iff carry or result >= p (helpers:87-104); the HIP path
    computes the same integer T, so even inputs >= p (undefined for the field, defined
    for the function) agree bit for bit with the restated CIOS."""
    rng = random.Random(5)
    for name in ("bn254_fr", "bls12_381_fq", "goldilocks", "secp256k1_p"):
        g = load_golden(name)
        n64 = g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        R = 1 << (64 * n64)
        a = ints_to_array([rng.randrange(R) for _ in range(4096)] + [R - 1] * 4, n64)
        b = ints_to_array([rng.randrange(R) for _ in range(4096)] + [R - 1, R - 2, 0, 1], n64)
        assert np.array_equal(f.mul(a, b), o.mul(a, b))
        assert np.array_equal(f.mul(a, b), o.mul_boxed(a, b))
        assert np.array_equal(f.lift(a), o.from_mont(a))


def test_boxed_path_equals_cios_path(cpa):
    """BoxedMontyField::mul (monty_mul_limbs + sub_assign_mod_with_carry) and
    MontyField::mul (CIOS) give the same bits; one kernel serves both adapters."""
    g = load_golden("bls12_381_fq")
    p, n64 = h2i(g["modulus"]), g["n64"]
    f, o = field_of(cpa, g), oracle_of(g)
    a = ints_to_array(synth_elements(p, n64, 5000, 21), n64)
    b = ints_to_array(synth_elements(p, n64, 5000, 22), n64)
    got = f.mul(a, b)
    assert np.array_equal(got, o.mul_boxed(a, b))
    assert np.array_equal(got, o.mul(a, b))


def test_mul_chain_matches_mul_assign_loop(cpa):
    """benches/field_ops_bench_common.rs:27-39 shape: acc = elem; 1000 x mul_assign(acc, elem)."""
    seeds = [695962179703, 2345695962179703, 111111111111111111, 12345678124578658568]  # :121-124
    for name in ("bench_120bit", "bn254_fr"):
        g = load_golden(name)
        f, o = field_of(cpa, g), oracle_of(g)
        elems = o.to_mont(ints_to_array(seeds, g["n64"]))
        got = f.mul_chain(elems, elems, 1000)
        for i in range(len(seeds)):
            assert np.array_equal(got[i], o.bench_mul_assign(elems[i], 1000))


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 2048, 2049, 5000, 70001])
def test_batch_inv_vs_oracle(cpa, n):
    for name in ("bls12_381_fq", "bn254_fr", "goldilocks"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        xs = synth_elements(p, n64, n, 31)
        rng = random.Random(n)
        for _ in range(max(1, n // 97)):
            xs[rng.randrange(n)] = 0
        xs[0] = 0
        xs[-1] = 0
        x = ints_to_array(xs, n64)
        got, flags = f.inv(x)
        exp, eflags = o.batch_inv(x)
        assert np.array_equal(flags, eflags)
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("clen", ["2", "3", "32", "100"])
def test_batch_inv_chain_lengths(cpa, clen, tuning):
    """The chain length per level is a tuning choice (FPGPU_INV_C pins it): every partition into
    chains gives the same inverses, ragged last chains included (these sizes plan ONE level: the top
    takes up to 65536 totals; multi-level plans are in test_batch_inv_multi_level_plans)."""
    tuning("FPGPU_INV_C", clen)
    for name, n in (("bls12_381_fq", 70001), ("bn254_fr", 40000), ("goldilocks", 100003), ("odd_composite_128", 6000)):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        xs = synth_elements(p, n64, n, 33)
        for i in range(0, n, 401):
            xs[i] = 0
        xs[-1] = 0
        if name == "odd_composite_128":
            xs[17] = 3
        x = ints_to_array(xs, n64)
        got, flags = f.inv(x)
        exp, eflags = o.batch_inv(x)
        assert np.array_equal(flags, eflags), (name, clen)
        assert np.array_equal(got, exp), (name, clen)


def test_batch_inv_composite_modulus_poisoned_groups(cpa):
    """Odd composite modulus: non-invertible non-zero elements must come back as None
    without disturbing their neighbours (invert_odd_mod_vartime semantics)."""
    g = load_golden("odd_composite_128")
    p, n64 = h2i(g["modulus"]), g["n64"]
    f, o = field_of(cpa, g), oracle_of(g)
    n = 6000
    xs = synth_elements(p, n64, n, 41)
    xs[5] = 3
    xs[777] = (2**107 - 1) * 2
    xs[778] = 0
    xs[4100] = 9
    x = ints_to_array(xs, n64)
    got, flags = f.inv(x)
    exp, eflags = o.batch_inv(x)
    assert np.array_equal(flags, eflags)
    assert np.array_equal(got, exp)
    assert flags.sum() >= 4


def test_div_by_zero_raises(cpa):
    g = load_golden("secp256k1_p")
    f = field_of(cpa, g)
    o = oracle_of(g)
    a = o.to_mont(ints_to_array([7, 11], 4))
    z = ints_to_array([0, 5], 4)
    with pytest.raises(ZeroDivisionError, match="Division by zero"):
        f.div(a, z)
    den = o.to_mont(ints_to_array([5, 5], 4))
    q = f.div(a, den)
    assert np.array_equal(f.mul(q, den), a)


@pytest.mark.parametrize("name,n", [("bn254_fr", 70001), ("bls12_381_fq", 4099), ("goldilocks", 100003), ("odd_composite_128", 9000)])
def test_checked_div_three_layers_vs_oracle(cpa, name, n):
    """FieldConfig::checked_div = mul(a, inv(b)?) (src/field.rs:168-170) as one call: host buffers, device
    pointers, handles (unset list and a three-entry list) -- quotient and None flags equal the oracle's
    per-element inv followed by mul; zero divisors at the ends and, for the composite modulus, divisors
    without an inverse."""
    import torch
    g = load_golden(name)
    p, n64 = h2i(g["modulus"]), g["n64"]
    f, o = field_of(cpa, g), oracle_of(g)
    a = ints_to_array(synth_elements(p, n64, n, 71), n64)
    bs = synth_elements(p, n64, n, 72)
    for i in (0, 1, n // 2, n - 1):
        bs[i] = 0
    if name == "odd_composite_128":
        bs[17] = 3
    b = ints_to_array(bs, n64)
    inv, want_flags = o.batch_inv(b)
    want = o.mul(a, inv)
    got, flags = f.checked_div(a, b)
    assert np.array_equal(flags, want_flags) and np.array_equal(got, want)
    assert int(flags.sum()) >= 4 and not got[flags == 1].any()
    # device pointers
    ta, tb = (torch.from_numpy(x.view(np.int64)).cuda() for x in (a, b))
    tout, tflags = f.alloc(n), torch.empty(n, dtype=torch.uint8, device="cuda")
    ws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    f.div_dev(ta, tb, tout, tflags, ws)
    assert np.array_equal(tout.cpu().numpy().view(np.uint64), want) and np.array_equal(tflags.cpu().numpy(), want_flags)
    # handles
    V = cpa.DeviceVector
    for ids in (None, [0, 0, 0]):
        cpa.set_devices(ids or [])
        try:
            where = cpa.PLACE_CURRENT if ids is None else cpa.PLACE_SHARDED
            va, vb, vo = V.from_host(f, a, where), V.from_host(f, b, where), V.alloc(f, n, where)
            vflags = f.div_vec(va, vb, vo)
            assert np.array_equal(vo.download(), want) and np.array_equal(vflags, want_flags)
            with pytest.raises(cpa.FpgpuError):
                f.div_vec(va, vb, va)                  # out must not alias an operand
        finally:
            cpa.set_devices([])


def test_even_modulus_rejected(cpa):
    with pytest.raises(cpa.InvalidModulus):
        cpa.MontyField(42, 4)


@pytest.mark.parametrize("shape", [(1, 1), (3, 1000), (64, 64), (130, 257), (257, 513)])
def test_matvec_vs_oracle(cpa, shape):
    rows, cols = shape
    for name in ("bn254_fr", "goldilocks", "bls12_381_fq"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        A = ints_to_array(synth_elements(p, n64, rows * cols, 51), n64)
        x = ints_to_array(synth_elements(p, n64, cols, 52), n64)
        assert np.array_equal(f.matvec(A, rows, cols, x), o.matvec(A, rows, cols, x, threads=0))


@pytest.mark.parametrize("T", ["2", "4"])
def test_matvec_multi_term_passes(cpa, T, tuning):
    """The T-products-per-pass mat-vec variants (selected with FPGPU_MATVEC_T; T = 1 is the
    default) must give the same canonical residues; moduli without headroom fall back to
    T = 1 inside the library."""
    tuning("FPGPU_MATVEC_T", T)
    for name in ("bn254_fr", "bls12_381_fq", "bench_120bit", "secp256k1_p", "goldilocks"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        for rows, cols in ((3, 2100), (5, 1024), (2, 777)):
            A = ints_to_array(synth_elements(p, n64, rows * cols, 53), n64)
            x = ints_to_array(synth_elements(p, n64, cols, 54), n64)
            assert np.array_equal(f.matvec(A, rows, cols, x), o.matvec(A, rows, cols, x, threads=0))


@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 9, 3), (300, 1000, 7), (33, 50, 64), (17, 400, 100), (9, 70, 200),
                                   (4, 6, 0), (70001, 4096, 4)])
@pytest.mark.parametrize("unroll,log2g", [(0, -1), (1, 2), (2, 0), (4, 1), (4, 3), (2, 6)])
def test_spmv_vs_oracle(cpa, tuning, shape, unroll, log2g):
    """SparseMatrix (src/matrix.rs:46-61: `density` (column, value) cells per row) times vector ==
    the oracle's fold of mul then add; repeated columns inside a row are legal and summed.  Default
    geometry and forced ones: 1, 2 or 4 cells per lane loaded together, 1 to 64 lanes per row (rows shorter
    than a lane's batch, batches that run past the end of the row)."""
    rows, cols, density = shape
    if unroll:
        tuning("FPGPU_SPMV_UNROLL", unroll)
        tuning("FPGPU_SPMV_LOG2G", log2g)
    rng = np.random.default_rng(rows + cols + density)
    for name in ("bn254_fr", "goldilocks", "bls12_381_fq"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        vals = ints_to_array(synth_elements(p, n64, rows * density, 91), n64)
        x = ints_to_array(synth_elements(p, n64, cols, 92), n64)
        col = rng.integers(0, cols, size=rows * density, dtype=np.uint32)
        want = o.spmv(vals, col, rows, cols, density, x)
        assert np.array_equal(f.spmv(vals, col, rows, cols, density, x), want), name


def test_spmv_rejects_out_of_range_column(cpa):
    f = cpa.MontyField(cpa.BN254_FR, 4)
    vals = np.ones((2, 4), dtype=np.uint64)
    x = np.ones((3, 4), dtype=np.uint64)
    with pytest.raises(cpa.FpgpuError):
        f.spmv(vals, np.array([0, 3], dtype=np.uint32), 1, 3, 2, x)


@pytest.mark.parametrize("shape", [(1, 1, 1), (16, 16, 16), (33, 70, 17), (100, 129, 65)])
def test_matmul_vs_oracle(cpa, shape):
    m, k, n = shape
    for name in ("goldilocks", "p64_ref_tests", "bn254_fr"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        A = ints_to_array(synth_elements(p, n64, m * k, 61), n64)
        B = ints_to_array(synth_elements(p, n64, k * n, 62), n64)
        assert np.array_equal(f.matmul(A, B, m, k, n), o.matmul(A, B, m, k, n, threads=0))


def _dev_matvec(cpa, f, A, rows, cols, x, use_ws):
    import torch
    dA = torch.from_numpy(A.view(np.int64)).cuda()
    dx = torch.from_numpy(x.view(np.int64)).cuda()
    dy = f.alloc(rows)
    ws = None
    if use_ws:
        nb = f.matvec_workspace_bytes(rows, cols)   # 0: shape left to the VALU kernel (tiny or ragged rows)
        assert nb > 0 or cols * f.n64 * 8 < 64 or (cols * f.n64 * 8) % 16
        ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda")
    f.matvec_dev(dA, rows, cols, dx, dy, workspace=ws)
    torch.cuda.synchronize()
    return dy.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("shape", [(1, 1), (3, 1000), (16, 2), (130, 257), (257, 513), (129, 31), (40, 4099)])
def test_matvec_matrix_core_path_vs_oracle(cpa, shape):
    """Mat-vec through the int8 matrix-core kernels (workspace given) == oracle == VALU kernel, on
    uniform elements and on byte patterns at the extremes of the signed digits."""
    rows, cols = shape
    rnd = random.Random(rows * 7 + cols)
    for name in ("bn254_fr", "goldilocks", "bls12_381_fq", "bench_120bit"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        nb = 8 * n64
        pats = [0, 1, p - 1, p - 2, p >> 1, (p >> 1) + 1] + [int.from_bytes(bytes([b]) * nb, "little") % p
                                                              for b in (0x7F, 0x80, 0xFF, 0x81, 0x01)]
        pats += [(p + int.from_bytes(bytes([0x7F]) * nb, "little") + 1) % p, (p - int.from_bytes(bytes([0x80]) * nb, "little")) % p]
        for mode in ("uniform", "edge"):
            if mode == "uniform":
                A = ints_to_array(synth_elements(p, n64, rows * cols, 81), n64)
                x = ints_to_array(synth_elements(p, n64, cols, 82), n64)
            else:
                A = ints_to_array([rnd.choice(pats) for _ in range(rows * cols)], n64)
                x = ints_to_array([rnd.choice(pats) for _ in range(cols)], n64)
            want = o.matvec(A, rows, cols, x, threads=0)
            got = _dev_matvec(cpa, f, A, rows, cols, x, True)
            assert np.array_equal(got.reshape(want.shape), want), (name, mode, "mfma")
            got = _dev_matvec(cpa, f, A, rows, cols, x, False)
            assert np.array_equal(got.reshape(want.shape), want), (name, mode, "valu")


def test_matvec_matrix_core_int32_flush_and_column_split(cpa, tuning):
    """One column part per row block (forced) so that the int32 tiles are flushed inside the loop
    (every 255 chunks = 4080 elements), then the default split over many workgroups; extreme bytes."""
    g = load_golden("bn254_fr")
    p, n64 = h2i(g["modulus"]), g["n64"]
    f, o = field_of(cpa, g), oracle_of(g)
    rows, cols = 20, 9001
    for va, vx in [(int.from_bytes(b"\xff" * 31 + b"\x2f", "little") % p, int.from_bytes(b"\x80" * 32, "little") % p),
                   (int.from_bytes(b"\x00" * 32, "little"), p - 1), (p - 1, int.from_bytes(b"\x7f" * 32, "little") % p)]:
        A = ints_to_array([va] * (rows * cols), n64)
        x = ints_to_array([vx] * cols, n64)
        want = o.matvec(A, rows, cols, x, threads=0)
        for ks in ("1", "3", None):
            tuning("FPGPU_MATVEC_KSPLIT", 0 if ks is None else ks)
            got = _dev_matvec(cpa, f, A, rows, cols, x, True)
            assert np.array_equal(got.reshape(want.shape), want), (hex(va), hex(vx), ks)


def test_matvec_matrix_core_falls_back_on_undigitisable_x(cpa):
    """x >= p whose x - p does not fit the signed digits: flagged on the device, the VALU kernel answers."""
    g = load_golden("bench_120bit")
    p, n64 = h2i(g["modulus"]), g["n64"]
    f = field_of(cpa, g)
    rows, cols = 9, 77
    A = ints_to_array(synth_elements(p, n64, rows * cols, 91), n64)
    x = ints_to_array(synth_elements(p, n64, cols, 92), n64)
    x[5, :] = np.uint64(2**64 - 1)
    want = _dev_matvec(cpa, f, A, rows, cols, x, False)
    assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols, x, True), want)


def _dev_matmul(cpa, f, A, B, m, k, n, use_ws):
    import torch
    dA = torch.from_numpy(A.view(np.int64)).cuda()
    dB = torch.from_numpy(B.view(np.int64)).cuda()
    dC = f.alloc(m * n)
    ws = None
    if use_ws:
        nb = f.matmul_workspace_bytes(m, k, n)
        assert nb > 0 or n == 1                          # one column: the mat-vec's workspace, 0 for a tiny one
        ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda")
    f.matmul_dev(dA, dB, dC, m, k, n, workspace=ws)
    torch.cuda.synchronize()
    return dC.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("shape", [(1, 1, 1), (64, 64, 64), (65, 63, 130), (17, 500, 33), (200, 1000, 70), (3, 16320, 5),
                                   (70, 130, 66), (130, 192, 70), (128, 2100, 129)])
def test_matmul_matrix_core_path_vs_oracle(cpa, shape):
    """Single-limb mat-mul through the int8 matrix-core kernel (workspace given) == oracle == VALU kernel,
    on uniform elements and on the values that make the largest signed digits."""
    m, k, n = shape
    rnd = random.Random(m * 1000 + n)
    for name in ("goldilocks", "p64_ref_tests"):
        g = load_golden(name)
        p = h2i(g["modulus"])
        f, o = field_of(cpa, g), oracle_of(g)
        edge = [0, 1, p - 1, p - 2, 0x7F7F7F7F7F7F7F7F % p, 0x7F7F7F7F7F7F7F80 % p, (p + 0x7F7F7F7F7F7F7F7F) % p,
                (p - 0x8080808080808080) % p, 0x8080808080808080 % p, p >> 1, (p >> 1) + 1]
        for mode in ("uniform", "edge"):
            if mode == "uniform":
                A = ints_to_array(synth_elements(p, 1, m * k, 71), 1)
                B = ints_to_array(synth_elements(p, 1, k * n, 72), 1)
            else:
                A = ints_to_array([rnd.choice(edge) for _ in range(m * k)], 1)
                B = ints_to_array([rnd.choice(edge) for _ in range(k * n)], 1)
            want = o.matmul(A, B, m, k, n, threads=0)
            assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, True).reshape(want.shape), want), (name, mode, "mfma")
            assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, False).reshape(want.shape), want), (name, mode, "valu")


def test_matmul_matrix_core_extreme_digits(cpa):
    """K = 16320 with every digit product at its extreme (+127*127 and -128*-128... the int32 bound)."""
    g = load_golden("goldilocks")
    p = h2i(g["modulus"])
    f, o = field_of(cpa, g), oracle_of(g)
    m, k, n = 2, 16320, 3
    for va, vb in [(0x7F7F7F7F7F7F7F7F, 0x7F7F7F7F7F7F7F7F), (0x7F7F7F7F7F7F7F80, 0x7F7F7F7F7F7F7F80),
                   (0x7F7F7F7F7F7F7F7F, 0x7F7F7F7F7F7F7F80)]:
        A = ints_to_array([va % p] * (m * k), 1)
        B = ints_to_array([vb % p] * (k * n), 1)
        want = o.matmul(A, B, m, k, n, threads=0)
        assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, True).reshape(want.shape), want)


def test_matmul_matrix_core_falls_back_on_undigitisable_inputs(cpa):
    """Non-canonical inputs x >= p with x - p beyond eight signed digits: the digitiser flags them and the
    VALU kernel (which takes any a < R, as monty_mul does) produces the result."""
    p = 2**61 - 1
    f = cpa.MontyField(p, 1)
    from oracle.fp_oracle import Oracle
    o = Oracle(p, 1)
    m, k, n = 5, 70, 9
    rnd = random.Random(3)
    A = ints_to_array([rnd.randrange(p) for _ in range(m * k)], 1)
    B = ints_to_array([rnd.randrange(p) for _ in range(k * n)], 1)
    A[7, 0] = np.uint64(2**64 - 1)
    want = _dev_matmul(cpa, f, A, B, m, k, n, False)
    assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, True), want)
    # the same numbers reduced first give the same residues through the oracle
    A2 = A.copy()
    A2[7, 0] = np.uint64((2**64 - 1) % p)
    got2 = _dev_matmul(cpa, f, A2, B, m, k, n, True)
    assert np.array_equal(got2.reshape(-1), o.matmul(A2, B, m, k, n, threads=0).reshape(-1))


def test_matmul_single_limb_beyond_16320_terms_on_the_matrix_cores(cpa, tuning):
    """k > 16320 is more than the int32 accumulators of k_matmul_mfma_1limb hold: with the workspace such a product runs
    through the limb kernel with one limb (k in parts of 255 k-steps, and split over work items where the tiles are few)
    instead of the VALU kernel; == oracle == VALU kernel, on uniform elements and on the largest signed digits."""
    rnd = random.Random(99)
    for name in ("goldilocks", "p64_ref_tests"):
        g = load_golden(name)
        p = h2i(g["modulus"])
        f, o = field_of(cpa, g), oracle_of(g)
        edge = [0, 1, p - 1, p - 2, 0x7F7F7F7F7F7F7F7F % p, 0x7F7F7F7F7F7F7F80 % p, (p + 0x7F7F7F7F7F7F7F7F) % p,
                (p - 0x8080808080808080) % p, 0x8080808080808080 % p, p >> 1]
        from soak_inversion_plans import rand_elems
        rng = np.random.default_rng(p % 1000)
        earr = np.array(edge, dtype=np.uint64)
        for m, k, n in ((200, 16400, 210), (300, 2 * 16320 + 77, 250)):
            for mode in ("uniform", "edge"):
                if mode == "uniform":
                    A, B = rand_elems(rng, m * k, 1, p), rand_elems(rng, k * n, 1, p)
                else:
                    A = earr[rng.integers(0, len(edge), size=m * k)].reshape(m * k, 1)
                    B = earr[rng.integers(0, len(edge), size=k * n)].reshape(k * n, 1)
                nb = f.matmul_workspace_bytes(m, k, n)
                assert nb >= 8 * (m * k + k * n), (m, k, n, nb)
                got = _dev_matmul(cpa, f, A, B, m, k, n, True).reshape(m * n, 1)
                for r0 in (0, m - 2):
                    want = o.matmul(A[r0 * k:(r0 + 2) * k], B, 2, k, n, threads=0)
                    assert np.array_equal(got[r0 * n:(r0 + 2) * n], want.reshape(2 * n, 1)), (name, m, k, n, mode, r0)
                assert np.array_equal(got, _dev_matmul(cpa, f, A, B, m, k, n, False).reshape(m * n, 1)), (name, m, k, n, mode, "vs VALU")
    # every digit product at its extreme over the longest run the int32 accumulators take, in BOTH matrix-core kernels at
    # full occupancy (225 tiles: no k split): k = 16320 is the single-limb kernel's bound, k = 16384 the limb kernel's
    # first k part of 255 k-steps plus one more
    g = load_golden("goldilocks")
    p = h2i(g["modulus"])
    f, o = field_of(cpa, g), oracle_of(g)
    m = n = 900
    for k in (16320, 16384):
        for va, vb in ((0x7F7F7F7F7F7F7F7F, 0x7F7F7F7F7F7F7F7F), (0x7F7F7F7F7F7F7F80, 0x7F7F7F7F7F7F7F80), (0x7F7F7F7F7F7F7F7F, 0x7F7F7F7F7F7F7F80)):
            A = np.full((m * k, 1), va % p, dtype=np.uint64)
            B = np.full((k * n, 1), vb % p, dtype=np.uint64)
            got = _dev_matmul(cpa, f, A, B, m, k, n, True).reshape(m * n, 1)
            want = o.matmul(A[:2 * k], B, 2, k, n, threads=0).reshape(2 * n, 1)
            assert np.array_equal(got[:2 * n], want) and np.array_equal(got[-2 * n:], want), (k, hex(va), hex(vb))
            assert (got == got[0]).all(), (k, hex(va), hex(vb))      # constant operands: every output is the same residue
    # a non-canonical input with no digit form: the VALU kernel answers (as for k <= 16320)
    p = 2**61 - 1
    f = cpa.MontyField(p, 1)
    m, k, n = 200, 16400, 210
    A, B = rand_elems(rng, m * k, 1, p), rand_elems(rng, k * n, 1, p)
    A[7, 0] = np.uint64(2**64 - 1)
    assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, True), _dev_matmul(cpa, f, A, B, m, k, n, False))


_MFL_MODULI = [("bn254", 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4),
               ("bls12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
               ("p127 in two words", 2**127 - 1, 2), ("p192", 2**192 - 2**64 - 1, 3), ("secp256k1 (tight)", 2**256 - 2**32 - 977, 4),
               ("p320", 2**320 - 197, 5), ("p448", 2**448 - 2**224 - 1, 7), ("p512", 2**512 - 569, 8),
               ("odd composite 384", 3 * (2**380 - 2**190 + 5) | 1, 6), ("120 bits in four words", 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFC7 , 4)]


@pytest.mark.parametrize("case", _MFL_MODULI, ids=[c[0] for c in _MFL_MODULI])
def test_matmul_multi_limb_matrix_core_path_vs_oracle(cpa, tuning, case):
    """Multi-limb mat-mul as n64 x n64 single-limb products on the int8 matrix cores (fp_matmul_mfma_limbs.cuh), forced for
    every shape with FPGPU_MATMUL_MFMA = 1 (the plan's cost model would leave these small ones to the VALU kernel): ragged
    tiles, one k-step, several persistent tiles per workgroup (more than 256 tiles), k split over work items where the
    tiles are few, k long enough for two k parts of 255 / n64 k-steps each per item (the scratch is then added to, not
    stored), uniform elements and the values with the largest signed digits; == oracle == VALU kernel, bit for bit."""
    from oracle.fp_oracle import Oracle
    from soak_inversion_plans import rand_elems
    _, p, n64 = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rng = np.random.default_rng(n64 + 77)
    rnd = random.Random(n64)
    full = int.from_bytes(b"\x7f" * (8 * n64), "little")
    edge = [0, 1, p - 1, p - 2, full % p, (full + 1) % p, (p + full) % p, int.from_bytes(b"\x80" * (8 * n64), "little") % p,
            (p - int.from_bytes(b"\x80" * (8 * n64), "little")) % p, p >> 1, (p >> 1) + 1,
            int.from_bytes(b"\xff" * (8 * n64 - 1) + b"\x00", "little") % p]
    part = 64 * (255 // n64)
    tuning("FPGPU_MATMUL_MFMA", 1)
    # few tiles: k split over work items (z parts per tile, added per output); 225 tiles over two k parts of 255 / n64
    # k-steps (the scratch is added to); 121 tiles in two z parts of more than one k part each; 324 tiles of one k-step
    shapes = [(1, 1, 2), (64, 64, 64), (65, 63, 130), (17, 300, 33), (200, part + 70, 210), (900, part + 70, 900),
              (700, 2 * part + 200, 700), (1100, 3, 1100)]
    for m, k, n in shapes:
        for mode in ("uniform", "edge"):
            if mode == "edge" and m * k * n > 3_000_000:
                continue
            if mode == "uniform":
                A, B = rand_elems(rng, m * k, n64, p), rand_elems(rng, k * n, n64, p)
            else:
                A = ints_to_array([rnd.choice(edge) for _ in range(m * k)], n64)
                B = ints_to_array([rnd.choice(edge) for _ in range(k * n)], n64)
            nb = f.matmul_workspace_bytes(m, k, n)
            assert nb >= 8 * n64 * (m * k + k * n), (m, k, n, nb)       # digit planes of both operands + scratch
            got = _dev_matmul(cpa, f, A, B, m, k, n, True).reshape(m * n, n64)
            if m * k * n > 20_000_000:
                # the long products: the oracle on the first and last rows, every output against the VALU kernel (no
                # workspace), which the other tests hold to the oracle
                for r0 in (0, m - 2):
                    want = o.matmul(A[r0 * k:(r0 + 2) * k], B, 2, k, n, threads=0)
                    assert np.array_equal(got[r0 * n:(r0 + 2) * n], want.reshape(2 * n, n64)), (m, k, n, mode, "matrix cores", r0)
                assert np.array_equal(got, _dev_matmul(cpa, f, A, B, m, k, n, False).reshape(m * n, n64)), (m, k, n, mode, "vs VALU")
            else:
                want = o.matmul(A, B, m, k, n, threads=0)
                assert np.array_equal(got, want.reshape(m * n, n64)), (m, k, n, mode, "matrix cores")
    m, k, n = 70, 130, 66
    A, B = rand_elems(rng, m * k, n64, p), rand_elems(rng, k * n, n64, p)
    want = o.matmul(A, B, m, k, n, threads=0)
    assert np.array_equal(f.matmul(A, B, m, k, n), want), "host call"
    tuning("FPGPU_MATMUL_MFMA", 0)
    assert f.matmul_workspace_bytes(m, k, n) == 0
    assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, False).reshape(want.shape), want), "VALU kernel"


def test_matmul_multi_limb_matrix_core_extreme_digits_and_fallback(cpa, tuning):
    """Every digit product at its extreme over a full k part (the int32 bound: 255 k-steps shared by the n64 pairs of a
    limb diagonal), and a non-canonical input whose x - p has no digit form: flagged on the device, k_matmul answers."""
    from oracle.fp_oracle import Oracle
    tuning("FPGPU_MATMUL_MFMA", 1)
    for p, n64 in ((2**256 - 2**32 - 977, 4), (2**127 - 1, 2), (2**512 - 569, 8)):
        f, o = cpa.MontyField(p, n64), Oracle(p, n64)
        full = int.from_bytes(b"\x7f" * (8 * n64), "little")
        m, k, n = 2, 64 * (255 // n64), 3
        for va, vb in ((full % p, full % p), ((full + 1) % p, (full + 1) % p), (full % p, (full + 1) % p)):
            A, B = ints_to_array([va] * (m * k), n64), ints_to_array([vb] * (k * n), n64)
            want = o.matmul(A, B, m, k, n, threads=0)
            assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, True).reshape(want.shape), want), (n64, hex(va), hex(vb))
    p, n64 = 2**127 - 1, 2
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    m, k, n = 5, 70, 9
    rnd = random.Random(5)
    A = ints_to_array([rnd.randrange(p) for _ in range(m * k)], n64)
    B = ints_to_array([rnd.randrange(p) for _ in range(k * n)], n64)
    A[7, :] = np.uint64(2**64 - 1)                       # 2^128 - 1 >= p, and 2^128 - 1 - p = 2^127 has no 16-digit form
    want = _dev_matmul(cpa, f, A, B, m, k, n, False)     # the VALU kernel takes any a < R, as monty_mul does
    assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, True), want)
    A2 = A.copy()
    A2[7] = ints_to_array([(2**128 - 1) % p], n64)[0]
    assert np.array_equal(_dev_matmul(cpa, f, A2, B, m, k, n, True).reshape(-1, n64), o.matmul(A2, B, m, k, n, threads=0).reshape(-1, n64))


@pytest.mark.parametrize("name", ["bn254_fr", "goldilocks", "bls12_381_fq", "odd_composite_128"])
def test_pow_vs_oracle(cpa, name):
    # batched pow, one exponent per batch; oracle = the reference's 4-bit-window AMM ladder
    g = load_golden(name)
    f, o = field_of(cpa, g), oracle_of(g)
    p, n64 = h2i(g["modulus"]), g["n64"]
    rnd = random.Random(11)
    xs = [0, 1, p - 1, p - 2] + [rnd.randrange(p) for _ in range(300)]
    x = o.to_mont(ints_to_array(xs, n64))
    for e in [0, 1, 2, 3, 15, 16, 17, 2**32 - 1, 2**64 - 1, 2**64, p - 2, p - 1, rnd.getrandbits(64 * n64)]:
        bits = max(64 * n64, 64 * ((e.bit_length() + 63) // 64))
        want = o.pow(x, e, exponent_bits=bits)
        got = f.pow(x, e)
        assert (got == want).all(), (name, hex(e))
    assert f.pow(x[:0], 5).shape == (0, n64)


def test_pow_u32_reference_kats(cpa):
    # crypto_bigint_monty.rs:726-760, 934-1004 through the GPU path
    f = cpa.MontyField(2**256 - 2**32 - 977, 4)
    pr = lambda v: f.project(ints_to_array([v], 4))
    for base, e, r in [(2, 0, 1), (2, 1, 2), (2, 3, 8), (2, 10, 1024), (3, 4, 81), (0, 0, 1), (0, 7, 0)]:
        assert (f.pow(pr(base), e) == pr(r)).all(), (base, e)
    g = cpa.MontyField(10064419296686275259, 1)
    pg = lambda v: g.project(ints_to_array([v], 1))
    for base, e in [(2, 8), (2, 9), (3, 5), (12345, 67), (2, 64), (123456789, 2**32 - 1)]:
        assert array_to_ints(g.lift(g.pow(pg(base), e)))[0] == pow(base, e, 10064419296686275259)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 256, 257, 4099, 600001])
def test_sum_product_vs_oracle(cpa, n):
    # SemiringConfig::sum_refs / product_refs (src/semiring.rs:213-241)
    for name in ["bn254_fr", "goldilocks", "bls12_381_fq"]:
        g = load_golden(name)
        f, o = field_of(cpa, g), oracle_of(g)
        p, n64 = h2i(g["modulus"]), g["n64"]
        v = np.random.default_rng(n + 3).integers(0, 2**64, size=(n, n64), dtype=np.uint64)
        top = p.bit_length() - 1 - 64 * (n64 - 1)      # keep every element below 2^(bitlen-1) < p
        v[:, n64 - 1] &= np.uint64((1 << top) - 1)
        if n > 2:
            v[1] = ints_to_array([p - 1], n64)[0]
        assert (f.sum(v) == o.sum(v)).all(), (name, n, "sum")
        assert (f.product(v) == o.product(v)).all(), (name, n, "product")


def test_fill_random_matches_host_generator(cpa):
    import torch
    for name in ("bn254_fr", "bls12_381_fq", "goldilocks"):
        g = load_golden(name)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f = field_of(cpa, g)
        t = f.random_dev(1000, 0x1234)
        torch.cuda.synchronize()
        got = array_to_ints(t.cpu().numpy().view(np.uint64))
        assert got == synth_elements(p, n64, 1000, 0x1234)


def test_large_device_resident_properties(cpa):
    """Size-independent laws on 2^22 HBM-resident BN254 elements (device API, no host copy):
    commutativity, (a+b)-b = a, a*a = sqr(a), lift(project(x)) = x, a*inv(a) = one,
    plus an oracle check of a strided sample."""
    import torch
    g = load_golden("bn254_fr")
    p, n64 = h2i(g["modulus"]), g["n64"]
    f, o = field_of(cpa, g), oracle_of(g)
    n = 1 << 22
    a, b = f.random_dev(n, 71), f.random_dev(n, 72)
    ab, ba, t, u = f.alloc(n), f.alloc(n), f.alloc(n), f.alloc(n)
    f.mul_dev(a, b, ab)
    f.mul_dev(b, a, ba)
    assert torch.equal(ab, ba)
    f.add_dev(a, b, t)
    f.sub_dev(t, b, u)
    assert torch.equal(u, a)
    f.sqr_dev(a, t)
    f.mul_dev(a, a, u)
    assert torch.equal(t, u)
    f.to_mont_dev(a, t)
    f.from_mont_dev(t, u)
    assert torch.equal(u, a)
    f.neg_dev(a, t)
    f.add_dev(a, t, u)
    assert int(u.count_nonzero()) == 0
    ws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    f.inv_dev(a, t, flags, ws)
    f.mul_dev(a, t, u)
    one = torch.from_numpy(f.one_words.view(np.int64)).cuda()
    nz = flags == 0
    assert torch.equal(u[nz], one.expand(int(nz.sum()), n64))
    torch.cuda.synchronize()
    idx = torch.arange(0, n, 4099, device="cuda")
    sa = a[idx].cpu().numpy().view(np.uint64)
    sb = b[idx].cpu().numpy().view(np.uint64)
    assert np.array_equal(ab[idx].cpu().numpy().view(np.uint64), o.mul(sa, sb))
    assert np.array_equal(t[idx].cpu().numpy().view(np.uint64), o.batch_inv(sa)[0])


# Moduli by headroom h = 64 n64 - bitlen: from 6 limbs on, the chain kernels (mul chain, pow, batch
# inversion, product fold) run on unsaturated 29-bit limbs and keep intermediates in [0, 2p) when
# h >= 2, canonical after every step otherwise; both regimes and every compiled width are covered.
_UNSAT_MODULI = [
    ("p192 h=0", 2**192 - 2**64 - 1, 3),
    ("secp256k1 h=0", 2**256 - 2**32 - 977, 4),
    ("curve25519 h=1", 2**255 - 19, 4),
    ("bn254 h=2", 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4),
    ("p127 in 256 h=129", 2**127 - 1, 4),
    ("p320 h=0", 2**320 - 197, 5),
    ("bls12-381 h=3", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
    ("ed448 h=0", 2**448 - 2**224 - 1, 7),
    ("p511 h=1", 2**511 - 187, 8),
    ("p512 h=0", 2**512 - 569, 8),
]


@pytest.mark.parametrize("case", _UNSAT_MODULI, ids=[c[0] for c in _UNSAT_MODULI])
def test_chain_kernels_tight_and_lazy_moduli(cpa, case):
    from oracle.fp_oracle import Oracle
    _, p, n64 = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(p % 1000003)
    edge = [0, 1, 2, p - 1, p - 2, p >> 1, (p >> 1) + 1]
    xs = edge + [rnd.randrange(p) for _ in range(600 - len(edge))]
    x = ints_to_array(xs, n64)
    y = ints_to_array([rnd.randrange(p) for _ in xs], n64)
    # dependent multiply chain == the oracle's mul_assign loop
    got = f.mul_chain(x, y, 37)
    want = x.copy()
    for _ in range(37):
        want = o.mul(want, y)
    assert np.array_equal(got, want)
    assert np.array_equal(f.mul_chain(x, y, 0), x)
    # pow (one exponent per batch)
    for e in (0, 1, 2, 65537, p - 2, rnd.getrandbits(64 * n64)):
        assert np.array_equal(f.pow(x, e), o.pow(x, e, exponent_bits=64 * n64)), hex(e)
    # batch inversion with zeros (one level of chains at this size; the multi-level plans are in
    # test_batch_inv_multi_level_plans)
    big = ints_to_array([0 if i % 53 == 7 else rnd.randrange(p) for i in range(4500)], n64)
    got, flags = f.inv(big)
    want, wflags = o.batch_inv(big)
    assert np.array_equal(flags, wflags) and np.array_equal(got, want)
    # product fold
    assert np.array_equal(f.product(big[8:53]), o.product(big[8:53]))
    assert np.array_equal(f.product(y), o.product(y))


def test_batch_inv_composite_modulus_wide(cpa):
    """Poisoned groups through the unsaturated-limb kernels: odd composite (2^127 - 1)(2^89 - 1)."""
    from oracle.fp_oracle import Oracle
    p, n64 = (2**127 - 1) * (2**89 - 1), 4
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(5)
    xs = [rnd.randrange(1, p) for _ in range(9000)]
    xs[3] = (2**89 - 1) * 12345
    xs[4000] = 0
    xs[4001] = (2**127 - 1) * 3
    xs[8999] = 2**89 - 1
    x = ints_to_array(xs, n64)
    got, flags = f.inv(x)
    want, wflags = o.batch_inv(x)
    assert np.array_equal(flags, wflags) and np.array_equal(got, want)
    assert flags.sum() >= 4


@pytest.mark.parametrize("case", _UNSAT_MODULI, ids=[c[0] for c in _UNSAT_MODULI])
def test_matmul_multi_limb_every_width(cpa, case):
    """Multi-limb mat-mul (column sums on unsaturated limbs carried across terms, one exact
    reduction per output) at every compiled width, ragged shapes, K beyond one LDS tile, and
    operands at the top of the range."""
    from oracle.fp_oracle import Oracle
    _, p, n64 = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(n64)
    for m, k, n in ((1, 1, 1), (3, 17, 5), (20, 100, 18), (2, 1000, 3)):
        A = ints_to_array([rnd.choice((p - 1, p - 2, rnd.randrange(p))) for _ in range(m * k)], n64)
        B = ints_to_array([rnd.choice((p - 1, 1, rnd.randrange(p))) for _ in range(k * n)], n64)
        assert np.array_equal(f.matmul(A, B, m, k, n), o.matmul(A, B, m, k, n, threads=0)), (m, k, n)


def test_empty_dimensions(cpa):
    """Empty folds: a mat-vec over zero columns and a mat-mul over k = 0 are all-zero (sum_refs of
    nothing, src/semiring.rs:213-226); zero rows write nothing."""
    import torch
    for p, n64 in ((cpa.BN254_FR, 4), (cpa.GOLDILOCKS, 1), (cpa.BLS12_381_FQ, 6)):
        f = cpa.MontyField(p, n64)
        e = np.zeros((0, n64), dtype=np.uint64)
        y = f.matvec(e, 5, 0, e)
        assert y.shape == (5, n64) and not y.any()
        C = f.matmul(e, e, 3, 0, 4)
        assert C.shape == (12, n64) and not C.any()
        assert f.matmul(e, np.ones((28, n64), dtype=np.uint64), 0, 7, 4).shape == (0, n64)
        assert f.matvec(e, 0, 9, np.ones((9, n64), dtype=np.uint64)).shape == (0, n64)
        dy = f.alloc(5)
        dy.fill_(-1)
        f.matvec_dev(f.alloc(0), 5, 0, f.alloc(0), dy, workspace=torch.empty(256, dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        assert int(dy.count_nonzero()) == 0


@pytest.mark.parametrize("hook", ["FPGPU_MATMUL_GENERIC=1", "FPGPU_MATMUL_MFMA=0", "FPGPU_MATVEC_MFMA=0"])
def test_forced_kernel_paths_agree(cpa, hook, tuning):
    """The path-forcing hooks of INTEGRATION.md section 4: single-limb mat-mul through the multi-limb
    kernel and through the VALU kernel, mat-vec through the VALU kernel although a workspace is
    given - all bit-identical to the oracle."""
    name, val = hook.split("=")
    tuning(name, val)
    for gname in ("goldilocks", "p64_ref_tests", "bn254_fr"):
        g = load_golden(gname)
        p, n64 = h2i(g["modulus"]), g["n64"]
        f, o = field_of(cpa, g), oracle_of(g)
        m, k, n = 37, 150, 21
        A = ints_to_array(synth_elements(p, n64, m * k, 201), n64)
        B = ints_to_array(synth_elements(p, n64, k * n, 202), n64)
        want = o.matmul(A, B, m, k, n, threads=0)
        assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, n64 == 1).reshape(want.shape), want), gname
        x = ints_to_array(synth_elements(p, n64, k, 203), n64)
        wy = o.matvec(A, m, k, x, threads=0)
        assert np.array_equal(_dev_matvec(cpa, f, A, m, k, x, True).reshape(wy.shape), wy), gname


@pytest.mark.parametrize("name,p,n64,rows,cols", [
    ("bn254", 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4, 1, 300001),
    ("bn254", 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4, 37, 262145),
    ("bls12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6, 3, 200000),
    ("goldilocks", 2**64 - 2**32 + 1, 1, 2, 1500007), ("p512", 2**512 - 569, 8, 5, 140000), ("p448", 2**448 - 2**224 - 1, 7, 1, 160003)])
def test_matvec_few_long_rows_split_over_columns(cpa, name, p, n64, rows, cols):
    """Rows longer than the matrix-core plan takes (more than 2^23 bytes) and too few to give every SIMD a wave: with
    the workspace fpgpu_matvec_workspace_bytes reports for such shapes the VALU kernel splits the columns over many
    waves per row (ragged last part, parts that start past the row's end) and k_matvec_combine adds the partial
    residues; without it the same kernel takes whole rows.  Both against the oracle, bit for bit."""
    from oracle.fp_oracle import Oracle
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rng = np.random.default_rng(rows * 1000 + n64)
    from soak_inversion_plans import rand_elems
    A, x = rand_elems(rng, rows * cols, n64, p), rand_elems(rng, cols, n64, p)
    A[:3], x[:2] = ints_to_array([p - 1, 0, 1], n64), ints_to_array([p - 1, p - 2], n64)
    want = o.matvec(A, rows, cols, x, threads=0)
    assert f.matvec_workspace_bytes(rows, cols) >= rows * 2 * 8 * n64, "no split workspace reported for this shape"
    assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols, x, True).reshape(want.shape), want), "column parts"
    assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols, x, False).reshape(want.shape), want), "whole rows"
    assert np.array_equal(f.matvec(A, rows, cols, x), want), "host call"


@pytest.mark.parametrize("name,p,n64", [("bn254", 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4),
                                        ("bls12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
                                        ("goldilocks", 2**64 - 2**32 + 1, 1), ("p512", 2**512 - 569, 8)])
def test_matmul_small_outputs_over_a_long_k_and_single_columns(cpa, name, p, n64):
    """Shapes whose 16 x 16 output tiles cannot fill the chip: with the scratch fpgpu_matmul_workspace_bytes reports for
    them, k is split over many workgroups per tile (ragged last part) and the partial residues are added per output;
    a product with ONE column is handed to the mat-vec kernels (with and without their workspace).  Oracle, bit for bit."""
    from oracle.fp_oracle import Oracle
    from soak_inversion_plans import rand_elems
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rng = np.random.default_rng(n64 + 40)
    for m, k, n in ((16, 5000, 16), (3, 8200, 5), (40, 4100, 33), (1, 70001, 2)):
        A, B = rand_elems(rng, m * k, n64, p), rand_elems(rng, k * n, n64, p)
        A[:2], B[:2] = ints_to_array([p - 1, 0], n64), ints_to_array([p - 1, 1], n64)
        want = o.matmul(A, B, m, k, n, threads=0)
        assert f.matmul_workspace_bytes(m, k, n) >= 2 * m * n * 8 * n64, (m, k, n)
        assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, True).reshape(want.shape), want), ("k parts", m, k, n)
        assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, n, False).reshape(want.shape), want), ("whole k", m, k, n)
        assert np.array_equal(f.matmul(A, B, m, k, n), want), ("host call", m, k, n)
    # a few vectors times a matrix: one lane per output column; k parts through the workspace when the columns are few
    for m, k, n in ((1, 2051, 2048), (3, 300, 5000), (2, 40, 70001), (4, 1500, 1024), (1, 1, 1024), (2, 129, 1500)):
        A, B = rand_elems(rng, m * k, n64, p), rand_elems(rng, k * n, n64, p)
        A[:1], B[:2] = ints_to_array([p - 1], n64), ints_to_array([p - 1, 1], n64)
        want = o.matmul(A, B, m, k, n, threads=0)
        dC = _dev_matmul(cpa, f, A, B, m, k, n, False).reshape(want.shape)
        assert np.array_equal(dC, want), ("vector times matrix", m, k, n)
        nb = f.matmul_workspace_bytes(m, k, n)
        import torch
        ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda")
        dA, dB, dC2 = torch.from_numpy(A.view(np.int64)).cuda(), torch.from_numpy(B.view(np.int64)).cuda(), f.alloc(m * n)
        f.matmul_dev(dA, dB, dC2, m, k, n, workspace=ws)
        torch.cuda.synchronize()
        assert np.array_equal(dC2.cpu().numpy().view(np.uint64).reshape(want.shape), want), ("vector times matrix, workspace", m, k, n, nb)
    for m, k in ((300, 700), (70003, 5), (2, 300001), (33, 1)):
        A, B = rand_elems(rng, m * k, n64, p), rand_elems(rng, k, n64, p)
        want = o.matmul(A, B, m, k, 1, threads=0)
        assert np.array_equal(want, o.matvec(A, m, k, B, threads=0))
        use_ws = f.matmul_workspace_bytes(m, k, 1) > 0
        assert f.matmul_workspace_bytes(m, k, 1) == f.matvec_workspace_bytes(m, k)
        if use_ws:
            assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, 1, True).reshape(want.shape), want), ("one column, workspace", m, k)
        assert np.array_equal(_dev_matmul(cpa, f, A, B, m, k, 1, False).reshape(want.shape), want), ("one column", m, k)
        assert np.array_equal(f.matmul(A, B, m, k, 1), want)


_MV_VALU_MODULI = _UNSAT_MODULI + [("goldilocks", 2**64 - 2**32 + 1, 1), ("p61 in one word", 2**61 - 1, 1), ("p128 h=1", 2**127 - 1, 2),
                                    ("odd composite 384", 3 * (2**380 - 2**190 + 5) | 1, 6)]


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("case", _MV_VALU_MODULI, ids=[c[0] for c in _MV_VALU_MODULI])
def test_matvec_both_valu_kernels_every_width(cpa, tuning, case, kernel):
    """The two VALU mat-vec kernels -- k_matvec (saturated words, a wide add per term) and k_matvec_cols (round 3:
    unsaturated 29-bit limbs, carry-free column sums across the terms of a row, one row per wave, the exact integer
    reduced once per row) -- forced with FPGPU_MATVEC_COLS at every compiled width, on tight, roomy and composite
    moduli: rows of 1 ... 4099 columns (shorter than a wave, ragged against 64 lanes, long enough for several
    normalisations of the columns: 63 / L terms each), all-extreme rows (every product (p - 1)^2: the largest column
    sums), rows of zeros, more rows than one launch's waves take in one pass."""
    from oracle.fp_oracle import Oracle
    _, p, n64 = case
    tuning("FPGPU_MATVEC_COLS", kernel)
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(p % 1000003 + kernel)
    edge = [0, 1, 2, p - 1, p - 2, p >> 1, (p >> 1) + 1]
    for rows, cols in ((1, 1), (3, 63), (5, 64), (7, 65), (2, 130), (9, 4099), (300, 17)):
        A = [rnd.choice(edge) if rnd.random() < 0.2 else rnd.randrange(p) for _ in range(rows * cols)]
        x = [rnd.choice(edge) if rnd.random() < 0.2 else rnd.randrange(p) for _ in range(cols)]
        A[:cols] = [p - 1] * cols                    # row 0: the largest products throughout (when x is)
        if rows > 1:
            A[cols:2 * cols] = [0] * cols            # row 1: nothing
        if rows == 9:
            x = [p - 1] * cols
        A, x = ints_to_array(A, n64), ints_to_array(x, n64)
        want = o.matvec(A, rows, cols, x, threads=0)
        assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols, x, False).reshape(want.shape), want), (rows, cols)
    rows, cols = 70003, 3                            # one row per wave: far more rows than waves in a pass of the grid cap
    A = ints_to_array([rnd.randrange(p) for _ in range(rows * cols)], n64)
    x = ints_to_array([rnd.randrange(p) for _ in range(cols)], n64)
    want = o.matvec(A, rows, cols, x, threads=0)
    assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols, x, False).reshape(want.shape), want)


@pytest.mark.parametrize("case", _MV_VALU_MODULI, ids=[c[0] for c in _MV_VALU_MODULI])
def test_matvec_one_row_per_lane_every_width(cpa, tuning, case):
    """k_matvec_rows (round 3: tall, narrow matrices -- a lane owns a row, x[j] is a scalar load, no cross-lane step):
    forced with FPGPU_MATVEC_ROWS = 1 on shapes of every kind (one column, column counts around its unroll of 4 / 2,
    more columns than one normalisation interval, fewer rows than a wave, all-extreme rows), then chosen by the
    launcher itself on a 70 003 x 5 matrix with and without a workspace; against the oracle bit for bit."""
    from oracle.fp_oracle import Oracle
    _, p, n64 = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(p % 1000003)
    edge = [0, 1, 2, p - 1, p - 2, p >> 1]
    tuning("FPGPU_MATVEC_ROWS", 1)
    for rows, cols in ((1, 1), (3, 2), (65, 3), (7, 4), (130, 5), (9, 33), (300, 70), (2, 1000)):
        A = [rnd.choice(edge) if rnd.random() < 0.2 else rnd.randrange(p) for _ in range(rows * cols)]
        x = [rnd.choice(edge) if rnd.random() < 0.2 else rnd.randrange(p) for _ in range(cols)]
        A[:cols] = [p - 1] * cols
        if rows == 9:
            x = [p - 1] * cols
        A, x = ints_to_array(A, n64), ints_to_array(x, n64)
        want = o.matvec(A, rows, cols, x, threads=0)
        assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols, x, False).reshape(want.shape), want), (rows, cols)
    tuning("FPGPU_MATVEC_ROWS", -1)                    # the launcher's own choice
    rows, cols = 70003, 5
    rng = np.random.default_rng(n64)
    from soak_inversion_plans import rand_elems
    A, x = rand_elems(rng, rows * cols, n64, p), rand_elems(rng, cols, n64, p)
    want = o.matvec(A, rows, cols, x, threads=0)
    for use_ws in (False, True):
        assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols, x, use_ws).reshape(want.shape), want), use_ws


@pytest.mark.parametrize("case", _UNSAT_MODULI, ids=[c[0] for c in _UNSAT_MODULI])
def test_every_width_elementwise_fold_matvec_spmv(cpa, case):
    """Every compiled width (n64 = 3, 5, 7, 8 included, tight and roomy moduli) through the kernels the
    chain tests above do not reach: k_elementwise for all seven operations (the unsaturated
    multiplier from 384 bits on), the sum fold, the dot product, k_spmv, and the mat-vec on both
    paths -- matrix cores (NT = n64, up to the 512-thread expander at NT = 8) and VALU."""
    from oracle.fp_oracle import Oracle
    _, p, n64 = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(p % 999983)
    edge = [0, 1, 2, p - 1, p - 2, p >> 1, (p >> 1) + 1]
    for n in (1, 65, 1031):
        xs = (edge + [rnd.randrange(p) for _ in range(n)])[:n]
        ys = ([p - 1, 1, p - 2] + [rnd.randrange(p) for _ in range(n)])[:n]
        a, b = ints_to_array(xs, n64), ints_to_array(ys, n64)
        assert np.array_equal(f.mul(a, b), o.mul(a, b))
        assert np.array_equal(f.sqr(a), o.sqr(a))
        assert np.array_equal(f.add(a, b), o.add(a, b))
        assert np.array_equal(f.sub(a, b), o.sub(a, b))
        assert np.array_equal(f.neg(a), o.neg(a))
        # double / div_by_2 (const_monty.rs:81-89): against the oracle, against Python ints on the
        # representatives, and half + half == a (the reference's own check, const_monty.rs:1184-1187)
        inv2 = pow(2, -1, p)
        assert np.array_equal(f.double(a), o.double(a)) and array_to_ints(f.double(a)) == [2 * v % p for v in xs]
        half = f.div_by_2(a)
        assert np.array_equal(half, o.div_by_2(a)) and array_to_ints(half) == [v * inv2 % p for v in xs]
        assert np.array_equal(f.add(half, half), a)
        assert np.array_equal(f.lift(a), o.from_mont(a))
        assert np.array_equal(f.sum(a), o.sum(a))
        assert np.array_equal(f.dot(a, b), o.sum(o.mul(a, b)))
        # project takes ANY n64-word value (also >= p), mul any a < R with b < p
        R = 1 << (64 * n64)
        wide = ints_to_array([R - 1, p, p + 1] + [rnd.randrange(R) for _ in range(n)], n64)[:n]
        assert np.array_equal(f.project(wide), o.to_mont(wide))
        assert np.array_equal(f.mul(wide, b), o.mul(wide, b))
    # mat-vec: row bytes a multiple of 16 and >= 64 so that the matrix-core plan is usable
    cols = 24 if n64 % 2 else 10
    assert cols * 8 * n64 >= 64 and (cols * 8 * n64) % 16 == 0
    for rows, cols_ in ((5, cols), (131, cols * 7), (3, 1)):
        A = ints_to_array([rnd.choice((p - 1, rnd.randrange(p))) for _ in range(rows * cols_)], n64)
        x = ints_to_array([rnd.choice((p - 1, p >> 1, rnd.randrange(p))) for _ in range(cols_)], n64)
        want = o.matvec(A, rows, cols_, x, threads=0)
        if cols_ > 1:
            assert f.matvec_workspace_bytes(rows, cols_) > 0
        assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols_, x, True).reshape(want.shape), want), ("mfma", rows, cols_)
        assert np.array_equal(_dev_matvec(cpa, f, A, rows, cols_, x, False).reshape(want.shape), want), ("valu", rows, cols_)
        assert np.array_equal(f.matvec(A, rows, cols_, x), want)
    rows, cols_, density = 37, 50, 6
    vals = ints_to_array([rnd.randrange(p) for _ in range(rows * density)], n64)
    x = ints_to_array([rnd.randrange(p) for _ in range(cols_)], n64)
    col = np.random.default_rng(n64).integers(0, cols_, size=rows * density, dtype=np.uint32)
    assert np.array_equal(f.spmv(vals, col, rows, cols_, density, x), o.spmv(vals, col, rows, cols_, density, x))


@pytest.mark.parametrize("name,p,n64", [("bls12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
                                        ("secp256k1 (tight)", 2**256 - 2**32 - 977, 4), ("p512 (tight)", 2**512 - 569, 8),
                                        ("odd composite", 3 * (2**107 - 1), 2)])
def test_batch_inv_multi_level_plans(cpa, name, p, n64, tuning):
    """Chains of 2 make 140001 elements plan TWO levels of prefix products below the top (70001 and
    35001 totals: k_binv_up / k_binv_down<N, false>, the per-level index map of the fallback kernel,
    a poisoned group reaching the top through two levels), compared bit for bit with the oracle."""
    from oracle.fp_oracle import Oracle
    tuning("FPGPU_INV_C", 2)
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    n = 140001
    assert f.inv_workspace_bytes(n) > (n // 2 + n // 4) * 8 * n64       # two stored levels of totals
    rnd = random.Random(n64)
    xs = [rnd.randrange(1, p) for _ in range(n)]
    for i in range(0, n, 997):
        xs[i] = 0
    xs[-1] = 0
    if "composite" in name:
        xs[5] = 3
        xs[70000] = (2**107 - 1) * 5
        xs[139999] = 9
    x = ints_to_array(xs, n64)
    got, flags = f.inv(x)
    want, wflags = o.batch_inv(x)
    assert np.array_equal(flags, wflags)
    assert np.array_equal(got, want)
    if "composite" in name:
        assert flags[5] == 1 and flags[70000] == 1 and flags[139999] == 1 and flags[6] == 0


@pytest.mark.parametrize("name,p,n64", [("bls12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
                                        ("p384 tight", 2**384 - 317, 6), ("p512 (tight)", 2**512 - 569, 8), ("p448", 2**448 - 2**224 - 1, 7),
                                        ("secp256k1 (tight)", 2**256 - 2**32 - 977, 4), ("goldilocks", 2**64 - 2**32 + 1, 1),
                                        ("odd composite 384", 3 * (2**380 - 2**190 + 5) | 1, 6)])
@pytest.mark.parametrize("fuse", [1, 0])
@pytest.mark.parametrize("fast_ws", [False, True])
def test_batch_inv_every_path_of_the_round3_kernels(cpa, tuning, name, p, n64, fuse, fast_ws):
    """The sweeps multiply in the Montgomery domain of R' = 2^(29 L), a chain starts as its first element (no
    multiplication by one, no prefix for it), the last stored level rides in the top kernel (k_binv_top_chain, LDS
    prefixes) unless FPGPU_INV_FUSE_TOP = 0, and with the larger workspace the level-0 prefixes are kept as limbs
    in 64-element tiles: every combination against the oracle, bit for bit, on tight, roomy and composite moduli,
    with zeros at the start, inside and at the end of chains, whole chains of zeros, and chains of one element."""
    import torch
    from oracle.fp_oracle import Oracle
    tuning("FPGPU_INV_FUSE_TOP", fuse)
    tuning("FPGPU_INV_C0", 8)
    tuning("FPGPU_INV_C", 4)                       # level 0: 8, above it: 4 (fused when fuse = 1)
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(n64 * 7 + fuse)
    for n in (1, 7, 8, 9, 4097, 70001):
        m = (n + 7) // 8                          # chains at level 0: chain c = {c + k m}
        xs = [rnd.randrange(1, p) for _ in range(n)]
        for i in range(0, n, 1013):
            xs[i] = 0
        xs[0] = 0                                 # the first element of chain 0
        if 4 * m + 1 < n:
            xs[4 * m + 1] = 0                     # inside chain 1
            for k in range(0, n - 3, m):          # chain 3: nothing but zeros
                xs[k + 3] = 0
        xs[-1] = 0
        if "composite" in name and n > 1000:
            xs[5], xs[n // 2] = 3, 9
        x = ints_to_array(xs, n64)
        want, wflags = o.batch_inv(x)
        dx = torch.from_numpy(x.view(np.int64)).cuda()
        out = f.alloc(n)
        flags = torch.empty(n, dtype=torch.uint8, device="cuda")
        ws = torch.empty(f.inv_workspace_bytes(n, fast=fast_ws), dtype=torch.uint8, device="cuda")
        f.inv_dev(dx, out, flags, ws)
        torch.cuda.synchronize()
        assert np.array_equal(flags.cpu().numpy(), wflags), (n, "flags")
        assert np.array_equal(out.cpu().numpy().view(np.uint64), want), n
    assert f.inv_workspace_bytes(1 << 20, fast=True) >= f.inv_workspace_bytes(1 << 20)
    if n64 >= 6:
        assert f.inv_workspace_bytes(1 << 20, fast=True) > f.inv_workspace_bytes(1 << 20) + (1 << 20) * 8 * n64


@pytest.mark.parametrize("name,p,n64", [("goldilocks", 2**64 - 2**32 + 1, 1), ("p127 in two words", 2**127 - 1, 2), ("p192 tight", 2**192 - 2**64 - 1, 3),
                                        ("bn254", 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4),
                                        ("secp256k1 (tight)", 2**256 - 2**32 - 977, 4), ("p320", 2**320 - 197, 5),
                                        ("bls12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
                                        ("odd composite 128", 3 * (2**125 - 2**63 + 5) | 1, 2)])
@pytest.mark.parametrize("c0", [2, 3, 4, 5, 9, 32])
def test_batch_inv_sparse_prefixes(cpa, tuning, name, p, n64, c0):
    """k_binv_up_sp / k_binv_down_sp (round 3): only every fourth prefix of a level-0 chain is stored, the down-sweep
    re-multiplies the others segment by segment.  Forced with FPGPU_INV_SP = 1 at every width the minimum workspace
    serves, chain lengths below, at and above the segment length (2, 3, 4, 5, 9, 32), lengths that leave ragged
    chains and ragged last segments, zeros at the start, at segment boundaries and inside segments, whole chains of
    zeros; against the oracle bit for bit, and against the dense sweeps."""
    import torch
    from oracle.fp_oracle import Oracle
    tuning("FPGPU_INV_SP", 1)
    tuning("FPGPU_INV_C0", c0)
    tuning("FPGPU_INV_C", 4)
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(n64 * 100 + c0)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 17, 4097, 20011):
        m = (n + c0 - 1) // c0                     # chains at level 0: chain c = {c + k m}
        xs = [rnd.randrange(1, p) for _ in range(n)]
        for i in range(0, n, 211):
            xs[i] = 0
        xs[0] = 0                                  # the first element of chain 0
        for k in (3, 4, 5):                        # around the first segment boundary of chain 1
            if k * m + 1 < n:
                xs[k * m + 1] = 0
        if 2 < m:
            for k in range(0, n - 2, m):           # chain 2: nothing but zeros
                xs[k + 2] = 0
        xs[-1] = 0
        if "composite" in name and n > 100:
            xs[50], xs[n // 2] = 3, 9
        x = ints_to_array(xs, n64)
        want, wflags = o.batch_inv(x)
        dx = torch.from_numpy(x.view(np.int64)).cuda()
        out = f.alloc(n)
        flags = torch.empty(n, dtype=torch.uint8, device="cuda")
        ws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")   # the minimum: no limb-form prefixes
        f.inv_dev(dx, out, flags, ws)
        torch.cuda.synchronize()
        assert np.array_equal(flags.cpu().numpy(), wflags), (n, "flags")
        assert np.array_equal(out.cpu().numpy().view(np.uint64).reshape(n, n64), want), n


def test_device_wrappers_validate_shapes(cpa):
    """The Python device wrappers refuse operands whose element counts do not match what the kernel
    will touch (a short `out` would otherwise be written past its end)."""
    import torch
    f = cpa.MontyField(cpa.BN254_FR, 4)
    a, b = f.alloc(100), f.alloc(100)
    with pytest.raises(ValueError, match="holds 99 elements"):
        f.mul_dev(a, b, f.alloc(99))
    with pytest.raises(ValueError, match="holds 99 elements"):
        f.add_dev(a, f.alloc(99), b)
    with pytest.raises(ValueError, match="holds"):
        f.matvec_dev(a, 10, 11, f.alloc(11), f.alloc(10))
    with pytest.raises(ValueError, match="holds"):
        f.matmul_dev(a, b, f.alloc(99), 10, 10, 10)
    with pytest.raises(ValueError, match="flags holds"):
        f.inv_dev(a, b, torch.empty(50, dtype=torch.uint8, device="cuda"),
                  torch.empty(f.inv_workspace_bytes(100), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError, match="workspace holds"):
        f.inv_dev(a, b, None, torch.empty(16, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError, match="CUDA tensor"):
        f.mul_dev(a.cpu(), b, a)
    with pytest.raises(ValueError, match="operand length mismatch"):
        f.mul_chain(np.ones((4, 4), dtype=np.uint64), np.ones((3, 4), dtype=np.uint64), 2)


@pytest.mark.parametrize("case", _UNSAT_MODULI + [("goldilocks", 2**64 - 2**32 + 1, 1), ("p127", 2**127 - 1, 2)],
                         ids=lambda c: c[0])
@pytest.mark.parametrize("window", [0, 1])
def test_pow_both_ladders_every_width(cpa, tuning, case, window):
    """The binary ladder (FPGPU_POW_WINDOW=0) and the sliding-window ladder over x, x^3, x^5, x^7 (=1) must both
    give the oracle's result (the reference's 4-bit-window ladder) at every compiled width, tight and lazy
    moduli: exponents that exercise every window value and length, runs of zeros and ones, a first window
    of one, two and three bits, and the exponent's top bit in every position of a word boundary."""
    from oracle.fp_oracle import Oracle
    _, p, n64 = case
    tuning("FPGPU_POW_WINDOW", window)
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(p % 999983 + window)
    xs = [0, 1, 2, p - 1, p - 2] + [rnd.randrange(p) for _ in range(200)]
    x = ints_to_array(xs, n64)
    bits = 64 * n64
    exps = [2, 3, 4, 5, 6, 7, 0b1011, 0b1101, 0b1111, 0b10001, 0b100101, 0b1110111, 0xAAAAAAAA, 0x55555555, 0xDB6DB6DB,
            2**31, 2**32, 2**32 + 1, 2**33 - 1, (1 << bits) - 1, (1 << (bits - 1)) | 1, p - 2, p - 1, p, (p - 1) // 2,
            rnd.getrandbits(bits), rnd.getrandbits(bits) | (1 << (bits - 1)), rnd.getrandbits(bits - 1), rnd.getrandbits(70)]
    for e in exps:
        ebits = max(bits, 64 * ((e.bit_length() + 63) // 64))      # the exponent may be wider than the field
        assert np.array_equal(f.pow(x, e), o.pow(x, e, exponent_bits=ebits)), hex(e)


@pytest.mark.parametrize("case", _UNSAT_MODULI + [("goldilocks", 2**64 - 2**32 + 1, 1), ("p127", 2**127 - 1, 2)], ids=lambda c: c[0])
def test_exact_inversion_special_values(cpa, case):
    """The exact inversion at the top of the batch inversion (binary GCD on word-sized approximations, update
    factors applied in product-scanning passes whose bounds rely on |f| + |g| <= 2^30) on values that drive the
    approximations and the updates to their extremes: powers of two and their neighbours, p minus those, small
    numbers, (p +- 1) / 2.  A batch of one element makes the top level invert that element itself; a batch of
    all of them checks the same values inside chains."""
    from oracle.fp_oracle import Oracle
    _, p, n64 = case
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    bits = p.bit_length()
    vals = {1, 2, 3, 5, p - 1, p - 2, p - 3, (p - 1) // 2, (p + 1) // 2, (p - 1) // 2 - 1}
    for k in range(1, bits):
        for d in (-1, 0, 1):
            for v in ((1 << k) + d, p - (1 << k) + d):
                if 0 < v < p:
                    vals.add(v)
    vals = sorted(vals)
    x = ints_to_array(vals, n64)
    want, wflags = o.batch_inv(x)
    got, flags = f.inv(x)
    assert np.array_equal(flags, wflags) and np.array_equal(got, want)
    step = max(1, len(vals) // 120)
    for i in range(0, len(vals), step):
        g1, f1 = f.inv(x[i:i + 1])
        assert np.array_equal(g1, want[i:i + 1]) and f1[0] == wflags[i], hex(vals[i])
