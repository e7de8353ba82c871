"""ctypes binding of the CPU oracle (oracle/fp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Element arrays are numpy uint64 arrays of shape (count, n64):
little-endian 64-bit words, the reference's `as_words()` layout
(src/semiring/crypto_bigint_uint.rs:65-68).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfp_oracle.so")

MUL, SQR, ADD, SUB, NEG, TO_MONT, FROM_MONT, MUL_BOXED, DOUBLE, DIV_BY_2 = range(10)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
        os.path.join(_HERE, "fp_oracle.c")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfp_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.fpo_mod_neg_inv.restype = ctypes.c_uint64
        _lib.fpo_mod_neg_inv.argtypes = [ctypes.c_uint64]
        _lib.fpo_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _arr(x, n64=None):
    a = np.ascontiguousarray(x, dtype=np.uint64)
    if n64 is not None:
        a = a.reshape(-1, n64)
    return a


def int_to_words(v, n64):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n64)], dtype=np.uint64)


def words_to_int(w):
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(w).reshape(-1)))


def ints_to_array(vals, n64):
    out = np.zeros((len(vals), n64), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = int_to_words(v, n64)
    return out


def array_to_ints(a):
    a = np.asarray(a)
    return [words_to_int(r) for r in a]


class Oracle:
    """The reference's scalar ops looped over element arrays, for one modulus."""

    def __init__(self, modulus: int, n64: int):
        self.n64 = n64
        self.p = modulus
        self.mod = int_to_words(modulus, n64)
        inv = ctypes.c_uint64()
        one = np.zeros(n64, dtype=np.uint64)
        r2 = np.zeros(n64, dtype=np.uint64)
        rc = lib().fpo_params(_p(self.mod), ctypes.c_size_t(n64), ctypes.byref(inv), _p(one), _p(r2))
        if rc != 0:
            raise ValueError("InvalidModulus")
        self.mod_neg_inv = inv.value
        self.one = one
        self.r2 = r2

    def batch(self, op, a, b=None, threads=1):
        a = _arr(a, self.n64)
        b = _arr(b, self.n64) if b is not None else None
        out = np.empty_like(a)
        rc = lib().fpo_batch_op(
            ctypes.c_int(op), _p(self.mod), ctypes.c_size_t(self.n64), _p(a), _p(b), _p(out),
            ctypes.c_size_t(a.shape[0]), ctypes.c_int(threads))
        assert rc == 0
        return out

    def mul(self, a, b, threads=1): return self.batch(MUL, a, b, threads)
    def mul_boxed(self, a, b, threads=1): return self.batch(MUL_BOXED, a, b, threads)
    def sqr(self, a, threads=1): return self.batch(SQR, a, None, threads)
    def add(self, a, b, threads=1): return self.batch(ADD, a, b, threads)
    def sub(self, a, b, threads=1): return self.batch(SUB, a, b, threads)
    def neg(self, a, threads=1): return self.batch(NEG, a, None, threads)
    def double(self, a, threads=1): return self.batch(DOUBLE, a, None, threads)
    def div_by_2(self, a, threads=1): return self.batch(DIV_BY_2, a, None, threads)
    def to_mont(self, a, threads=1): return self.batch(TO_MONT, a, None, threads)
    def from_mont(self, a, threads=1): return self.batch(FROM_MONT, a, None, threads)

    def batch_inv(self, x, threads=1):
        x = _arr(x, self.n64)
        out = np.empty_like(x)
        flags = np.zeros(x.shape[0], dtype=np.uint8)
        rc = lib().fpo_batch_inv(_p(self.mod), ctypes.c_size_t(self.n64), _p(x), _p(out), _p(flags),
                                 ctypes.c_size_t(x.shape[0]), ctypes.c_int(threads))
        assert rc == 0
        return out, flags

    def sum(self, v):
        v = _arr(v, self.n64)
        out = np.zeros(self.n64, dtype=np.uint64)
        lib().fpo_sum(_p(self.mod), ctypes.c_size_t(self.n64), _p(v), ctypes.c_size_t(v.shape[0]), _p(out))
        return out

    def product(self, v):
        v = _arr(v, self.n64)
        out = np.zeros(self.n64, dtype=np.uint64)
        lib().fpo_product(_p(self.mod), ctypes.c_size_t(self.n64), _p(v), ctypes.c_size_t(v.shape[0]), _p(out))
        return out

    def matvec(self, A, rows, cols, x, threads=1):
        A = _arr(A, self.n64)
        x = _arr(x, self.n64)
        assert A.shape[0] == rows * cols and x.shape[0] == cols
        y = np.zeros((rows, self.n64), dtype=np.uint64)
        lib().fpo_matvec(_p(self.mod), ctypes.c_size_t(self.n64), _p(A), ctypes.c_size_t(rows),
                         ctypes.c_size_t(cols), _p(x), _p(y), ctypes.c_int(threads))
        return y

    def spmv(self, vals, col_idx, rows, cols, density, x):
        vals, x = _arr(vals, self.n64), _arr(x, self.n64)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        assert vals.shape[0] == rows * density == col_idx.shape[0] and x.shape[0] == cols
        y = np.zeros((rows, self.n64), dtype=np.uint64)
        rc = lib().fpo_spmv(_p(self.mod), ctypes.c_size_t(self.n64), _p(vals), col_idx.ctypes.data_as(ctypes.c_void_p),
                            ctypes.c_size_t(rows), ctypes.c_size_t(cols), ctypes.c_size_t(density), _p(x), _p(y))
        if rc:
            raise IndexError("column index out of range")
        return y

    def matmul(self, A, B, m, k, n, threads=1):
        A = _arr(A, self.n64)
        B = _arr(B, self.n64)
        assert A.shape[0] == m * k and B.shape[0] == k * n
        C = np.zeros((m * n, self.n64), dtype=np.uint64)
        lib().fpo_matmul(_p(self.mod), ctypes.c_size_t(self.n64), _p(A), _p(B), _p(C),
                         ctypes.c_size_t(m), ctypes.c_size_t(k), ctypes.c_size_t(n), ctypes.c_int(threads))
        return C

    def pow(self, x, exponent: int, exponent_bits=None, threads=1):
        """x[i]^exponent for every i (one exponent for the batch), as pow_bounded_exp does it."""
        x = _arr(x, self.n64)
        if exponent_bits is None:
            exponent_bits = 64 * self.n64
        nw = max(1, (exponent_bits + 63) // 64)
        ew = int_to_words(exponent & ((1 << (64 * nw)) - 1), nw)
        out = np.empty_like(x)
        rc = lib().fpo_batch_pow(_p(self.mod), ctypes.c_size_t(self.n64), _p(x), _p(ew),
                                 ctypes.c_uint32(exponent_bits), _p(out), ctypes.c_size_t(x.shape[0]),
                                 ctypes.c_int(threads))
        assert rc == 0
        return out

    def bench_mul_assign(self, elem, iters):
        elem = _arr(elem).reshape(-1)
        out = np.zeros(self.n64, dtype=np.uint64)
        lib().fpo_bench_mul_assign(_p(self.mod), ctypes.c_size_t(self.n64), _p(elem),
                                   ctypes.c_size_t(iters), _p(out))
        return out


REF_BENCH_GROUPS = ("Mul ref by ref", "Mul assign", "Add ref to ref", "Add assign", "Div ref by ref", "Negation",
                    "Sum", "Product")
REF_BENCH_SEEDS = (695962179703, 2345695962179703, 111111111111111111, 12345678124578658568)  # field_ops_bench_common.rs:121-124


def bench_reference_shape(modulus: int, n64: int, iters: int = 10000):
    """The reference's criterion groups (benches/field_ops_bench_common.rs) on this CPU: for each of its
    four seeds projected into the field, nanoseconds per `iters` operations of each group, single core."""
    o = Oracle(modulus, n64)
    out = {}
    for seed in REF_BENCH_SEEDS:
        elem = o.to_mont(int_to_words(seed % (1 << (64 * n64)), n64).reshape(1, -1)).reshape(-1)
        ns = np.zeros(8, dtype=np.float64)
        sink = np.zeros(1, dtype=np.uint64)
        rc = lib().fpo_bench_reference_shape(_p(o.mod), ctypes.c_size_t(n64), _p(elem), ctypes.c_size_t(iters),
                                             ns.ctypes.data_as(ctypes.c_void_p), _p(sink))
        assert rc == 0
        out[str(seed)] = {g: float(v) for g, v in zip(REF_BENCH_GROUPS, ns)}
    return out


def max_threads():
    return lib().fpo_max_threads()
