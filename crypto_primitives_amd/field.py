"""Python view of the batched field backend: a thin object layer over the C ABI.

Mirrors the reference's config-style interface for the hot path
(`MontyField::{add, sub, mul, neg, inv, project, lift}`,
src/field/crypto_bigint_monty.rs:134-250,319-385) in batch form: every method
takes and returns arrays of Montgomery representatives laid out exactly as the
reference's `Vec<MontyFieldElement<LIMBS>>` (count x n64 little-endian u64 words).

Host arrays are numpy uint64 (count, n64); device arrays are torch int64 CUDA
tensors of the same shape (torch has no uint64 arithmetic; the bits are what
matter).  Nothing here computes on the CPU.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check


class InvalidModulus(ValueError):
    """FieldError::InvalidModulus (src/field.rs:345-349)."""


def _np(a, n64):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, n64)


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class MontyField:
    """Runtime-modulus prime field, batched (BaseFieldConfig::new + batch ops)."""

    def __init__(self, modulus: int, n64: int):
        lib = _lib.load()
        self.n64 = n64
        self.modulus_int = modulus
        words = np.array([(modulus >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n64)], dtype=np.uint64)
        if modulus >> (64 * n64):
            raise ValueError("modulus does not fit n64 words")
        h = ctypes.c_void_p()
        rc = lib.fpgpu_ctx_create(_ptr(words), n64, ctypes.byref(h))
        if rc == _lib.ERR_INVALID_MODULUS:
            raise InvalidModulus(lib.fpgpu_last_error_string().decode())
        check(rc)
        self._h = h
        self._lib = lib
        m = np.zeros(n64, dtype=np.uint64)
        one = np.zeros(n64, dtype=np.uint64)
        r2 = np.zeros(n64, dtype=np.uint64)
        inv = ctypes.c_uint64()
        check(lib.fpgpu_ctx_params(h, _ptr(m), _ptr(one), _ptr(r2), ctypes.byref(inv)))
        self.modulus_words, self.one_words, self.r2_words, self.mod_neg_inv = m, one, r2, inv.value

    @classmethod
    def new(cls, modulus: int, n64: int):
        return cls(modulus, n64)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.fpgpu_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- identities ------------------------------------------------------------------
    def zero(self):
        return np.zeros(self.n64, dtype=np.uint64)

    def one(self):
        return self.one_words.copy()

    # -- host-array batch ops (synchronous; H2D + kernel + D2H inside the library) ----
    def _bin(self, fn, a, b):
        a, b = _np(a, self.n64), _np(b, self.n64)
        if a.shape != b.shape:
            raise ValueError("operand length mismatch")
        out = np.empty_like(a)
        check(fn(self._h, _ptr(a), _ptr(b), _ptr(out), a.shape[0]))
        return out

    def _un(self, fn, a):
        a = _np(a, self.n64)
        out = np.empty_like(a)
        check(fn(self._h, _ptr(a), _ptr(out), a.shape[0]))
        return out

    def mul(self, a, b): return self._bin(self._lib.fpgpu_mul, a, b)
    def add(self, a, b): return self._bin(self._lib.fpgpu_add, a, b)
    def sub(self, a, b): return self._bin(self._lib.fpgpu_sub, a, b)
    def sqr(self, a): return self._un(self._lib.fpgpu_sqr, a)
    def neg(self, a): return self._un(self._lib.fpgpu_neg, a)
    def double(self, a): return self._un(self._lib.fpgpu_double, a)      # ConstMontyField::double, const_monty.rs:86-89
    def div_by_2(self, a): return self._un(self._lib.fpgpu_div_by_2, a)  # ConstMontyField::div_by_2, const_monty.rs:81-84
    def project(self, a): return self._un(self._lib.fpgpu_to_mont, a)    # integers -> Montgomery form
    def lift(self, a): return self._un(self._lib.fpgpu_from_mont, a)     # Montgomery form -> integers

    def mul_chain(self, a, b, iters):
        a, b = _np(a, self.n64), _np(b, self.n64)
        if a.shape != b.shape:
            raise ValueError("operand length mismatch")
        out = np.empty_like(a)
        check(self._lib.fpgpu_mul_chain(self._h, _ptr(a), _ptr(b), _ptr(out), a.shape[0], iters))
        return out

    def pow(self, a, exponent: int):
        """Batch FieldConfig::pow / pow_u32: a[i]^exponent, one exponent for the whole batch."""
        a = _np(a, self.n64)
        nw = max(1, (exponent.bit_length() + 63) // 64)
        ew = np.array([(exponent >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nw)], dtype=np.uint64)
        out = np.empty_like(a)
        check(self._lib.fpgpu_pow(self._h, _ptr(a), _ptr(ew), nw, _ptr(out), a.shape[0]))
        return out

    def sum(self, v):
        v = _np(v, self.n64)
        out = np.zeros(self.n64, dtype=np.uint64)
        check(self._lib.fpgpu_sum(self._h, _ptr(v) if v.shape[0] else None, v.shape[0], _ptr(out)))
        return out

    def product(self, v):
        v = _np(v, self.n64)
        out = np.zeros(self.n64, dtype=np.uint64)
        check(self._lib.fpgpu_product(self._h, _ptr(v) if v.shape[0] else None, v.shape[0], _ptr(out)))
        return out

    def inv(self, x):
        """Batch FieldConfig::inv.  Returns (out, none_flags): none_flags[i] = 1 where the
        reference returns None (out[i] = 0 there)."""
        x = _np(x, self.n64)
        out = np.empty_like(x)
        flags = np.zeros(x.shape[0], dtype=np.uint8)
        check(self._lib.fpgpu_batch_inv(self._h, _ptr(x), _ptr(out), _ptr(flags), x.shape[0]))
        return out, flags

    def checked_div(self, a, b):
        """Batch FieldConfig::checked_div (src/field.rs:168-170): (a / b, flags); flags[i] = 1 where the reference
        returns None (out[i] = 0 there).  One call: inversion and multiplication on the device."""
        a, b = _np(a, self.n64), _np(b, self.n64)
        if a.shape != b.shape:
            raise ValueError("operand length mismatch")
        out = np.zeros_like(a)
        flags = np.zeros(a.shape[0], dtype=np.uint8)
        check(self._lib.fpgpu_div(self._h, _ptr(a), _ptr(b), _ptr(out), _ptr(flags), a.shape[0]))
        return out, flags

    def div(self, a, b):
        """Batch FieldConfig::div (src/field.rs:155-157): raises on a zero divisor."""
        out, flags = self.checked_div(a, b)
        if flags.any():
            raise ZeroDivisionError("Division by zero")
        return out

    def matvec(self, A, rows, cols, x):
        A, x = _np(A, self.n64), _np(x, self.n64)
        if A.shape[0] != rows * cols or x.shape[0] != cols:
            raise ValueError("matrix / vector shape mismatch")
        y = np.zeros((rows, self.n64), dtype=np.uint64)
        check(self._lib.fpgpu_matvec(self._h, _ptr(A), rows, cols, _ptr(x), _ptr(y)))
        return y

    def spmv(self, vals, col_idx, rows, cols, density, x):
        """SparseMatrix (fixed `density` cells per row) times vector; col_idx is uint32[rows*density]."""
        vals, x = _np(vals, self.n64), _np(x, self.n64)
        col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        if vals.shape[0] != rows * density or col_idx.shape[0] != rows * density or x.shape[0] != cols:
            raise ValueError("sparse matrix / vector shape mismatch")
        y = np.zeros((rows, self.n64), dtype=np.uint64)
        check(self._lib.fpgpu_spmv(self._h, _ptr(vals), col_idx.ctypes.data_as(ctypes.c_void_p), rows, cols, density,
                                   _ptr(x), _ptr(y)))
        return y

    def matmul(self, A, B, m, k, n):
        A, B = _np(A, self.n64), _np(B, self.n64)
        if A.shape[0] != m * k or B.shape[0] != k * n:
            raise ValueError("matrix shape mismatch")
        C = np.zeros((m * n, self.n64), dtype=np.uint64)
        check(self._lib.fpgpu_matmul(self._h, _ptr(A), _ptr(B), _ptr(C), m, k, n))
        return C

    # -- handle API: operands are DeviceVector (crypto_primitives_amd.vec); asynchronous, no host copies --
    def _v3(self, fn, a, b, out):
        check(fn(self._h, a._h, b._h, out._h))
        return out

    def _v2(self, fn, a, out):
        check(fn(self._h, a._h, out._h))
        return out

    def mul_vec(self, a, b, out): return self._v3(self._lib.fpgpu_vec_mul, a, b, out)
    def add_vec(self, a, b, out): return self._v3(self._lib.fpgpu_vec_add, a, b, out)
    def sub_vec(self, a, b, out): return self._v3(self._lib.fpgpu_vec_sub, a, b, out)
    def sqr_vec(self, a, out): return self._v2(self._lib.fpgpu_vec_sqr, a, out)
    def neg_vec(self, a, out): return self._v2(self._lib.fpgpu_vec_neg, a, out)
    def double_vec(self, a, out): return self._v2(self._lib.fpgpu_vec_double, a, out)
    def div_by_2_vec(self, a, out): return self._v2(self._lib.fpgpu_vec_div_by_2, a, out)
    def project_vec(self, a, out): return self._v2(self._lib.fpgpu_vec_to_mont, a, out)
    def lift_vec(self, a, out): return self._v2(self._lib.fpgpu_vec_from_mont, a, out)

    def mul_chain_vec(self, a, b, out, iters):
        check(self._lib.fpgpu_vec_mul_chain(self._h, a._h, b._h, out._h, iters))
        return out

    def pow_vec(self, a, exponent: int, out):
        nw = max(1, (exponent.bit_length() + 63) // 64)
        ew = np.array([(exponent >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nw)], dtype=np.uint64)
        check(self._lib.fpgpu_vec_pow(self._h, a._h, _ptr(ew), nw, out._h))
        return out

    def inv_vec(self, x, out, want_flags=True):
        """Batch FieldConfig::inv on handles.  Returns the host flag array (1 = the reference's None)
        when want_flags, else None (and the call stays asynchronous)."""
        if not want_flags:
            check(self._lib.fpgpu_vec_batch_inv(self._h, x._h, out._h, None, None))
            return None
        flags = np.zeros(len(x), dtype=np.uint8)
        cnt = ctypes.c_size_t()
        check(self._lib.fpgpu_vec_batch_inv(self._h, x._h, out._h, _ptr(flags), ctypes.byref(cnt)))
        assert cnt.value == int(flags.sum())
        return flags

    def div_vec(self, a, b, out):
        """Batch FieldConfig::checked_div on handles: out = a / b; returns the host flag array (1 = None)."""
        flags = np.zeros(len(b), dtype=np.uint8)
        cnt = ctypes.c_size_t()
        check(self._lib.fpgpu_vec_div(self._h, a._h, b._h, out._h, _ptr(flags), ctypes.byref(cnt)))
        assert cnt.value == int(flags.sum())
        return flags

    def _vred(self, fn, *vs):
        out = np.zeros(self.n64, dtype=np.uint64)
        check(fn(self._h, *[v._h for v in vs], _ptr(out)))
        return out

    def sum_vec(self, v): return self._vred(self._lib.fpgpu_vec_sum, v)
    def product_vec(self, v): return self._vred(self._lib.fpgpu_vec_product, v)
    def dot_vec(self, a, b): return self._vred(self._lib.fpgpu_vec_dot, a, b)

    def matvec_vec(self, A, rows, cols, x, y):
        check(self._lib.fpgpu_vec_matvec(self._h, A._h, rows, cols, x._h, y._h))
        return y

    def spmv_vec(self, vals, col_idx, rows, cols, density, x, y):
        """SparseMatrix times vector on handles: vals a DeviceVector (alloc_rows(rows, density)), col_idx a DeviceIndices."""
        check(self._lib.fpgpu_vec_spmv(self._h, vals._h, col_idx._h, rows, cols, density, x._h, y._h))
        return y

    def matmul_vec(self, A, B, C, m, k, n):
        check(self._lib.fpgpu_vec_matmul(self._h, A._h, B._h, C._h, m, k, n))
        return C

    def dot(self, a, b):
        """sum_i mul(a[i], b[i]) on host arrays (fpgpu_dot)."""
        a, b = _np(a, self.n64), _np(b, self.n64)
        if a.shape != b.shape:
            raise ValueError("operand length mismatch")
        out = np.zeros(self.n64, dtype=np.uint64)
        check(self._lib.fpgpu_dot(self._h, _ptr(a) if a.shape[0] else None, _ptr(b) if b.shape[0] else None, a.shape[0], _ptr(out)))
        return out

    # -- device-tensor batch ops (asynchronous on torch's current stream) --------------
    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _chk_t(self, t, count=None, name="tensor"):
        """Device pointer of `t` after checking what the kernels assume: a contiguous CUDA tensor of 8-byte
        words with last dimension n64, on the current HIP device (whose stream the launch uses), holding
        exactly `count` elements when a count is given."""
        import torch
        if not t.is_cuda or not t.is_contiguous() or t.element_size() != 8 or t.dim() < 1 or t.shape[-1] != self.n64:
            raise ValueError("%s: expected a contiguous CUDA tensor of 8-byte words with last dim n64" % name)
        if t.device.index != torch.cuda.current_device():
            raise ValueError("%s lives on cuda:%d but the current device is cuda:%d" % (name, t.device.index, torch.cuda.current_device()))
        if count is not None and t.numel() != count * self.n64:
            raise ValueError("%s holds %d elements, expected %d" % (name, t.numel() // self.n64, count))
        return ctypes.c_void_p(t.data_ptr())

    @staticmethod
    def _chk_bytes(t, nbytes, name):
        """Device pointer and byte size of a raw buffer (flags, workspace) of at least `nbytes` bytes."""
        import torch
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError("%s: expected a contiguous CUDA tensor" % name)
        if t.device.index != torch.cuda.current_device():
            raise ValueError("%s lives on cuda:%d but the current device is cuda:%d" % (name, t.device.index, torch.cuda.current_device()))
        have = t.numel() * t.element_size()
        if have < nbytes:
            raise ValueError("%s holds %d bytes, needs %d" % (name, have, nbytes))
        return ctypes.c_void_p(t.data_ptr()), have

    def alloc(self, n, device=None):
        import torch
        return torch.empty((n, self.n64), dtype=torch.int64, device=device or "cuda")

    def random_dev(self, n, seed, device=None):
        t = self.alloc(n, device)
        check(self._lib.fpgpu_fill_random_dev(self._h, self._chk_t(t, n, "out"), n, seed, self._stream()))
        return t

    def _bin_dev(self, fn, a, b, out):
        n = a.numel() // self.n64
        check(fn(self._h, self._chk_t(a, n, "a"), self._chk_t(b, n, "b"), self._chk_t(out, n, "out"), n, self._stream()))
        return out

    def _un_dev(self, fn, a, out):
        n = a.numel() // self.n64
        check(fn(self._h, self._chk_t(a, n, "a"), self._chk_t(out, n, "out"), n, self._stream()))
        return out

    def mul_dev(self, a, b, out): return self._bin_dev(self._lib.fpgpu_mul_dev, a, b, out)
    def add_dev(self, a, b, out): return self._bin_dev(self._lib.fpgpu_add_dev, a, b, out)
    def sub_dev(self, a, b, out): return self._bin_dev(self._lib.fpgpu_sub_dev, a, b, out)
    def sqr_dev(self, a, out): return self._un_dev(self._lib.fpgpu_sqr_dev, a, out)
    def neg_dev(self, a, out): return self._un_dev(self._lib.fpgpu_neg_dev, a, out)
    def double_dev(self, a, out): return self._un_dev(self._lib.fpgpu_double_dev, a, out)
    def div_by_2_dev(self, a, out): return self._un_dev(self._lib.fpgpu_div_by_2_dev, a, out)
    def to_mont_dev(self, a, out): return self._un_dev(self._lib.fpgpu_to_mont_dev, a, out)
    def from_mont_dev(self, a, out): return self._un_dev(self._lib.fpgpu_from_mont_dev, a, out)

    def mul_chain_dev(self, a, b, out, iters):
        n = a.numel() // self.n64
        check(self._lib.fpgpu_mul_chain_dev(self._h, self._chk_t(a, n, "a"), self._chk_t(b, n, "b"), self._chk_t(out, n, "out"),
                                            n, iters, self._stream()))
        return out

    def pow_dev(self, a, exponent: int, out):
        n = a.numel() // self.n64
        nw = max(1, (exponent.bit_length() + 63) // 64)
        ew = np.array([(exponent >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nw)], dtype=np.uint64)
        check(self._lib.fpgpu_pow_dev(self._h, self._chk_t(a, n, "a"), _ptr(ew), nw, self._chk_t(out, n, "out"), n, self._stream()))
        return out

    def fold_workspace_bytes(self):
        return int(self._lib.fpgpu_fold_workspace_bytes(self._h))

    def _fold_dev(self, fn, operands, n, out, workspace):
        wp, wb = (None, 0) if workspace is None else self._chk_bytes(workspace, self.fold_workspace_bytes() if n > 256 else 0, "workspace")
        if n > 256 and workspace is None:
            raise ValueError("workspace needed for more than 256 elements (fold_workspace_bytes())")
        ptrs = [self._chk_t(t, n, "operand") for t in operands]
        check(fn(self._h, *ptrs, n, self._chk_t(out, 1, "out"), wp, wb, self._stream()))
        return out

    def sum_dev(self, v, out, workspace=None): return self._fold_dev(self._lib.fpgpu_sum_dev, [v], v.numel() // self.n64, out, workspace)
    def product_dev(self, v, out, workspace=None): return self._fold_dev(self._lib.fpgpu_product_dev, [v], v.numel() // self.n64, out, workspace)

    def dot_dev(self, a, b, out, workspace=None):
        """out[0] = sum_i mul(a[i], b[i]) (fpgpu_dot_dev)."""
        return self._fold_dev(self._lib.fpgpu_dot_dev, [a, b], a.numel() // self.n64, out, workspace)

    def inv_workspace_bytes(self, n, fast=False):
        """Device scratch of the batch inversion: the minimum, or (fast) the size that also holds the level-0
        prefixes as limbs -- the call picks its path by the size of the workspace it is given."""
        if fast:
            return self._lib.fpgpu_batch_inv_fast_workspace_bytes(self._h, n)
        return self._lib.fpgpu_batch_inv_workspace_bytes(self._h, n)

    def inv_dev(self, x, out, flags, workspace):
        n = x.numel() // self.n64
        fp = self._chk_bytes(flags, n, "flags")[0] if flags is not None else None
        wp, wb = self._chk_bytes(workspace, self.inv_workspace_bytes(n), "workspace")
        check(self._lib.fpgpu_batch_inv_dev(self._h, self._chk_t(x, n, "x"), self._chk_t(out, n, "out"), fp, n, wp, wb,
                                            self._stream()))
        return out

    def div_dev(self, a, b, out, flags, workspace):
        """fpgpu_div_dev: out = a / b (checked_div: flags[i] = 1 and out[i] = 0 where b[i] has no inverse)."""
        n = a.numel() // self.n64
        fp = self._chk_bytes(flags, n, "flags")[0] if flags is not None else None
        wp, wb = self._chk_bytes(workspace, self.inv_workspace_bytes(n), "workspace")
        check(self._lib.fpgpu_div_dev(self._h, self._chk_t(a, n, "a"), self._chk_t(b, n, "b"), self._chk_t(out, n, "out"),
                                      fp, n, wp, wb, self._stream()))
        return out

    def inv_static_dev(self, x, out, flags):
        """fpgpu_batch_inv_static_dev: batch inversion that allocates nothing and takes no workspace (the
        library's static device array); same results as inv_dev."""
        n = x.numel() // self.n64
        fp = self._chk_bytes(flags, n, "flags")[0] if flags is not None else None
        check(self._lib.fpgpu_batch_inv_static_dev(self._h, self._chk_t(x, n, "x"), self._chk_t(out, n, "out"), fp, n, self._stream()))
        return out

    def matvec_workspace_bytes(self, rows, cols):
        return int(self._lib.fpgpu_matvec_workspace_bytes(self._h, rows, cols))

    def matvec_dev(self, A, rows, cols, x, y, workspace=None):
        """`workspace`: optional uint8 device tensor of matvec_workspace_bytes(rows, cols) bytes; with it
        the products run on the matrix cores (same results)."""
        wp, wb = None, 0
        if workspace is not None:
            wp, wb = self._chk_bytes(workspace, 0, "workspace")
        check(self._lib.fpgpu_matvec_dev(self._h, self._chk_t(A, rows * cols, "A"), rows, cols, self._chk_t(x, cols, "x"),
                                         self._chk_t(y, rows, "y"), wp, wb, self._stream()))
        return y

    def spmv_dev(self, vals, col_idx, rows, cols, density, x, y):
        """SparseMatrix (fixed `density` cells per row) times vector on device tensors; col_idx is a
        contiguous 4-byte integer CUDA tensor of rows*density column indices (< cols)."""
        if col_idx.element_size() != 4 or col_idx.numel() != rows * density:
            raise ValueError("expected a contiguous CUDA tensor of rows*density 4-byte column indices")
        cp, _ = self._chk_bytes(col_idx, rows * density * 4, "col_idx")
        check(self._lib.fpgpu_spmv_dev(self._h, self._chk_t(vals, rows * density, "vals"), cp, rows, cols, density,
                                       self._chk_t(x, cols, "x"), self._chk_t(y, rows, "y"), self._stream()))
        return y

    def matmul_workspace_bytes(self, m, k, n):
        return int(self._lib.fpgpu_matmul_workspace_bytes(self._h, m, k, n))

    def matmul_dev(self, A, B, C, m, k, n, workspace=None):
        """`workspace`: optional uint8 device tensor of matmul_workspace_bytes(m, k, n) bytes; with it
        single-limb fields run on the matrix cores (same results)."""
        wp, wb = None, 0
        if workspace is not None:
            wp, wb = self._chk_bytes(workspace, 0, "workspace")
        check(self._lib.fpgpu_matmul_dev(self._h, self._chk_t(A, m * k, "A"), self._chk_t(B, k * n, "B"), self._chk_t(C, m * n, "C"),
                                         m, k, n, wp, wb, self._stream()))
        return C
