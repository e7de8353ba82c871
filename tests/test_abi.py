"""CPU: the C-ABI shared library loads and exports every symbol include/fpgpu.h
declares; context set-up (host code) matches the golden Montgomery constants; argument
errors are reported through status codes; with no GPU the batch calls fail loudly
(there is no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, array_to_ints, h2i

import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "fpgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fpgpu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = cpa.load()
    names = _declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libfpgpu.so does not export {n}"
    # and the Python prototypes cover the same set
    assert sorted(_lib.SIGNATURES) == names


def test_ctx_params_match_golden(golden):
    f = cpa.MontyField(h2i(golden["modulus"]), golden["n64"])
    assert f.mod_neg_inv == h2i(golden["mod_neg_inv"])
    assert array_to_ints(f.one_words.reshape(1, -1))[0] == h2i(golden["one"])
    assert array_to_ints(f.r2_words.reshape(1, -1))[0] == h2i(golden["r2"])
    assert array_to_ints(f.modulus_words.reshape(1, -1))[0] == h2i(golden["modulus"])


def test_invalid_modulus_and_unsupported_width():
    with pytest.raises(cpa.InvalidModulus):
        cpa.MontyField(42, 4)          # even: FieldError::InvalidModulus (crypto_bigint_monty.rs:1050-1055)
    with pytest.raises(cpa.InvalidModulus):
        cpa.MontyField(0, 1)
    with pytest.raises(cpa.FpgpuError) as e:
        cpa.MontyField(3, 9)           # wider than any compiled kernel
    assert e.value.code == _lib.ERR_UNSUPPORTED


def test_argument_checks_need_no_gpu():
    lib = cpa.load()
    f = cpa.MontyField(cpa.BN254_FR, 4)
    # n == 0 is a no-op for every entry point
    assert lib.fpgpu_mul_dev(f._h, None, None, None, 0, None) == _lib.OK
    assert lib.fpgpu_mul(f._h, None, None, None, 0) == _lib.OK
    assert lib.fpgpu_batch_inv(f._h, None, None, None, 0) == _lib.OK
    # null / misaligned device pointers are rejected before any launch
    assert lib.fpgpu_mul_dev(f._h, None, None, None, 4, None) == _lib.ERR_SHAPE
    assert lib.fpgpu_add_dev(f._h, ctypes.c_void_p(8), ctypes.c_void_p(16), ctypes.c_void_p(32), 4, None) == _lib.ERR_SHAPE
    assert b"misaligned" in lib.fpgpu_last_error_string()
    assert lib.fpgpu_batch_inv_dev(f._h, ctypes.c_void_p(64), ctypes.c_void_p(64), None, 4, None, 0, None) == _lib.ERR_SHAPE
    assert lib.fpgpu_mul_dev(None, None, None, None, 4, None) == _lib.ERR_SHAPE
    # workspace query is pure host arithmetic and grows with n
    w1 = lib.fpgpu_batch_inv_workspace_bytes(f._h, 1 << 10)
    w2 = lib.fpgpu_batch_inv_workspace_bytes(f._h, 1 << 20)
    assert 0 < w1 < w2 < (1 << 20) * 32


@pytest.mark.skipif(cpa.load().fpgpu_device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_gpu():
    f = cpa.MontyField(cpa.BN254_FR, 4)
    a = np.ones((4, 4), dtype=np.uint64)
    with pytest.raises(cpa.FpgpuError) as e:
        f.mul(a, a)
    assert e.value.code == _lib.ERR_NO_DEVICE
    with pytest.raises(cpa.FpgpuError):
        f.inv(a)


_C_TO_RUST = {
    "int": "c_int", "long": "c_long", "size_t": "usize", "uint32_t": "u32", "uint64_t": "u64", "void": "()",
    "const char *": "*const c_char", "const int *": "*const c_int", "int *": "*mut c_int", "size_t *": "*mut usize",
    "const void *": "*const c_void", "void *": "*mut c_void", "void **": "*mut *mut c_void",
    "const uint64_t *": "*const u64", "uint64_t *": "*mut u64", "const uint32_t *": "*const u32", "uint8_t *": "*mut u8",
    "const fpgpu_ctx *": "*const fpgpu_ctx", "fpgpu_ctx *": "*mut fpgpu_ctx", "fpgpu_ctx **": "*mut *mut fpgpu_ctx",
    "const fpgpu_vec *": "*const fpgpu_vec", "fpgpu_vec *": "*mut fpgpu_vec", "fpgpu_vec **": "*mut *mut fpgpu_vec",
    "const fpgpu_idx *": "*const fpgpu_idx", "fpgpu_idx *": "*mut fpgpu_idx", "fpgpu_idx **": "*mut *mut fpgpu_idx",
    "const fpgpu_devset *": "*const fpgpu_devset", "fpgpu_devset *": "*mut fpgpu_devset",
    "fpgpu_devset **": "*mut *mut fpgpu_devset",
}


def _c_param_type(decl):
    """'const uint64_t *modulus_le' -> 'const uint64_t *' (the declarator's name dropped, spacing normalised)."""
    decl = " ".join(decl.replace("*", " * ").split())
    toks = decl.split(" ")
    if toks[-1] != "*" and len(toks) > 1 and re.fullmatch(r"[A-Za-z_][A-Za-z0-9_]*", toks[-1]) and \
            toks[-1] not in ("int", "long", "void", "size_t", "uint32_t", "uint64_t", "uint8_t", "char"):
        toks = toks[:-1]
    t = " ".join(toks).replace(" * *", " **")
    return t


def test_rust_sys_crate_binds_every_declared_symbol():
    """rust/fpgpu-sys cannot be compiled in this image (no cargo); keep its extern block in step with
    the header: the same set of functions, and for every function the same parameter TYPES in the
    same order (pointer constness, usize / u32 / u64 / c_int widths) and the same return type."""
    hdr = open(os.path.join(ROOT, "include", "fpgpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    rs = open(os.path.join(ROOT, "rust", "fpgpu-sys", "src", "lib.rs")).read()
    rs = re.sub(r"//.*", "", rs)
    c_decl = {m.group(2): (m.group(1), m.group(3))
              for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(fpgpu_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr)}
    r_decl = {m.group(1): (m.group(3) or "()", m.group(2))
              for m in re.finditer(r"pub fn (fpgpu_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+?))?\s*;", rs)}
    assert sorted(c_decl) == sorted(r_decl)
    assert len(c_decl) >= 70
    for name, (cret, cargs) in c_decl.items():
        rret, rargs = r_decl[name]
        cargs = cargs.strip()
        ctypes_ = [] if cargs in ("", "void") else [_c_param_type(a) for a in cargs.split(",")]
        rtypes = [a.split(":", 1)[1].strip() for a in rargs.split(",") if a.strip()]
        assert [_C_TO_RUST[t] for t in ctypes_] == rtypes, (name, ctypes_, rtypes)
        assert _C_TO_RUST[_c_param_type(cret + " x")] == rret.strip(), (name, cret, rret)


def test_batch_inv_workspace_plan_is_sane(tuning):
    """Host arithmetic of the inversion plan: the workspace grows with n, stays a small fraction of
    the batch for the default chain lengths, and follows the FPGPU_INV_C tuning hook."""
    lib = cpa.load()
    f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
    sizes = [lib.fpgpu_batch_inv_workspace_bytes(f._h, n) for n in (1, 2, 255, 256, 257, 70001, 1 << 20, 1 << 24)]
    assert all(a <= b for a, b in zip(sizes, sizes[1:])) and sizes[0] > 0
    assert sizes[-1] < (1 << 24) * 48 // 8          # chains of 32 at the big level: totals are n/32 elements
    tuning("FPGPU_INV_C", 2)
    assert lib.fpgpu_batch_inv_workspace_bytes(f._h, 1 << 20) > (1 << 20) * 48 // 2
    assert lib.fpgpu_batch_inv_workspace_bytes(f._h, 0) == 0
    # an unknown hook is an error, not a silent no-op
    assert lib.fpgpu_tuning_set(b"FPGPU_NO_SUCH_HOOK", 1) == _lib.ERR_UNSUPPORTED


def test_static_scratch_entry_point_argument_checks():
    """fpgpu_batch_inv_static_dev: sizes beyond the static array are rejected with a clear error
    before anything touches a device; n == 0 is a no-op."""
    lib = cpa.load()
    f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
    cap = lib.fpgpu_static_scratch_bytes()
    assert cap == 64 << 20
    assert lib.fpgpu_batch_inv_workspace_bytes(f._h, 1 << 24) <= cap      # BASELINE configs[2] fits
    assert lib.fpgpu_batch_inv_static_dev(f._h, None, None, None, 0, None) == _lib.OK
    big = 1 << 27
    assert lib.fpgpu_batch_inv_workspace_bytes(f._h, big) > cap
    rc = lib.fpgpu_batch_inv_static_dev(f._h, ctypes.c_void_p(256), ctypes.c_void_p(512), None, big, None)
    assert rc == _lib.ERR_SHAPE and b"static array holds" in lib.fpgpu_last_error_string()


def test_spmv_argument_checks_need_no_gpu():
    lib = cpa.load()
    f = cpa.MontyField(cpa.BN254_FR, 4)
    vals = np.ones((2, 4), dtype=np.uint64)
    x = np.ones((3, 4), dtype=np.uint64)
    y = np.zeros((1, 4), dtype=np.uint64)
    col = np.array([0, 3], dtype=np.uint32)      # 3 is out of range for 3 columns: the reference would panic
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.fpgpu_spmv(f._h, p(vals), p(col), 1, 3, 2, p(x), p(y)) == _lib.ERR_SHAPE
    assert b"column index" in lib.fpgpu_last_error_string()
    assert lib.fpgpu_spmv(f._h, None, None, 0, 3, 2, None, None) == _lib.OK          # no rows: nothing to do
    assert lib.fpgpu_spmv_dev(f._h, None, None, 4, 3, 2, None, None, None) == _lib.ERR_SHAPE


def test_handle_api_argument_checks_need_no_gpu():
    """fpgpu_vec_* / fpgpu_set_devices: null handles and bad lists are reported through status codes;
    without a GPU an allocation is FPGPU_ERR_NO_DEVICE (no CPU fallback); clearing the list is always allowed."""
    lib = cpa.load()
    f = cpa.MontyField(cpa.BN254_FR, 4)
    assert lib.fpgpu_set_devices(None, 0) == _lib.OK
    assert lib.fpgpu_get_devices(None, 0) == 0
    assert lib.fpgpu_set_devices(None, 3) == _lib.ERR_SHAPE
    assert lib.fpgpu_vec_len(None) == 0 and lib.fpgpu_vec_parts(None) == 0
    lib.fpgpu_vec_free(None)
    assert lib.fpgpu_vec_mul(f._h, None, None, None) == _lib.ERR_SHAPE
    assert lib.fpgpu_vec_sync(None) == _lib.ERR_SHAPE
    assert lib.fpgpu_vec_upload(None, None, 0, 0) == _lib.ERR_SHAPE
    h = ctypes.c_void_p()
    rc = lib.fpgpu_vec_alloc(f._h, 16, _lib.PLACE_CURRENT, ctypes.byref(h))
    if lib.fpgpu_device_count() == 0:
        assert rc == _lib.ERR_NO_DEVICE and not h.value
    else:
        assert rc == _lib.OK
        lib.fpgpu_vec_free(h)
    assert lib.fpgpu_vec_alloc(f._h, 16, -77, ctypes.byref(h)) in (_lib.ERR_SHAPE, _lib.ERR_NO_DEVICE)
    assert lib.fpgpu_dot(f._h, None, None, 0, None) == _lib.ERR_SHAPE          # out is required even for n = 0
