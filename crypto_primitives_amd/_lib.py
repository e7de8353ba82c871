"""ctypes loader for libfpgpu.so (the C ABI declared in include/fpgpu.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
crypto_primitives_amd``.  There is no fallback: if the shared object is missing
or no MI355X is visible, calls fail loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FPGPU_LIB") or os.path.join(_HERE, "libfpgpu.so")  # FPGPU_LIB: tuning builds

OK, ERR_INVALID_MODULUS, ERR_UNSUPPORTED, ERR_SHAPE, ERR_HIP, ERR_NO_DEVICE, ERR_RCCL = range(7)
PLACE_CURRENT, PLACE_SHARDED, PLACE_REPLICATED = -1, -2, -3

_c = ctypes
_vp, _sz, _u32, _u64, _int = _c.c_void_p, _c.c_size_t, _c.c_uint32, _c.c_uint64, _c.c_int

# name -> (restype, argtypes); mirrors include/fpgpu.h one to one
SIGNATURES = {
    "fpgpu_ctx_create": (_int, [_vp, _u32, _c.POINTER(_vp)]),
    "fpgpu_ctx_destroy": (None, [_vp]),
    "fpgpu_ctx_n64": (_u32, [_vp]),
    "fpgpu_ctx_params": (_int, [_vp, _vp, _vp, _vp, _c.POINTER(_u64)]),
    "fpgpu_mul_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "fpgpu_sqr_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "fpgpu_add_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "fpgpu_sub_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "fpgpu_neg_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "fpgpu_double_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "fpgpu_div_by_2_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "fpgpu_to_mont_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "fpgpu_from_mont_dev": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "fpgpu_mul_chain_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _u32, _vp]),
    "fpgpu_batch_inv_workspace_bytes": (_sz, [_vp, _sz]),
    "fpgpu_batch_inv_fast_workspace_bytes": (_sz, [_vp, _sz]),
    "fpgpu_batch_inv_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "fpgpu_div_dev": (_int, [_vp, _vp, _vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "fpgpu_static_scratch_bytes": (_sz, []),
    "fpgpu_batch_inv_static_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _vp]),
    "fpgpu_matvec_workspace_bytes": (_sz, [_vp, _sz, _sz]),
    "fpgpu_matvec_dev": (_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp, _sz, _vp]),
    "fpgpu_spmv_dev": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp, _vp]),
    "fpgpu_spmv": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "fpgpu_matmul_workspace_bytes": (_sz, [_vp, _sz, _sz, _sz]),
    "fpgpu_matmul_dev": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _vp, _sz, _vp]),
    "fpgpu_fill_random_dev": (_int, [_vp, _vp, _sz, _u64, _vp]),
    "fpgpu_pow_dev": (_int, [_vp, _vp, _vp, _u32, _vp, _sz, _vp]),
    "fpgpu_fold_workspace_bytes": (_sz, [_vp]),
    "fpgpu_sum_dev": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "fpgpu_product_dev": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "fpgpu_pow": (_int, [_vp, _vp, _vp, _u32, _vp, _sz]),
    "fpgpu_sum": (_int, [_vp, _vp, _sz, _vp]),
    "fpgpu_product": (_int, [_vp, _vp, _sz, _vp]),
    "fpgpu_mul": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "fpgpu_sqr": (_int, [_vp, _vp, _vp, _sz]),
    "fpgpu_add": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "fpgpu_sub": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "fpgpu_neg": (_int, [_vp, _vp, _vp, _sz]),
    "fpgpu_double": (_int, [_vp, _vp, _vp, _sz]),
    "fpgpu_div_by_2": (_int, [_vp, _vp, _vp, _sz]),
    "fpgpu_to_mont": (_int, [_vp, _vp, _vp, _sz]),
    "fpgpu_from_mont": (_int, [_vp, _vp, _vp, _sz]),
    "fpgpu_mul_chain": (_int, [_vp, _vp, _vp, _vp, _sz, _u32]),
    "fpgpu_batch_inv": (_int, [_vp, _vp, _vp, _vp, _sz]),
    "fpgpu_div": (_int, [_vp, _vp, _vp, _vp, _vp, _sz]),
    "fpgpu_matvec": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "fpgpu_matmul": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz]),
    "fpgpu_dot_dev": (_int, [_vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "fpgpu_dot": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "fpgpu_set_devices": (_int, [_c.POINTER(_int), _int]),
    "fpgpu_get_devices": (_int, [_c.POINTER(_int), _int]),
    "fpgpu_devset_create": (_int, [_c.POINTER(_int), _int, _c.POINTER(_vp)]),
    "fpgpu_devset_destroy": (None, [_vp]),
    "fpgpu_devset_devices": (_int, [_vp, _c.POINTER(_int), _int]),
    "fpgpu_vec_alloc_on": (_int, [_vp, _vp, _sz, _int, _c.POINTER(_vp)]),
    "fpgpu_vec_alloc_rows_on": (_int, [_vp, _vp, _sz, _sz, _int, _c.POINTER(_vp)]),
    "fpgpu_idx_alloc_rows_on": (_int, [_vp, _sz, _sz, _int, _c.POINTER(_vp)]),
    "fpgpu_vec_alloc": (_int, [_vp, _sz, _int, _c.POINTER(_vp)]),
    "fpgpu_vec_alloc_rows": (_int, [_vp, _sz, _sz, _int, _c.POINTER(_vp)]),
    "fpgpu_vec_free": (None, [_vp]),
    "fpgpu_vec_len": (_sz, [_vp]),
    "fpgpu_vec_parts": (_int, [_vp]),
    "fpgpu_vec_part": (_int, [_vp, _int, _c.POINTER(_int), _c.POINTER(_sz), _c.POINTER(_sz), _c.POINTER(_vp),
                              _c.POINTER(_vp)]),
    "fpgpu_vec_upload": (_int, [_vp, _vp, _sz, _sz]),
    "fpgpu_vec_download": (_int, [_vp, _vp, _sz, _sz]),
    "fpgpu_vec_sync": (_int, [_vp]),
    "fpgpu_vec_copy": (_int, [_vp, _vp]),
    "fpgpu_vec_fill_random": (_int, [_vp, _vp, _u64]),
    "fpgpu_vec_mul": (_int, [_vp, _vp, _vp, _vp]),
    "fpgpu_vec_sqr": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_add": (_int, [_vp, _vp, _vp, _vp]),
    "fpgpu_vec_sub": (_int, [_vp, _vp, _vp, _vp]),
    "fpgpu_vec_neg": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_double": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_div_by_2": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_to_mont": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_from_mont": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_mul_chain": (_int, [_vp, _vp, _vp, _vp, _u32]),
    "fpgpu_vec_pow": (_int, [_vp, _vp, _vp, _u32, _vp]),
    "fpgpu_vec_batch_inv": (_int, [_vp, _vp, _vp, _vp, _c.POINTER(_sz)]),
    "fpgpu_vec_div": (_int, [_vp, _vp, _vp, _vp, _vp, _c.POINTER(_sz)]),
    "fpgpu_vec_sum": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_product": (_int, [_vp, _vp, _vp]),
    "fpgpu_vec_dot": (_int, [_vp, _vp, _vp, _vp]),
    "fpgpu_vec_matvec": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "fpgpu_vec_matmul": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _sz]),
    "fpgpu_idx_alloc_rows": (_int, [_sz, _sz, _int, _c.POINTER(_vp)]),
    "fpgpu_idx_free": (None, [_vp]),
    "fpgpu_idx_upload": (_int, [_vp, _vp, _sz, _sz]),
    "fpgpu_vec_spmv": (_int, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "fpgpu_tuning_set": (_int, [_c.c_char_p, _c.c_long]),
    "fpgpu_last_error_string": (_c.c_char_p, []),
    "fpgpu_device_count": (_int, []),
    "fpgpu_version": (_c.c_char_p, []),
}

_lib = None


def load():
    """Load libfpgpu.so and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7.  If
        # torch is going to be used beside this library (device tensors, RCCL), it must be
        # loaded first so that our DT_NEEDED libamdhip64.so.7 binds to the copy torch
        # already mapped; two runtimes in one process leave the second with no devices.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                if os.environ.get("FPGPU_LIB"):   # an older tuning build beside the tree's library (A/B timing tools)
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


TUNING_DEFAULT = -0x7fffffff - 1


def tuning_set(name, value=TUNING_DEFAULT):
    """Set one tuning / test hook; TUNING_DEFAULT restores the library's choice."""
    check(load().fpgpu_tuning_set(name.encode(), value))


class FpgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"fpgpu error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != OK:
        raise FpgpuError(rc, load().fpgpu_last_error_string().decode())
