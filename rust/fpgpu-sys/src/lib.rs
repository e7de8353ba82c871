//! Raw bindings to `include/fpgpu.h`, one item per declaration, in the header's order.
//! Element layout everywhere: element `i` = `n64` little-endian `u64` words at index `i * n64`
//! (`Uint::as_words()`, src/semiring/crypto_bigint_uint.rs:65-68), Montgomery form, R = 2^(64 n64).
#![no_std]
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_int, c_long, c_void};

/// Opaque `fpgpu_ctx`: immutable after creation, shareable between threads.
#[repr(C)]
pub struct fpgpu_ctx {
    _private: [u8; 0],
}

/// Opaque `fpgpu_vec`: a device-resident vector (one device, sharded, or replicated over the device list).
#[repr(C)]
pub struct fpgpu_vec {
    _private: [u8; 0],
}

/// Opaque `fpgpu_idx`: device-resident 32-bit column indices of a `SparseMatrix`, laid out like its values.
#[repr(C)]
pub struct fpgpu_idx {
    _private: [u8; 0],
}

/// Opaque `fpgpu_devset`: a device list as a handle (the `*_on` allocators lay vectors out over it).
#[repr(C)]
pub struct fpgpu_devset {
    _private: [u8; 0],
}

pub const FPGPU_OK: c_int = 0;
/// Even or zero modulus: `FieldError::InvalidModulus` (src/field.rs:345-349).
pub const FPGPU_ERR_INVALID_MODULUS: c_int = 1;
pub const FPGPU_ERR_UNSUPPORTED: c_int = 2;
pub const FPGPU_ERR_SHAPE: c_int = 3;
pub const FPGPU_ERR_HIP: c_int = 4;
/// No gfx950 device visible: there is no CPU fallback.
pub const FPGPU_ERR_NO_DEVICE: c_int = 5;
/// librccl could not be loaded or a collective failed.
pub const FPGPU_ERR_RCCL: c_int = 6;
pub const FPGPU_PLACE_CURRENT: c_int = -1;
pub const FPGPU_PLACE_SHARDED: c_int = -2;
pub const FPGPU_PLACE_REPLICATED: c_int = -3;
pub const FPGPU_MAX_N64: u32 = 8;
pub const FPGPU_STATIC_SCRATCH_BYTES: usize = 64 << 20;
pub const FPGPU_TUNING_DEFAULT: c_long = -0x7fff_ffff - 1;

extern "C" {
    // ---- field configuration
    pub fn fpgpu_ctx_create(modulus_le: *const u64, n64: u32, out: *mut *mut fpgpu_ctx) -> c_int;
    pub fn fpgpu_ctx_destroy(ctx: *mut fpgpu_ctx);
    pub fn fpgpu_ctx_n64(ctx: *const fpgpu_ctx) -> u32;
    pub fn fpgpu_ctx_params(ctx: *const fpgpu_ctx, modulus_le: *mut u64, one_le: *mut u64, r2_le: *mut u64,
                            mod_neg_inv: *mut u64) -> c_int;

    // ---- elementwise batches, device buffers (asynchronous on `stream`, a hipStream_t)
    pub fn fpgpu_mul_dev(ctx: *const fpgpu_ctx, a: *const c_void, b: *const c_void, out: *mut c_void, n: usize,
                         stream: *mut c_void) -> c_int;
    pub fn fpgpu_sqr_dev(ctx: *const fpgpu_ctx, a: *const c_void, out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_add_dev(ctx: *const fpgpu_ctx, a: *const c_void, b: *const c_void, out: *mut c_void, n: usize,
                         stream: *mut c_void) -> c_int;
    pub fn fpgpu_sub_dev(ctx: *const fpgpu_ctx, a: *const c_void, b: *const c_void, out: *mut c_void, n: usize,
                         stream: *mut c_void) -> c_int;
    pub fn fpgpu_neg_dev(ctx: *const fpgpu_ctx, a: *const c_void, out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_double_dev(ctx: *const fpgpu_ctx, a: *const c_void, out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_div_by_2_dev(ctx: *const fpgpu_ctx, a: *const c_void, out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_to_mont_dev(ctx: *const fpgpu_ctx, a: *const c_void, out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_from_mont_dev(ctx: *const fpgpu_ctx, a: *const c_void, out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_mul_chain_dev(ctx: *const fpgpu_ctx, a: *const c_void, b: *const c_void, out: *mut c_void, n: usize,
                               iters: u32, stream: *mut c_void) -> c_int;

    // ---- pow, folds
    pub fn fpgpu_pow_dev(ctx: *const fpgpu_ctx, a: *const c_void, exponent_le: *const u64, exponent_words: u32,
                         out: *mut c_void, n: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_fold_workspace_bytes(ctx: *const fpgpu_ctx) -> usize;
    pub fn fpgpu_sum_dev(ctx: *const fpgpu_ctx, v: *const c_void, n: usize, out: *mut c_void, workspace: *mut c_void,
                         workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_product_dev(ctx: *const fpgpu_ctx, v: *const c_void, n: usize, out: *mut c_void, workspace: *mut c_void,
                             workspace_bytes: usize, stream: *mut c_void) -> c_int;

    pub fn fpgpu_dot_dev(ctx: *const fpgpu_ctx, a: *const c_void, b: *const c_void, n: usize, out: *mut c_void,
                         workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;

    // ---- batch inversion
    pub fn fpgpu_batch_inv_workspace_bytes(ctx: *const fpgpu_ctx, n: usize) -> usize;
    pub fn fpgpu_batch_inv_fast_workspace_bytes(ctx: *const fpgpu_ctx, n: usize) -> usize;
    pub fn fpgpu_batch_inv_dev(ctx: *const fpgpu_ctx, x: *const c_void, out: *mut c_void, zero_flags: *mut u8, n: usize,
                               workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_div_dev(ctx: *const fpgpu_ctx, a: *const c_void, b: *const c_void, out: *mut c_void, zero_flags: *mut u8, n: usize,
                         workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    /// No counterpart in the reference: batch inversion for callers that cannot allocate (HIP graph
    /// capture, serving loops); the workspace is a static device array of `FPGPU_STATIC_SCRATCH_BYTES`.
    pub fn fpgpu_static_scratch_bytes() -> usize;
    pub fn fpgpu_batch_inv_static_dev(ctx: *const fpgpu_ctx, x: *const c_void, out: *mut c_void, zero_flags: *mut u8,
                                      n: usize, stream: *mut c_void) -> c_int;

    // ---- matrix products
    pub fn fpgpu_matvec_workspace_bytes(ctx: *const fpgpu_ctx, rows: usize, cols: usize) -> usize;
    pub fn fpgpu_matvec_dev(ctx: *const fpgpu_ctx, a: *const c_void, rows: usize, cols: usize, x: *const c_void,
                            y: *mut c_void, workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;
    pub fn fpgpu_spmv_dev(ctx: *const fpgpu_ctx, vals: *const c_void, col_idx: *const u32, rows: usize, cols: usize,
                          density: usize, x: *const c_void, y: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn fpgpu_matmul_workspace_bytes(ctx: *const fpgpu_ctx, m: usize, k: usize, n: usize) -> usize;
    pub fn fpgpu_matmul_dev(ctx: *const fpgpu_ctx, a: *const c_void, b: *const c_void, c: *mut c_void, m: usize, k: usize,
                            n: usize, workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;

    // ---- synthetic inputs
    pub fn fpgpu_fill_random_dev(ctx: *const fpgpu_ctx, out: *mut c_void, n: usize, seed: u64, stream: *mut c_void) -> c_int;

    // ---- host-buffer calls (synchronous: H2D, kernel, D2H)
    pub fn fpgpu_mul(ctx: *const fpgpu_ctx, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_sqr(ctx: *const fpgpu_ctx, a: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_add(ctx: *const fpgpu_ctx, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_sub(ctx: *const fpgpu_ctx, a: *const u64, b: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_neg(ctx: *const fpgpu_ctx, a: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_double(ctx: *const fpgpu_ctx, a: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_div_by_2(ctx: *const fpgpu_ctx, a: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_to_mont(ctx: *const fpgpu_ctx, a: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_from_mont(ctx: *const fpgpu_ctx, a: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn fpgpu_mul_chain(ctx: *const fpgpu_ctx, a: *const u64, b: *const u64, out: *mut u64, n: usize, iters: u32) -> c_int;
    pub fn fpgpu_pow(ctx: *const fpgpu_ctx, a: *const u64, exponent_le: *const u64, exponent_words: u32, out: *mut u64,
                     n: usize) -> c_int;
    pub fn fpgpu_sum(ctx: *const fpgpu_ctx, v: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn fpgpu_product(ctx: *const fpgpu_ctx, v: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn fpgpu_dot(ctx: *const fpgpu_ctx, a: *const u64, b: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn fpgpu_batch_inv(ctx: *const fpgpu_ctx, x: *const u64, out: *mut u64, zero_flags: *mut u8, n: usize) -> c_int;
    pub fn fpgpu_div(ctx: *const fpgpu_ctx, a: *const u64, b: *const u64, out: *mut u64, zero_flags: *mut u8, n: usize) -> c_int;
    pub fn fpgpu_matvec(ctx: *const fpgpu_ctx, a: *const u64, rows: usize, cols: usize, x: *const u64, y: *mut u64) -> c_int;
    pub fn fpgpu_spmv(ctx: *const fpgpu_ctx, vals: *const u64, col_idx: *const u32, rows: usize, cols: usize,
                      density: usize, x: *const u64, y: *mut u64) -> c_int;
    pub fn fpgpu_matmul(ctx: *const fpgpu_ctx, a: *const u64, b: *const u64, c: *mut u64, m: usize, k: usize, n: usize) -> c_int;

    // ---- device-resident vectors and multi-GPU sharding (one host thread drives every device)
    pub fn fpgpu_set_devices(ids: *const c_int, count: c_int) -> c_int;
    pub fn fpgpu_get_devices(ids: *mut c_int, capacity: c_int) -> c_int;
    pub fn fpgpu_devset_create(ids: *const c_int, count: c_int, out: *mut *mut fpgpu_devset) -> c_int;
    pub fn fpgpu_devset_destroy(set: *mut fpgpu_devset);
    pub fn fpgpu_devset_devices(set: *const fpgpu_devset, ids: *mut c_int, capacity: c_int) -> c_int;
    pub fn fpgpu_vec_alloc(ctx: *const fpgpu_ctx, n: usize, placement: c_int, out: *mut *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_alloc_rows(ctx: *const fpgpu_ctx, rows: usize, cols: usize, placement: c_int,
                                out: *mut *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_alloc_on(ctx: *const fpgpu_ctx, set: *const fpgpu_devset, n: usize, placement: c_int,
                              out: *mut *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_alloc_rows_on(ctx: *const fpgpu_ctx, set: *const fpgpu_devset, rows: usize, cols: usize,
                                   placement: c_int, out: *mut *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_free(v: *mut fpgpu_vec);
    pub fn fpgpu_vec_len(v: *const fpgpu_vec) -> usize;
    pub fn fpgpu_vec_parts(v: *const fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_part(v: *const fpgpu_vec, i: c_int, device: *mut c_int, first: *mut usize, count: *mut usize,
                          dev_ptr: *mut *mut c_void, stream: *mut *mut c_void) -> c_int;
    pub fn fpgpu_vec_upload(v: *mut fpgpu_vec, host: *const u64, first: usize, n: usize) -> c_int;
    pub fn fpgpu_vec_download(v: *const fpgpu_vec, host: *mut u64, first: usize, n: usize) -> c_int;
    pub fn fpgpu_vec_sync(v: *const fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_copy(dst: *mut fpgpu_vec, src: *const fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_fill_random(ctx: *const fpgpu_ctx, v: *mut fpgpu_vec, seed: u64) -> c_int;
    pub fn fpgpu_vec_mul(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, b: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_sqr(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_add(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, b: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_sub(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, b: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_neg(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_double(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_div_by_2(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_to_mont(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_from_mont(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_mul_chain(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, b: *const fpgpu_vec, out: *mut fpgpu_vec,
                               iters: u32) -> c_int;
    pub fn fpgpu_vec_pow(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, exponent_le: *const u64, exponent_words: u32,
                         out: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_batch_inv(ctx: *const fpgpu_ctx, x: *const fpgpu_vec, out: *mut fpgpu_vec, none_flags_host: *mut u8,
                               none_count: *mut usize) -> c_int;
    pub fn fpgpu_vec_div(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, b: *const fpgpu_vec, out: *mut fpgpu_vec,
                         none_flags_host: *mut u8, none_count: *mut usize) -> c_int;
    pub fn fpgpu_vec_sum(ctx: *const fpgpu_ctx, v: *const fpgpu_vec, out_host: *mut u64) -> c_int;
    pub fn fpgpu_vec_product(ctx: *const fpgpu_ctx, v: *const fpgpu_vec, out_host: *mut u64) -> c_int;
    pub fn fpgpu_vec_dot(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, b: *const fpgpu_vec, out_host: *mut u64) -> c_int;
    pub fn fpgpu_vec_matvec(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, rows: usize, cols: usize, x: *const fpgpu_vec,
                            y: *mut fpgpu_vec) -> c_int;
    pub fn fpgpu_vec_matmul(ctx: *const fpgpu_ctx, a: *const fpgpu_vec, b: *const fpgpu_vec, c: *mut fpgpu_vec, m: usize,
                            k: usize, n: usize) -> c_int;

    pub fn fpgpu_idx_alloc_rows(rows: usize, density: usize, placement: c_int, out: *mut *mut fpgpu_idx) -> c_int;
    pub fn fpgpu_idx_alloc_rows_on(set: *const fpgpu_devset, rows: usize, density: usize, placement: c_int,
                                   out: *mut *mut fpgpu_idx) -> c_int;
    pub fn fpgpu_idx_free(idx: *mut fpgpu_idx);
    pub fn fpgpu_idx_upload(idx: *mut fpgpu_idx, host: *const u32, first: usize, n: usize) -> c_int;
    pub fn fpgpu_vec_spmv(ctx: *const fpgpu_ctx, vals: *const fpgpu_vec, col_idx: *const fpgpu_idx, rows: usize, cols: usize,
                          density: usize, x: *const fpgpu_vec, y: *mut fpgpu_vec) -> c_int;

    // ---- tuning / test hooks (process-wide; FPGPU_TUNING_DEFAULT restores the default)
    pub fn fpgpu_tuning_set(name: *const c_char, value: c_long) -> c_int;

    // ---- diagnostics
    pub fn fpgpu_last_error_string() -> *const c_char;
    pub fn fpgpu_device_count() -> c_int;
    pub fn fpgpu_version() -> *const c_char;
}
