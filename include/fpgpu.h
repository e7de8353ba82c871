/*
 * fpgpu.h -- C ABI of libfpgpu: the MI355X (gfx950) backend for crypto-primitives'
 * prime-field Montgomery hot path.
 *
 * The reference (Rust) has no slice-level entry point: its Field / FieldConfig
 * methods are scalar (src/semiring.rs:138-242, src/ring.rs:45-56,
 * src/field.rs:146-187).  Each function below is the batched form of one of
 * those scalar methods and is what a Rust `fpgpu-sys` crate would bind
 * (INTEGRATION.md shows the extern block and the adapter).  Plain pointers and
 * sizes only.
 *
 * Element layout (both host and device buffers): element i occupies n64
 * consecutive little-endian uint64_t words at index i*n64 -- exactly
 * `Uint::as_words()` (src/semiring/crypto_bigint_uint.rs:65-68) laid out as a
 * `Vec<T>` / `DenseRowMatrix<T>::data` (src/matrix.rs:135-145).  No transposition,
 * no copy-in format: a `&[MontyFieldElement<LIMBS>]` can be passed as is.
 * All field elements are Montgomery representatives with R = 2^(64*n64), as in
 * the reference.  Results are canonical ([0, p)) and bit-identical to the
 * reference's for inputs that are field elements (< p).  Inputs outside the
 * field: mul (a < R, b < p), sqr, to_mont (any n64-word value) and from_mont
 * follow the reference's integer arithmetic exactly for every input -- the
 * reference itself projects wide values this way (crypto_bigint_monty.rs:319-340).
 * Everything else -- add, sub, neg, the chains, folds, dot, batch inversion, the
 * matrix products (lazily reduced accumulators) -- is defined for canonical
 * inputs only, as crypto-bigint's add_mod / sub_mod / neg_mod are.
 * On a little-endian host this layout is also the elements' wire format (the
 * reference's serde impls hand each element's Uint to crypto-bigint's serialiser,
 * crypto_bigint_monty.rs:391-415, i.e. its little-endian bytes).  PARITY UNPINNED: the
 * reference holds no byte fixture of a serialised element, and crypto-bigint is not in
 * this tree, so tests/test_wire_format.py checks the statement against a restatement only.
 *
 * Device pointers must be 16-byte aligned (8-byte for n64 odd).  Every *_dev
 * call is asynchronous on `stream` (a hipStream_t, NULL = default stream) of the
 * current HIP device and allocates nothing.  The host-buffer calls are
 * synchronous and run on the current device; their host <-> device copies go
 * through pinned staging buffers of the library, filled and drained by a few
 * host threads (pageable caller memory cannot be pinned in place).  The
 * elementwise ones (fpgpu_mul ... fpgpu_pow), fpgpu_batch_inv and fpgpu_div
 * stream their operands through the device in chunks -- copy in, kernels and
 * copy out overlap, `out` may be `a` or `b`, and the arrays may exceed device
 * memory (a batch inversion is cut into independent ranges: an inverse does not
 * depend on the cut); the others hold whole operands in device memory they
 * allocate and free themselves.  Staging is organised in LANES (two 6 MiB pinned
 * buffers and, on each device a lane has worked with, two 6 MiB device buffers): a
 * transfer takes FPGPU_COPY_THREADS (default 4) free lanes per device it spans out of a
 * process-wide table of 16, so the library keeps 48 MiB of pinned host memory and
 * 48 MiB per device in single-threaded use, and at most 192 MiB of each when sixteen
 * lanes have been in use at once; all of it is held until the process ends.
 *
 * Threads.  The reference's configs are Send + Sync and shared across rayon workers
 * (src/lib.rs:62-65); here a context is immutable and every entry point may be called
 * from any number of host threads at once.  There is no library-wide lock on a call
 * path: host-buffer calls of different threads each take their own staging lanes and
 * run side by side (a call waits only while all sixteen lanes are taken); handle
 * operations enqueue on their list entries' streams without a lock, except that one
 * call at a time enqueues the kernels that use an entry's scratch block (inversion,
 * folds, the matrix products) and one all-gather / fpgpu_vec_copy protocol runs at a
 * time.  What remains process-wide: the DEFAULT device list (fpgpu_set_devices; pass an
 * fpgpu_devset to the *_on allocators to be independent of it), the tuning hooks, and
 * the static array of fpgpu_batch_inv_static_dev.  Two threads that operate on the SAME
 * vector are ordered only by the stream of the entry it lives on, in the order their
 * calls happened to enqueue.
 */
#ifndef FPGPU_H
#define FPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fpgpu_ctx fpgpu_ctx;

enum {
    FPGPU_OK = 0,
    FPGPU_ERR_INVALID_MODULUS = 1, /* even or zero modulus: FieldError::InvalidModulus (src/field.rs:345-349) */
    FPGPU_ERR_UNSUPPORTED = 2,     /* limb count without a compiled kernel */
    FPGPU_ERR_SHAPE = 3,           /* null pointer / size / alignment mismatch */
    FPGPU_ERR_HIP = 4,             /* a HIP runtime call failed; see fpgpu_last_error_string() */
    FPGPU_ERR_NO_DEVICE = 5,       /* no gfx950 device visible: there is no CPU fallback */
    FPGPU_ERR_RCCL = 6             /* librccl could not be loaded or a collective failed */
};

#define FPGPU_MAX_N64 8 /* largest compiled element width: 512 bits */

/* ---- field configuration -------------------------------------------------
 * Replaces BaseFieldConfig::new (src/field.rs:197-209; MontyField impl
 * src/field/crypto_bigint_monty.rs:346-363, BoxedMontyField impl
 * src/field/crypto_bigint_boxed_monty.rs:406-412) together with
 * crypto-bigint's MontyParams::new: computes -m^-1 mod 2^64
 * (src/helpers/crypto_bigint.rs:56-67), R mod p and R^2 mod p.
 * `modulus_le` holds n64 little-endian words.  An even or zero modulus returns
 * FPGPU_ERR_INVALID_MODULUS.  The context is immutable and may be shared
 * between threads (the reference's configs are Send + Sync, src/lib.rs:62-65). */
int fpgpu_ctx_create(const uint64_t *modulus_le, uint32_t n64, fpgpu_ctx **out);
void fpgpu_ctx_destroy(fpgpu_ctx *ctx);
uint32_t fpgpu_ctx_n64(const fpgpu_ctx *ctx);
/* modulus(), params.one(), params.r2(), params.mod_neg_inv()
 * (crypto_bigint_monty.rs:144,169,234,355-357).  Any output pointer may be NULL. */
int fpgpu_ctx_params(const fpgpu_ctx *ctx, uint64_t *modulus_le, uint64_t *one_le,
                     uint64_t *r2_le, uint64_t *mod_neg_inv);

/* ---- elementwise batches, device buffers ----------------------------------
 * out[i] = op(a[i], b[i]) for i in [0, n).  `out` may alias `a` or `b`. */

/* SemiringConfig::mul -> helpers::mul::monty_mul (crypto_bigint_monty.rs:159-161,
 * src/helpers/crypto_bigint.rs:74-159); ConstMontyField MulAssign
 * (crypto_bigint_const_monty.rs:337-347); BoxedMontyField::mul
 * (crypto_bigint_boxed_monty.rs:182-196). */
int fpgpu_mul_dev(const fpgpu_ctx *ctx, const void *a, const void *b, void *out, size_t n, void *stream);
/* Square (crypto_bigint_const_monty.rs:736-740): a*a*R^-1 mod p. */
int fpgpu_sqr_dev(const fpgpu_ctx *ctx, const void *a, void *out, size_t n, void *stream);
/* SemiringConfig::add -> add_mod (crypto_bigint_monty.rs:147-151, const_monty.rs:317-325). */
int fpgpu_add_dev(const fpgpu_ctx *ctx, const void *a, const void *b, void *out, size_t n, void *stream);
/* SemiringConfig::sub -> sub_mod (crypto_bigint_monty.rs:153-157, const_monty.rs:327-335). */
int fpgpu_sub_dev(const fpgpu_ctx *ctx, const void *a, const void *b, void *out, size_t n, void *stream);
/* RingConfig::neg -> neg_mod, 0 -> 0 (crypto_bigint_monty.rs:198-202, const_monty.rs:191-202). */
int fpgpu_neg_dev(const fpgpu_ctx *ctx, const void *a, void *out, size_t n, void *stream);
/* project(&Uint): integer -> Montgomery form = monty_mul(x, R^2); x may be any n64-word
 * value, also >= p (crypto_bigint_monty.rs:319-340, const_monty.rs:497-506). */
int fpgpu_to_mont_dev(const fpgpu_ctx *ctx, const void *a, void *out, size_t n, void *stream);
/* lift: Montgomery form -> canonical integer = monty_retrieve_inner
 * (crypto_bigint_monty.rs:369-385, src/helpers/crypto_bigint.rs:13-37). */
int fpgpu_from_mont_dev(const fpgpu_ctx *ctx, const void *a, void *out, size_t n, void *stream);
/* ConstMontyField::double (crypto_bigint_const_monty.rs:86-89): a + a mod p. */
int fpgpu_double_dev(const fpgpu_ctx *ctx, const void *a, void *out, size_t n, void *stream);
/* ConstMontyField::div_by_2 (crypto_bigint_const_monty.rs:81-84): the x with x + x = a, i.e. a / 2 for even
 * a and (a + p) / 2 for odd a (on the Montgomery representative, which halves the value it stands for). */
int fpgpu_div_by_2_dev(const fpgpu_ctx *ctx, const void *a, void *out, size_t n, void *stream);
/* "Mul assign" bench shape (benches/field_ops_bench_common.rs:27-39), batched:
 * acc[i] = a[i]; repeat `iters` times acc[i] = mul(acc[i], b[i]); out[i] = acc[i]. */
int fpgpu_mul_chain_dev(const fpgpu_ctx *ctx, const void *a, const void *b, void *out, size_t n,
                        uint32_t iters, void *stream);

/* ---- rows SURVEY.md section 8f marks "next" ------------------------------------
 * Batched SemiringConfig::pow_u32 / FieldConfig::pow with ONE exponent for the whole
 * batch (crypto_bigint_monty.rs:163-172,239-249; the 4-bit-window almost-Montgomery
 * ladder of src/helpers/crypto_bigint.rs:220-455): out[i] = a[i]^e in Montgomery form,
 * fully reduced; e == 0 gives one (also for a[i] == 0).  `exponent_le` is a HOST array
 * of `exponent_words` (<= 8) little-endian 64-bit words. */
int fpgpu_pow_dev(const fpgpu_ctx *ctx, const void *a, const uint64_t *exponent_le, uint32_t exponent_words,
                  void *out, size_t n, void *stream);
/* SemiringConfig::sum_refs / product_refs (src/semiring.rs:213-241): folds of add / mul
 * over v[0..n); the empty fold is zero / one.  `out` is ONE element (device).
 * `workspace`: fpgpu_fold_workspace_bytes(ctx) bytes of device scratch (needed when n > 256). */
size_t fpgpu_fold_workspace_bytes(const fpgpu_ctx *ctx);
int fpgpu_sum_dev(const fpgpu_ctx *ctx, const void *v, size_t n, void *out, void *workspace,
                  size_t workspace_bytes, void *stream);
int fpgpu_product_dev(const fpgpu_ctx *ctx, const void *v, size_t n, void *out, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---- batch inversion -------------------------------------------------------
 * out[i] = FieldConfig::inv(x[i]) (crypto_bigint_monty.rs:223-237,
 * boxed_monty.rs:273-284, const_monty.rs:277-283) for every i, computed with
 * Montgomery's trick.  Where the reference returns None (x[i] == 0, or not
 * invertible for a composite modulus) out[i] = 0 and zero_flags[i] = 1;
 * elsewhere zero_flags[i] = 0.  zero_flags (n bytes, device) may be NULL.
 * `workspace` is device scratch of fpgpu_batch_inv_workspace_bytes(ctx, n) bytes.  Both sizes below never shrink
 * as n grows: a workspace sized for the largest batch serves every shorter one. */
size_t fpgpu_batch_inv_workspace_bytes(const fpgpu_ctx *ctx, size_t n);
/* A workspace of at least this size (instead of the minimum above; about 4 * (n64 * 2 + 2 .. 3) more bytes
 * per element) lets the level-0 prefix products stay in the multiplier's own 29-bit-limb form in the
 * workspace instead of going through the output array as words: same results, a few per cent faster
 * (2^24 BLS12-381 elements: 1.07 -> 1.04 ms).  Equal to the minimum below 384 bits (n64 <= 5), where the limb form does not pay.  fpgpu_batch_inv_dev and
 * fpgpu_div_dev pick the path by the `workspace_bytes` they are given. */
size_t fpgpu_batch_inv_fast_workspace_bytes(const fpgpu_ctx *ctx, size_t n);
int fpgpu_batch_inv_dev(const fpgpu_ctx *ctx, const void *x, void *out, uint8_t *zero_flags, size_t n,
                        void *workspace, size_t workspace_bytes, void *stream);
/* FieldConfig::checked_div / div (src/field.rs:155-170): out[i] = mul(a[i], inv(b[i])); where b[i] has no
 * inverse the reference returns None (div panics "Division by zero"): zero_flags[i] = 1 and out[i] = 0.
 * One batch inversion and one multiplication on `stream`; out must alias neither a nor b; the workspace is
 * the batch inversion's (fpgpu_batch_inv_workspace_bytes). */
int fpgpu_div_dev(const fpgpu_ctx *ctx, const void *a, const void *b, void *out, uint8_t *zero_flags, size_t n,
                  void *workspace, size_t workspace_bytes, void *stream);

/* Batch inversion that allocates nothing and takes no workspace.  NO COUNTERPART IN THE
 * REFERENCE: it exists for callers that cannot allocate device memory or carry a workspace --
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same
 * results and flags as fpgpu_batch_inv_dev; the workspace is a static device array of
 * FPGPU_STATIC_SCRATCH_BYTES inside the library (one copy per GPU, held until the process ends).
 * A batch whose fpgpu_batch_inv_workspace_bytes(ctx, n) exceeds that returns FPGPU_ERR_SHAPE
 * (64 MiB covers 2^24 384-bit elements).  Every call on a GPU uses that GPU's one array: calls that
 * overlap on the same GPU (different streams or threads) race; calls in stream order do not.
 * The array's device address is looked up on the first call on each device (not an allocation;
 * make that first call outside graph capture). */
#define FPGPU_STATIC_SCRATCH_BYTES (64u << 20)
size_t fpgpu_static_scratch_bytes(void);
int fpgpu_batch_inv_static_dev(const fpgpu_ctx *ctx, const void *x, void *out, uint8_t *zero_flags, size_t n,
                               void *stream);

/* ---- dense matrix products over DenseRowMatrix<T> (src/matrix.rs:135-164) --
 * y[i] = sum_j mul(A[i*cols + j], x[j]): the SemiringConfig::sum_refs fold
 * (src/semiring.rs:213-226) of monty_mul over DenseRowMatrix::as_rows().
 * A is rows*cols elements row-major; x has cols, y has rows elements.
 * `workspace` is optional device scratch (256-byte aligned) of at least
 * fpgpu_matvec_workspace_bytes(ctx, rows, cols) bytes; with it the products run on the matrix
 * cores (A streamed once, as it lies) and the call is HBM-bound; without it (NULL), or where
 * that size is 0, a VALU kernel runs.  Results are identical.  The launcher picks the kernel by
 * shape: tall, narrow matrices go one row per lane (no workspace needed); a few very long rows
 * (more than 2^23 bytes each: outside the matrix-core plan) are split over columns through the
 * workspace -- such shapes NEED it to fill the GPU (one row of 2^27 elements: 2.8 TB/s with it,
 * one wave without). */
size_t fpgpu_matvec_workspace_bytes(const fpgpu_ctx *ctx, size_t rows, size_t cols);
int fpgpu_matvec_dev(const fpgpu_ctx *ctx, const void *A, size_t rows, size_t cols, const void *x,
                     void *y, void *workspace, size_t workspace_bytes, void *stream);
/* Sparse mat-vec over SparseMatrix<T> (src/matrix.rs:46-61: a fixed number `density` of
 * (column index, value) cells per row, rows back to back):
 * y[i] = sum_{j<density} mul(vals[i*density + j], x[col_idx[i*density + j]]).  The reference's
 * cells are Rust tuples (usize, T) with no defined layout, so the binding passes the two halves
 * as separate arrays; indices are 32-bit and must be < cols (the host variant checks them and
 * returns FPGPU_ERR_SHAPE where the reference would panic; the device variant skips such cells). */
int fpgpu_spmv_dev(const fpgpu_ctx *ctx, const void *vals, const uint32_t *col_idx, size_t rows, size_t cols,
                   size_t density, const void *x, void *y, void *stream);
/* C[i*n + j] = sum_l mul(A[i*k + l], B[l*n + j]); A is m x k, B is k x n, C is m x n.
 * `workspace` is optional device scratch (256-byte aligned) of at least
 * fpgpu_matmul_workspace_bytes(ctx, m, k, n) bytes; with it, single-limb fields (n64 == 1)
 * run on the matrix cores (int8 digit planes of A and B^T live there; any k: beyond 16320 terms
 * in parts, 1024 x 32768 x 1024 in 1.8 ms against 10.0), and so do
 * multi-limb fields (n64 = 2 ... 8, any k) from about 256 x 1024 x 256 on, as n64 x n64
 * single-limb products (digit planes of every limb plus up to 256 x (2 n64 - 1) x 96 KiB of scratch:
 * 232 MiB for 1024^3 at 256 bits, 3.4 against 0.87 ms; products of few output tiles split k over
 * workgroups through it as well; below that size the function reports what the VALU paths need).  Without it (NULL), or where that size is 0, the VALU kernels run.  Results are identical.  Two shapes
 * use the workspace at every width: a small output over a long k (at most 128 tiles of 16 x 16,
 * k >= 4096) splits k over workgroups through it -- without it such a product runs on a few
 * workgroups (16 x 2^20 x 16: 335 ms against 1.0) -- and a product with one column (n == 1) is
 * the mat-vec y = A x and takes the mat-vec's workspace (the size reported is that one).  One to
 * four vectors times a matrix (m <= 4, n >= 1024: linear combinations of the k rows of B) run a lane
 * per output column and stream B once per vector; the workspace serves them when n < 2^18. */
size_t fpgpu_matmul_workspace_bytes(const fpgpu_ctx *ctx, size_t m, size_t k, size_t n);
int fpgpu_matmul_dev(const fpgpu_ctx *ctx, const void *A, const void *B, void *C, size_t m, size_t k,
                     size_t n, void *workspace, size_t workspace_bytes, void *stream);

/* Dot product sum_i mul(a[i], b[i]): SemiringConfig::sum_refs (src/semiring.rs:213-226) over
 * helpers::mul::monty_mul -- one row of the mat-vec, spread over the whole GPU.
 * `out` is ONE element (device); n == 0 gives zero.  `workspace`: fpgpu_fold_workspace_bytes(ctx) bytes
 * of device scratch (needed when n > 256). */
int fpgpu_dot_dev(const fpgpu_ctx *ctx, const void *a, const void *b, size_t n, void *out, void *workspace,
                  size_t workspace_bytes, void *stream);

/* ---- synthetic inputs -------------------------------
 * Fills `out` with n near-uniform elements of [0, p): word l of element i is
 * splitmix64(seed ^ (i*n64 + l)), the value masked to bitlen(p) bits, minus p once if >= p
 * (so residues below 2^bitlen - p are twice as likely: synthetic bench input, not a sampler).
 * Identical to the host generator in bench.py / tests. */
int fpgpu_fill_random_dev(const fpgpu_ctx *ctx, void *out, size_t n, uint64_t seed, void *stream);

/* ---- host-buffer convenience calls (synchronous; H2D, kernel, D2H) -------- */
int fpgpu_mul(const fpgpu_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int fpgpu_sqr(const fpgpu_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int fpgpu_add(const fpgpu_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int fpgpu_sub(const fpgpu_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n);
int fpgpu_neg(const fpgpu_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int fpgpu_to_mont(const fpgpu_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int fpgpu_from_mont(const fpgpu_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int fpgpu_double(const fpgpu_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int fpgpu_div_by_2(const fpgpu_ctx *ctx, const uint64_t *a, uint64_t *out, size_t n);
int fpgpu_mul_chain(const fpgpu_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n,
                    uint32_t iters);
int fpgpu_pow(const fpgpu_ctx *ctx, const uint64_t *a, const uint64_t *exponent_le, uint32_t exponent_words,
              uint64_t *out, size_t n);
int fpgpu_sum(const fpgpu_ctx *ctx, const uint64_t *v, size_t n, uint64_t *out);
int fpgpu_product(const fpgpu_ctx *ctx, const uint64_t *v, size_t n, uint64_t *out);
int fpgpu_dot(const fpgpu_ctx *ctx, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out);
int fpgpu_batch_inv(const fpgpu_ctx *ctx, const uint64_t *x, uint64_t *out, uint8_t *zero_flags, size_t n);
int fpgpu_div(const fpgpu_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, uint8_t *zero_flags, size_t n);
int fpgpu_matvec(const fpgpu_ctx *ctx, const uint64_t *A, size_t rows, size_t cols, const uint64_t *x,
                 uint64_t *y);
int fpgpu_spmv(const fpgpu_ctx *ctx, const uint64_t *vals, const uint32_t *col_idx, size_t rows, size_t cols,
               size_t density, const uint64_t *x, uint64_t *y);
int fpgpu_matmul(const fpgpu_ctx *ctx, const uint64_t *A, const uint64_t *B, uint64_t *C, size_t m,
                 size_t k, size_t n);

/* ---- device-resident vectors and multi-GPU sharding -------------------------------------
 * The reference's containers are host memory -- Vec<T>, DenseRowMatrix<T>::data
 * (src/matrix.rs:135-145), and one heap allocation per element for BoxedMontyField
 * (src/field/crypto_bigint_boxed_monty.rs:62-64) -- so a host-buffer call pays PCIe for every
 * operand.  An fpgpu_vec keeps a vector (or a matrix's row-major data) in HBM between calls: a host
 * without HIP bindings (the Rust adapter) allocates, uploads, chains any number of operations
 * with no host copy, and downloads the result.
 *
 * Placement: a device index >= 0, or
 *   FPGPU_PLACE_CURRENT     the calling thread's current HIP device,
 *   FPGPU_PLACE_SHARDED     contiguous, balanced element ranges over the device list (whole rows
 *                           per device with fpgpu_vec_alloc_rows: the reference's own parallel axis,
 *                           rayon over par_chunks_exact(num_cols), src/matrix.rs:170-184),
 *   FPGPU_PLACE_REPLICATED  one full copy on every device of the list (mat-vec's x, mat-mul's B,
 *                           gathered results).
 * The device list is set with fpgpu_set_devices (ids may repeat: every entry gets its own stream
 * and range, which lets one GPU rehearse a multi-GPU layout); until it is set, "the list" is the
 * current device alone.  One host thread drives all devices: every list entry owns a stream and all
 * handle operations are asynchronous on those streams and ordered per device.  Host memory is only
 * touched by upload / download and by the calls that return a value (sum, product, dot, flags).
 *
 * Operands of one call must be laid out alike: same partition as the output, or replicated (a
 * replica on the output's device serves any range).  Workspaces are the library's (a grow-only
 * block per list entry).  fpgpu_vec_part exposes each part's device pointer and stream so that
 * the *_dev entry points and other HIP code can work on the same memory in stream order. */
typedef struct fpgpu_vec fpgpu_vec;
#define FPGPU_PLACE_CURRENT (-1)
#define FPGPU_PLACE_SHARDED (-2)
#define FPGPU_PLACE_REPLICATED (-3)

/* count == 0 (ids may be NULL) returns to "current device alone".  Existing vectors stay valid
 * (they keep the list they were laid out over, its streams and its communicators). */
int fpgpu_set_devices(const int *ids, int count);
int fpgpu_get_devices(int *ids, int capacity); /* returns the list's length (0 = not set) */

/* A device list as a handle, for callers that must not depend on process-wide state (a library inside a
 * larger program, several pipelines with different device sets in one process): the *_on allocators lay a
 * vector out over `set` instead of the default list (set == NULL: the default list, exactly
 * fpgpu_vec_alloc / fpgpu_vec_alloc_rows / fpgpu_idx_alloc_rows).  Everything else follows from the vector:
 * operations, all-gathers and their RCCL communicators belong to the list the OUTPUT was laid out over.
 * Two lists that name the same device share that device's stream(s): their work stays ordered.
 * No counterpart in the reference (its parallel axis, rayon, has one global pool: src/matrix.rs:170-184).
 * Destroying the handle leaves vectors laid out over it valid. */
typedef struct fpgpu_devset fpgpu_devset;
int fpgpu_devset_create(const int *ids, int count, fpgpu_devset **out);
void fpgpu_devset_destroy(fpgpu_devset *set);
int fpgpu_devset_devices(const fpgpu_devset *set, int *ids, int capacity); /* returns the list's length */

int fpgpu_vec_alloc(const fpgpu_ctx *ctx, size_t n, int placement, fpgpu_vec **out);
/* rows * cols elements; a sharded placement gives every device whole rows */
int fpgpu_vec_alloc_rows(const fpgpu_ctx *ctx, size_t rows, size_t cols, int placement, fpgpu_vec **out);
int fpgpu_vec_alloc_on(const fpgpu_ctx *ctx, const fpgpu_devset *set, size_t n, int placement, fpgpu_vec **out);
int fpgpu_vec_alloc_rows_on(const fpgpu_ctx *ctx, const fpgpu_devset *set, size_t rows, size_t cols, int placement,
                            fpgpu_vec **out);
void fpgpu_vec_free(fpgpu_vec *v);
size_t fpgpu_vec_len(const fpgpu_vec *v);
int fpgpu_vec_parts(const fpgpu_vec *v);
/* part i: its device, element range [first, first + count), device pointer and stream (any output may be NULL) */
int fpgpu_vec_part(const fpgpu_vec *v, int i, int *device, size_t *first, size_t *count, void **dev_ptr,
                   void **stream);
/* elements [first, first + n) <- host / -> host; both return when the host buffer is the caller's again
 * (an upload: when the data is in device memory).  The parts of a sharded or replicated vector are
 * transferred together, every device over its own link.
 * `host` is n * n64 words in the reference's layout (a &[MontyFieldElement] as it lies). */
int fpgpu_vec_upload(fpgpu_vec *v, const uint64_t *host, size_t first, size_t n);
int fpgpu_vec_download(const fpgpu_vec *v, uint64_t *host, size_t first, size_t n);
int fpgpu_vec_sync(const fpgpu_vec *v); /* wait for everything queued on the vector's devices */
/* dst <- src between any two layouts of equal length: scatter, gather (sharded -> replicated is an
 * all-gather by direct peer copies), re-shard, replicate; device to device */
int fpgpu_vec_copy(fpgpu_vec *dst, const fpgpu_vec *src);
/* the synthetic generator of fpgpu_fill_random_dev; element i of the VECTOR gets sequence item i whatever the layout */
int fpgpu_vec_fill_random(const fpgpu_ctx *ctx, fpgpu_vec *v, uint64_t seed);

/* handle forms of the batch operations above (same reference methods, same results) */
int fpgpu_vec_mul(const fpgpu_ctx *ctx, const fpgpu_vec *a, const fpgpu_vec *b, fpgpu_vec *out);
int fpgpu_vec_sqr(const fpgpu_ctx *ctx, const fpgpu_vec *a, fpgpu_vec *out);
int fpgpu_vec_add(const fpgpu_ctx *ctx, const fpgpu_vec *a, const fpgpu_vec *b, fpgpu_vec *out);
int fpgpu_vec_sub(const fpgpu_ctx *ctx, const fpgpu_vec *a, const fpgpu_vec *b, fpgpu_vec *out);
int fpgpu_vec_neg(const fpgpu_ctx *ctx, const fpgpu_vec *a, fpgpu_vec *out);
int fpgpu_vec_to_mont(const fpgpu_ctx *ctx, const fpgpu_vec *a, fpgpu_vec *out);
int fpgpu_vec_from_mont(const fpgpu_ctx *ctx, const fpgpu_vec *a, fpgpu_vec *out);
int fpgpu_vec_double(const fpgpu_ctx *ctx, const fpgpu_vec *a, fpgpu_vec *out);
int fpgpu_vec_div_by_2(const fpgpu_ctx *ctx, const fpgpu_vec *a, fpgpu_vec *out);
int fpgpu_vec_mul_chain(const fpgpu_ctx *ctx, const fpgpu_vec *a, const fpgpu_vec *b, fpgpu_vec *out, uint32_t iters);
int fpgpu_vec_pow(const fpgpu_ctx *ctx, const fpgpu_vec *a, const uint64_t *exponent_le, uint32_t exponent_words,
                  fpgpu_vec *out);
/* every device inverts its own range with Montgomery's trick: no collective.  none_flags_host (n
 * bytes, HOST) and none_count may be NULL; asking for either makes the call synchronous. */
int fpgpu_vec_batch_inv(const fpgpu_ctx *ctx, const fpgpu_vec *x, fpgpu_vec *out, uint8_t *none_flags_host,
                        size_t *none_count);
/* a / b elementwise (checked_div): flags and count as fpgpu_vec_batch_inv; out must alias neither operand */
int fpgpu_vec_div(const fpgpu_ctx *ctx, const fpgpu_vec *a, const fpgpu_vec *b, fpgpu_vec *out, uint8_t *none_flags_host,
                  size_t *none_count);
/* one element to HOST memory (synchronous); per-device partials are folded on the first device */
int fpgpu_vec_sum(const fpgpu_ctx *ctx, const fpgpu_vec *v, uint64_t *out_host);
int fpgpu_vec_product(const fpgpu_ctx *ctx, const fpgpu_vec *v, uint64_t *out_host);
int fpgpu_vec_dot(const fpgpu_ctx *ctx, const fpgpu_vec *a, const fpgpu_vec *b, uint64_t *out_host);
/* y = A x.  A: rows * cols elements with whole rows per part (fpgpu_vec_alloc_rows); x: cols elements,
 * whole on every device that holds rows (replicated, or on A's device); y: rows elements, either
 * partitioned like A's rows (no communication) or replicated: then every device's row block is
 * all-gathered into every copy -- ncclAllGather / ncclBroadcast over xGMI through communicators made
 * by ncclCommInitAll for the device list (librccl is loaded at the first such call); direct peer
 * copies instead when the list names a device twice or FPGPU_ALLGATHER_COPY is set. */
int fpgpu_vec_matvec(const fpgpu_ctx *ctx, const fpgpu_vec *A, size_t rows, size_t cols, const fpgpu_vec *x,
                     fpgpu_vec *y);
/* C = A B.  A (m x k) and C (m x n) in row blocks, B (k x n) whole on every device; a replicated C is
 * all-gathered as above. */
int fpgpu_vec_matmul(const fpgpu_ctx *ctx, const fpgpu_vec *A, const fpgpu_vec *B, fpgpu_vec *C, size_t m, size_t k,
                     size_t n);

/* y = A x over SparseMatrix<T> (src/matrix.rs:46-61) on handles: `vals` holds rows * density values in
 * row blocks (fpgpu_vec_alloc_rows(ctx, rows, density, placement)), `col_idx` the cells' 32-bit column
 * indices in the same layout (an fpgpu_idx: same placements, 4-byte elements), x whole on every device
 * that holds rows, y partitioned like the rows or replicated (then all-gathered as above). */
typedef struct fpgpu_idx fpgpu_idx;
int fpgpu_idx_alloc_rows(size_t rows, size_t density, int placement, fpgpu_idx **out);
int fpgpu_idx_alloc_rows_on(const fpgpu_devset *set, size_t rows, size_t density, int placement, fpgpu_idx **out);
void fpgpu_idx_free(fpgpu_idx *idx);
int fpgpu_idx_upload(fpgpu_idx *idx, const uint32_t *host, size_t first, size_t n);
int fpgpu_vec_spmv(const fpgpu_ctx *ctx, const fpgpu_vec *vals, const fpgpu_idx *col_idx, size_t rows, size_t cols,
                   size_t density, const fpgpu_vec *x, fpgpu_vec *y);

/* ---- tuning / test hooks -----------------------------------------------------
 * No counterpart in the reference.  The hooks of INTEGRATION.md section 4 (FPGPU_GRID_CAP,
 * FPGPU_INV_C, FPGPU_INV_C0, FPGPU_INV_FUSE_TOP, FPGPU_INV_LP, FPGPU_INV_SP, FPGPU_MATVEC_MFMA, FPGPU_MATVEC_KSPLIT, FPGPU_MATVEC_T, FPGPU_MATVEC_COLS, FPGPU_MATVEC_ROWS, FPGPU_MATMUL_MFMA,
 * FPGPU_MATMUL_GENERIC, FPGPU_SPMV_LOG2G, FPGPU_SPMV_UNROLL, FPGPU_ALLGATHER_COPY, FPGPU_COPY_THREADS, FPGPU_POW_WINDOW) are read from the environment once, when the library
 * is first used; this call sets one afterwards (process-wide; FPGPU_TUNING_DEFAULT restores the
 * library's choice).  No hook changes a result -- they pick between bit-identical paths.  Not
 * synchronised against calls running on other threads (set them before the threads start). */
#define FPGPU_TUNING_DEFAULT (-0x7fffffffL - 1)
int fpgpu_tuning_set(const char *name, long value);

/* ---- diagnostics ----------------------------------------------------------- */
const char *fpgpu_last_error_string(void); /* thread-local, never NULL */
int fpgpu_device_count(void);              /* gfx950 devices visible to this process */
const char *fpgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FPGPU_H */
