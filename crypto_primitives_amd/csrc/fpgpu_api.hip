// fpgpu_api.hip -- the C ABI of include/fpgpu.h: context, host-side Montgomery
// parameters, argument checks and kernel dispatch.  There is no CPU fallback:
// every batch entry point launches a gfx950 kernel or returns an error.
#include "../../include/fpgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fp_launch.hpp"
#include "fpgpu_internal.hpp"

using namespace fpgpu;

namespace fpgpu {
thread_local std::string g_last_error = "";
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(FPGPU_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
}  // namespace fpgpu

namespace {

// ---- host-side big integers (product code: parameter set-up only) ----------------
using u128 = unsigned __int128;

int cmp_words(const uint64_t* a, const uint64_t* b, int n) {
    for (int i = n - 1; i >= 0; i--)
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
void sub_words(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
    uint64_t bo = 0;
    for (int i = 0; i < n; i++) {
        u128 t = (u128)a[i] - b[i] - bo;
        r[i] = (uint64_t)t;
        bo = (uint64_t)(t >> 64) & 1;
    }
}
// a = 2a mod p for a < p
void double_mod(uint64_t* a, const uint64_t* p, int n) {
    uint64_t top = a[n - 1] >> 63;
    for (int i = n - 1; i > 0; i--) a[i] = (a[i] << 1) | (a[i - 1] >> 63);
    a[0] <<= 1;
    if (top || cmp_words(a, p, n) >= 0) sub_words(a, a, p, n);
}
// -m^-1 mod 2^64 by Newton iteration (the reference does the same six steps,
// src/helpers/crypto_bigint.rs:56-67)
uint64_t neg_inv64(uint64_t m0) {
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - m0 * inv;
    return 0 - inv;
}


// ---- static device scratch of fpgpu_batch_inv_static_dev -------------------------------
// Workspace of the batch inversion for callers that cannot allocate (HIP graph capture,
// serving loops): a file-scope device array, one copy per GPU at its own address, held until
// the process ends.  Host code never uses the array's name as an address: the per-device
// pointer comes from hipGetSymbolAddress, once per device, and is what the kernels receive.
static __device__ __align__(256) unsigned char g_static_scratch[FPGPU_STATIC_SCRATCH_BYTES];
constexpr int kMaxDevices = 64;
std::mutex g_static_scratch_mutex;
void* g_static_scratch_ptr[kMaxDevices] = {nullptr};

// Device address of g_static_scratch on the CURRENT device (looked up on the first call there).
int static_scratch_for_current_device(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(FPGPU_ERR_NO_DEVICE, "no HIP device visible: libfpgpu has no CPU fallback");
    }
    if (dev < 0 || dev >= kMaxDevices) return fail(FPGPU_ERR_UNSUPPORTED, "device index beyond the static scratch table");
    std::lock_guard<std::mutex> lock(g_static_scratch_mutex);
    if (!g_static_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch));
        if (e != hipSuccess || !p) return hip_fail(e, "hipGetSymbolAddress(g_static_scratch)");
        g_static_scratch_ptr[dev] = p;
    }
    *out = g_static_scratch_ptr[dev];
    return FPGPU_OK;
}

}  // namespace

namespace fpgpu {

// The tuning words: environment once (first use), fpgpu_tuning_set afterwards.
struct TuningEntry {
    const char* name;
    long Tuning::*field;
};
static const TuningEntry kTuningEntries[] = {
    {"FPGPU_GRID_CAP", &Tuning::grid_cap},         {"FPGPU_INV_C", &Tuning::inv_c},
    {"FPGPU_MATVEC_MFMA", &Tuning::matvec_mfma},   {"FPGPU_MATVEC_KSPLIT", &Tuning::matvec_ksplit},
    {"FPGPU_MATVEC_T", &Tuning::matvec_t},         {"FPGPU_MATMUL_MFMA", &Tuning::matmul_mfma},
    {"FPGPU_MATMUL_GENERIC", &Tuning::matmul_generic}, {"FPGPU_SPMV_LOG2G", &Tuning::spmv_log2g},
    {"FPGPU_ALLGATHER_COPY", &Tuning::allgather_copy}, {"FPGPU_COPY_THREADS", &Tuning::copy_threads},
    {"FPGPU_POW_WINDOW", &Tuning::pow_window},         {"FPGPU_SPMV_UNROLL", &Tuning::spmv_unroll},
    {"FPGPU_INV_C0", &Tuning::inv_c0},                 {"FPGPU_INV_FUSE_TOP", &Tuning::inv_fuse_top},
    {"FPGPU_INV_LP", &Tuning::inv_lp},                 {"FPGPU_MATVEC_COLS", &Tuning::matvec_cols},
    {"FPGPU_MATVEC_ROWS", &Tuning::matvec_rows},        {"FPGPU_INV_SP", &Tuning::inv_sp},
};
static Tuning g_tuning_defaults;
static Tuning g_tuning;
static std::once_flag g_tuning_once;
static Tuning& tuning_mutable() {
    std::call_once(g_tuning_once, [] {
        for (const TuningEntry& t : kTuningEntries)
            if (const char* e = getenv(t.name); e && *e) g_tuning.*(t.field) = atol(e);
    });
    return g_tuning;
}
const Tuning& tuning() { return tuning_mutable(); }

int grid_for(size_t n) {
    // One element per lane up to 2^20 workgroups, grid-stride beyond.  Measured on the
    // 2^28-element BN254 mul (MI355X): cap 2048 4.99 TB/s, 16384 5.22, 65536 5.38,
    // 2^20 (no striding) 5.67-5.77, 2^22 5.68: short-lived workgroups keep the memory
    // pipeline fuller than long grid-stride loops for these VALU-heavy streams.
    size_t blocks = (n + kBlock - 1) / kBlock;
    size_t cap = (size_t)1 << 20;
    if (const long v = tuning().grid_cap; v > 0) cap = (size_t)v;  // tuning only
    return (int)std::min<size_t>(std::max<size_t>(blocks, 1), cap);
}


InvPlan make_inv_plan(uint32_t n64, size_t n) {
    InvPlan p;
    const size_t esz = (size_t)n64 * 8;
    p.count[0] = n;
    // at least one level, so the top always works on chain totals
    while (p.levels < kInvMaxLevels && (p.levels == 0 || p.count[p.levels] > kInvTop)) {
        size_t c = p.count[p.levels];
        int len = kInvCMin;
        while (len < kInvCMax && c / (2 * (size_t)len) >= kInvLanesWanted) len *= 2;
        if (p.levels == 0) {
            // A tree level costs 80-100 us beside sweeps of 100-300 us (tools/exp_binv_threshold.py, BLS12-381: 2^21 + 1
            // elements 291 us with chains of 32 and a second level, 192 us with chains of 33 and none; 600 000: 145 -> 108;
            // 70 000: 97 -> 79 with chains of 2).  Up to kInvOneLevelMax elements the level-0 chains are therefore just
            // long enough for the totals to fit the top kernel (one level, the fewest multiplications that allows);
            // beyond, where chains of 90 and more lose to two levels, they are stretched only as far as it takes for
            // the level above to finish the tree (2^24 + 1 elements: 33 and 8, not 32, 8 and 8).
            const size_t one = (c + kInvTop - 1) / kInvTop;
            const size_t two = (c + (size_t)kInvCMin * kInvTop - 1) / ((size_t)kInvCMin * kInvTop);
            if (c <= kInvOneLevelMax) len = (int)(one < 2 ? 2 : one);
            else if (two > (size_t)len && two <= kInvStretchMax) len = (int)two;
        }
        if (const long v = tuning().inv_c; v >= 2 && v <= 1024) len = (int)v;  // tuning / tests
        if (const long v = tuning().inv_c0; p.levels == 0 && v >= 2 && v <= 1024) len = (int)v;
        p.c[p.levels] = len;
        p.count[p.levels + 1] = (c + len - 1) / len;   // chains of `len` elements, see k_binv_up
        p.levels++;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    for (int l = 1; l <= p.levels; l++) p.x_off[l] = take(p.count[l] * esz);
    for (int l = 1; l < p.levels; l++) p.o_off[l] = take(p.count[l] * esz);
    p.poison_off = take(p.count[p.levels]);
    p.any_off = take(sizeof(uint32_t));
    // The size REPORTED never shrinks as n grows: a caller sizes a workspace for its largest batch and then inverts
    // shorter ones in it (the chunk pipeline's last chunk, the static scratch, a serving loop) -- while the layout
    // above does shrink where a longer chain takes over (65 537 totals at 131 073 elements in chains of 2, 43 691 at
    // 131 074 in chains of 3).  The bound: the most totals any shorter batch can have at each level under the rules
    // above (hooks aside), laid out the same way.
    {
        size_t bound = 0;
        auto put = [&](size_t bytes) { bound += (bytes + 255) & ~(size_t)255; };
        size_t cl = n <= kInvOneLevelMax ? ((n + 1) / 2 < kInvTop ? (n + 1) / 2 : kInvTop) : (n + kInvCMax - 1) / kInvCMax;
        for (int l = 1; l <= kInvMaxLevels; l++) {
            put(cl * esz);                       // X_l
            if (cl <= kInvTop) break;
            put(cl * esz);                       // O_l (every level but the last)
            cl = (cl + kInvCMin - 1) / kInvCMin;
        }
        put(kInvTop);                            // poison flags of the top
        put(sizeof(uint32_t));
        if (bound > off) off = bound;
    }
    p.bytes = off;
    // the last stored level rides in the top kernel when its chains are short (then that level's own prefix
    // array O_l is still what the level below reads its inverted totals from; X_(l+1) goes unused)
    p.fused_top = p.levels >= 2 && p.c[p.levels - 1] <= kInvTopChainMaxC;
    // level-0 prefixes as 29-bit limbs, 4 * STRIDE bytes each (fp_kernels.cuh RawForm).  Measured (tools/exp_binv_ab.py,
    // 2^24 elements): it pays from 384 bits on (BLS12-381 1.065 -> 1.047 ms, 512 bits 1.62 -> 1.57), where the limb form
    // is 12-17 % larger than the words; at 256 bits it is 25 % larger and loses (0.618 -> 0.638 ms)
    if (n64 >= 6) {
        const size_t n32 = 2 * (size_t)n64, raw = (32 * n32 + 28) / 29, stride = (raw + 1) & ~(size_t)1;
        p.lp_off = take(((n + 63) / 64 * 64) * stride * 4);   // whole tiles of 64 elements
        p.bytes_lp = off;
    }
    return p;
}

}  // namespace fpgpu

namespace {

int check_common(const fpgpu_ctx* ctx, size_t n) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (n > ((size_t)1 << 40)) return fail(FPGPU_ERR_SHAPE, "element count too large");
    return FPGPU_OK;
}
int check_ptr(const fpgpu_ctx* ctx, const void* p, const char* name) {
    if (!p) return fail(FPGPU_ERR_SHAPE, std::string("null pointer: ") + name);
    uintptr_t align = (ctx->n64 % 2 == 0) ? 16 : 8;
    if ((uintptr_t)p % align) return fail(FPGPU_ERR_SHAPE, std::string("misaligned device pointer: ") + name);
    return FPGPU_OK;
}
template <int OP>
int elementwise_dev(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    CHECK(check_ptr(ctx, a, "a"));
    if (OP == OP_MUL || OP == OP_ADD || OP == OP_SUB) CHECK(check_ptr(ctx, b, "b"));
    CHECK(check_ptr(ctx, out, "out"));
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::elementwise(OP, ctx, a, b, out, n, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

// ---- host staging helper --------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc"); }
        return FPGPU_OK;
    }
};

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(FPGPU_ERR_NO_DEVICE, "no HIP device visible: libfpgpu has no CPU fallback");
    }
    return FPGPU_OK;
}

template <typename F>
int host_binary(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, bool has_b,
                F&& dev_call) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    if (!a || !out || (has_b && !b)) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    CHECK(require_device());
    return host_stream_elementwise(a, has_b ? b : nullptr, out, n, (size_t)ctx->n64 * 8, dev_call);
}

}  // namespace

extern "C" {

int fpgpu_ctx_create(const uint64_t* modulus_le, uint32_t n64, fpgpu_ctx** out) {
    if (!modulus_le || !out) return fail(FPGPU_ERR_SHAPE, "null pointer");
    *out = nullptr;
    if (n64 == 0 || n64 > FPGPU_MAX_N64) return fail(FPGPU_ERR_UNSUPPORTED, "n64 must be in 1..8");
    if ((modulus_le[0] & 1) == 0) return fail(FPGPU_ERR_INVALID_MODULUS, "modulus must be odd");
    fpgpu_ctx* c = new (std::nothrow) fpgpu_ctx();
    if (!c) return fail(FPGPU_ERR_HIP, "out of host memory");
    c->n64 = n64;
    std::memset(c->modulus, 0, sizeof c->modulus);
    std::memcpy(c->modulus, modulus_le, n64 * 8);
    c->mod_neg_inv = neg_inv64(modulus_le[0]);
    // R mod p and R^2 mod p by 64*n64 modular doublings each, starting from 1 mod p.
    uint64_t t[FPGPU_MAX_N64] = {0};
    t[0] = 1;
    if (cmp_words(t, c->modulus, n64) >= 0) sub_words(t, t, c->modulus, n64);  // p == 1
    for (uint32_t i = 0; i < 64 * n64; i++) double_mod(t, c->modulus, n64);
    std::memcpy(c->one, t, sizeof t);
    for (uint32_t i = 0; i < 64 * n64; i++) double_mod(t, c->modulus, n64);
    std::memcpy(c->r2, t, sizeof t);
    // bit length of the modulus (for the synthetic generator)
    c->bitlen = 0;
    for (int i = (int)n64 - 1; i >= 0; i--)
        if (c->modulus[i]) { c->bitlen = 64u * i + 64u - (uint32_t)__builtin_clzll(c->modulus[i]); break; }
    *out = c;
    return FPGPU_OK;
}

void fpgpu_ctx_destroy(fpgpu_ctx* ctx) { delete ctx; }

uint32_t fpgpu_ctx_n64(const fpgpu_ctx* ctx) { return ctx ? ctx->n64 : 0; }

int fpgpu_ctx_params(const fpgpu_ctx* ctx, uint64_t* modulus_le, uint64_t* one_le, uint64_t* r2_le,
                     uint64_t* mod_neg_inv) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (modulus_le) std::memcpy(modulus_le, ctx->modulus, ctx->n64 * 8);
    if (one_le) std::memcpy(one_le, ctx->one, ctx->n64 * 8);
    if (r2_le) std::memcpy(r2_le, ctx->r2, ctx->n64 * 8);
    if (mod_neg_inv) *mod_neg_inv = ctx->mod_neg_inv;
    return FPGPU_OK;
}

int fpgpu_mul_dev(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_MUL>(ctx, a, b, out, n, stream);
}
int fpgpu_sqr_dev(const fpgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_SQR>(ctx, a, nullptr, out, n, stream);
}
int fpgpu_add_dev(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_ADD>(ctx, a, b, out, n, stream);
}
int fpgpu_sub_dev(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_SUB>(ctx, a, b, out, n, stream);
}
int fpgpu_neg_dev(const fpgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_NEG>(ctx, a, nullptr, out, n, stream);
}
int fpgpu_to_mont_dev(const fpgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_TO_MONT>(ctx, a, nullptr, out, n, stream);
}
int fpgpu_from_mont_dev(const fpgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_FROM_MONT>(ctx, a, nullptr, out, n, stream);
}
int fpgpu_double_dev(const fpgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_DOUBLE>(ctx, a, nullptr, out, n, stream);
}
int fpgpu_div_by_2_dev(const fpgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    return elementwise_dev<OP_HALVE>(ctx, a, nullptr, out, n, stream);
}

int fpgpu_mul_chain_dev(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, uint32_t iters,
                        void* stream) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    CHECK(check_ptr(ctx, a, "a"));
    CHECK(check_ptr(ctx, b, "b"));
    CHECK(check_ptr(ctx, out, "out"));
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::mul_chain(ctx, a, b, out, n, iters, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

size_t fpgpu_batch_inv_workspace_bytes(const fpgpu_ctx* ctx, size_t n) {
    if (!ctx || n == 0) return 0;
    return make_inv_plan(ctx->n64, n).bytes;
}

size_t fpgpu_batch_inv_fast_workspace_bytes(const fpgpu_ctx* ctx, size_t n) {
    if (!ctx || n == 0) return 0;
    const InvPlan p = make_inv_plan(ctx->n64, n);
    return p.bytes_lp ? p.bytes_lp : p.bytes;
}

int fpgpu_batch_inv_dev(const fpgpu_ctx* ctx, const void* x, void* out, uint8_t* zero_flags, size_t n,
                        void* workspace, size_t workspace_bytes, void* stream) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    CHECK(check_ptr(ctx, x, "x"));
    CHECK(check_ptr(ctx, out, "out"));
    if (x == out) return fail(FPGPU_ERR_SHAPE, "batch_inv: out must not alias x");
    if (!workspace || (uintptr_t)workspace % 16) return fail(FPGPU_ERR_SHAPE, "batch_inv: workspace null or misaligned");
    if (workspace_bytes < fpgpu_batch_inv_workspace_bytes(ctx, n))
        return fail(FPGPU_ERR_SHAPE, "batch_inv: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::batch_inv(ctx, x, out, zero_flags, n, (char*)workspace, workspace_bytes, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

int fpgpu_div_dev(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, uint8_t* zero_flags, size_t n,
                  void* workspace, size_t workspace_bytes, void* stream) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    CHECK(check_ptr(ctx, a, "a"));
    if (a == out) return fail(FPGPU_ERR_SHAPE, "div: out must not alias a");
    CHECK(fpgpu_batch_inv_dev(ctx, b, out, zero_flags, n, workspace, workspace_bytes, stream));   // out = 1 / b, 0 where None
    return fpgpu_mul_dev(ctx, a, out, out, n, stream);
}

size_t fpgpu_static_scratch_bytes(void) { return FPGPU_STATIC_SCRATCH_BYTES; }

int fpgpu_batch_inv_static_dev(const fpgpu_ctx* ctx, const void* x, void* out, uint8_t* zero_flags, size_t n,
                               void* stream) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    CHECK(check_ptr(ctx, x, "x"));
    CHECK(check_ptr(ctx, out, "out"));
    if (x == out) return fail(FPGPU_ERR_SHAPE, "batch_inv: out must not alias x");
    const size_t need = fpgpu_batch_inv_workspace_bytes(ctx, n);
    if (need > (size_t)FPGPU_STATIC_SCRATCH_BYTES)
        return fail(FPGPU_ERR_SHAPE, "batch_inv_static: " + std::to_string(n) + " elements need " + std::to_string(need) +
                                         " bytes of scratch, the static array holds " +
                                         std::to_string((size_t)FPGPU_STATIC_SCRATCH_BYTES) +
                                         " (use fpgpu_batch_inv_dev with a caller workspace)");
    void* scratch = nullptr;
    CHECK(static_scratch_for_current_device(&scratch));
    hipStream_t s = (hipStream_t)stream;
    // the launcher clears what must start at zero (the poison word); everything else in the
    // scratch is written before it is read, so the previous call's contents do not matter
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::batch_inv(ctx, x, out, zero_flags, n, (char*)scratch, FPGPU_STATIC_SCRATCH_BYTES, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

size_t fpgpu_matvec_workspace_bytes(const fpgpu_ctx* ctx, size_t rows, size_t cols) {
    if (!ctx) return 0;
    const MvPlan pl = make_mv_plan(ctx->n64, rows, cols);
    return pl.usable ? pl.bytes : mv_split_bytes(ctx->n64, rows, cols);   // few long rows outside the plan: scratch for column parts
}

int fpgpu_matvec_dev(const fpgpu_ctx* ctx, const void* A, size_t rows, size_t cols, const void* x, void* y,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (workspace && (uintptr_t)workspace % 256) return fail(FPGPU_ERR_SHAPE, "matvec: workspace not 256-byte aligned");
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (rows == 0) return FPGPU_OK;
    if (cols != 0 && rows > ((size_t)1 << 40) / cols) return fail(FPGPU_ERR_SHAPE, "matrix too large");
    CHECK(check_ptr(ctx, y, "y"));
    if (cols != 0) {
        CHECK(check_ptr(ctx, A, "A"));
        CHECK(check_ptr(ctx, x, "x"));
    }
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::matvec(ctx, A, rows, cols, x, y, workspace, workspace_bytes, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

int fpgpu_spmv_dev(const fpgpu_ctx* ctx, const void* vals, const uint32_t* col_idx, size_t rows, size_t cols,
                   size_t density, const void* x, void* y, void* stream) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (rows == 0) return FPGPU_OK;
    if (density > 0xffffffffu || cols > 0xffffffffu || rows > ((size_t)1 << 40) / (density ? density : 1))
        return fail(FPGPU_ERR_SHAPE, "sparse matrix too large");
    CHECK(check_ptr(ctx, y, "y"));
    if (density != 0) {
        CHECK(check_ptr(ctx, vals, "vals"));
        if (!col_idx || (uintptr_t)col_idx % 4) return fail(FPGPU_ERR_SHAPE, "null or misaligned pointer: col_idx");
        if (cols) CHECK(check_ptr(ctx, x, "x"));
    }
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::spmv(ctx, vals, col_idx, rows, cols, (uint32_t)density, x, y, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

size_t fpgpu_matmul_workspace_bytes(const fpgpu_ctx* ctx, size_t m, size_t k, size_t n) {
    if (!ctx) return 0;
    if (n == 1) return fpgpu_matvec_workspace_bytes(ctx, m, k);   // a mat-vec: served by its kernels
    if (vm_shape(m, k, n)) {   // a few vectors times a matrix: k parts only when the columns alone cannot fill the chip
        if (const size_t vm = vm_split_bytes(ctx->n64, k, n)) return vm;
        if (n >= 16384) return 0;
    }
    if (const size_t split = mm_split_bytes(ctx->n64, m, k, n)) return split;   // a small output over a long k: scratch for the k parts
    if (ctx->n64 >= 2) {   // multi-limb: digit planes and scratch of the matrix-core product, where it is expected to win (or the hook asks)
        const MflPlan ml = make_mfl_plan(ctx->n64, m, k, n, tuning().matmul_mfma > 0);
        return ml.usable && tuning().matmul_mfma != 0 ? ml.bytes : 0;
    }
    const MfmaPlan pl = make_mfma_plan(ctx->n64, m, k, n);
    // a single-limb product with k beyond the int32 accumulators, or with too few tiles for the chip: the limb kernel's planes
    // and scratch (never less than the single-limb kernel's planes: a caller may still force that one)
    const MflPlan ml = make_mfl_plan(ctx->n64, m, k, n, pl.usable || tuning().matmul_mfma > 0);
    if (pl.usable) return one_limb_takes_limb_kernel(pl, ml) && ml.bytes > pl.bytes ? ml.bytes : pl.bytes;
    return ml.usable && tuning().matmul_mfma != 0 ? ml.bytes : 0;
}

int fpgpu_matmul_dev(const fpgpu_ctx* ctx, const void* A, const void* B, void* C, size_t m, size_t k, size_t n,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (workspace && (uintptr_t)workspace % 256) return fail(FPGPU_ERR_SHAPE, "matmul: workspace not 256-byte aligned");
    if (m == 0 || n == 0) return FPGPU_OK;
    // one column: C = A B is the mat-vec y = A x with x = B, y = C (4096 x 4096 x 1: 12 G terms/s through the tiled
    // kernel, the mat-vec kernels' rate through this)
    if (n == 1) return fpgpu_matvec_dev(ctx, A, m, k, B, C, workspace, workspace_bytes, stream);
    CHECK(check_ptr(ctx, C, "C"));
    if (k != 0) {
        CHECK(check_ptr(ctx, A, "A"));
        CHECK(check_ptr(ctx, B, "B"));
    }
    size_t gx = (n + kMMTile - 1) / kMMTile, gy = (m + kMMTile - 1) / kMMTile;
    if (gy > 65535 || gx > ((size_t)1 << 31) - 1) return fail(FPGPU_ERR_SHAPE, "matrix too large");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::matmul(ctx, A, B, C, m, k, n, workspace, workspace_bytes, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

int fpgpu_pow_dev(const fpgpu_ctx* ctx, const void* a, const uint64_t* exponent_le, uint32_t exponent_words, void* out,
                  size_t n, void* stream) {
    CHECK(check_common(ctx, n));
    if (exponent_words > 8) return fail(FPGPU_ERR_UNSUPPORTED, "pow: exponent wider than 512 bits");
    if (exponent_words && !exponent_le) return fail(FPGPU_ERR_SHAPE, "null pointer: exponent");
    if (n == 0) return FPGPU_OK;
    CHECK(check_ptr(ctx, a, "a"));
    CHECK(check_ptr(ctx, out, "out"));
    uint32_t e32[16] = {0};
    int top = -1;
    for (uint32_t i = 0; i < exponent_words; i++) {
        e32[2 * i] = (uint32_t)exponent_le[i];
        e32[2 * i + 1] = (uint32_t)(exponent_le[i] >> 32);
        if (exponent_le[i]) top = 64 * (int)i + 63 - __builtin_clzll(exponent_le[i]);
    }
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::pow(ctx, a, e32, top, out, n, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

size_t fpgpu_fold_workspace_bytes(const fpgpu_ctx* ctx) {
    return ctx ? (size_t)kFoldBlocks * ctx->n64 * 8 : 0;
}

static int fold_dev(bool product, const fpgpu_ctx* ctx, const void* v, size_t n, void* out, void* workspace,
                    size_t workspace_bytes, void* stream) {
    CHECK(check_common(ctx, n));
    CHECK(check_ptr(ctx, out, "out"));
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {  // empty fold: the identity (sum -> zero, product -> one), src/semiring.rs:213-241
        const uint64_t zero[FPGPU_MAX_N64] = {0};
        HIP_TRY(hipMemcpyAsync(out, product ? ctx->one : zero, ctx->n64 * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));  // the source is on this call's stack / in the context
        return FPGPU_OK;
    }
    CHECK(check_ptr(ctx, v, "v"));
    if (n > (size_t)kBlock) {
        if (!workspace || (uintptr_t)workspace % 16) return fail(FPGPU_ERR_SHAPE, "fold: workspace null or misaligned");
        if (workspace_bytes < fpgpu_fold_workspace_bytes(ctx)) return fail(FPGPU_ERR_SHAPE, "fold: workspace too small");
    }
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::fold(product, ctx, v, n, out, workspace, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}
int fpgpu_sum_dev(const fpgpu_ctx* ctx, const void* v, size_t n, void* out, void* workspace, size_t workspace_bytes,
                  void* stream) {
    return fold_dev(false, ctx, v, n, out, workspace, workspace_bytes, stream);
}
int fpgpu_product_dev(const fpgpu_ctx* ctx, const void* v, size_t n, void* out, void* workspace,
                      size_t workspace_bytes, void* stream) {
    return fold_dev(true, ctx, v, n, out, workspace, workspace_bytes, stream);
}

int fpgpu_dot_dev(const fpgpu_ctx* ctx, const void* a, const void* b, size_t n, void* out, void* workspace,
                  size_t workspace_bytes, void* stream) {
    CHECK(check_common(ctx, n));
    CHECK(check_ptr(ctx, out, "out"));
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {  // the empty sum is zero (src/semiring.rs:213-226)
        HIP_TRY(hipMemsetAsync(out, 0, ctx->n64 * 8, s));
        return FPGPU_OK;
    }
    CHECK(check_ptr(ctx, a, "a"));
    CHECK(check_ptr(ctx, b, "b"));
    if (n > (size_t)kBlock) {
        if (!workspace || (uintptr_t)workspace % 16) return fail(FPGPU_ERR_SHAPE, "dot: workspace null or misaligned");
        if (workspace_bytes < fpgpu_fold_workspace_bytes(ctx)) return fail(FPGPU_ERR_SHAPE, "dot: workspace too small");
    }
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::dot(ctx, a, b, n, out, workspace, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

int fpgpu_fill_random_dev(const fpgpu_ctx* ctx, void* out, size_t n, uint64_t seed, void* stream) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    CHECK(check_ptr(ctx, out, "out"));
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::fill_random(ctx, out, n, seed, 0, s)));
    HIP_TRY(hipGetLastError());
    return FPGPU_OK;
}

// ---- host-buffer calls -----------------------------------------------------------------
int fpgpu_mul(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    return host_binary(ctx, a, b, out, n, true,
                       [&](void* da, void* db, void* dout, size_t cnt, hipStream_t s) { return fpgpu_mul_dev(ctx, da, db, dout, cnt, s); });
}
int fpgpu_sqr(const fpgpu_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return host_binary(ctx, a, nullptr, out, n, false,
                       [&](void* da, void*, void* dout, size_t cnt, hipStream_t s) { return fpgpu_sqr_dev(ctx, da, dout, cnt, s); });
}
int fpgpu_add(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    return host_binary(ctx, a, b, out, n, true,
                       [&](void* da, void* db, void* dout, size_t cnt, hipStream_t s) { return fpgpu_add_dev(ctx, da, db, dout, cnt, s); });
}
int fpgpu_sub(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    return host_binary(ctx, a, b, out, n, true,
                       [&](void* da, void* db, void* dout, size_t cnt, hipStream_t s) { return fpgpu_sub_dev(ctx, da, db, dout, cnt, s); });
}
int fpgpu_neg(const fpgpu_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return host_binary(ctx, a, nullptr, out, n, false,
                       [&](void* da, void*, void* dout, size_t cnt, hipStream_t s) { return fpgpu_neg_dev(ctx, da, dout, cnt, s); });
}
int fpgpu_to_mont(const fpgpu_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return host_binary(ctx, a, nullptr, out, n, false,
                       [&](void* da, void*, void* dout, size_t cnt, hipStream_t s) { return fpgpu_to_mont_dev(ctx, da, dout, cnt, s); });
}
int fpgpu_from_mont(const fpgpu_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return host_binary(ctx, a, nullptr, out, n, false,
                       [&](void* da, void*, void* dout, size_t cnt, hipStream_t s) { return fpgpu_from_mont_dev(ctx, da, dout, cnt, s); });
}
int fpgpu_double(const fpgpu_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return host_binary(ctx, a, nullptr, out, n, false,
                       [&](void* da, void*, void* dout, size_t cnt, hipStream_t s) { return fpgpu_double_dev(ctx, da, dout, cnt, s); });
}
int fpgpu_div_by_2(const fpgpu_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n) {
    return host_binary(ctx, a, nullptr, out, n, false,
                       [&](void* da, void*, void* dout, size_t cnt, hipStream_t s) { return fpgpu_div_by_2_dev(ctx, da, dout, cnt, s); });
}
int fpgpu_mul_chain(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n,
                    uint32_t iters) {
    return host_binary(ctx, a, b, out, n, true, [&](void* da, void* db, void* dout, size_t cnt, hipStream_t s) {
        return fpgpu_mul_chain_dev(ctx, da, db, dout, cnt, iters, s);
    });
}

int fpgpu_pow(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* exponent_le, uint32_t exponent_words,
              uint64_t* out, size_t n) {
    return host_binary(ctx, a, nullptr, out, n, false, [&](void* da, void*, void* dout, size_t cnt, hipStream_t s) {
        return fpgpu_pow_dev(ctx, da, exponent_le, exponent_words, dout, cnt, s);
    });
}

static int fold_host(bool product, const fpgpu_ctx* ctx, const uint64_t* v, size_t n, uint64_t* out) {
    CHECK(check_common(ctx, n));
    if (!out || (n && !v)) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    CHECK(require_device());
    size_t esz = ctx->n64 * 8;
    DevBuf dv, dout, ws;
    CHECK(dv.alloc(n * esz));
    CHECK(dout.alloc(esz));
    CHECK(ws.alloc(fpgpu_fold_workspace_bytes(ctx)));
    if (n) CHECK(upload_staged(dv.p, v, n * esz, nullptr));
    CHECK(fold_dev(product, ctx, dv.p, n, dout.p, ws.p, fpgpu_fold_workspace_bytes(ctx), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    CHECK(download_staged(out, dout.p, esz, nullptr));
    return FPGPU_OK;
}
int fpgpu_sum(const fpgpu_ctx* ctx, const uint64_t* v, size_t n, uint64_t* out) { return fold_host(false, ctx, v, n, out); }
int fpgpu_product(const fpgpu_ctx* ctx, const uint64_t* v, size_t n, uint64_t* out) {
    return fold_host(true, ctx, v, n, out);
}

int fpgpu_dot(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
    CHECK(check_common(ctx, n));
    if (!out || (n && (!a || !b))) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    CHECK(require_device());
    size_t esz = ctx->n64 * 8;
    DevBuf da, db, dout, ws;
    CHECK(da.alloc(n * esz));
    CHECK(db.alloc(n * esz));
    CHECK(dout.alloc(esz));
    CHECK(ws.alloc(fpgpu_fold_workspace_bytes(ctx)));
    if (n) {
        CHECK(upload_staged(da.p, a, n * esz, nullptr));
        CHECK(upload_staged(db.p, b, n * esz, nullptr));
    }
    CHECK(fpgpu_dot_dev(ctx, da.p, db.p, n, dout.p, ws.p, fpgpu_fold_workspace_bytes(ctx), nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    CHECK(download_staged(out, dout.p, esz, nullptr));
    return FPGPU_OK;
}

// Large host batches of inversions / divisions: independent ranges through the chunk pipeline (each range its
// own Montgomery trick; an inverse does not depend on how the batch is cut), copies in both directions
// overlapping the kernels.  FPGPU_ERR_UNSUPPORTED = take the whole-operand path.
static int inv_or_div_streamed(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* zero_flags,
                               size_t n) {
    if (!host_stream_usable(n * (size_t)ctx->n64 * 8)) return FPGPU_ERR_UNSUPPORTED;
    StreamJob job;
    job.a = a ? (const void*)a : (const void*)b;   // division: a | b; inversion: x alone
    job.b = a ? b : nullptr;
    job.out = out;
    job.flags = zero_flags;
    job.want_flags = true;
    job.n = n;
    job.esz = (size_t)ctx->n64 * 8;
    job.lane_compute = true;
    job.workspace_for = [ctx](size_t count) { return fpgpu_batch_inv_workspace_bytes(ctx, count); };
    const bool div = a != nullptr;
    job.call = [ctx, div](void* da, void* db, void* dout, uint8_t* dflags, void* ws, size_t wsb, size_t cnt, hipStream_t s) {
        return div ? fpgpu_div_dev(ctx, da, db, dout, dflags, cnt, ws, wsb, s)
                   : fpgpu_batch_inv_dev(ctx, da, dout, dflags, cnt, ws, wsb, s);
    };
    return host_stream_map(job);
}

int fpgpu_batch_inv(const fpgpu_ctx* ctx, const uint64_t* x, uint64_t* out, uint8_t* zero_flags, size_t n) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    if (!x || !out) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    CHECK(require_device());
    if (const int rc = inv_or_div_streamed(ctx, nullptr, x, out, zero_flags, n); rc != FPGPU_ERR_UNSUPPORTED) return rc;
    size_t bytes = n * ctx->n64 * 8;
    size_t wsb = fpgpu_batch_inv_fast_workspace_bytes(ctx, n);
    DevBuf dx, dout, dflags, ws;
    CHECK(dx.alloc(bytes));
    CHECK(dout.alloc(bytes));
    CHECK(dflags.alloc(n));
    CHECK(ws.alloc(wsb));
    CHECK(upload_staged(dx.p, x, bytes, nullptr));
    CHECK(fpgpu_batch_inv_dev(ctx, dx.p, dout.p, (uint8_t*)dflags.p, n, ws.p, wsb, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    CHECK(download_staged(out, dout.p, bytes, nullptr));
    if (zero_flags) CHECK(download_staged(zero_flags, dflags.p, n, nullptr));
    return FPGPU_OK;
}

int fpgpu_div(const fpgpu_ctx* ctx, const uint64_t* a, const uint64_t* b, uint64_t* out, uint8_t* zero_flags, size_t n) {
    CHECK(check_common(ctx, n));
    if (n == 0) return FPGPU_OK;
    if (!a || !b || !out) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    CHECK(require_device());
    if (const int rc = inv_or_div_streamed(ctx, a, b, out, zero_flags, n); rc != FPGPU_ERR_UNSUPPORTED) return rc;
    size_t bytes = n * ctx->n64 * 8;
    size_t wsb = fpgpu_batch_inv_fast_workspace_bytes(ctx, n);
    DevBuf da, db, dout, dflags, ws;
    CHECK(da.alloc(bytes));
    CHECK(db.alloc(bytes));
    CHECK(dout.alloc(bytes));
    CHECK(dflags.alloc(n));
    CHECK(ws.alloc(wsb));
    CHECK(upload_staged(db.p, b, bytes, nullptr));
    CHECK(fpgpu_batch_inv_dev(ctx, db.p, dout.p, (uint8_t*)dflags.p, n, ws.p, wsb, nullptr));
    CHECK(upload_staged(da.p, a, bytes, nullptr));
    CHECK(fpgpu_mul_dev(ctx, da.p, dout.p, dout.p, n, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    CHECK(download_staged(out, dout.p, bytes, nullptr));
    if (zero_flags) CHECK(download_staged(zero_flags, dflags.p, n, nullptr));
    return FPGPU_OK;
}

int fpgpu_matvec(const fpgpu_ctx* ctx, const uint64_t* A, size_t rows, size_t cols, const uint64_t* x, uint64_t* y) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (rows == 0) return FPGPU_OK;
    if (!y || (cols && (!A || !x))) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    CHECK(require_device());
    size_t esz = ctx->n64 * 8;
    DevBuf dA, dx, dy;
    CHECK(dA.alloc(rows * cols * esz));
    CHECK(dx.alloc(cols * esz));
    CHECK(dy.alloc(rows * esz));
    if (cols) {
        CHECK(upload_staged(dA.p, A, rows * cols * esz, nullptr));
        CHECK(upload_staged(dx.p, x, cols * esz, nullptr));
    }
    DevBuf ws;
    size_t ws_bytes = fpgpu_matvec_workspace_bytes(ctx, rows, cols);
    if (ws_bytes && ws.alloc(ws_bytes) != FPGPU_OK) {   // optional: without it the VALU kernels run
        ws_bytes = 0;
        (void)hipGetLastError();
    }
    CHECK(fpgpu_matvec_dev(ctx, dA.p, rows, cols, dx.p, dy.p, ws_bytes ? ws.p : nullptr, ws_bytes, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    CHECK(download_staged(y, dy.p, rows * esz, nullptr));
    return FPGPU_OK;
}

int fpgpu_spmv(const fpgpu_ctx* ctx, const uint64_t* vals, const uint32_t* col_idx, size_t rows, size_t cols,
               size_t density, const uint64_t* x, uint64_t* y) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (rows == 0) return FPGPU_OK;
    if (!y || (density && (!vals || !col_idx)) || (density && cols && !x)) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    if (density && rows > ((size_t)1 << 40) / density) return fail(FPGPU_ERR_SHAPE, "sparse matrix too large");
    // the reference indexes x[col] and panics out of bounds (src/matrix.rs cells are (usize, T)): reject here
    for (size_t i = 0; i < rows * density; i++)
        if (col_idx[i] >= cols) return fail(FPGPU_ERR_SHAPE, "spmv: column index out of range");
    CHECK(require_device());
    size_t esz = ctx->n64 * 8, cells = rows * density;
    DevBuf dv, dc, dx, dy;
    CHECK(dv.alloc(cells * esz));
    CHECK(dc.alloc(cells * 4));
    CHECK(dx.alloc(cols * esz));
    CHECK(dy.alloc(rows * esz));
    if (cells) {
        CHECK(upload_staged(dv.p, vals, cells * esz, nullptr));
        CHECK(upload_staged(dc.p, col_idx, cells * 4, nullptr));
    }
    if (cols) CHECK(upload_staged(dx.p, x, cols * esz, nullptr));
    CHECK(fpgpu_spmv_dev(ctx, dv.p, (const uint32_t*)dc.p, rows, cols, density, dx.p, dy.p, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    CHECK(download_staged(y, dy.p, rows * esz, nullptr));
    return FPGPU_OK;
}

int fpgpu_matmul(const fpgpu_ctx* ctx, const uint64_t* A, const uint64_t* B, uint64_t* C, size_t m, size_t k,
                 size_t n) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (m == 0 || n == 0) return FPGPU_OK;
    if (!C || (k && (!A || !B))) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    CHECK(require_device());
    size_t esz = ctx->n64 * 8;
    DevBuf dA, dB, dC;
    CHECK(dA.alloc(m * k * esz));
    CHECK(dB.alloc(k * n * esz));
    CHECK(dC.alloc(m * n * esz));
    if (k) {
        CHECK(upload_staged(dA.p, A, m * k * esz, nullptr));
        CHECK(upload_staged(dB.p, B, k * n * esz, nullptr));
    }
    DevBuf ws;
    size_t ws_bytes = fpgpu_matmul_workspace_bytes(ctx, m, k, n);
    if (ws_bytes && ws.alloc(ws_bytes) != FPGPU_OK) {   // the workspace is optional (it can be gigabytes): without it the VALU kernels run
        ws_bytes = 0;
        (void)hipGetLastError();
    }
    CHECK(fpgpu_matmul_dev(ctx, dA.p, dB.p, dC.p, m, k, n, ws_bytes ? ws.p : nullptr, ws_bytes, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    CHECK(download_staged(C, dC.p, m * n * esz, nullptr));
    return FPGPU_OK;
}

int fpgpu_tuning_set(const char* name, long value) {
    if (!name) return fail(FPGPU_ERR_SHAPE, "null pointer: name");
    Tuning& t = tuning_mutable();
    for (const TuningEntry& e : kTuningEntries)
        if (std::strcmp(name, e.name) == 0) {
            t.*(e.field) = value == FPGPU_TUNING_DEFAULT ? g_tuning_defaults.*(e.field) : value;
            return FPGPU_OK;
        }
    return fail(FPGPU_ERR_UNSUPPORTED, std::string("unknown tuning hook: ") + name);
}

const char* fpgpu_last_error_string(void) { return g_last_error.c_str(); }

int fpgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* fpgpu_version(void) { return "fpgpu 0.1.0 (gfx950)"; }

}  // extern "C"
