// fp_inst.hip -- instantiates every kernel launcher for ONE limb count (-DFPGPU_N=N).
#include "fp_kernels.cuh"
#include "fp_launch.hpp"
#include "fp_matmul_mfma_limbs.cuh"   // (includes fp_matmul_mfma.cuh, whose kernel only N = 2 instantiates)
#include "fp_matvec_mfma.cuh"

#ifndef FPGPU_N
#error "compile with -DFPGPU_N=<number of 32-bit limbs>"
#endif

namespace fpgpu {

template <int N>
static FpParams<N> make_params(const fpgpu_ctx* c) {
    FpParams<N> P;
    for (int i = 0; i < N / 2; i++) {
        P.m[2 * i] = (uint32_t)c->modulus[i];
        P.m[2 * i + 1] = (uint32_t)(c->modulus[i] >> 32);
        P.r2[2 * i] = (uint32_t)c->r2[i];
        P.r2[2 * i + 1] = (uint32_t)(c->r2[i] >> 32);
        P.one[2 * i] = (uint32_t)c->one[i];
        P.one[2 * i + 1] = (uint32_t)(c->one[i] >> 32);
    }
    P.n0inv = (uint32_t)c->mod_neg_inv;
    P.bitlen = c->bitlen;
    return P;
}

template <int N, int OP>
static void launch_ew(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL((k_elementwise<N, OP>), dim3(grid_for(n)), dim3(kBlock), 0, s, (const uint32_t*)a,
                       (const uint32_t*)b, (uint32_t*)out, n, make_params<N>(ctx));
}

template <int N>
hipError_t Launch<N>::elementwise(int op, const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n,
                            hipStream_t s) {
    switch (op) {
        case OP_MUL: launch_ew<N, OP_MUL>(ctx, a, b, out, n, s); break;
        case OP_SQR: launch_ew<N, OP_SQR>(ctx, a, b, out, n, s); break;
        case OP_ADD: launch_ew<N, OP_ADD>(ctx, a, b, out, n, s); break;
        case OP_SUB: launch_ew<N, OP_SUB>(ctx, a, b, out, n, s); break;
        case OP_NEG: launch_ew<N, OP_NEG>(ctx, a, b, out, n, s); break;
        case OP_TO_MONT: launch_ew<N, OP_TO_MONT>(ctx, a, b, out, n, s); break;
        case OP_FROM_MONT: launch_ew<N, OP_FROM_MONT>(ctx, a, b, out, n, s); break;
        case OP_DOUBLE: launch_ew<N, OP_DOUBLE>(ctx, a, b, out, n, s); break;
        case OP_HALVE: launch_ew<N, OP_HALVE>(ctx, a, b, out, n, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::mul_chain(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, uint32_t iters,
                          hipStream_t s) {
    hipLaunchKernelGGL((k_mul_chain<N>), dim3(grid_for(n)), dim3(kBlock), 0, s, (const uint32_t*)a,
                       (const uint32_t*)b, (uint32_t*)out, n, iters, make_params<N>(ctx));
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::batch_inv(const fpgpu_ctx* ctx, const void* x, void* out, uint8_t* flags, size_t n, char* ws,
                          size_t ws_bytes, hipStream_t s) {
    const InvPlan p = make_inv_plan(ctx->n64, n);
    const FpParams<N> P = make_params<N>(ctx);
    auto blocks_for = [&](int l) { return (int)((p.count[l + 1] + kBlock - 1) / kBlock); };  // one lane per chain
    auto X = [&](int l) -> const uint32_t* { return l == 0 ? (const uint32_t*)x : (const uint32_t*)(ws + p.x_off[l]); };
    auto O = [&](int l) -> uint32_t* { return l == 0 ? (uint32_t*)out : (uint32_t*)(ws + p.o_off[l]); };
    uint8_t* poisoned = (uint8_t*)(ws + p.poison_off);
    uint32_t* any_poisoned = (uint32_t*)(ws + p.any_off);
    // level-0 prefixes as limbs in their own array when the caller's workspace has room for it
    const bool lp = p.bytes_lp != 0 && ws_bytes >= p.bytes_lp && tuning().inv_lp != 0;
    uint32_t* praw = lp ? (uint32_t*)(ws + p.lp_off) : nullptr;
    // sparse level-0 prefixes where memory binds the sweeps: up to 256 bits, batches large enough for a second level
    const bool sp = !lp && (tuning().inv_sp >= 0 ? tuning().inv_sp == 1 : (N <= 8 && p.levels >= 2));
    if (hipError_t e = hipMemsetAsync(any_poisoned, 0, sizeof(uint32_t), s); e != hipSuccess) return e;
    // the last stored level and the exact inversions share one kernel when that level's chains are short
    const bool fused = p.fused_top && tuning().inv_fuse_top != 0;
    const int sweeps = fused ? p.levels - 1 : p.levels;
    // up-sweeps
    for (int l = 0; l < sweeps; l++) {
        int blocks = blocks_for(l);
        uint32_t* totals = (uint32_t*)(ws + p.x_off[l + 1]);
        if (l == 0 && lp)
            hipLaunchKernelGGL((k_binv_up<N, true, true>), dim3(blocks), dim3(kBlock), 0, s, X(0), O(0), praw, totals,
                               flags, p.count[0], p.c[0], P);
        else if (l == 0 && sp)
            hipLaunchKernelGGL((k_binv_up_sp<N>), dim3(blocks), dim3(kBlock), 0, s, X(0), O(0), totals, flags, p.count[0],
                               p.c[0], P);
        else if (l == 0)
            hipLaunchKernelGGL((k_binv_up<N, true, false>), dim3(blocks), dim3(kBlock), 0, s, X(0), O(0),
                               (uint32_t*)nullptr, totals, flags, p.count[0], p.c[0], P);
        else
            hipLaunchKernelGGL((k_binv_up<N, false, false>), dim3(blocks), dim3(kBlock), 0, s, X(l), O(l),
                               (uint32_t*)nullptr, totals, (uint8_t*)nullptr, p.count[l], p.c[l], P);
    }
    if (fused) {
        // top: chains over the last level's elements, exact inverse of each chain total, walk back: O(l) = inverses of X(l)
        const int l = p.levels - 1;
        const size_t chains = p.count[l + 1];
        const int blocks = (int)((chains + kTopChainBlock - 1) / kTopChainBlock);
        hipLaunchKernelGGL((k_binv_top_chain<N>), dim3(blocks), dim3(kTopChainBlock), 0, s, X(l), O(l), poisoned,
                           any_poisoned, p.count[l], p.c[l], P);
    } else {
        // top: exact inverses of the remaining totals, in place
        size_t c = p.count[p.levels];
        int blocks = (int)((c + kBlock - 1) / kBlock);
        hipLaunchKernelGGL((k_binv_top<N>), dim3(blocks), dim3(kBlock), 0, s, (uint32_t*)(ws + p.x_off[p.levels]),
                           poisoned, any_poisoned, c, P);
    }
    // down-sweeps
    for (int l = sweeps - 1; l >= 0; l--) {
        int blocks = blocks_for(l);
        const uint32_t* inv_totals =
            (l + 1 == p.levels) ? (const uint32_t*)(ws + p.x_off[l + 1]) : (const uint32_t*)O(l + 1);
        if (l == 0 && lp)
            hipLaunchKernelGGL((k_binv_down<N, true, true>), dim3(blocks), dim3(kBlock), 0, s, X(0), O(0),
                               (const uint32_t*)praw, inv_totals, p.count[0], p.c[0], P);
        else if (l == 0 && sp)
            hipLaunchKernelGGL((k_binv_down_sp<N>), dim3(blocks), dim3(kBlock), 0, s, X(0), O(0), inv_totals, p.count[0],
                               p.c[0], P);
        else if (l == 0)
            hipLaunchKernelGGL((k_binv_down<N, true, false>), dim3(blocks), dim3(kBlock), 0, s, X(0), O(0),
                               (const uint32_t*)nullptr, inv_totals, p.count[0], p.c[0], P);
        else
            hipLaunchKernelGGL((k_binv_down<N, false, false>), dim3(blocks), dim3(kBlock), 0, s, X(l), O(l),
                               (const uint32_t*)nullptr, inv_totals, p.count[l], p.c[l], P);
    }
    // composite-modulus safety net (a no-op unless a top-level total was not invertible)
    InvChains chains{};
    for (int l = 0; l < p.levels; l++) chains.m[l] = p.count[l + 1];
    // (grid-stride over at most 2048 workgroups: when nothing is poisoned every wave exits at once, and a
    // million empty workgroups would still cost 17 us of dispatch)
    const int fb_blocks = grid_for(n) < 2048 ? grid_for(n) : 2048;
    hipLaunchKernelGGL((k_binv_fallback<N>), dim3(fb_blocks), dim3(kBlock), 0, s, (const uint32_t*)x,
                       (uint32_t*)out, flags, poisoned, any_poisoned, n, p.levels, chains, P);
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::matvec(const fpgpu_ctx* ctx, const void* A, size_t rows, size_t cols, const void* x, void* y,
                       void* workspace, size_t workspace_bytes, hipStream_t s) {
    // Matrix-core path (fp_matvec_mfma.cuh) when the caller brings the workspace; the tuning hook
    // matvec_mfma = 0 forces the VALU kernel (A/B checks and timing).  The VALU kernel still runs
    // after it, gated by the digitiser's flag (x not representable in signed digits).
    const Tuning& tune = tuning();
    const uint32_t* gate = nullptr;
    {
        MvPlan pl = make_mv_plan(ctx->n64, rows, cols);
        const long ksp = tune.matvec_ksplit;  // tuning / test hook: column parts per row block
        if (ksp > 0 && pl.usable) pl.ksplit = (uint32_t)ksp < pl.chunks_total ? (uint32_t)ksp : pl.chunks_total;
        const bool cores = pl.usable && workspace && workspace_bytes >= pl.bytes && tune.matvec_mfma != 0 && (uintptr_t)A % 16 == 0;
        // Many short rows: one row per lane (k_matvec_rows: no cross-lane sum, the per-row reduction once per lane).
        // Measured on 4 GiB matrices (tools/time_matvec_tall.py, TB/s, row per lane / matrix cores / row per wave):
        // BN254 16 columns 3.59 / 0.81 / 0.19, 64: 3.93 / 2.53 / 0.73, 128: 3.87 / 3.33 / 1.32, 256: 4.06 / 4.02 / 2.11,
        // 512: 4.25 / 4.27 / 2.98; BLS12-381 64: 2.92 / 2.46 / 0.46, 128: 2.92 / 3.12 / 0.80, 512: 3.02 / 3.91 / 2.24;
        // Goldilocks 64: 2.77 / 2.30 / 0.72, 128: 2.42 / 3.24 / 1.35, 256: 2.30 / 4.01 / 2.31.
        const size_t widest = cores ? (N == 8 ? 128 : 64) : (N >= 8 ? 512 : 256);
        if (tune.matvec_rows >= 0 ? tune.matvec_rows == 1 : (cols <= widest && rows >= kMvRowsMinRows)) {
            hipLaunchKernelGGL((k_matvec_rows<N>), dim3(grid_for(rows)), dim3(kBlock), 0, s, (const uint32_t*)A, (const uint32_t*)x,
                               (uint32_t*)y, rows, cols, make_params<N>(ctx), cols, (size_t)1, 1u, cols);
            return hipSuccess;
        }
        if (cores) {
            constexpr int NT = N / 2;
            uint8_t* ws = (uint8_t*)workspace;
            uint32_t* flag = (uint32_t*)ws;
            unsigned long long* colsum = (unsigned long long*)(ws + pl.colsum_off);
            unsigned long long* Dbuf = (unsigned long long*)(ws + pl.d_off);
            uint8_t* Bw = ws + pl.b_off;
            if (hipError_t e = hipMemsetAsync(ws, 0, pl.b_off, s); e != hipSuccess) return e;
            const size_t ksteps = (size_t)pl.chunks_total * mv_chunk_ksteps(NT);
            const size_t ebatches = (ksteps + kMV_EXPAND_BATCH - 1) / kMV_EXPAND_BATCH;
            const unsigned eg = (unsigned)(ebatches < 512 ? ebatches : 512);
            hipLaunchKernelGGL((k_mv_expand<N, NT>), dim3(eg), dim3(NT * 64), 0, s, (const uint32_t*)x, cols, ksteps, Bw,
                               colsum, flag, make_params<N>(ctx));
            hipLaunchKernelGGL((k_matvec_mfma<N, NT>), dim3(pl.rowblocks * pl.ksplit), dim3(kMV_WAVES * 64), 0, s,
                               (const uint8_t*)A, rows, cols * (size_t)(4 * N), Bw, pl.chunks_total, pl.ksplit, Dbuf,
                               (const uint32_t*)flag);
            hipLaunchKernelGGL((k_mv_finish<N, NT>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, Dbuf, colsum,
                               (uint32_t*)y, rows, (const uint32_t*)flag, make_params<N>(ctx));
            gate = flag;
        }
    }
    // Terms per product-scanning pass: T products may be summed before the wide add only
    // if T*p <= R, i.e. the modulus leaves log2(T) zero bits at the top of its container.
    // Measured on MI355X (2^15 x 2^15, BN254): T = 1 (66 VGPRs, 7 waves/SIMD) 6.96 ms,
    // T = 2 7.3 ms, T = 4 (125 VGPRs, 4 waves) 7.9 ms -- occupancy beats the saved column
    // shifts, so T = 1 is the default; the tuning hook matvec_t selects 2 or 4 for experiments.
    const int t_env = (int)tune.matvec_t;
    int lz = __builtin_clzll(ctx->modulus[ctx->n64 - 1] | 1);
    if (ctx->modulus[ctx->n64 - 1] == 0) lz = 64;
    int Tmax = lz >= 2 ? 4 : (lz >= 1 ? 2 : 1);
    if (N > 8 && Tmax > 2) Tmax = 2;  // register budget (T = 4 at N = 12 needs 182 VGPRs)
    int T = 1;
    if (t_env > 1) T = t_env < Tmax ? t_env : Tmax;
    // (as the gated fallback of the matrix-core path the kernel normally exits at once: a small
    // grid then keeps the empty dispatch at a few microseconds; rows are walked grid-stride)
    const size_t grid_cap = gate ? 2048 : 65536;
    int grid = (int)(rows < grid_cap ? rows : grid_cap);
    const uint32_t* Ap = (const uint32_t*)A;
    const uint32_t* xp = (const uint32_t*)x;
    uint32_t* yp = (uint32_t*)y;
    // Which VALU kernel: k_matvec_cols (unsaturated column sums, one row per wave) from 192 to 384 bits, where it is
    // ahead (8 GiB matrices, TB/s of matrix bytes, k_matvec / k_matvec_cols: 192 bits 4.88 / 4.95, BN254 4.80 / 5.33,
    // 320 bits 4.20 / 4.14, BLS12-381 3.78 / 4.09; 448 bits 3.44 / 3.19 and 512 bits 3.16 / 3.12 with 256 registers
    // and one wave per SIMD; one and two words 4.69 / 4.66 and 5.37 / 5.39).  FPGPU_MATVEC_COLS forces either.
    const bool cols_kernel = tune.matvec_cols >= 0 ? tune.matvec_cols == 1 : (N >= 6 && N <= 12);
    // few long rows (not the matrix cores' shapes) with a workspace: column parts, then a sum per row -- at every width
    uint32_t parts = 1;
    if (!gate && workspace && tune.matvec_cols != 0) {
        parts = mv_split_parts(rows, cols);
        if (parts > 1 && workspace_bytes < mv_split_bytes(ctx->n64, rows, cols)) parts = 1;
    }
    if ((cols_kernel || parts > 1) && t_env <= 1) {
        const size_t per_part = parts > 1 ? ((cols + parts - 1) / parts + 63) / 64 * 64 : cols;
        const size_t items = rows * parts, wgs = (items + kBlock / 64 - 1) / (kBlock / 64);
        const int g2 = (int)(wgs < grid_cap ? (wgs ? wgs : 1) : grid_cap);
        hipLaunchKernelGGL((k_matvec_cols<N>), dim3(g2), dim3(kBlock), 0, s, Ap, xp, parts > 1 ? (uint32_t*)workspace : yp, rows, cols,
                           gate, make_params<N>(ctx), parts, per_part);
        if (parts > 1) {
            const size_t cw = (rows + kBlock / 64 - 1) / (kBlock / 64);
            hipLaunchKernelGGL((k_matvec_combine<N>), dim3((unsigned)(cw ? cw : 1)), dim3(kBlock), 0, s, (const uint32_t*)workspace, yp,
                               rows, parts, make_params<N>(ctx));
        }
        return hipSuccess;
    }
    if (T >= 4)
        hipLaunchKernelGGL((k_matvec<N, 4>), dim3(grid), dim3(kBlock), 0, s, Ap, xp, yp, rows, cols, gate, make_params<N>(ctx));
    else if (T >= 2)
        hipLaunchKernelGGL((k_matvec<N, 2>), dim3(grid), dim3(kBlock), 0, s, Ap, xp, yp, rows, cols, gate, make_params<N>(ctx));
    else
        hipLaunchKernelGGL((k_matvec<N, 1>), dim3(grid), dim3(kBlock), 0, s, Ap, xp, yp, rows, cols, gate, make_params<N>(ctx));
    return hipSuccess;
}

// The limb kernel's launches (fp_matmul_mfma_limbs.cuh) over a workspace laid out by make_mfl_plan: the flag word first (it
// stays the caller's gate for the VALU fallback), the digitisers of A and B^T (one limb per blockIdx.y), the product, and
// the sum of the k parts where the plan split k over work items (it sums whatever the scratch holds when the flag is up:
// the gated VALU kernel then overwrites C).
template <int N>
static hipError_t launch_limb_matmul(const fpgpu_ctx* ctx, const MflPlan& pl, const void* A, const void* B, void* C, size_t m,
                                     size_t k, size_t n, void* workspace, hipStream_t s) {
    uint8_t* ws = (uint8_t*)workspace;
    uint32_t* flag = (uint32_t*)ws;
    if (hipError_t e = hipMemsetAsync(flag, 0, 4, s); e != hipSuccess) return e;
    const unsigned wa = pl.rblocks * pl.ksteps, wb = pl.cblocks * pl.ksteps;  // x4 waves = one workgroup each
    hipLaunchKernelGGL((k_mfl_digitize<N, false>), dim3(wa, N / 2), dim3(256), 0, s, (const uint32_t*)A, m, k, k, ws + pl.a_off,
                       pl.plane_a, pl.rblocks, pl.ksteps, flag, make_params<N>(ctx));
    hipLaunchKernelGGL((k_mfl_digitize<N, true>), dim3(wb, N / 2), dim3(256), 0, s, (const uint32_t*)B, n, k, n, ws + pl.b_off,
                       pl.plane_b, pl.cblocks, pl.ksteps, flag, make_params<N>(ctx));
    uint32_t* dst = pl.zparts > 1 ? (uint32_t*)(ws + pl.p_off) : (uint32_t*)C;
    hipLaunchKernelGGL((k_matmul_mfma_limbs<N>), dim3(pl.grid), dim3(256), 0, s, ws + pl.a_off, ws + pl.b_off, pl.plane_a,
                       pl.plane_b, dst, m, n, pl.rblocks, pl.cblocks, pl.ksteps, pl.zparts, pl.zsteps,
                       (uint32_t*)(ws + pl.t_off), (const uint32_t*)flag, make_params<N>(ctx));
    if (pl.zparts > 1) {
        const size_t outs = m * n, cw = (outs + kBlock / 64 - 1) / (kBlock / 64);
        hipLaunchKernelGGL((k_matvec_combine<N>), dim3((unsigned)cw), dim3(kBlock), 0, s, (const uint32_t*)dst, (uint32_t*)C, outs,
                           pl.zparts, make_params<N>(ctx));
    }
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::matmul(const fpgpu_ctx* ctx, const void* A, const void* B, void* C, size_t m, size_t k, size_t n,
                       void* workspace, size_t workspace_bytes, hipStream_t s) {
    // a few vectors times a matrix: one lane per output column, B streamed once per vector, as it lies
    if (vm_shape(m, k, n)) {
        uint32_t parts = vm_split_parts(k, n);
        if (parts > 1 && !(workspace && workspace_bytes >= vm_split_bytes(ctx->n64, k, n))) parts = 1;
        if (parts > 1 || n >= 16384) {
            const size_t per = parts > 1 ? (k + parts - 1) / parts : k;
            for (size_t i = 0; i < m; i++) {
                uint32_t* dst = (uint32_t*)C + i * n * N;
                hipLaunchKernelGGL((k_matvec_rows<N>), dim3(grid_for(n * parts)), dim3(kBlock), 0, s, (const uint32_t*)B,
                                   (const uint32_t*)A + i * k * N, parts > 1 ? (uint32_t*)workspace : dst, n, k, make_params<N>(ctx),
                                   (size_t)1, n, parts, per);
                if (parts > 1) {
                    const size_t cw = (n + kBlock / 64 - 1) / (kBlock / 64);
                    hipLaunchKernelGGL((k_matvec_combine<N>), dim3((unsigned)cw), dim3(kBlock), 0, s, (const uint32_t*)workspace, dst, n,
                                       parts, make_params<N>(ctx));
                }
            }
            return hipSuccess;
        }
    }
    // a small output over a long k, with the scratch for it: k split over workgroups (every width, the multi-limb kernel)
    if (const uint32_t parts = mm_split_parts(m, k, n);
        parts > 1 && workspace && workspace_bytes >= mm_split_bytes(ctx->n64, m, k, n) ) {
        const size_t kper = ((k + parts - 1) / parts + kMMTile - 1) / kMMTile * kMMTile;
        const size_t gx = (n + kMMTile - 1) / kMMTile, gy = (m + kMMTile - 1) / kMMTile;
        hipLaunchKernelGGL((k_matmul<N>), dim3((unsigned)gx, (unsigned)gy, parts), dim3(kMMTile * kMMTile), 0, s,
                           (const uint32_t*)A, (const uint32_t*)B, (uint32_t*)workspace, m, k, n, make_params<N>(ctx), parts, kper,
                           (const uint32_t*)nullptr);
        const size_t outs = m * n, cw = (outs + kBlock / 64 - 1) / (kBlock / 64);
        hipLaunchKernelGGL((k_matvec_combine<N>), dim3((unsigned)cw), dim3(kBlock), 0, s, (const uint32_t*)workspace, (uint32_t*)C,
                           outs, parts, make_params<N>(ctx));
        return hipSuccess;
    }
#if FPGPU_N == 2
    {
        // single-limb paths; the tuning hooks matmul_generic = 1 force the generic kernel,
        // matmul_mfma = 0 the VALU kernel (A/B checks and timing)
        const Tuning& tune = tuning();
        if (tune.matmul_generic <= 0) {
            const size_t gx = (n + kMM1_BN - 1) / kMM1_BN, gy = (m + kMM1_BM - 1) / kMM1_BM;
            const MfmaPlan pl = make_mfma_plan(ctx->n64, m, k, n);
            // k beyond the int32 accumulators of k_matmul_mfma_1limb, or too few tiles for the chip: the limb kernel with one
            // limb (k in parts of 255 k-steps, k split over work items where the tiles are few), the VALU kernel behind its flag
            MflPlan ml = make_mfl_plan(ctx->n64, m, k, n, pl.usable || tune.matmul_mfma > 0);
            if (pl.usable && !one_limb_takes_limb_kernel(pl, ml)) ml = MflPlan{};
            const bool limbs = ml.usable && workspace && workspace_bytes >= ml.bytes && tune.matmul_mfma != 0;
            const bool mfma = !limbs && pl.usable && workspace && workspace_bytes >= pl.bytes && tune.matmul_mfma != 0;
            if (!mfma) {
                const uint32_t* gate1 = nullptr;
                if (limbs) {
                    if (hipError_t e = launch_limb_matmul<N>(ctx, ml, A, B, C, m, k, n, workspace, s); e != hipSuccess) return e;
                    gate1 = (const uint32_t*)workspace;
                }
                hipLaunchKernelGGL((k_matmul_1limb<N>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s,
                                   (const uint2*)A, (const uint2*)B, (uint2*)C, m, k, n, gate1, make_params<N>(ctx));
                return hipSuccess;
            }
            uint8_t* ws = (uint8_t*)workspace;
            uint32_t* flag = (uint32_t*)ws;
            if (hipError_t e = hipMemsetAsync(flag, 0, 4, s); e != hipSuccess) return e;
            const unsigned wa = pl.rblocks * pl.ksteps, wb = pl.cblocks * pl.ksteps;  // x4 waves = one workgroup each
            hipLaunchKernelGGL((k_mf_digitize<false>), dim3(wa), dim3(256), 0, s, (const uint64_t*)A, m, k, k,
                               ws + pl.a_off, pl.rblocks, pl.ksteps, ctx->modulus[0], flag);
            hipLaunchKernelGGL((k_mf_digitize<true>), dim3(wb), dim3(256), 0, s, (const uint64_t*)B, n, k, n,
                               ws + pl.b_off, pl.cblocks, pl.ksteps, ctx->modulus[0], flag);
            const unsigned tiles = pl.rblocks * pl.cblocks, grid = (tiles + 7) / 8 * 8;
            hipLaunchKernelGGL((k_matmul_mfma_1limb<N>), dim3(grid), dim3(256), 0, s, ws + pl.a_off, ws + pl.b_off,
                               (uint2*)C, m, n, pl.rblocks, pl.cblocks, pl.ksteps, (const uint32_t*)flag,
                               make_params<N>(ctx));
            hipLaunchKernelGGL((k_matmul_1limb<N>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, s, (const uint2*)A,
                               (const uint2*)B, (uint2*)C, m, k, n, (const uint32_t*)flag, make_params<N>(ctx));
            return hipSuccess;
        }
    }
#endif
    size_t gx = (n + kMMTile - 1) / kMMTile, gy = (m + kMMTile - 1) / kMMTile;
    const uint32_t* gate = nullptr;
#if FPGPU_N >= 4
    {
        // multi-limb product on the matrix cores (fp_matmul_mfma_limbs.cuh) where the plan's cost model prefers it and
        // the workspace holds its digit planes and scratch; the hook matmul_mfma = 0 keeps the VALU kernel, 1 takes the
        // matrix cores for every shape the plan can lay out (parity tests on small shapes)
        const Tuning& tune = tuning();
        const MflPlan pl = make_mfl_plan(ctx->n64, m, k, n, tune.matmul_mfma > 0);
        if (pl.usable && workspace && workspace_bytes >= pl.bytes && tune.matmul_mfma != 0 && tune.matmul_generic <= 0) {
            if (hipError_t e = launch_limb_matmul<N>(ctx, pl, A, B, C, m, k, n, workspace, s); e != hipSuccess) return e;
            gate = (const uint32_t*)workspace;   // the flag: the VALU kernel below then runs only if some input had no digit form
        }
    }
#endif
    hipLaunchKernelGGL((k_matmul<N>), dim3((unsigned)gx, (unsigned)gy), dim3(kMMTile * kMMTile), 0, s,
                       (const uint32_t*)A, (const uint32_t*)B, (uint32_t*)C, m, k, n, make_params<N>(ctx), 1u, k, gate);
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::fill_random(const fpgpu_ctx* ctx, void* out, size_t n, uint64_t seed, size_t first,
                                  hipStream_t s) {
    hipLaunchKernelGGL((k_fill_random<N>), dim3(grid_for(n)), dim3(kBlock), 0, s, (uint32_t*)out, n, seed, first,
                       make_params<N>(ctx));
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::spmv(const fpgpu_ctx* ctx, const void* vals, const uint32_t* colidx, size_t rows, size_t cols,
                     uint32_t density, const void* x, void* y, hipStream_t s) {
    // G = 2^log2g lanes share a row, about two cells each.  The kernel is bound by the latency and
    // the line granularity of the gather of x (each 32-byte element drags a 128-byte line through
    // L2 / Infinity Cache), not by arithmetic: more lanes per row = more gathers in flight.
    // Measured on MI355X, 2^20 rows x 32 cells, BN254 (G cells/s): G = 1: 19.6, 2: 29.9, 4: 40.6,
    // 8: 45.7, 16: 48.2, 32: 47.6.
    uint32_t log2g = 0;
    while ((2u << log2g) < density && log2g < 6) log2g++;
    if (const long v = tuning().spmv_log2g; v >= 0 && v <= 6) log2g = (uint32_t)v;  // tuning / tests
    const size_t lanes = rows << log2g;
    size_t blocks = (lanes + kBlock - 1) / kBlock;
    if (blocks > (1u << 20)) blocks = 1u << 20;
    if (blocks < 1) blocks = 1;
    // cells a lane loads together: what it owns, at most 4
    const uint32_t per_lane = (density + (1u << log2g) - 1) >> log2g;
    long u = per_lane >= 4 ? 4 : (per_lane >= 2 ? 2 : 1);
    if (const long v = tuning().spmv_unroll; v == 1 || v == 2 || v == 4) u = v;  // tuning / tests
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, (const uint32_t*)vals, colidx,
                           (const uint32_t*)x, (uint32_t*)y, rows, cols, density, log2g, make_params<N>(ctx));
    };
    if (u == 4) launch(k_spmv<N, 4>);
    else if (u == 2) launch(k_spmv<N, 2>);
    else launch(k_spmv<N, 1>);
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::pow(const fpgpu_ctx* ctx, const void* a, const uint32_t* exp32, int top_bit, void* out, size_t n,
                    hipStream_t s) {
    PowExp e;
    for (int i = 0; i < 16; i++) e.w[i] = exp32[i];
    e.top = top_bit;
    int ones = 0;
    for (int i = 0; i < 16; i++) ones += __builtin_popcount(exp32[i]);
    const long hook = tuning().pow_window;   // tuning / tests: 0 forces the binary ladder, 1 the windowed one
    e.windowed = hook >= 0 ? (hook > 0 && top_bit > 0) : ones > kPowWindowMinOnes;
    hipLaunchKernelGGL((k_pow<N>), dim3(grid_for(n)), dim3(kBlock), 0, s, (const uint32_t*)a, (uint32_t*)out, n, e,
                       make_params<N>(ctx));
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::fold(bool product, const fpgpu_ctx* ctx, const void* v, size_t n, void* out, void* ws,
                     hipStream_t s) {
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > (size_t)kFoldBlocks) blocks = kFoldBlocks;
    if (blocks < 1) blocks = 1;
    const FpParams<N> P = make_params<N>(ctx);
    // pass 1: slices -> partials (straight to `out` when one workgroup covers everything)
    uint32_t* p1 = blocks == 1 ? (uint32_t*)out : (uint32_t*)ws;
    if (product) hipLaunchKernelGGL((k_fold<N, true>), dim3((unsigned)blocks), dim3(kBlock), 0, s, (const uint32_t*)v, n, p1, P);
    else hipLaunchKernelGGL((k_fold<N, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s, (const uint32_t*)v, n, p1, P);
    if (blocks > 1) {
        if (product) hipLaunchKernelGGL((k_fold<N, true>), dim3(1), dim3(kBlock), 0, s, (const uint32_t*)ws, blocks, (uint32_t*)out, P);
        else hipLaunchKernelGGL((k_fold<N, false>), dim3(1), dim3(kBlock), 0, s, (const uint32_t*)ws, blocks, (uint32_t*)out, P);
    }
    return hipSuccess;
}

template <int N>
hipError_t Launch<N>::dot(const fpgpu_ctx* ctx, const void* a, const void* b, size_t n, void* out, void* ws,
                          hipStream_t s) {
    size_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > (size_t)kFoldBlocks) blocks = kFoldBlocks;
    if (blocks < 1) blocks = 1;
    const FpParams<N> P = make_params<N>(ctx);
    // pass 1: one canonical partial per workgroup (straight to `out` when one workgroup covers everything)
    uint32_t* p1 = blocks == 1 ? (uint32_t*)out : (uint32_t*)ws;
    hipLaunchKernelGGL((k_dot<N>), dim3((unsigned)blocks), dim3(kBlock), 0, s, (const uint32_t*)a, (const uint32_t*)b, n,
                       p1, P);
    if (blocks > 1)
        hipLaunchKernelGGL((k_fold<N, false>), dim3(1), dim3(kBlock), 0, s, (const uint32_t*)ws, blocks, (uint32_t*)out, P);
    return hipSuccess;
}

template struct Launch<FPGPU_N>;

}  // namespace fpgpu
