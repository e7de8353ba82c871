// fp_launch.hpp -- per-limb-count kernel launchers.  Each limb count N is compiled in
// its own translation unit (fp_inst.hip with -DFPGPU_N=N) so the build parallelises;
// fpgpu_api.hip only sees these declarations.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/fpgpu.h"

struct fpgpu_ctx {
    uint32_t n64;
    uint64_t modulus[FPGPU_MAX_N64];
    uint64_t one[FPGPU_MAX_N64];
    uint64_t r2[FPGPU_MAX_N64];
    uint64_t mod_neg_inv;
    uint32_t bitlen;    // bit length of the modulus
};

namespace fpgpu {

enum Op : int { OP_MUL = 0, OP_SQR = 1, OP_ADD = 2, OP_SUB = 3, OP_NEG = 4, OP_TO_MONT = 5, OP_FROM_MONT = 6, OP_DOUBLE = 7, OP_HALVE = 8 };

#ifndef FPGPU_BLOCK
#define FPGPU_BLOCK 256
#endif
constexpr int kBlock = FPGPU_BLOCK;   // threads per workgroup of every streaming kernel (4 waves)
constexpr int kMMTile = 16;   // k_matmul output tile edge

// Batch-inversion level plan (shared by the workspace-size query and the launcher).
// Chain length per lane: long chains at a big level (fewer totals to recurse on: 2^24 elements
// cost 1.47 ms with 8 everywhere, 1.38 ms with 32 everywhere), short ones once a level has too few
// chains to fill the chip, where a lane's serial chain is pure latency.
constexpr int kInvCMin = 8, kInvCMax = 32;
constexpr size_t kInvLanesWanted = 65536;   // one wave per SIMD
constexpr size_t kInvTop = 65536; // stop recursing when this few totals remain: one exact inversion per lane is
                                  // pure latency up to one wave per SIMD (2^13 lanes 91 us, 2^16 lanes 103 us on
                                  // 2^24 BLS12-381 elements), a tree level costs 45 us
constexpr size_t kInvOneLevelMax = 84 * kInvTop;   // 5.5 M elements: up to here one level with chains of n / 65536 wins (make_inv_plan)
constexpr size_t kInvStretchMax = 48;              // longest level-0 chain chosen to spare a third level
constexpr int kInvMaxLevels = 16;
constexpr int kInvTopChainMaxC = 8;   // longest chain of the last stored level that rides in the top kernel (k_binv_top_chain's LDS is sized by it)
struct InvPlan {
    int levels = 0;                         // number of up-sweeps
    size_t count[kInvMaxLevels + 1] = {0};  // count[0] = n, count[l+1] = chains at level l
    int c[kInvMaxLevels] = {0};             // chain length at level l
    size_t x_off[kInvMaxLevels + 1] = {0};  // byte offset of X_l (l >= 1) in the workspace
    size_t o_off[kInvMaxLevels + 1] = {0};  // byte offset of O_l (1 <= l < levels)
    size_t poison_off = 0;
    size_t any_off = 0;                      // one word: did any top-level inversion fail
    size_t bytes = 0;
    bool fused_top = false;                  // the last level's sweeps and the exact inversions run as one kernel (k_binv_top_chain)
    size_t lp_off = 0;                       // level-0 prefixes as limbs (k_binv_up<.., LP>): their array sits behind `bytes`
    size_t bytes_lp = 0;                     // workspace size with that array; 0 where limbs are the words (narrow fields)
};
InvPlan make_inv_plan(uint32_t n64, size_t n);
// Workspace of the matrix-core mat-vec (fp_matvec_mfma.cuh): flag, column sums, the 64-bit
// diagonal sums D[rows][16 NT], then the Toeplitz expansion of x in 64-byte k-steps x NT KiB.
#ifndef FPGPU_MV_WAVES
#define FPGPU_MV_WAVES 8
#endif
constexpr int kMV_WAVES = FPGPU_MV_WAVES;  // waves per workgroup of k_matvec_mfma: 16 rows each
// k-steps (64 bytes of every row) per staged chunk of the Toeplitz operand, by n64 (see k_matvec_mfma)
constexpr int mv_chunk_ksteps(uint32_t n64) { return n64 <= 4 ? 8 : 4; }
struct MvPlan {
    bool usable;
    uint32_t nt, rowblocks, ksplit, chunks_total;
    size_t colsum_off, d_off, b_off, bytes;
};
inline MvPlan make_mv_plan(uint32_t n64, size_t rows, size_t cols) {
    MvPlan pl{};
    const size_t eb = 8 * (size_t)n64;
    // the finishing kernel's 64-bit diagonal groups hold cols * EB <= 2^23 exactly
    // and rows must be whole 16-byte pieces (odd n64 with odd cols is left to the VALU kernel)
    pl.usable = rows > 0 && cols > 0 && cols * eb <= ((size_t)1 << 23) && rows <= ((size_t)1 << 24) &&
                (cols * eb) % 16 == 0 && cols * eb >= 64;
    if (!pl.usable) return pl;
    pl.nt = n64;  // ceil((2 EB - 1) / 16)
    pl.rowblocks = (uint32_t)((rows + kMV_WAVES * 16 - 1) / (kMV_WAVES * 16));
    const size_t ksteps = (cols * eb + 63) / 64;
    const size_t kc = (size_t)mv_chunk_ksteps(n64);
    pl.chunks_total = (uint32_t)((ksteps + kc - 1) / kc);
    // Column parts per row block.  A CU streams this access pattern (64 bytes of each of 16 rows per wave instruction) at
    // a bounded rate whatever it holds -- tools/ubench_rowtile_stride.hip: 128 workgroups 4.1 TB/s, 340 4.9, 256 or 510
    // 6.0-6.2 -- so the launch is as fast as its fullest CU: cost(k) = ceil(row blocks * k / 256) / k.  Take the SMALLEST
    // split within 5 % of the best cost among splits that keep a part at 8 chunks or more and the grid near 3072
    // workgroups (every part adds a prologue and a round of atomics per row; tools/time_matvec_ksplit.py: 170 row
    // blocks 3.5 TB/s with 1 or 2 parts, 4.9 with 3; 171: 3.5 with 3, 4.5 with 4 ... 16; 32: 5.3 with 8).
    uint32_t ks = 1;
    {
        uint32_t most = pl.chunks_total / 8;
        if (most > 256) most = 256;
        if (const uint32_t by_grid = 3072 / pl.rowblocks; most > by_grid) most = by_grid;
        if (most < 1) most = 1;
        auto cost = [&](uint32_t k) { return (double)(((uint64_t)pl.rowblocks * k + 255) / 256) / k; };
        double best = cost(1);
        for (uint32_t k = 2; k <= most; k++) best = cost(k) < best ? cost(k) : best;
        while (ks < most && cost(ks) > 1.05 * best) ks++;
    }
    pl.ksplit = ks;
    pl.colsum_off = 256;
    pl.d_off = pl.colsum_off + ((16 * pl.nt * 8 + 255) / 256) * 256;
    pl.b_off = pl.d_off + (((size_t)pl.rowblocks * (kMV_WAVES * 16) * 16 * pl.nt * 8 + 1023) / 1024) * 1024;
    pl.bytes = pl.b_off + (size_t)pl.chunks_total * kc * pl.nt * 1024;
    return pl;
}
// VALU mat-vec with few, long rows (shapes the matrix-core plan does not take: more than 2^23 bytes per row): one row per
// wave would leave the chip empty (128 rows x 2^20 BN254 columns: 0.5 TB/s; one row: one wave), so the columns are split
// over `parts` waves per row and the canonical partial residues go through the workspace (k_matvec_cols, k_matvec_combine).
constexpr size_t kMvRowsMinRows = 65536;   // k_matvec_rows (one row per lane) takes matrices at least this tall: a wave per SIMD
inline uint32_t mv_split_parts(size_t rows, size_t cols) {
    constexpr size_t kWavesWanted = 8192;   // two per SIMD
    if (rows == 0 || rows >= kWavesWanted / 2 || cols < 1024) return 1;
    size_t parts = (kWavesWanted + rows - 1) / rows;
    const size_t most = cols / 512;         // a part is at least 8 elements per lane
    if (parts > most) parts = most;
    return parts < 2 ? 1 : (uint32_t)parts;
}
inline size_t mv_split_bytes(uint32_t n64, size_t rows, size_t cols) {
    const uint32_t parts = mv_split_parts(rows, cols);
    return parts < 2 ? 0 : ((rows * parts * 8 * (size_t)n64 + 255) / 256) * 256;
}
// Mat-mul with a small output over a long k: the 16 x 16 output tiles are too few to fill the chip (16 x 2^20 x 16: ONE
// workgroup), so k is split over `parts` workgroups per tile and the canonical partial residues go through the workspace
// (k_matmul with blockIdx.z, k_matvec_combine).  Every field width takes this path for such shapes.
inline uint32_t mm_split_parts(size_t m, size_t k, size_t n) {
    const size_t tiles = ((m + 15) / 16) * ((n + 15) / 16);
    if (tiles == 0 || tiles > 128 || k < 4096) return 1;
    size_t parts = (1024 + tiles - 1) / tiles;   // ~four workgroups per CU
    const size_t most = k / 512;
    if (parts > most) parts = most;
    if (parts > 65535) parts = 65535;
    return parts < 2 ? 1 : (uint32_t)parts;
}
inline size_t mm_split_bytes(uint32_t n64, size_t m, size_t k, size_t n) {
    const uint32_t parts = mm_split_parts(m, k, n);
    return parts < 2 ? 0 : ((m * n * parts * 8 * (size_t)n64 + 255) / 256) * 256;
}
// A few vectors times a matrix (m <= 4 rows of A against B k x n: random linear combinations of k rows): one lane per
// output column (k_matvec_rows with B's strides); when the columns are too few for that, the k terms are split over
// `parts` lanes per column through the workspace.
constexpr size_t kVmMaxRows = 4, kVmMinCols = 1024;
inline uint32_t vm_split_parts(size_t k, size_t n) {
    constexpr size_t kLanesWanted = 262144;   // four waves per SIMD
    if (n == 0 || n >= kLanesWanted || k < 128) return 1;
    size_t parts = (kLanesWanted + n - 1) / n;
    const size_t most = k / 64;
    if (parts > most) parts = most;
    return parts < 2 ? 1 : (uint32_t)parts;
}
inline size_t vm_split_bytes(uint32_t n64, size_t k, size_t n) {
    const uint32_t parts = vm_split_parts(k, n);
    return parts < 2 ? 0 : ((n * parts * 8 * (size_t)n64 + 255) / 256) * 256;
}
inline bool vm_shape(size_t m, size_t k, size_t n) { return m >= 1 && m <= kVmMaxRows && n >= kVmMinCols && k >= 1; }
// Workspace of the matrix-core mat-mul (fp_matmul_mfma.cuh): a flag word, then the int8 digit
// planes of A and of B^T in 64-row blocks x 64-deep k-steps x 32 KiB.
struct MfmaPlan {
    bool usable;  // n64 == 1 shapes the int32 accumulators hold exactly
    uint32_t rblocks, cblocks, ksteps;
    size_t a_off, b_off, bytes;
};
inline MfmaPlan make_mfma_plan(uint32_t n64, size_t m, size_t k, size_t n) {
    MfmaPlan pl{};
    pl.usable = n64 == 1 && m > 0 && n > 0 && k > 0 && k <= 16320 && (m + 63) / 64 * ((n + 63) / 64) < (1u << 28);
    if (!pl.usable) return pl;
    pl.rblocks = (uint32_t)((m + 63) / 64);
    pl.cblocks = (uint32_t)((n + 63) / 64);
    pl.ksteps = (uint32_t)((k + 63) / 64);
    pl.a_off = 256;
    pl.b_off = pl.a_off + (size_t)pl.rblocks * pl.ksteps * 32768;
    pl.bytes = pl.b_off + (size_t)pl.cblocks * pl.ksteps * 32768;
    return pl;
}
// Workspace of the multi-limb matrix-core mat-mul (fp_matmul_mfma_limbs.cuh): a flag word, the digit planes of every limb of
// A and of B^T (each limb a plane set in the single-limb layout), and (2 n64 - 1) x 6 words x 4096 outputs of scratch per
// workgroup of a persistent grid.  `usable` also says that the matrix cores are expected to WIN: the VALU kernel runs
// n64^2 x 0.21 ps per term on the whole chip (296 G terms/s at 256 bits), a workgroup here n64^2 x 3.0 us per 64-deep
// k-step of a 64 x 64 tile on ONE CU, so the tiles must come near the CUs' number and k must amortise the launches.
struct MflPlan {
    bool usable;
    uint32_t rblocks, cblocks, ksteps, grid, zparts, zsteps;
    size_t plane_a, plane_b, a_off, b_off, t_off, p_off, bytes;
};
inline MflPlan make_mfl_plan(uint32_t n64, size_t m, size_t k, size_t n, bool forced = false) {
    MflPlan pl{};
    if (n64 < 1 || n64 > 8 || m == 0 || n == 0 || k == 0) return pl;   // (n64 == 1: only asked for k beyond make_mfma_plan's 16320)
    const size_t rb = (m + 63) / 64, cb = (n + 63) / 64, ks = (k + 63) / 64;
    if (rb * cb >= (1u << 28) || ks >= (1u << 24) || rb * ks >= (1u << 31) || cb * ks >= (1u << 31)) return pl;   // (grids of the digitisers)
    // too few tiles for the chip's 256 CUs: k is split over z parts per tile, each a work item of its own, and the parts'
    // canonical residues are added per output (k_matvec_combine) -- 512 x 4096 x 512 BN254: 64 tiles, 2.7 ms on 64 CUs
    const size_t tiles = rb * cb;
    size_t z = tiles >= 192 ? 1 : 256 / tiles;
    if (z > ks) z = ks;
    const size_t zs = (ks + z - 1) / z;
    z = (ks + zs - 1) / zs;
    if (!forced) {
        const double nl2 = (double)n64 * n64;
        // the VALU kernel: the chip's rate, but never less than ONE workgroup's walk over k (a 16 x 16 tile, a lane per
        // output: 0.25 us x n64^2 / 16 per term -- 128 x 2048 x 128 BN254 takes 0.63 ms, not the 0.11 the rate alone gives)
        const double valu_rate_us = (double)m * n * k * nl2 * 0.21e-6, valu_walk_us = (double)k * nl2 * 0.0156;
        const double valu_us = valu_rate_us > valu_walk_us ? valu_rate_us : valu_walk_us;
        const double rounds = (double)((tiles * z + 255) / 256);
        // measured (tools/exp_matmul_limbs.py, 256 ... 4096 on a side, 128 to 512 bits): 3.0 us per pair and k-step of a tile
        // with the epilogues and pipeline restarts of a limb diagonal in it, ~5 us x n64^2 per work item for its diagonals'
        // epilogues and its 16 reductions per lane, 1.65 ps x n64^2 per digitised element, 40 us of launches
        const double mfma_us = rounds * (zs * nl2 * 3.0 + nl2 * 5.0) + (double)(m + n) * k * nl2 * 1.65e-6 + 40.0 + (z > 1 ? 10.0 : 0.0);
        if (mfma_us >= valu_us) return pl;
    }
    pl.usable = true;
    pl.rblocks = (uint32_t)rb;
    pl.cblocks = (uint32_t)cb;
    pl.ksteps = (uint32_t)ks;
    pl.zparts = (uint32_t)z;
    pl.zsteps = (uint32_t)zs;
    const size_t items8 = (tiles + 7) / 8 * 8 * z;
    pl.grid = (uint32_t)(items8 < 256 ? items8 : 256);
    pl.plane_a = rb * ks * 32768;
    pl.plane_b = cb * ks * 32768;
    pl.a_off = 256;
    pl.b_off = pl.a_off + n64 * pl.plane_a;
    pl.t_off = pl.b_off + n64 * pl.plane_b;
    pl.p_off = pl.t_off + (size_t)pl.grid * (2 * n64 - 1) * 6 * 4096 * 4;
    pl.bytes = pl.p_off + (z > 1 ? ((m * n * z * 8 * n64 + 255) / 256) * 256 : 0);
    return pl;
}
// Single-limb products k_matmul_mfma_1limb can take (k <= 16320) but whose 64 x 64 tiles are too few for the chip: the limb
// kernel with one limb splits k over work items (256 x 8192 x 256 Goldilocks: 16 tiles, 128 k-steps each on 16 CUs).  Taken
// when its model time (a work item's k-steps at 3.0 us + its epilogue, + the combine) is under 0.7 of the single-limb
// kernel's (a tile's k-steps at 2.6 us).
inline bool one_limb_takes_limb_kernel(const MfmaPlan& pl, const MflPlan& ml) {
    if (!pl.usable || !ml.usable || ml.zparts < 2) return false;
    const double t1 = (double)pl.ksteps * 2.6, tl = (double)ml.zsteps * 3.0 + 5.0 + 10.0;
    return tl < 0.7 * t1;
}
constexpr int kFoldBlocks = 2048;  // partials of the first fold pass (workspace = kFoldBlocks elements)
int grid_for(size_t n);

// Tuning / test hooks.  None changes a result.  They are read from the
// environment ONCE, when the library is first used, and can be set afterwards with
// fpgpu_tuning_set(); the launchers only read these words (no getenv on any call path).
// 0 (or -1 where 0 is a legal value) = the library's default.
struct Tuning {
    long grid_cap = 0;        // FPGPU_GRID_CAP: workgroup cap of the streaming kernels
    long inv_c = 0;           // FPGPU_INV_C: chain length of every batch-inversion level
    long inv_c0 = 0;          // FPGPU_INV_C0: chain length of level 0 only
    long inv_fuse_top = -1;   // FPGPU_INV_FUSE_TOP: 0 = the last level as separate up / top / down kernels
    long inv_lp = -1;         // FPGPU_INV_LP: 0 = level-0 prefixes as words in the output array even when the workspace has room for limbs
    long matvec_mfma = -1;    // FPGPU_MATVEC_MFMA: 0 forces the VALU mat-vec
    long matvec_ksplit = 0;   // FPGPU_MATVEC_KSPLIT: column parts per row block (matrix-core mat-vec)
    long matvec_t = 0;        // FPGPU_MATVEC_T: products per pass of the VALU mat-vec
    long inv_sp = -1;         // FPGPU_INV_SP: 0 = dense level-0 prefixes at every width, 1 = sparse prefixes (k_binv_*_sp) wherever the limb form is not in use
    long matvec_rows = -1;    // FPGPU_MATVEC_ROWS: 1 = one row per lane (k_matvec_rows) whatever the shape, 0 = never; -1 = tall narrow matrices
    long matvec_cols = -1;    // FPGPU_MATVEC_COLS: 1 = VALU mat-vec on unsaturated column sums (k_matvec_cols), 0 = k_matvec; -1 = the launcher's choice
    long matmul_mfma = -1;    // FPGPU_MATMUL_MFMA: 0 forces the VALU mat-mul kernels; 1 takes the matrix cores for every multi-limb shape they can lay out, not only where they are expected to win
    long matmul_generic = 0;  // FPGPU_MATMUL_GENERIC: 1 forces the multi-limb mat-mul kernel
    long spmv_log2g = -1;     // FPGPU_SPMV_LOG2G: log2 of the lanes sharing a sparse row
    long spmv_unroll = 0;     // FPGPU_SPMV_UNROLL: cells a lane of the sparse mat-vec loads together (1, 2 or 4)
    long pow_window = -1;     // FPGPU_POW_WINDOW: 0 forces the binary pow ladder, 1 the sliding-window one
    long copy_threads = 0;    // FPGPU_COPY_THREADS: host threads that fill / drain the pinned staging buffers (1 = plain hipMemcpy)
    long allgather_copy = 0;  // FPGPU_ALLGATHER_COPY: 1 = peer copies instead of RCCL for the all-gather
};
const Tuning& tuning();

// Every launcher returns the first HIP error of the calls it makes (memsets included); the kernel
// launches themselves are checked by the caller with hipGetLastError().
template <int N>
struct Launch {
    static hipError_t elementwise(int op, const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n,
                            hipStream_t s);
    static hipError_t mul_chain(const fpgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, uint32_t iters,
                          hipStream_t s);
    static hipError_t batch_inv(const fpgpu_ctx* ctx, const void* x, void* out, uint8_t* flags, size_t n, char* ws, size_t ws_bytes,
                          hipStream_t s);
    static hipError_t matvec(const fpgpu_ctx* ctx, const void* A, size_t rows, size_t cols, const void* x, void* y,
                       void* workspace, size_t workspace_bytes,
                       hipStream_t s);
    static hipError_t matmul(const fpgpu_ctx* ctx, const void* A, const void* B, void* C, size_t m, size_t k, size_t n,
                       void* workspace, size_t workspace_bytes,
                       hipStream_t s);
    static hipError_t fill_random(const fpgpu_ctx* ctx, void* out, size_t n, uint64_t seed, size_t first,
                                  hipStream_t s);
    static hipError_t spmv(const fpgpu_ctx* ctx, const void* vals, const uint32_t* colidx, size_t rows, size_t cols,
                     uint32_t density, const void* x, void* y, hipStream_t s);
    static hipError_t pow(const fpgpu_ctx* ctx, const void* a, const uint32_t* exp32, int top_bit, void* out, size_t n,
                    hipStream_t s);
    static hipError_t fold(bool product, const fpgpu_ctx* ctx, const void* v, size_t n, void* out, void* ws,
                           hipStream_t s);
    static hipError_t dot(const fpgpu_ctx* ctx, const void* a, const void* b, size_t n, void* out, void* ws,
                          hipStream_t s);
};

extern template struct Launch<2>;
extern template struct Launch<4>;
extern template struct Launch<6>;
extern template struct Launch<8>;
extern template struct Launch<10>;
extern template struct Launch<12>;
extern template struct Launch<14>;
extern template struct Launch<16>;

}  // namespace fpgpu
