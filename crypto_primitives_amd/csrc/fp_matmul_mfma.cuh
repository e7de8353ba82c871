// Single-limb (n64 = 1) Fp mat-mul on the matrix cores: C = A B (mod p) by byte slicing.
//
// BASELINE.json configs[4] (64-bit prime, 8192^2) is a real GEMM and VALU-bound in
// k_matmul_1limb (8 VALU instructions per 64x64-bit term).  Here every 64-bit element is
// written once as 8 signed base-256 digits, x = sum_i d_i 256^i with d_i in [-128, 127];
// then
//     sum_k a_k b_k  =  sum_{d=0..14} 256^d  *  sum_{i+j=d} sum_k a_k,i b_k,j
// and every inner sum is an int8 x int8 -> int32 matrix product, which is what
// v_mfma_i32_16x16x64_i8 computes: 16384 multiply-accumulates per instruction, one every
// 8.4 ns per SIMD measured (tools/ubench_mfma_i8.hip), so a 64x64-bit term = 64 of them costs
// 0.033 ns of SIMD time against 0.22 ns on the VALU (8 instructions x 4.3 cycles / 64 lanes).
// (The 32x32x32 form is 17 % faster per multiply in that register-only loop on constant data,
// but ran 10 % slower in this kernel on random operands, so the 16x16 form stays.)
// The 15 diagonal sums per output stay in int32 accumulators over the whole K
// loop (|digit product| <= 2^14, at most 8 products per k per diagonal: exact for
// K <= 16383), are recombined exactly in 192-bit two's complement at the end and reduced
// with the same three Montgomery multiplications as k_matmul_1limb, so the result is the
// canonical residue: bit-identical to the reference's fold of mul and add.
//
// Signed digits of an unsigned element: x itself if x <= 0x7f7f..7f (the largest value 8
// balanced digits hold), else the negative representative x - p, which always fits
// (x - p > -0x8080..80).  A non-canonical input x >= p whose x - p is still too large has
// no 8-digit form: the digitiser raises a flag, this kernel returns at once and
// k_matmul_1limb (which takes any a < R) runs instead - decided on the device, no host sync.
//
// Data movement.  A pre-pass writes A and B^T as int8 digit planes already in MFMA fragment
// order, so the main loop stages them with global_load_lds (no staging registers, lane-linear
// LDS image, conflict-free ds_read_b128):
//   chunk(block b, k-step s) = 32 KiB = [digit 0..7][16-row subtile 0..3][1 KiB fragment],
//   fragment = [k group g = 0..3][row r = 0..15][16 bytes: k = 64 s + 16 g + 0..15],
// i.e. lane l = 16 g + r of a wave reads its 16 operand bytes at offset 16 l.  The MFMA sums
// over all 64 k of a fragment pair, and A and B use the same (g, byte) -> k assignment, so the
// product does not depend on how the hardware orders k inside the instruction.
// Workgroup = 4 waves (one per SIMD, up to 512 registers each) = 64 x 64 outputs; a wave owns
// 32 x 32 = 2 x 2 MFMA tiles x 15 diagonals = 240 accumulator registers.  Per 64-deep k-step
// a wave issues 256 MFMAs against 32 ds_read_b128 and 16 global_load_lds.
#pragma once
#include "fp_device.cuh"

namespace fpgpu {

constexpr int kMF_TILE = 64;            // workgroup tile edge (rows and columns)
constexpr int kMF_KSTEP = 64;           // k per MFMA step
constexpr int kMF_CHUNK = 8 * 4 * 1024; // bytes of one operand per (block, k-step)
constexpr size_t kMF_MAXK = 16320;      // int32 accumulators are exact up to K = 16383; padded K is a multiple of 64
constexpr uint64_t kMF_MAXPOS = 0x7F7F7F7F7F7F7F7Full;
constexpr uint64_t kMF_BIAS = 0x8080808080808080ull;

typedef int mf_v4i __attribute__((ext_vector_type(4)));

// Digit planes of the logical matrix Y[rows][kdim]: Y[r][kk] = TRANS ? X[kk*ld + r] : X[r*ld + kk]
// (A as it is; B transposed, so that both operands have k innermost).  One wave per
// (block, k-step, 16-row subtile); lane (g = l>>4, r = l&15) owns elements (row r, k = 16g .. 16g+15).
template <bool TRANS>
__global__ __launch_bounds__(256) void k_mf_digitize(const uint64_t* __restrict__ X, size_t rows, size_t kdim, size_t ld,
                                                     uint8_t* __restrict__ W, uint32_t blocks, uint32_t ksteps,
                                                     uint64_t p, uint32_t* __restrict__ flag) {
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t sub = (uint32_t)(gw & 3);
    const uint32_t ks = (uint32_t)((gw >> 2) % ksteps);
    const size_t blk = (gw >> 2) / ksteps;
    if (blk >= blocks) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const size_t row = blk * kMF_TILE + sub * 16 + r;
    const size_t k0 = (size_t)ks * kMF_KSTEP + 16 * g;
    uint64_t d[16];
    uint32_t bad = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const size_t kk = k0 + j;
        uint64_t x = 0;
        if (row < rows && kk < kdim) x = TRANS ? X[kk * ld + row] : X[row * ld + kk];
        uint64_t t = x;
        if (x > kMF_MAXPOS) {
            t = x - p;
            if (x >= p && t > kMF_MAXPOS) bad = 1;
        }
        d[j] = (t + kMF_BIAS) ^ kMF_BIAS;  // byte i = balanced digit i
    }
    if (bad) atomicOr(flag, 1u);
    uint8_t* out = W + ((blk * ksteps + ks) * 8) * 4096 + sub * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            w[q] = (uint32_t)((d[4 * q] >> (8 * i)) & 0xff) | ((uint32_t)((d[4 * q + 1] >> (8 * i)) & 0xff) << 8) |
                   ((uint32_t)((d[4 * q + 2] >> (8 * i)) & 0xff) << 16) | ((uint32_t)((d[4 * q + 3] >> (8 * i)) & 0xff) << 24);
        }
        *reinterpret_cast<uint4*>(out + (size_t)i * 4096) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Consecutive workgroup ids land on the 8 XCDs round-robin; give each XCD a contiguous run of
// tiles, ordered in bands of 4 row-blocks so that the ~32 tiles an XCD works on at a time
// share 4 A panels and 8 B panels in its L2.
__device__ __forceinline__ bool mf_tile_of(uint32_t b, uint32_t rblocks, uint32_t cblocks, uint32_t& rb, uint32_t& cb) {
    const uint32_t total = rblocks * cblocks;
    const uint32_t per_xcd = (total + 7) / 8;
    const uint32_t L = (b & 7) * per_xcd + (b >> 3);
    if ((b >> 3) >= per_xcd || L >= total) return false;
    const uint32_t band = L / (4 * cblocks), rem = L % (4 * cblocks);
    const uint32_t rows_in_band = rblocks - 4 * band < 4 ? rblocks - 4 * band : 4;
    cb = rem / rows_in_band;
    rb = 4 * band + rem % rows_in_band;
    return true;
}

// The main loop is a fixed instruction order (fp_matmul_kstep.inc, tools/gen_matmul_kstep.py): MFMAs,
// LDS reads and their waits are asm statements.  Accumulators are pinned to AGPRs by the "+a"
// constraint -- with the builtin, hipcc moves them between the two register files around every MFMA
// as soon as a DMA shares the basic block -- so the next k-step's DMAs can be issued from inside the
// MFMA stream instead of ahead of it (16 DMAs x ~60 cycles of issue per k-step were not hidden at
// one wave per SIMD).
#define MF_MFMA(ACC, A, B) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define MF_LDS_READ(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(DST) : "v"(ADDR) : "memory")
#define MF_WAIT_LGKM(CNT) asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory")
// Nothing ties a bare `s_nop` asm to the accumulators: the compiler is free to hoist its own v_accvgpr_read / _mov of an
// MFMA result ABOVE the wait (it did in k_matmul_mfma_limbs<2>: one output in sixteen wrong).  An empty asm that "modifies"
// every accumulator, placed after the s_nop (volatile asms keep their order), makes every later read depend on it.
#define MF_FENCE_ACC(ACC)                                                               \
    _Pragma("unroll") for (int d_ = 0; d_ < 15; d_++)                                   \
        _Pragma("unroll") for (int r_ = 0; r_ < 2; r_++)                                \
            _Pragma("unroll") for (int c_ = 0; c_ < 2; c_++) asm volatile("" : "+a"(ACC[d_][r_][c_]))

// The three places where a timing-only build may alter the main loop (tools/variants/matmul_timing.hpp,
// force-included by tools/build_variant.sh; such builds give wrong results and are never shipped).
// The library itself only ever sees these definitions.
#ifndef MF_STAGE
#define MF_STAGE(J) stage_part(ks_next, (ks + 1) & 1, J)
#endif
#ifndef MF_KSTEPS
#define MF_KSTEPS(n) (n)
#endif
#ifndef MF_KSTEP_BARRIER
#define MF_KSTEP_BARRIER() __syncthreads()
#endif
template <int N>  // N == 2 only (see k_matmul_1limb)
__global__ __launch_bounds__(256) void k_matmul_mfma_1limb(const uint8_t* __restrict__ Aw, const uint8_t* __restrict__ Bw,
                                                           uint2* __restrict__ Cm, size_t m, size_t n, uint32_t rblocks,
                                                           uint32_t cblocks, uint32_t ksteps,
                                                           const uint32_t* __restrict__ flag, const FpParams<N> P) {
    static_assert(N == 2, "single-limb kernel");
    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * 2 * kMF_CHUNK];  // two buffers of (A chunk, B chunk)
    if (*flag) return;  // some input has no 8-digit form: k_matmul_1limb takes over
    uint32_t rb, cb;
    if (!mf_tile_of(blockIdx.x, rblocks, cblocks, rb, cb)) return;
    // the wave index as a scalar: every LDS address and every DMA source below is then a scalar
    // base plus the lane's 16-byte slot, so a DMA costs two scalar adds, s_mov m0 and the load
    // itself instead of VALU address arithmetic and a v_readfirstlane per piece
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint8_t* ga = Aw + (size_t)rb * ksteps * kMF_CHUNK;
    const uint8_t* gb = Bw + (size_t)cb * ksteps * kMF_CHUNK;

    // one eighth of the staging of k-step ks into LDS buffer buf: 32 pieces of 1 KiB per operand,
    // wave w moves pieces w, w+4, ...; part q = 0..7 moves piece 4q + w of A and of B
    auto stage_part = [&](uint32_t ks, int buf, int q) {
        const int piece = q * 4 + wave;
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(ga + (size_t)ks * kMF_CHUNK + piece * 1024 + lane_off),
            (__attribute__((address_space(3))) void*)(lds + buf * 2 * kMF_CHUNK + piece * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(gb + (size_t)ks * kMF_CHUNK + piece * 1024 + lane_off),
            (__attribute__((address_space(3))) void*)(lds + buf * 2 * kMF_CHUNK + kMF_CHUNK + piece * 1024), 16, 0, 0);
    };

    mf_v4i acc[15][2][2];
#pragma unroll
    for (int d = 0; d < 15; d++)
#pragma unroll
        for (int rt = 0; rt < 2; rt++)
#pragma unroll
            for (int ct = 0; ct < 2; ct++) acc[d][rt][ct] = mf_v4i{0, 0, 0, 0};

#pragma unroll
    for (int q = 0; q < 8; q++) stage_part(0, 0, q);
    __syncthreads();  // (waits for this wave's DMAs, then the barrier)
    for (uint32_t ks = 0; ks < MF_KSTEPS(ksteps); ks++) {
        // the next k-step's DMAs are issued from inside the MFMA stream, two per digit of B; the
        // last k-step re-stages itself into the idle buffer so that no branch is needed
        const uint32_t ks_next = ks + 1 < ksteps ? ks + 1 : ks;
        mf_v4i a[2][8], b[2][2];  // b: [parity of digit j][column tile], read one digit ahead
        const uint32_t a_addr = lds_addr + (ks & 1) * 2 * kMF_CHUNK + lane_off + 2 * wr * 1024;
        const uint32_t b_addr = lds_addr + (ks & 1) * 2 * kMF_CHUNK + kMF_CHUNK + lane_off + 2 * wc * 1024;
#include "fp_matmul_kstep.inc"
        MF_KSTEP_BARRIER();  // next buffer has landed (vmcnt(0) per wave) and this one is free to refill
    }
    // the compiler does not see the asm MFMAs' writes: cover the MFMA-write -> accvgpr-read hazard by hand
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    MF_FENCE_ACC(acc);

    // ---- recombine the diagonals and reduce: C/D layout col = lane & 15, row = 4 (lane >> 4) + reg
    uint32_t onep[2] = {1u, 0u}, oneR[2], r2[2], zero[2] = {0u, 0u};
    set_limbs<2>(oneR, P.one);
    set_limbs<2>(r2, P.r2);
#pragma unroll
    for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const size_t row = (size_t)rb * kMF_TILE + (2 * wr + rt) * 16 + 4 * (lane >> 4) + e;
                const size_t col = (size_t)cb * kMF_TILE + (2 * wc + ct) * 16 + (lane & 15);
                // S = sum_d D_d 256^d in 192-bit two's complement, four diagonals (32 bits of weight) at a time
                uint32_t S[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int64_t v = 0;
#pragma unroll
                    for (int dd = 3; dd >= 0; dd--)
                        if (4 * q + dd < 15) v = v * 256 + (int64_t)acc[4 * q + dd][rt][ct][e];
                    const uint32_t lo = (uint32_t)v, hi = (uint32_t)((uint64_t)v >> 32), sg = (uint32_t)(v >> 63);
                    unsigned cy = 0;
                    S[q] = __builtin_addc(S[q], lo, cy, &cy);
                    S[q + 1] = __builtin_addc(S[q + 1], hi, cy, &cy);
#pragma unroll
                    for (int t = q + 2; t < 6; t++) S[t] = __builtin_addc(S[t], sg, cy, &cy);
                }
                if (row >= m || col >= n) continue;
                uint32_t s0[2] = {S[0], S[1]}, s1[2] = {S[2], S[3]}, s2[2] = {S[4], S[5]};
                uint32_t t0[2], t1[2], t2[2], u[2], v2[2], w[2], corr[2];
                mont_mul<2>(t0, s0, onep, P);
                mont_mul<2>(t1, s1, oneR, P);
                mont_mul<2>(t2, s2, r2, P);
                mod_add<2>(u, t0, t1, P);
                mod_add<2>(v2, u, t2, P);
                // a negative S was read as S + 2^192: take 2^192 R^-1 = R^2 (mod p) back out
                const bool neg = (S[5] >> 31) != 0;
                corr[0] = neg ? r2[0] : zero[0];
                corr[1] = neg ? r2[1] : zero[1];
                mod_sub<2>(w, v2, corr, P);
                Cm[row * n + col] = make_uint2(w[0], w[1]);
            }
}

}  // namespace fpgpu
