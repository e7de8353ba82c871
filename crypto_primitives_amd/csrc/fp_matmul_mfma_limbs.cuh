// Multi-limb (n64 >= 2) Fp mat-mul on the matrix cores: C = A B (mod p) as NL x NL single-limb products.
//
// k_matmul<N> (fp_kernels.cuh) is at the VALU's issue ceiling: 81 multiplies + 37 other instructions per 256-bit
// term, 296 G terms/s.  The single-limb kernel (fp_matmul_mfma.cuh) forms sum_k a_k b_k for 64-bit operands
// EXACTLY on v_mfma_i32_16x16x64_i8 (8 x 8 signed base-256 digits, 15 diagonal sums in int32).  An element of
// NL = n64 words is NL such operands: with the element's 8 NL balanced digits cut into NL groups of 8,
//     x = sum_u X_u 2^(64 u),   X_u = sum_{i<8} d_{8u+i} 256^i   (a signed 64-bit-sized value),
//     sum_k a_k b_k = sum_{w=0}^{2NL-2} 2^(64 w)  sum_{u+v=w} sum_k A_u,k B_v,k ,
// so the product is NL^2 passes of the SAME main loop (fp_matmul_kstep.inc, untouched) over digit planes of limb u
// of A and limb v of B^T -- 64 NL^2 int8 multiply-accumulates per term, the minimum for byte digits -- and the
// pairs of one limb diagonal w share their int32 accumulators (same weights), which are exact while the k-steps a
// diagonal has absorbed stay <= 255: k is walked in parts of floor(255 / NL) k-steps.  The k-steps of all diagonals and
// parts of a tile form ONE stream, staged one step ahead of the MFMAs across their boundaries.  After each (k part, w) the
// 15 digit diagonals are recombined into a 192-bit two's-complement value T_w per output and added into the
// workgroup's scratch (6 words per w per output, plain stores by the lane that owns the output: no atomics, the
// scratch belongs to the workgroup, which walks tiles persistently).  A tile ends with V = sum_w T_w 2^(64 w),
// exact in 2N + 2 words of two's complement, reduced as k_mv_finish does:
//     V = W0 + W1 R + W2 R^2  =>  V R^-1 = mont(W0, 1) + mont(W1, R) + mont(W2, R^2)  (- R^2 if V < 0),
// the canonical residue: bit-identical to the reference's fold of mul and add.
// Signed digits of an element: x itself when x <= 0x7f7f..7f, else x - p; an input that fits neither (non-canonical)
// raises the flag, this kernel returns at once and k_matmul<N>, gated on the flag, runs instead.
// Single-limb fields (N = 2) come here when k exceeds what k_matmul_mfma_1limb's int32 accumulators hold (16320): one pair,
// k in parts of 255 k-steps through the same scratch.
// A product of few tiles would occupy as many CUs: the plan then splits k over z work items per tile (fp_launch.hpp),
// each with its own reduction, and k_matvec_combine adds the z canonical residues of every output.
#pragma once
#include "fp_matmul_mfma.cuh"

namespace fpgpu {

// Digit planes of limb u of the logical matrix Y[rows][kdim] (Y[r][kk] = TRANS ? X[kk*ld + r] : X[r*ld + kk]) at
// W + u * plane_stride, each plane set in k_mf_digitize's layout.  Same wave / lane assignment as there, with the limb u
// in blockIdx.y (every limb re-reads the elements: they stay in L2).  2048^2 BN254 elements: 0.35 ms with a branch per
// element and dword loads, 0.11 ms with the loads batched (profiles/r3_experiments.md).
template <int N, bool TRANS>
__global__ __launch_bounds__(256) void k_mfl_digitize(const uint32_t* __restrict__ X, size_t rows, size_t kdim, size_t ld,
                                                      uint8_t* __restrict__ W, size_t plane_stride, uint32_t blocks,
                                                      uint32_t ksteps, uint32_t* __restrict__ flag, const FpParams<N> P) {
    const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t sub = (uint32_t)(gw & 3);
    const uint32_t ks = (uint32_t)((gw >> 2) % ksteps);
    const size_t blk = (gw >> 2) / ksteps;
    if (blk >= blocks) return;
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const size_t row = blk * kMF_TILE + sub * 16 + r;
    const size_t k0 = (size_t)ks * kMF_KSTEP + 16 * g;
    uint32_t bad = 0;
    uint8_t* out = W + ((blk * ksteps + ks) * 8) * 4096 + sub * 1024 + lane * 16;
    const int u = (int)blockIdx.y;
    {
        uint64_t d[16];
        // the loads of a batch of elements first, all in flight together (as whole 16-byte pieces), then branch-free
        // arithmetic: with a branch per element the compiler waits for each element's loads before it issues the next
        constexpr int JB = N <= 8 ? 16 : 8;
#pragma unroll
        for (int j0 = 0; j0 < 16; j0 += JB) {
            uint32_t wv[JB][N];
#pragma unroll
            for (int j = 0; j < JB; j++) {
                const size_t kk = k0 + j0 + j;
                const bool in = row < rows && kk < kdim;
                load_elem<N>(wv[j], X, in ? (TRANS ? kk * ld + row : row * ld + kk) : 0);   // (element 0 exists; zeroed below)
            }
#pragma unroll
            for (int j = 0; j < JB; j++) {
                const size_t kk = k0 + j0 + j;
                const bool in = row < rows && kk < kdim;
                uint32_t w[N], t[N];
#pragma unroll
                for (int l = 0; l < N; l++) w[l] = in ? wv[j][l] : 0u;
                // x > 0x7f7f..7f ?  then x - p (negative); flag if that is still too large (only a non-canonical x can be)
                bool gt = false, decided = false, gt2 = false, dec2 = false;
#pragma unroll
                for (int l = N - 1; l >= 0; l--) {
                    const bool ne = w[l] != 0x7F7F7F7Fu;
                    gt = !decided && ne ? w[l] > 0x7F7F7F7Fu : gt;
                    decided = decided || ne;
                }
                const uint32_t borrow = sub_limbs_s<N>(t, w, P.m);   // t = x - p mod 2^(32N); borrow = (x < p)
#pragma unroll
                for (int l = N - 1; l >= 0; l--) {
                    const bool ne = t[l] != 0x7F7F7F7Fu;
                    gt2 = !dec2 && ne ? t[l] > 0x7F7F7F7Fu : gt2;
                    dec2 = dec2 || ne;
                }
                bad |= gt && !borrow && gt2 ? 1u : 0u;
                // balanced digits of the whole element, four to a word; keep the two words of limb u
                unsigned cy = 0;
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int l = 0; l < N; l++) {
                    const uint32_t dw = __builtin_addc(gt ? t[l] : w[l], 0x80808080u, cy, &cy) ^ 0x80808080u;
                    lo = l == 2 * u ? dw : lo;
                    hi = l == 2 * u + 1 ? dw : hi;
                }
                d[j0 + j] = ((uint64_t)hi << 32) | lo;
            }
        }
        uint8_t* o = out + (size_t)u * plane_stride;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t q4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                q4[q] = (uint32_t)((d[4 * q] >> (8 * i)) & 0xff) | ((uint32_t)((d[4 * q + 1] >> (8 * i)) & 0xff) << 8) |
                        ((uint32_t)((d[4 * q + 2] >> (8 * i)) & 0xff) << 16) | ((uint32_t)((d[4 * q + 3] >> (8 * i)) & 0xff) << 24);
            }
            *reinterpret_cast<uint4*>(o + (size_t)i * 4096) = make_uint4(q4[0], q4[1], q4[2], q4[3]);
        }
    }
    if (bad) atomicOr(flag, 1u);
}

// int32 diagonals absorb 255 k-steps (fp_matmul_mfma.cuh: exact for K <= 16383); NL pairs share them
__host__ __device__ constexpr uint32_t mfl_part_ksteps(int nl) { return 255u / (uint32_t)nl; }
// words of scratch per workgroup: (2 NL - 1) limb diagonals x 6 words x 4096 outputs
__host__ __device__ constexpr size_t mfl_scratch_words(int nl) { return (size_t)(2 * nl - 1) * 6 * 4096; }

template <int N>
__global__ __launch_bounds__(256) void k_matmul_mfma_limbs(const uint8_t* __restrict__ Aw, const uint8_t* __restrict__ Bw,
                                                           size_t plane_a, size_t plane_b, uint32_t* __restrict__ Cm, size_t m,
                                                           size_t n, uint32_t rblocks, uint32_t cblocks, uint32_t ksteps,
                                                           uint32_t zparts, uint32_t zsteps, uint32_t* __restrict__ scratch,
                                                           const uint32_t* __restrict__ flag, const FpParams<N> P) {
    // zparts > 1 (too few tiles to fill the chip): work item (z, tile) takes the k-steps [z * zsteps, (z + 1) * zsteps) of its
    // tile and writes the canonical residue of that partial sum to Cm[(row * n + col) * zparts + z] -- Cm is then the
    // launcher's scratch and k_matvec_combine adds the parts of every output (as for k_matmul's k parts)
    static_assert(N >= 2 && N % 2 == 0, "whole 64-bit limbs (N = 2: single-limb fields with k beyond k_matmul_mfma_1limb's 16320)");
    constexpr int NL = N / 2, NW = 2 * NL - 1, SL = 2 * N + 2;
    constexpr uint32_t kPart = mfl_part_ksteps(NL);
    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * 2 * kMF_CHUNK];
    if (*flag) return;   // some input has no digit form: k_matmul<N> takes over
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    // this lane's 16 outputs of a tile live at slot (wave * 16 + o) * 64 + lane of each scratch word
    uint32_t* const mine = scratch + (size_t)blockIdx.x * mfl_scratch_words(NL) + (size_t)wave * 16 * 64 + lane;
    const uint32_t total8 = 8 * ((rblocks * cblocks + 7) / 8);

    // items z * total8 + tile: gridDim.x and total8 are multiples of 8, so a workgroup stays on its XCD's run of tiles
    for (uint32_t item = blockIdx.x; item < total8 * zparts; item += gridDim.x) {
        const uint32_t tb = item % total8, z = item / total8;
        uint32_t rb, cb;
        if (!mf_tile_of(tb, rblocks, cblocks, rb, cb)) continue;     // (uniform over the workgroup)
        const uint32_t ks_lo = z * zsteps, ks_hi = ks_lo + zsteps < ksteps ? ks_lo + zsteps : ksteps;
        // ONE stream of k-steps per item: (k part, limb diagonal w, pair u of the diagonal, k-step of the part), staged one step
        // ahead of the MFMAs ACROSS the diagonals and k parts -- the first chunk of the next diagonal is in flight while this
        // one's accumulators are recombined, so only the item's first k-step waits for its data.  `gs` counts the item's
        // k-steps (buffer parity); the cursor c_* names the step staged last.
        uint32_t gs = 0, c_kp0 = ks_lo, c_k = 0;
        int c_w = 0, c_u = 0;
        auto stage_part_at = [&](uint32_t kp0s, int ws, int u, uint32_t kl, int buf, int q) {
            const int piece = q * 4 + wave;
            const uint8_t* ga = Aw + (size_t)u * plane_a + ((size_t)rb * ksteps + kp0s + kl) * kMF_CHUNK;
            const uint8_t* gb = Bw + (size_t)(ws - u) * plane_b + ((size_t)cb * ksteps + kp0s + kl) * kMF_CHUNK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + piece * 1024 + lane_off),
                                             (__attribute__((address_space(3))) void*)(lds + buf * 2 * kMF_CHUNK + piece * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + piece * 1024 + lane_off),
                                             (__attribute__((address_space(3))) void*)(lds + buf * 2 * kMF_CHUNK + kMF_CHUNK + piece * 1024), 16, 0, 0);
        };
        // (the previous item's last barrier came after every wave's last LDS read and DMA: both buffers are free)
#pragma unroll
        for (int q = 0; q < 8; q++) stage_part_at(c_kp0, c_w, c_u, c_k, 0, q);
        __syncthreads();
        for (uint32_t kp0 = ks_lo; kp0 < ks_hi; kp0 += kPart) {
            const uint32_t kpn = ks_hi - kp0 < kPart ? ks_hi - kp0 : kPart;
#pragma unroll 1
            for (int w = 0; w < NW; w++) {
                const int u0 = w - (NL - 1) > 0 ? w - (NL - 1) : 0, u1 = w < NL - 1 ? w : NL - 1;
                const uint32_t vsteps = (uint32_t)(u1 - u0 + 1) * kpn;
                mf_v4i acc[15][2][2];
#pragma unroll
                for (int d = 0; d < 15; d++)
#pragma unroll
                    for (int rt = 0; rt < 2; rt++)
#pragma unroll
                        for (int ct = 0; ct < 2; ct++) acc[d][rt][ct] = mf_v4i{0, 0, 0, 0};
                for (uint32_t ks = 0; ks < vsteps; ks++, gs++) {
                    // the cursor moves to the step after the one computed now; past the item's end it stays (the last step
                    // re-stages itself into the idle buffer, so that no branch surrounds the DMAs)
                    {
                        uint32_t n_kp0 = c_kp0, n_k = c_k + 1;
                        int n_w = c_w, n_u = c_u;
                        const uint32_t c_kpn = ks_hi - c_kp0 < kPart ? ks_hi - c_kp0 : kPart;
                        if (n_k == c_kpn) {
                            n_k = 0;
                            if (++n_u > (n_w < NL - 1 ? n_w : NL - 1)) {
                                if (++n_w == NW) { n_w = 0; n_kp0 += kPart; }
                                n_u = n_w - (NL - 1) > 0 ? n_w - (NL - 1) : 0;
                            }
                        }
                        if (n_kp0 < ks_hi) { c_kp0 = n_kp0; c_k = n_k; c_w = n_w; c_u = n_u; }
                    }
                    const uint32_t s_kp0 = c_kp0, s_k = c_k;
                    const int s_w = c_w, s_u = c_u;
                    mf_v4i a[2][8], b[2][2];
                    const uint32_t a_addr = lds_addr + (gs & 1) * 2 * kMF_CHUNK + lane_off + 2 * wr * 1024;
                    const uint32_t b_addr = lds_addr + (gs & 1) * 2 * kMF_CHUNK + kMF_CHUNK + lane_off + 2 * wc * 1024;
#pragma push_macro("MF_STAGE")
#undef MF_STAGE
#define MF_STAGE(J) stage_part_at(s_kp0, s_w, s_u, s_k, (gs + 1) & 1, J)
#include "fp_matmul_kstep.inc"
#pragma pop_macro("MF_STAGE")
                    __syncthreads();   // the next step's chunk has landed (vmcnt(0) per wave) and this buffer is free to refill
                }
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA write -> accvgpr read, invisible to the compiler
                MF_FENCE_ACC(acc);
                // T_w (+)= sum_d D_d 256^d in 192-bit two's complement
#pragma unroll
                for (int rt = 0; rt < 2; rt++)
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
#pragma unroll
                        for (int e = 0; e < 4; e++) {
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
                            uint32_t* slot = mine + (size_t)w * 6 * 4096 + (size_t)((rt * 2 + ct) * 4 + e) * 64;
                            if (kp0 != ks_lo) {
                                unsigned cy = 0;
#pragma unroll
                                for (int t = 0; t < 6; t++) S[t] = __builtin_addc(S[t], slot[(size_t)t * 4096], cy, &cy);
                            }
#pragma unroll
                            for (int t = 0; t < 6; t++) slot[(size_t)t * 4096] = S[t];
                        }
            }
        }
        // ---- the tile's outputs: V = sum_w T_w 2^(64 w), reduced once
        uint32_t onep[N], oneR[N], r2[N];
#pragma unroll
        for (int l = 0; l < N; l++) onep[l] = l == 0 ? 1u : 0u;
        set_limbs<N>(oneR, P.one);
        set_limbs<N>(r2, P.r2);
#pragma unroll 1
        for (int o = 0; o < 16; o++) {
            const int rt = o >> 3, ct = (o >> 2) & 1, e = o & 3;
            const size_t row = (size_t)rb * kMF_TILE + (2 * wr + rt) * 16 + 4 * (lane >> 4) + e;
            const size_t col = (size_t)cb * kMF_TILE + (2 * wc + ct) * 16 + (lane & 15);
            uint32_t S[SL];
#pragma unroll
            for (int l = 0; l < SL; l++) S[l] = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const uint32_t* slot = mine + (size_t)w * 6 * 4096 + (size_t)o * 64;
                uint32_t t6[6];
#pragma unroll
                for (int t = 0; t < 6; t++) t6[t] = slot[(size_t)t * 4096];
                const uint32_t sg = (uint32_t)((int32_t)t6[5] >> 31);
                unsigned cy = 0;
#pragma unroll
                for (int t = 2 * w; t < SL; t++) S[t] = __builtin_addc(S[t], t - 2 * w < 6 ? t6[t - 2 * w] : sg, cy, &cy);
            }
            if (row >= m || col >= n) continue;
            uint32_t W0[N], W1[N], W2[N], t0[N], t1[N], t2[N], uu[N], v2[N], res[N], corr[N];
            const uint32_t sg = (uint32_t)((int32_t)S[SL - 1] >> 31);
#pragma unroll
            for (int l = 0; l < N; l++) {
                W0[l] = S[l];
                W1[l] = S[N + l];
                W2[l] = l < 2 ? S[2 * N + l] : sg;   // sign-extended: a negative V is read as V + 2^(32*3N)
            }
            mont_mul<N>(t0, W0, onep, P);
            mont_mul<N>(t1, W1, oneR, P);
            mont_mul<N>(t2, W2, r2, P);
            mod_add<N>(uu, t0, t1, P);
            mod_add<N>(v2, uu, t2, P);
#pragma unroll
            for (int l = 0; l < N; l++) corr[l] = sg ? r2[l] : 0u;   // 2^(32*3N) R^-1 = R^2 (mod p) comes back out
            mod_sub<N>(res, v2, corr, P);
            store_elem<N>(Cm, zparts > 1 ? (row * n + col) * zparts + z : row * n + col, res);
        }
    }
}

}  // namespace fpgpu
