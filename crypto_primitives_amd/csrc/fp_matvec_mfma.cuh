// Fp dense mat-vec on the matrix cores: y = A x (mod p) with A streamed ONCE from HBM, as it
// lies, and no VALU multiplier in the loop.
//
// BASELINE.json configs[3] (2^15 x 2^15 over BN254 Fr) moves 32 GiB; the VALU kernel k_matvec
// needs ~190 instructions per 32-byte element beside that stream and ends at 4.9 TB/s, while a
// bare read of the same bytes runs at 6.0-6.2 TB/s (tools/ubench_rowtile_read.hip).  Here the
// products go to v_mfma_i32_16x16x64_i8 instead.
//
// An element is EB = 8*n64 bytes, little-endian: A[i][c] = sum_s a_{i,c,s} 256^s with the
// a's being exactly the bytes in memory.  Write x[c] in balanced signed digits,
// x[c] = sum_t xd_{c,t} 256^t, xd in [-128, 127] (x itself, or x - p when x is too large for
// EB such digits, as in fp_matmul_mfma.cuh).  Then
//     sum_c A[i][c] x[c]  =  sum_{d=0}^{2EB-2} 256^d  D[i][d],
//     D[i][d] = sum_{k} a_{i,k} B[k][d],      k = c*EB + s = byte offset in row i,
//     B[k][d] = xd_{c, d-s}  (0 outside 0..EB-1),
// an int8 GEMM  (rows x cols*EB) . (cols*EB x 2EB-1)  whose left operand IS matrix A's memory
// and whose right operand is a Toeplitz expansion of x (cols*EB x 64 bytes at EB = 32: 64 MiB
// for C4, built once per call, read through L2 and LDS).  MFMA wants signed bytes: the kernel
// flips the top bit of every byte of A (a - 128 = a ^ 0x80 as int8, one v_xor per dword), and
// the finishing kernel adds 128 * sum_k B[k][d] back, a per-call constant vector.
//
// A wave owns 16 rows: lane (r = l & 15, g = l >> 4) loads the 16 bytes at offset 64 t + 16 g of
// row r for k-step t, i.e. a wave instruction reads 64 contiguous bytes of each of 16 rows; that
// pattern streams HBM as fast as a linear sweep (ubench above).  The int32 tiles (4 column
// tiles of 16 diagonals) are exact for 4096 elements; they are then added into a 64-bit
// D[row][64] array with atomics, which also lets the columns be split over workgroups when
// there are few rows (row-sharded multi-GPU runs).  k_mv_finish recombines the diagonals of a
// row exactly (two's complement, 2N+2 limbs) and reduces with three Montgomery
// multiplications: S = W0 + W1 R + W2 R^2  =>  S R^-1 = mont(W0,1) + mont(W1,R) + mont(W2,R^2),
// the canonical residue, bit-identical to the reference's fold of mul and add.
#pragma once
#include "fp_device.cuh"
#include "fp_launch.hpp"

namespace fpgpu {

constexpr int kMV_EXPAND_BATCH = 32;  // k-steps a workgroup of k_mv_expand digitises at a time
typedef int mv_v4i __attribute__((ext_vector_type(4)));

// ---- pre-pass: signed digits of x, their Toeplitz expansion in MFMA fragment order, column sums
// B chunk layout: k-step ks, column tile nt -> 1 KiB fragment at ((ks * NT) + nt) * 1024, lane
// l = 16 g + n holds B[64 ks + 16 g + 0..15][16 nt + n].
// One workgroup of NT waves walks batches of kMV_EXPAND_BATCH k-steps: blockIdx.x, blockIdx.x + gridDim.x, ...
template <int N, int NT>
__global__ __launch_bounds__(NT * 64) void k_mv_expand(const uint32_t* __restrict__ x, size_t cols, size_t ksteps,
                                                       uint8_t* __restrict__ Bw, unsigned long long* __restrict__ colsum,
                                                       uint32_t* __restrict__ flag, const FpParams<N> P) {
    constexpr int EB = 4 * N;
    constexpr int MAXEL = 64 * kMV_EXPAND_BATCH / EB + 2;  // elements a batch of k-steps can touch
    // digits of the batch's elements, EB zero bytes on either side of each: B[k][d] = xd[c][d - s] for k = c EB + s is
    // then a plain (reversed) window of the row, zero where d - s leaves [0, EB)
    constexpr int ROW = 3 * EB;
    __shared__ __attribute__((aligned(16))) int8_t xd[MAXEL][ROW];
    const int nt = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, n = lane & 15;
    const int d = 16 * nt + n;
    long long csum = 0;
    for (int i = threadIdx.x; i < MAXEL * ROW / 4; i += NT * 64) reinterpret_cast<uint32_t*>(&xd[0][0])[i] = 0;   // the padding stays zero
    // A workgroup takes kMV_EXPAND_BATCH consecutive k-steps at a time: their elements are digitised
    // into LDS together (two barriers per batch, not per k-step), then every wave writes its column
    // tile's fragment of each k-step.
    for (size_t ks0 = (size_t)blockIdx.x * kMV_EXPAND_BATCH; ks0 < ksteps; ks0 += (size_t)gridDim.x * kMV_EXPAND_BATCH) {
        const size_t c_first = (64 * ks0) / EB;
        __syncthreads();  // previous batch's readers are done (and, the first time, the zero fill)
        for (int e = threadIdx.x; e < MAXEL; e += NT * 64) {
            const size_t c = c_first + e;
            uint32_t w[N];
#pragma unroll
            for (int l = 0; l < N; l++) w[l] = c < cols ? x[c * N + l] : 0u;
            // x > 0x7f7f..7f ?  then use x - p (negative); flag if that is still too large
            bool gt = false, decided = false;
#pragma unroll
            for (int l = N - 1; l >= 0; l--)
                if (!decided && w[l] != 0x7F7F7F7Fu) { gt = w[l] > 0x7F7F7F7Fu; decided = true; }
            if (gt) {
                uint32_t t[N];
                const uint32_t borrow = sub_limbs_s<N>(t, w, P.m);  // t = x - p mod 2^(32N); borrow = (x < p)
                if (!borrow) {
                    bool gt2 = false, dec2 = false;
#pragma unroll
                    for (int l = N - 1; l >= 0; l--)
                        if (!dec2 && t[l] != 0x7F7F7F7Fu) { gt2 = t[l] > 0x7F7F7F7Fu; dec2 = true; }
                    if (gt2) atomicOr(flag, 1u);
                }
#pragma unroll
                for (int l = 0; l < N; l++) w[l] = t[l];
            }
            unsigned cy = 0;
#pragma unroll
            for (int l = 0; l < N; l++)   // balanced digits, four to a word (little-endian: digit 4l + b is byte b)
                reinterpret_cast<uint32_t*>(&xd[e][EB])[l] = __builtin_addc(w[l], 0x80808080u, cy, &cy) ^ 0x80808080u;
        }
        __syncthreads();
        const size_t ks_end = ks0 + kMV_EXPAND_BATCH < ksteps ? ks0 + kMV_EXPAND_BATCH : ksteps;
        for (size_t ks = ks0; ks < ks_end; ks++) {
            uint32_t wds[4];
            if constexpr (EB % 16 == 0) {
                // the lane's 16 values of k lie in ONE element (16 | EB): bytes j = 0..15 are xd[c][p0 - j], the 16-byte
                // window of the padded row that ends at p0 = d - s0, reversed.  Five aligned words, a funnel shift, a byte swap.
                const size_t k0 = 64 * ks + 16 * g;
                const size_t c = k0 / EB;
                const int p0 = d - (int)(k0 % EB);
                const int w0 = EB + p0 - 15;                 // first byte of the window in the row: 1 .. 3 EB - 16
                const uint32_t* row = reinterpret_cast<const uint32_t*>(&xd[c - c_first][0]) + (w0 >> 2);
                uint32_t q[5];
#pragma unroll
                for (int i = 0; i < 5; i++) q[i] = row[i < 4 || (w0 & 3) ? i : 3];   // (the fifth word only exists off alignment)
                const uint32_t sh = (uint32_t)(w0 & 3);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t win = __builtin_amdgcn_alignbyte(q[i + 1], q[i], sh);   // window bytes 4i .. 4i + 3
                    wds[3 - i] = __builtin_bswap32(win);
                    csum += __builtin_amdgcn_sdot4((int)win, 0x01010101, 0, false);
                }
                if (c >= cols) {   // k past the last column: B is zero there (the rows of xd beyond cols are zero already)
#pragma unroll
                    for (int i = 0; i < 4; i++) wds[i] = 0;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint32_t word = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const size_t k = 64 * ks + 16 * g + 4 * q + b;
                        const size_t c = k / EB;   // c - c_first <= 64 * kMV_EXPAND_BATCH / EB < MAXEL
                        const int s = (int)(k % EB), t = d - s;
                        int v = 0;
                        if (c < cols && t >= 0 && t < EB) v = xd[c - c_first][EB + t];
                        csum += v;
                        word |= (uint32_t)(v & 0xff) << (8 * b);
                    }
                    wds[q] = word;
                }
            }
            *reinterpret_cast<uint4*>(Bw + (ks * NT + nt) * 1024 + lane * 16) = make_uint4(wds[0], wds[1], wds[2], wds[3]);
        }
    }
    // column sums: lanes n, n+16, n+32, n+48 of a wave share d
    csum += __shfl_xor(csum, 16, 64);
    csum += __shfl_xor(csum, 32, 64);
    if (g == 0) atomicAdd(colsum + d, (unsigned long long)csum);
}

// ---- main kernel ----------------------------------------------------------------------------
template <int N, int NT>
__global__ __launch_bounds__(kMV_WAVES * 64) void k_matvec_mfma(const uint8_t* __restrict__ A, size_t rows, size_t rowbytes,
                                                                const uint8_t* __restrict__ Bw, uint32_t chunks_total,
                                                                uint32_t ksplit, unsigned long long* __restrict__ Dbuf,
                                                                const uint32_t* __restrict__ flag) {
    // k-steps per staged chunk of B: 8 up to 256 bits; 4 beyond, where two 8-step buffers (16 NT KiB) would leave room
    // for ONE workgroup per CU (BLS12-381: 96 of 160 KiB) and too few loads in flight to cover HBM's latency
    constexpr int KC = mv_chunk_ksteps(NT);
    constexpr int CHUNK_BYTES = KC * NT * 1024;
    // |a - 128| <= 128, |xd| <= 128: a k-step adds at most 64 * 2^14 = 2^20 to a diagonal sum,
    // so 2047 k-steps are exact in int32; flush every 2040 k-steps
    constexpr uint32_t flush_every = 2040 / KC;
    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * CHUNK_BYTES];
    if (*flag) return;  // x has no EB-digit form: the VALU kernel k_matvec runs instead
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t rowblock = blockIdx.x / ksplit, part = blockIdx.x % ksplit;
    const uint32_t per_part = (chunks_total + ksplit - 1) / ksplit;
    const uint32_t c0 = part * per_part;
    const uint32_t c1 = c0 + per_part < chunks_total ? c0 + per_part : chunks_total;
    if (c0 >= c1) return;
    const size_t row = (size_t)rowblock * (kMV_WAVES * 16) + wave * 16 + (lane & 15);
    const uint8_t* arow = A + (row < rows ? row : rows - 1) * rowbytes;  // rows past the end: read a valid row, add nothing
    const size_t lane_off = 16 * (lane >> 4);
    const size_t last_off = rowbytes - 16;  // k past the end of the row meets B = 0: any readable bytes will do

    auto stage = [&](uint32_t c, int buf) {  // chunk c of B -> LDS buffer buf: NT KiB per k-step, 8 KiB per DMA round
#pragma unroll
        for (int q = 0; q < (KC * NT + kMV_WAVES - 1) / kMV_WAVES; q++) {
            const int piece = q * kMV_WAVES + wave;
            if ((KC * NT) % kMV_WAVES != 0 && piece >= KC * NT) break;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(Bw + (size_t)c * CHUNK_BYTES + piece * 1024 + lane * 16),
                (__attribute__((address_space(3))) void*)(lds + buf * CHUNK_BYTES + piece * 1024), 16, 0, 0);
        }
    };
    // A loads are inline asm so that their waits can be counted by hand: with an LDS-DMA in the same
    // loop hipcc would wait vmcnt(0) at the first use of any ordinary load and at every __syncthreads(),
    // draining the chunk that should stay in flight across the barrier.
    // (One address register pair per chunk and immediate offsets.  Only the last chunk of a row can
    // reach past the row's end, where B is zero: those lanes read the row's last 16 bytes instead.)
    const bool ragged = (size_t)chunks_total * (KC * 64) > rowbytes;
    auto load_a = [&](uint32_t c, mv_v4i (&v)[KC]) {
        if (!ragged || c + 1 < chunks_total) {
            const uint8_t* pc = arow + (size_t)c * (KC * 64) + lane_off;
#pragma unroll
            for (int t = 0; t < KC; t++)
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v[t]) : "v"(pc), "n"(t * 64));
        } else {
#pragma unroll
            for (int t = 0; t < KC; t++) {
                size_t off = ((size_t)c * KC + t) * 64 + lane_off;
                off = off < last_off ? off : last_off;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[t]) : "v"(arow + off));
            }
        }
    };
    mv_v4i acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[nt] = mv_v4i{0, 0, 0, 0};
    auto flush = [&]() {
        // (the lane id goes through an empty asm so that the 16 row addresses are formed here, in this
        // rarely run block, instead of being hoisted out of the loop into 32 live registers)
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        unsigned long long* drow = Dbuf + ((size_t)rowblock * (kMV_WAVES * 16) + wave * 16 + 4 * (l2 >> 4)) * (NT * 16) + (l2 & 15);
        const size_t r0 = (size_t)rowblock * (kMV_WAVES * 16) + wave * 16 + 4 * (l2 >> 4);
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (r0 + e < rows) atomicAdd(drow + e * (NT * 16) + 16 * nt, (unsigned long long)(long long)acc[nt][e]);
                acc[nt][e] = 0;
            }
    };

    // Three register sets of KC x 16 bytes rotate by name (no copies: a register with a load in flight
    // must not be read): one computes, one was issued a chunk ago, one is issued now.  At the end of a
    // phase the outstanding memory operations are, oldest first, the previous phase's KC loads, this
    // phase's DMAs and its KC loads: vmcnt(KC) retires all but the newest KC, i.e. the next operand set
    // and the next B chunk, and leaves a whole chunk in flight across the barrier.
    mv_v4i s0[KC], s1[KC], s2[KC];
    uint32_t since_flush = 0;
    uint32_t c = c0;
    auto compute = [&](mv_v4i (&cur)[KC], int buf) {
        const uint8_t* base = lds + buf * CHUNK_BYTES + lane * 16;
#pragma unroll
        for (int t = 0; t < KC; t++) {
            const mv_v4i a = cur[t] ^ (int)0x80808080;  // unsigned bytes -> a - 128 as int8
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const mv_v4i b = *reinterpret_cast<const mv_v4i*>(base + (t * NT + nt) * 1024);
                acc[nt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[nt], 0, 0, 0);
            }
        }
        if (++since_flush == flush_every) {
            flush();
            since_flush = 0;
        }
    };
    auto retire = [&](mv_v4i (&landed)[KC]) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KC) : "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int t = 0; t < KC; t++) asm volatile("" : "+v"(landed[t]));  // no use of it may move above the wait
    };
#define MV_PHASE(CUR, NXT, NX2)                                    \
    {                                                              \
        const int buf = (c - c0) & 1;                              \
        if (c + 1 < c1) stage(c + 1, buf ^ 1);                     \
        load_a(c + 2 < c1 ? c + 2 : c, NX2);                       \
        compute(CUR, buf);                                         \
        retire(NXT);                                               \
        c++;                                                       \
    }
    stage(c0, 0);
    load_a(c0, s0);
    load_a(c0 + 1 < c1 ? c0 + 1 : c0, s1);
    retire(s0);
    while (true) {
        MV_PHASE(s0, s1, s2)
        if (c >= c1) break;
        MV_PHASE(s1, s2, s0)
        if (c >= c1) break;
        MV_PHASE(s2, s0, s1)
        if (c >= c1) break;
    }
#undef MV_PHASE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped tail loads
    flush();
}

// ---- finish: y[row] = (sum_d 256^d (D[row][d] + 128 colsum[d])) R^-1 mod p ---------------------
template <int N, int NT>
__global__ __launch_bounds__(256) void k_mv_finish(const unsigned long long* __restrict__ Dbuf,
                                                   const unsigned long long* __restrict__ colsum, uint32_t* __restrict__ y,
                                                   size_t rows, const uint32_t* __restrict__ flag, const FpParams<N> P) {
    if (*flag) return;
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    constexpr int SL = 2 * N + 2;  // limbs of S (two's complement)
    static_assert(NT * 4 <= 2 * N, "diagonal groups must fit below the two spare limbs");
    uint32_t S[SL];
#pragma unroll
    for (int l = 0; l < SL; l++) S[l] = 0;
#pragma unroll
    for (int q = 0; q < NT * 4; q++) {
        long long v = 0;  // four diagonals = 32 bits of weight
#pragma unroll
        for (int dd = 3; dd >= 0; dd--) {
            const int d = 4 * q + dd;
            const long long D = (long long)Dbuf[row * (NT * 16) + d] + 128 * (long long)colsum[d];
            v = v * 256 + D;
        }
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)((unsigned long long)v >> 32), sg = (uint32_t)(v >> 63);
        unsigned cy = 0;
        S[q] = __builtin_addc(S[q], lo, cy, &cy);
        S[q + 1] = __builtin_addc(S[q + 1], hi, cy, &cy);
#pragma unroll
        for (int t = q + 2; t < SL; t++) S[t] = __builtin_addc(S[t], sg, cy, &cy);
    }
    uint32_t W0[N], W1[N], W2[N], onep[N], oneR[N], r2[N], t0[N], t1[N], t2[N], u[N], v2[N], w[N], corr[N];
    const uint32_t sg = (uint32_t)((int32_t)S[SL - 1] >> 31);
#pragma unroll
    for (int l = 0; l < N; l++) {
        W0[l] = S[l];
        W1[l] = S[N + l];
        W2[l] = l < 2 ? S[2 * N + l] : sg;  // sign-extended: a negative S is read as S + 2^(32*3N)
        onep[l] = l == 0 ? 1u : 0u;
    }
    set_limbs<N>(oneR, P.one);
    set_limbs<N>(r2, P.r2);
    mont_mul<N>(t0, W0, onep, P);
    mont_mul<N>(t1, W1, oneR, P);
    mont_mul<N>(t2, W2, r2, P);
    mod_add<N>(u, t0, t1, P);
    mod_add<N>(v2, u, t2, P);
    // 2^(32*3N) R^-1 = R^2 (mod p) comes back out when S was negative
#pragma unroll
    for (int l = 0; l < N; l++) corr[l] = sg ? r2[l] : 0u;
    mod_sub<N>(w, v2, corr, P);
    store_elem<N>(y, row, w);
}

}  // namespace fpgpu
