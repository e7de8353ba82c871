// fp_kernels.cuh -- gfx950 kernels over AoS element arrays (see include/fpgpu.h for
// the layout contract).  Every kernel is HBM-streaming unless noted; DESIGN.md
// lists algorithmic bytes per element and the roofline that bounds each.
#pragma once
#include "fp_device.cuh"
#include "fp_launch.hpp"
#include "fp_unsat.cuh"

namespace fpgpu {


// Which elementwise operations take the unsaturated multiplier (fp_unsat.cuh): the multiplying
// ones, from 384 bits on, where the saturated multiplier makes them VALU-bound (BLS12-381: sqr
// 4.26 TB/s and mul 5.13 against 5.4 for add).  Up to 320 bits the saturated multiplier hides
// behind the memory time; the unsaturated one was tried at 256 bits for the one-read-one-write
// kernels (round 1: sqr 3.35 -> 3.74 ms, to_mont 3.20 -> 3.30 ms over 2^28 elements; round 2, with R^2
// as scalar limbs: to_mont 3.10 -> 3.23 ms although the kernel has 190 four-cycle instructions
// instead of 275 -- one serial 64-bit column chain per lane overlaps worse with the loads than the
// saturated mad / addc pairs) and is not used there.  At 256 bits sqr takes the symmetric saturated
// square (mont_sqr: 36 limb products instead of 64) and to_mont multiplies by R^2 out of scalar
// registers (mont_mul_const).
template <int N, int OP>
struct UnsatElementwise {
    static constexpr bool value = N >= 12 && (OP == OP_MUL || OP == OP_SQR || OP == OP_TO_MONT);
};

// ---- elementwise: out[i] = op(a[i], b[i]) ------------------------------------------
// One element per lane per iteration, grid-stride.  Lane i of a wave reads its
// whole 4N-byte element with 16-byte loads; a wave covers 64 consecutive elements
// (2 KiB at N = 8), so every fetched 128-byte line is fully consumed by the wave
// that touched it (measured 5.3 TB/s on the 2R+1W pattern, profiles/).
// One elementwise operation on registers (shared by every shape of the streaming kernel).
template <int N, int OP>
__device__ __forceinline__ void elementwise_op(uint32_t (&r)[N], const uint32_t (&x)[N], const uint32_t (&y)[N],
                                               const FpParams<N>& P) {
    if constexpr (UnsatElementwise<N, OP>::value) {
        // unsaturated multiplier (fp_unsat.cuh) when the multiplier operand is canonical; an
        // operand >= p (outside the field, still defined for monty_mul) takes the saturated path,
        // which forms the reference's integer T for every input
        UParams<UF<N>::L> U;
        uparams_from<N>(U, P);
        if constexpr (OP == OP_TO_MONT) {
            uint32_t c[N];
            set_limbs<N>(c, P.r2);          // x may be any N-word value; R^2 mod p is canonical
            umul_words<N>(r, x, c, U);
        } else {
            uint32_t d[N];
            const bool canonical = sub_limbs_s<N>(d, OP == OP_MUL ? y : x, P.m) != 0;
            if (canonical) {
                if constexpr (OP == OP_MUL) umul_words<N>(r, x, y, U);
                else usqr_words<N>(r, x, U);
            } else {
                if constexpr (OP == OP_MUL) mont_mul<N>(r, x, y, P);
                else mont_sqr<N>(r, x, P);
            }
        }
    } else if constexpr (OP == OP_MUL) mont_mul<N>(r, x, y, P);
    else if constexpr (OP == OP_SQR) mont_sqr<N>(r, x, P);
    else if constexpr (OP == OP_ADD) mod_add<N>(r, x, y, P);
    else if constexpr (OP == OP_SUB) mod_sub<N>(r, x, y, P);
    else if constexpr (OP == OP_NEG) mod_neg<N>(r, x, P);
    else if constexpr (OP == OP_TO_MONT) mont_mul_const<N>(r, x, P.r2, P);
    else if constexpr (OP == OP_FROM_MONT) mont_retrieve<N>(r, x, P);
    else if constexpr (OP == OP_DOUBLE) mod_add<N>(r, x, x, P);
    else if constexpr (OP == OP_HALVE) mod_half<N>(r, x, P);
}

// One element per lane per iteration, grid-stride.  (Two elements per lane with both loads issued up
// front -- more bytes in flight per wave -- was measured and lost 1-4 % on every operation: the
// million short-lived workgroups already keep the memory pipeline full.)
template <int N, int OP>
__global__ __launch_bounds__(kBlock) void k_elementwise(const uint32_t* __restrict__ a,
                                                        const uint32_t* __restrict__ b,
                                                        uint32_t* __restrict__ out, size_t n,
                                                        const FpParams<N> P) {
    constexpr bool BIN = OP == OP_MUL || OP == OP_ADD || OP == OP_SUB;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t x[N], y[N], r[N];
        load_elem<N>(x, a, i);
        if constexpr (BIN) load_elem<N>(y, b, i);
        elementwise_op<N, OP>(r, x, BIN ? y : x, P);
        store_elem<N>(out, i, r);
    }
}

// ---- dependent multiply chain (VALU-bound; "Mul assign" bench shape) -----------
// Chain<N> (fp_unsat.cuh): from N = 6 limbs on the chain runs on unsaturated 29-bit limbs (no
// carry instruction inside a column) and the running value stays in [0, 2p) between steps where
// the modulus leaves two spare bits; it is made canonical once, before the store.
template <int N>
__global__ __launch_bounds__(kBlock) void k_mul_chain(const uint32_t* __restrict__ a,
                                                      const uint32_t* __restrict__ b,
                                                      uint32_t* __restrict__ out, size_t n, uint32_t iters,
                                                      const FpParams<N> P) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    Chain<N> acc(P);
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t w[N];
        typename Chain<N>::Op y;
        load_elem<N>(w, b, i);
        Chain<N>::prep(y, w);
        load_elem<N>(w, a, i);
        if (iters) {   // zero iterations hand the input back untouched
            acc.set(w);
            for (uint32_t it = 0; it < iters; it++) acc.mul_op(y);
            acc.get_canonical(w);
        }
        store_elem<N>(out, i, w);
    }
}

// ---- synthetic uniform field elements (SURVEY.md 8d generator) -----------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <int N>
__global__ __launch_bounds__(kBlock) void k_fill_random(uint32_t* __restrict__ out, size_t n, uint64_t seed,
                                                        size_t first, const FpParams<N> P) {
    // out[i] is element first + i of the generator's sequence (a shard of a larger vector passes its offset)
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t x[N], d[N];
#pragma unroll
        for (int l = 0; l < N / 2; l++) {
            uint64_t w = splitmix64(seed ^ ((first + i) * (N / 2) + l));
            x[2 * l] = (uint32_t)w;
            x[2 * l + 1] = (uint32_t)(w >> 32);
        }
#pragma unroll
        for (int l = 0; l < N; l++) {  // keep bitlen(p) bits: candidate < 2^bitlen <= 2p
            const int nb = (int)P.bitlen - 32 * l;
            x[l] &= nb >= 32 ? ~0u : (nb <= 0 ? 0u : ((1u << nb) - 1u));
        }
        uint32_t bo = sub_limbs_s<N>(d, x, P.m);
#pragma unroll
        for (int l = 0; l < N; l++) x[l] = bo ? x[l] : d[l];
        store_elem<N>(out, i, x);
    }
}

// ---- modular inversion of single elements (top level of batch inversion) -----------
// Binary GCD on word-sized approximations (Pornin, "Optimized Binary GCD for Modular
// Inversion", 2020), one element per lane, one instruction stream for the whole wave.
// Each outer round runs K = 30 divsteps on 62-bit approximations of (a, b) (top 32 bits
// + low 30 bits; exact once both fit 64 bits), collecting update factors f0,g0,f1,g1
// (|.| <= 2^30), then applies them once to the full-width values:
//     (a, b) <- (|a f0 + b g0|, |a f1 + b g1|) / 2^30          (exact division)
//     (u, v) <- ( u f0 + v g0 ,  u f1 + v g1 ) / 2^30  mod p   (Montgomery step on 2^30)
// with a = y u, b = y v (mod p) invariant, so b = gcd ends as 1 and v = y^-1 mod p.
// ceil((2 len(p) - 1) / 30) + 2 rounds always suffice (extra rounds are no-ops); that is
// 28 rounds of ~1.4k instructions for a 381-bit modulus, against 762 full-width steps of
// ~270 instructions for the textbook loop (1.4 ms -> 0.36 ms -> this).
// Returns false when gcd(y, p) != 1 (the reference's invert_odd_mod_vartime -> None).
// The update factors of K = 30 divsteps satisfy |f| + |g| <= 2^30 (each step at most doubles the sum of a
// row of the transition matrix), which is what bounds everything below.
//
// |x f + y g| / 2^30 for non-negative N-limb x, y; returns true if the signed value was negative.
// x f + y g = +-(|f| x + s |g| y) with s = -1 where the signs differ: y is negated in two's complement on
// N + 1 limbs, so one product-scanning pass -- two multiply-adds per limb into a 64-bit column, no carry
// instruction -- yields the signed inner value modulo 2^(32 (N + 1)) (|inner| < 2^(32 N + 30) fits).
template <int N>
__device__ __forceinline__ bool bingcd_update_ab(uint32_t (&out)[N], const uint32_t (&x)[N], const uint32_t (&y)[N],
                                                 int32_t f, int32_t g) {
    const bool sf = f < 0, sg = g < 0;
    const uint32_t af = sf ? 0u - (uint32_t)f : (uint32_t)f, ag = sg ? 0u - (uint32_t)g : (uint32_t)g;
    const uint32_t nm = sf != sg ? ~0u : 0u;
    uint32_t y2[N + 1], Z[N + 1];
    unsigned cy = nm & 1u;
#pragma unroll
    for (int j = 0; j < N; j++) y2[j] = __builtin_addc(y[j] ^ nm, 0u, cy, &cy);
    y2[N] = __builtin_addc(nm, 0u, cy, &cy);   // the sign limb (0 again when y == 0)
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < N; j++) {
        c = (c >> 32) + (uint64_t)af * x[j] + (uint64_t)ag * y2[j];   // < 2^32 + 2^62
        Z[j] = (uint32_t)c;
    }
    c = (c >> 32) + (uint64_t)ag * y2[N];
    Z[N] = (uint32_t)c;
    const bool inner_neg = (Z[N] >> 31) != 0;
    const uint32_t zm = inner_neg ? ~0u : 0u;
    cy = zm & 1u;
#pragma unroll
    for (int j = 0; j <= N; j++) Z[j] = __builtin_addc(Z[j] ^ zm, 0u, cy, &cy);
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = (Z[j] >> 30) | (Z[j + 1] << 2);
    return sf != inner_neg;
}
// (u f + v g) / 2^30 mod p for u, v in [0, p); result in [0, p).  A negative factor times u is the positive
// factor times (p - u), so Z = |f| U + |g| V <= 2^30 p; q makes Z + q p divisible by 2^30 and
// (Z + q p) / 2^30 < 2 p: one product-scanning pass of three multiply-adds per limb, one conditional subtraction.
template <int N>
__device__ __forceinline__ void bingcd_update_uv(uint32_t (&out)[N], const uint32_t (&u)[N], const uint32_t (&v)[N],
                                                 int32_t f, int32_t g, const uint32_t (&m)[N], uint32_t n0inv) {
    const bool sf = f < 0, sg = g < 0;
    const uint32_t af = sf ? 0u - (uint32_t)f : (uint32_t)f, ag = sg ? 0u - (uint32_t)g : (uint32_t)g;
    uint32_t U[N], V[N], t[N];
    sub_limbs<N>(t, m, u);
    const uint32_t mf = sf ? ~0u : 0u, mg = sg ? ~0u : 0u;
#pragma unroll
    for (int j = 0; j < N; j++) U[j] = (t[j] & mf) | (u[j] & ~mf);
    sub_limbs<N>(t, m, v);
#pragma unroll
    for (int j = 0; j < N; j++) V[j] = (t[j] & mg) | (v[j] & ~mg);
    uint32_t Z[N + 1];
    uint64_t c = (uint64_t)af * U[0] + (uint64_t)ag * V[0];
    const uint32_t q = ((uint32_t)c * n0inv) & 0x3FFFFFFFu;
    c += (uint64_t)q * m[0];                                          // each product < 2^62
    Z[0] = (uint32_t)c;
#pragma unroll
    for (int j = 1; j < N; j++) {
        c = (c >> 32) + (uint64_t)af * U[j] + (uint64_t)ag * V[j] + (uint64_t)q * m[j];
        Z[j] = (uint32_t)c;
    }
    Z[N] = (uint32_t)(c >> 32);
    uint32_t w[N + 1], d[N + 1], mm[N + 1];
#pragma unroll
    for (int j = 0; j < N; j++) { w[j] = (Z[j] >> 30) | (Z[j + 1] << 2); mm[j] = m[j]; }
    w[N] = Z[N] >> 30;
    mm[N] = 0;
    const uint32_t bo = sub_limbs<N + 1>(d, w, mm);   // w < 2p: subtract p if it fits
    const uint32_t keep = 0u - bo;                     // all-ones: w < p, keep w
#pragma unroll
    for (int j = 0; j < N; j++) out[j] = (w[j] & keep) | (d[j] & ~keep);
}

template <int N>
__device__ __forceinline__ bool inv_bingcd(uint32_t (&out)[N], const uint32_t (&y)[N], const FpParams<N>& P) {
    constexpr int K = 30;
    uint32_t a[N], b[N], u[N], v[N], m[N], t[N];
    set_limbs<N>(m, P.m);
#pragma unroll
    for (int i = 0; i < N; i++) { a[i] = y[i]; b[i] = m[i]; u[i] = 0; v[i] = 0; }
    u[0] = 1;
    // keep the input below p (expected canonical; stay total)
    for (int guard = 0; guard < 2; guard++) {
        const uint32_t bo = sub_limbs<N>(t, a, m);
        const uint32_t keep = 0u - bo;
#pragma unroll
        for (int i = 0; i < N; i++) a[i] = (a[i] & keep) | (t[i] & ~keep);
    }
    const int rounds = (2 * (int)P.bitlen - 1 + K - 1) / K + 2;   // wave-uniform
    for (int it = 0; it < rounds; it++) {
        // ---- 62-bit approximations: top 32 bits (aligned on the longer of a, b) + low 30 bits
        uint32_t ax2 = a[1], ax1 = a[0], bx2 = b[1], bx1 = b[0], wtop = 1;
        bool wide = false;
        if constexpr (N > 2) {
#pragma unroll
            for (int i = 2; i < N; i++) {
                const uint32_t o = a[i] | b[i];
                const bool nz = o != 0;
                ax2 = nz ? a[i] : ax2; ax1 = nz ? a[i - 1] : ax1;
                bx2 = nz ? b[i] : bx2; bx1 = nz ? b[i - 1] : bx1;
                wtop = nz ? o : wtop;
                wide = wide || nz;
            }
        }
        const int lz = __builtin_clz(wtop);
        const uint64_t ta = ((((uint64_t)ax2 << 32) | ax1) << lz) >> 32;
        const uint64_t tb = ((((uint64_t)bx2 << 32) | bx1) << lz) >> 32;
        uint64_t abar = wide ? ((ta << 30) | (a[0] & 0x3FFFFFFFu)) : (((uint64_t)a[1] << 32) | a[0]);
        uint64_t bbar = wide ? ((tb << 30) | (b[0] & 0x3FFFFFFFu)) : (((uint64_t)b[1] << 32) | b[0]);
        // ---- K divsteps on the approximations
        uint32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;   // two's complement, |.| <= 2^30
        // One 32-bit mask per decision, applied to both halves of a, b with bit-field inserts.  The swap mask
        // comes out of a three-instruction asm so that the compiler keeps it a VGPR mask (seen through, the
        // comparison turns every select below into a v_cndmask on an SGPR pair and the masks are built twice).
#pragma unroll 2
        for (int i = 0; i < K; i++) {
            const uint32_t oddm = 0u - ((uint32_t)abar & 1u);
            uint32_t swm;
            asm("v_cmp_lt_u64 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, 0, %3, vcc" : "=v"(swm) : "v"(abar), "v"(bbar), "v"(oddm) : "vcc");
            const uint32_t alo = (uint32_t)abar, ahi = (uint32_t)(abar >> 32), blo = (uint32_t)bbar, bhi = (uint32_t)(bbar >> 32);
            const uint32_t nalo = (blo & swm) | (alo & ~swm), nahi = (bhi & swm) | (ahi & ~swm);
            uint32_t nblo = (alo & swm) | (blo & ~swm), nbhi = (ahi & swm) | (bhi & ~swm);
            uint32_t nf0 = (f1 & swm) | (f0 & ~swm), nf1 = (f0 & swm) | (f1 & ~swm);
            uint32_t ng0 = (g1 & swm) | (g0 & ~swm), ng1 = (g0 & swm) | (g1 & ~swm);
            asm("" : "+v"(nblo), "+v"(nbhi), "+v"(nf1), "+v"(ng1));   // keep the selected values as they are (no re-expansion into the masks)
            bbar = ((uint64_t)nbhi << 32) | nblo;
            unsigned bo = 0;
            const uint32_t dlo = __builtin_subc(nalo, nblo & oddm, bo, &bo);
            const uint32_t dhi = __builtin_subc(nahi, nbhi & oddm, bo, &bo);
            abar = (((uint64_t)dhi << 32) | dlo) >> 1;
            f0 = nf0 - (nf1 & oddm);
            g0 = ng0 - (ng1 & oddm);
            f1 = nf1 << 1;
            g1 = ng1 << 1;
        }
        // ---- apply to the full-width values
        uint32_t na[N], nb[N], nu[N], nv[N];
        int32_t F0 = (int32_t)f0, G0 = (int32_t)g0, F1 = (int32_t)f1, G1 = (int32_t)g1;
        const bool nega = bingcd_update_ab<N>(na, a, b, F0, G0);
        const bool negb = bingcd_update_ab<N>(nb, a, b, F1, G1);
        F0 = nega ? -F0 : F0; G0 = nega ? -G0 : G0;
        F1 = negb ? -F1 : F1; G1 = negb ? -G1 : G1;
        bingcd_update_uv<N>(nu, u, v, F0, G0, m, P.n0inv);
        bingcd_update_uv<N>(nv, u, v, F1, G1, m, P.n0inv);
#pragma unroll
        for (int i = 0; i < N; i++) { a[i] = na[i]; b[i] = nb[i]; u[i] = nu[i]; v[i] = nv[i]; }
        // once a == 0, b holds the gcd and further rounds change nothing: leave as soon as that is
        // true in every lane of the wave (the bound above is the worst case; random inputs need
        // about two thirds of it)
        uint32_t anz = 0;
#pragma unroll
        for (int i = 0; i < N; i++) anz |= a[i];
        if (__ballot(anz != 0) == 0) break;
    }
    uint32_t nz = b[0] ^ 1u;
#pragma unroll
    for (int i = 1; i < N; i++) nz |= b[i];
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = v[i];
    return nz == 0;
}

// Montgomery-form inverse exactly as MontyField::inv (crypto_bigint_monty.rs:223-237):
// invert the representative, then two Montgomery multiplications by R^2.
template <int N>
__device__ __forceinline__ bool mont_inv_gcd(uint32_t (&out)[N], const uint32_t (&x)[N], const FpParams<N>& P) {
    uint32_t inv[N], r2[N], t[N];
    const bool ok = inv_bingcd<N>(inv, x, P);
    set_limbs<N>(r2, P.r2);
    mont_mul<N>(t, inv, r2, P);
    mont_mul<N>(out, t, r2, P);
    return ok;
}

// ---- batch inversion: Montgomery's trick over interleaved chains -----------------
// n elements form M = ceil(n / C) chains (C is chosen per level by make_inv_plan); chain c is {c + k M : k < C, c + k M < n} (the product
// is commutative, any partition into chains is valid).  Lane t of workgroup b owns chain
// 256 b + t, so a wave reads 64 consecutive elements, and the workgroups in flight at step k
// sweep ONE dense range of memory [k M, (k + 1) M): every HBM channel is busy.  (The first
// version gave each workgroup its own tile of 256 C consecutive elements; in lockstep those
// tile strides land on a fraction of the channels, and the level-0 sweeps ran at 4.2 TB/s.)
//   k_binv_up:   prefix[e] = product of the chain's earlier elements (zeros taken
//                as a unit), total[c] = product of the whole chain.
//   (recurse on the totals until few remain, invert those with the GCD kernel)
//   k_binv_down: walks each chain backwards with the inverted total.
// Chain index of element e at a level with M chains: e % M.
// The running products are Chain<N> values (fp_unsat.cuh) multiplied in the Montgomery domain of
// R' = 2^(29 L) from N = 6 limbs on (Chain::mulr: no pre-shifted operand, values in [0, 2p) for every
// modulus; the derivation is there), so prefixes and totals are stored unreduced where they fit the
// container and only the results k_binv_down writes are canonical.  A chain of c elements costs
// 3 (c - 1) multiplications: the running product STARTS as the first element (no multiplication by one,
// no prefix stored for it), and the walk back ends with out[first] = the running inverse itself.
// Both kernels load the next element of the chain before they multiply by the current one, so a wave
// always has a load in flight behind its arithmetic.
// LP ("limb prefixes", level 0 only): prefixes go to a separate array `praw` in the chain's raw form
// (RawForm<N>::STRIDE 32-bit words per element: the 29-bit limbs as they are), which removes the limbs -> words
// conversion here and the words -> limbs one in k_binv_down at 4 * (STRIDE - N) more bytes per element each way.
template <int N>
struct RawForm {   // a running product as its limbs in memory: RAW 32-bit words, padded to whole 8-byte pairs
    static constexpr int RAW = Chain<N>::RAW, STRIDE = (RAW + 1) & ~1;
};
template <int N>
__device__ __forceinline__ void load_raw(uint32_t (&t)[Chain<N>::RAW], const uint32_t* __restrict__ base, size_t e) {
    constexpr int R = RawForm<N>::RAW, S = RawForm<N>::STRIDE;
    // tiles of 64 consecutive elements, limb pair i of the tile's elements side by side: a wave's access is one 512-byte row
    const uint2* p = reinterpret_cast<const uint2*>(base) + (e >> 6) * (size_t)(S / 2) * 64 + (e & 63);
#pragma unroll
    for (int i = 0; i < S / 2; i++) {
        const uint2 q = p[(size_t)i * 64];
        t[2 * i] = q.x;
        if (2 * i + 1 < R) t[2 * i + 1] = q.y;
    }
}
template <int N>
__device__ __forceinline__ void store_raw(uint32_t* __restrict__ base, size_t e, const uint32_t (&t)[Chain<N>::RAW]) {
    constexpr int R = RawForm<N>::RAW, S = RawForm<N>::STRIDE;
    uint2* p = reinterpret_cast<uint2*>(base) + (e >> 6) * (size_t)(S / 2) * 64 + (e & 63);
#pragma unroll
    for (int i = 0; i < S / 2; i++) p[(size_t)i * 64] = make_uint2(t[2 * i], 2 * i + 1 < R ? t[2 * i + 1] : 0u);
}

template <int N, bool LEVEL0, bool LP>
__global__ __launch_bounds__(kBlock) void k_binv_up(const uint32_t* __restrict__ x, uint32_t* __restrict__ prefix,
                                                    uint32_t* __restrict__ praw, uint32_t* __restrict__ totals,
                                                    uint8_t* __restrict__ zero_flags, size_t n, const int C,
                                                    const FpParams<N> P) {
    const size_t M = (n + C - 1) / C;   // chains at this level
    const size_t chain = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t e = chain;
    if (chain >= M) return;
    Chain<N> run(P);
    uint32_t w[N], v[N], vn[N];
    load_elem<N>(v, x, e);
    {   // the first element: the running product starts as it.  Level 0 takes a zero as a unit (it is flagged
        // and its slot gets 0): ANY unit will do -- the other elements' inverses do not depend on it -- so R mod p.
        const bool more = 1 < C && e + M < n;
        load_elem<N>(vn, x, more ? e + M : e);
        if constexpr (LEVEL0) {
            const bool z = is_zero<N>(v);
            if (zero_flags) zero_flags[e] = z ? 1 : 0;
#pragma unroll
            for (int l = 0; l < N; l++) v[l] = z ? P.one[l] : v[l];
        }
        run.set(v);
#pragma unroll
        for (int l = 0; l < N; l++) v[l] = vn[l];
        e += M;
    }
#pragma unroll 1
    for (int k = 1; k < C && e < n; k++, e += M) {
        // next element of the chain (the current one again at the end: an unconditional load
        // keeps the loop free of a conditional register merge)
        const bool more = k + 1 < C && e + M < n;
        load_elem<N>(vn, x, more ? e + M : e);
        if constexpr (LP) {
            uint32_t t[Chain<N>::RAW];
            run.get_raw(t);
            store_raw<N>(praw, e, t);
        } else {
            run.get_words(w);
            store_elem<N>(prefix, e, w);
        }
        bool z = is_zero<N>(v);
        if constexpr (LEVEL0) {
            if (zero_flags) zero_flags[e] = z ? 1 : 0;
        }
        // Level 0 skips a zero (flagged, its slot gets 0).  Above level 0 a zero total (composite modulus
        // only) must poison its group: keep multiplying so the zero reaches the top, where the GCD inverse
        // rejects it.
        if (!z || !LEVEL0) run.mulr(v);
#pragma unroll
        for (int l = 0; l < N; l++) v[l] = vn[l];
    }
    run.get_words(w);
    store_elem<N>(totals, chain, w);
}

// inv_totals[chain] holds R^2 / (the chain total); poisoned[] is handled by the fallback kernel, not here.
template <int N, bool LEVEL0, bool LP>
__global__ __launch_bounds__(kBlock) void k_binv_down(const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                      const uint32_t* __restrict__ praw,
                                                      const uint32_t* __restrict__ inv_totals, size_t n, const int C,
                                                      const FpParams<N> P) {
    constexpr int RW = LP ? Chain<N>::RAW : N;
    const size_t M = (n + C - 1) / C;   // chains at this level
    const size_t chain = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t first = chain;
    if (chain >= M) return;
    // number of elements in this chain (at most C, because C M >= n)
    const size_t cnt = (n - first + M - 1) / M;
    Chain<N> run(P);
    uint32_t w[N], v[N], vn[N], pre[RW], pn[RW];
    auto load_pre = [&](uint32_t (&t)[RW], size_t e) {
        if constexpr (LP) load_raw<N>(t, praw, e);
        else load_elem<N>(t, out, e);   // the exclusive prefix k_binv_up left in the output slot
    };
    load_elem<N>(w, inv_totals, chain);
    run.set(w);
    load_elem<N>(v, x, first + (cnt - 1) * M);
    if (cnt > 1) load_pre(pre, first + (cnt - 1) * M);
#pragma unroll 1
    for (size_t k = cnt; k-- > 1;) {
        const size_t e = first + k * M;
        const size_t en = e - M;               // next element down the chain
        load_elem<N>(vn, x, en);
        load_pre(pn, k > 1 ? en : e);          // the first element has no prefix: read a valid slot again
        bool z = LEVEL0 && is_zero<N>(v);
        if (z) {
#pragma unroll
            for (int l = 0; l < N; l++) w[l] = 0;
            store_elem<N>(out, e, w);
        } else {
            // Above level 0 a zero total only occurs in a poisoned group (composite
            // modulus); whatever is written below it is recomputed by k_binv_fallback.
            if constexpr (LP) run.mulr_raw_to(w, pre);
            else run.mulr_to(w, pre);
            store_elem<N>(out, e, w);
            run.mulr(v);
        }
#pragma unroll
        for (int l = 0; l < N; l++) v[l] = vn[l];
#pragma unroll
        for (int l = 0; l < RW; l++) pre[l] = pn[l];
    }
    // the first element of the chain: what is left of the running inverse is its inverse
    if (LEVEL0 && is_zero<N>(v)) {
#pragma unroll
        for (int l = 0; l < N; l++) w[l] = 0;
    } else {
        run.get_canonical(w);
    }
    store_elem<N>(out, first, w);
}

// ---- level 0 with SPARSE prefixes (large batches over fields of up to 256 bits, where memory binds the sweeps) ----
// BN254: 2^24 elements move 160 bytes each through the two sweeps above (x, prefix out; prefix, x again, result) for
// 3 multiplications of ~200 issue slots.  Here the up-sweep stores only every kInvSeg-th prefix of a chain, and the
// down-sweep takes a chain in segments of kInvSeg elements: it loads the segment's elements (all in flight together)
// and the one stored prefix, RE-MULTIPLIES the prefixes inside the segment, and walks the segment back.
// 3 + (S - 1) / S multiplications per element instead of 3 for 64 - 64 / S bytes of prefix traffic less: 112 instead
// of 160 bytes per element at S = 4.  Same results as the dense sweeps (a prefix is the same residue however it is
// re-derived; what is written is canonical).  Measured (tools/exp_binv_sparse.py, 2^24 elements, dense -> sparse):
// Goldilocks 0.141 -> 0.112 ms, 128 bits 0.290 -> 0.244, 192 bits 0.471 -> 0.432, BN254 0.641 -> 0.591; 320 bits
// 0.906 -> 0.929 and up lose (arithmetic binds there), and so do batches small enough for a one-level plan
// (3 M BN254 elements 0.145 -> 0.163): the launcher takes this path from 2 levels on and up to 256 bits.
// (Loading the next segment before working on the current one costs a wave of occupancy and loses: 0.630.)
constexpr int kInvSeg = 4;
template <int N>
__global__ __launch_bounds__(kBlock) void k_binv_up_sp(const uint32_t* __restrict__ x, uint32_t* __restrict__ prefix,
                                                       uint32_t* __restrict__ totals, uint8_t* __restrict__ zero_flags,
                                                       size_t n, const int C, const FpParams<N> P) {
    const size_t M = (n + C - 1) / C;
    const size_t chain = (size_t)blockIdx.x * kBlock + threadIdx.x;
    size_t e = chain;
    if (chain >= M) return;
    Chain<N> run(P);
    uint32_t w[N], v[N], vn[N];
    load_elem<N>(v, x, e);
    {
        const bool more = 1 < C && e + M < n;
        load_elem<N>(vn, x, more ? e + M : e);
        const bool z = is_zero<N>(v);
        if (zero_flags) zero_flags[e] = z ? 1 : 0;
#pragma unroll
        for (int l = 0; l < N; l++) v[l] = z ? P.one[l] : v[l];
        run.set(v);
#pragma unroll
        for (int l = 0; l < N; l++) v[l] = vn[l];
        e += M;
    }
#pragma unroll 1
    for (int k = 1; k < C && e < n; k++, e += M) {
        const bool more = k + 1 < C && e + M < n;
        load_elem<N>(vn, x, more ? e + M : e);
        if (k % kInvSeg == 0) {   // the prefix a segment of the down-sweep starts from
            run.get_words(w);
            store_elem<N>(prefix, e, w);
        }
        const bool z = is_zero<N>(v);
        if (zero_flags) zero_flags[e] = z ? 1 : 0;
        if (!z) run.mulr(v);
#pragma unroll
        for (int l = 0; l < N; l++) v[l] = vn[l];
    }
    run.get_words(w);
    store_elem<N>(totals, chain, w);
}
template <int N>
__global__ __launch_bounds__(kBlock) void k_binv_down_sp(const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                         const uint32_t* __restrict__ inv_totals, size_t n, const int C,
                                                         const FpParams<N> P) {
    constexpr int S = kInvSeg;
    const size_t M = (n + C - 1) / C;
    const size_t chain = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t first = chain;
    if (chain >= M) return;
    const size_t cnt = (n - first + M - 1) / M;
    Chain<N> run(P), fw(P);
    uint32_t w[N], xs[S][N], pre[S][N];
    load_elem<N>(w, inv_totals, chain);
    run.set(w);
#pragma unroll 1
    for (size_t kb = ((cnt - 1) / S) * S;; kb -= S) {
        const int len = (int)(cnt - kb < (size_t)S ? cnt - kb : (size_t)S);
        // the segment's elements and its stored prefix: up to S + 1 loads in flight (slots past the end re-read the first)
#pragma unroll
        for (int j = 0; j < S; j++) load_elem<N>(xs[j], x, first + (kb + (j < len ? j : 0)) * M);
        if (kb > 0) load_elem<N>(pre[0], out, first + kb * M);   // the exclusive prefix k_binv_up_sp left in the output slot
        // forward: pre[j] = product of the chain's elements before kb + j (zeros as units)
#pragma unroll
        for (int j = 0; j + 1 < S; j++) {
            if (j + 1 < len) {
                const bool z = is_zero<N>(xs[j]);
                if (kb == 0 && j == 0) {
                    uint32_t v0[N];
#pragma unroll
                    for (int l = 0; l < N; l++) v0[l] = z ? P.one[l] : xs[0][l];
                    fw.set(v0);
                } else {
                    if (j == 0) fw.set(pre[0]);
                    if (!z) fw.mulr(xs[j]);
                }
                fw.get_words(pre[j + 1]);
            }
        }
        // backward
#pragma unroll
        for (int j = S - 1; j >= 0; j--) {
            if (j < len) {
                const size_t k = kb + j;
                const bool z = is_zero<N>(xs[j]);
                if (k == 0) {   // the first element of the chain: what is left of the running inverse is its inverse
                    if (z) {
#pragma unroll
                        for (int l = 0; l < N; l++) w[l] = 0;
                    } else {
                        run.get_canonical(w);
                    }
                    store_elem<N>(out, first, w);
                } else if (z) {
#pragma unroll
                    for (int l = 0; l < N; l++) w[l] = 0;
                    store_elem<N>(out, first + k * M, w);
                } else {
                    run.mulr_to(w, pre[j]);
                    store_elem<N>(out, first + k * M, w);
                    run.mulr(xs[j]);
                }
            }
        }
        if (kb == 0) break;
    }
}

// The last stored level and the exact inversions in ONE kernel (no separate up- and down-sweep over the
// totals, no prefixes through HBM): lane c multiplies its chain {c + k M : k < cnt <= C} of the level's
// elements together, keeping the running products in LDS (raw form, one column per lane, no barrier),
// inverts the chain total exactly and walks back.  The level is small by construction (at most one wave
// per SIMD), so every lane's work is pure latency either way; this takes two launches and two HBM round
// trips of latency out of it.  Zero totals are not skipped here (composite moduli: they poison the group).
constexpr int kTopChainBlock = 64;   // one wave per workgroup: (C - 1) * RAW * 256 bytes of LDS each, four workgroups per CU
constexpr int kTopChainMaxC = kInvTopChainMaxC;   // fp_launch.hpp: the plan fuses a level only up to this chain length
template <int N>
__global__ __launch_bounds__(kTopChainBlock) void k_binv_top_chain(const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                           uint8_t* __restrict__ poisoned,
                                                           uint32_t* __restrict__ any_poisoned, size_t n, const int C,
                                                           const FpParams<N> P) {
    constexpr int RW = Chain<N>::RAW;
    // [C - 1][RW][lane], padded to 40 KiB: at most four of these one-wave workgroups then fit a CU's 160 KiB, so
    // 2^16 chains spread as one wave per SIMD over the whole chip instead of piling up on some CUs
    constexpr int kWords = (kTopChainMaxC - 1) * RW * kTopChainBlock > 10240 ? (kTopChainMaxC - 1) * RW * kTopChainBlock : 10240;
    __shared__ uint32_t binv_lds[kWords];
    const size_t M = (n + C - 1) / C;
    const size_t chain = (size_t)blockIdx.x * kTopChainBlock + threadIdx.x;
    if (chain >= M) return;   // no barrier below: a lane only ever touches its own LDS column
    const int cnt = (int)((n - chain + M - 1) / M);
    uint32_t* col = binv_lds + threadIdx.x;
    Chain<N> run(P);
    uint32_t w[N], v[N], t[RW];
    load_elem<N>(v, x, chain);
    run.set(v);
#pragma unroll 1
    for (int k = 1; k < cnt; k++) {
        load_elem<N>(v, x, chain + (size_t)k * M);
        run.get_raw(t);
#pragma unroll
        for (int l = 0; l < RW; l++) col[((k - 1) * RW + l) * kTopChainBlock] = t[l];
        run.mulr(v);
    }
    run.get_canonical(w);
    uint32_t r[N];
    const bool ok = mont_inv_gcd<N>(r, w, P);
    if (!ok) {
#pragma unroll
        for (int l = 0; l < N; l++) r[l] = 0;
        any_poisoned[0] = 1;   // benign race: every writer stores the same value
    }
    poisoned[chain] = ok ? 0 : 1;
    run.set(r);
#pragma unroll 1
    for (int k = cnt; k-- > 1;) {
        load_elem<N>(v, x, chain + (size_t)k * M);
#pragma unroll
        for (int l = 0; l < RW; l++) t[l] = col[((k - 1) * RW + l) * kTopChainBlock];
        run.mulr_raw_to(w, t);
        store_elem<N>(out, chain + (size_t)k * M, w);
        run.mulr(v);
    }
    run.get_canonical(w);
    store_elem<N>(out, chain, w);
}

// Top level: invert each remaining total with the exact GCD inverse.  A total that
// is not invertible (composite modulus only) marks its group as poisoned.
template <int N>
__global__ __launch_bounds__(kBlock) void k_binv_top(uint32_t* __restrict__ totals, uint8_t* __restrict__ poisoned,
                                                     uint32_t* __restrict__ any_poisoned, size_t n,
                                                     const FpParams<N> P) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t v[N], r[N];
    load_elem<N>(v, totals, i);
    bool ok = mont_inv_gcd<N>(r, v, P);
    if (!ok) {
#pragma unroll
        for (int l = 0; l < N; l++) r[l] = 0;
    }
    poisoned[i] = ok ? 0 : 1;
    if (!ok) any_poisoned[0] = 1;   // benign race: every writer stores the same value
    store_elem<N>(totals, i, r);
}

// Fallback for composite moduli: every level-0 element whose top-level group is
// poisoned is inverted on its own (exactly the reference's per-element inv).
// The chain map e -> e % M_l, applied once per level, gives the element's top-level group.
struct InvChains {
    unsigned long long m[kInvMaxLevels];  // chains per level
};
template <int N>
__global__ __launch_bounds__(kBlock) void k_binv_fallback(const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                          uint8_t* __restrict__ zero_flags,
                                                          const uint8_t* __restrict__ poisoned,
                                                          const uint32_t* __restrict__ any_poisoned, size_t n,
                                                          int levels, const InvChains chains, const FpParams<N> P) {
    if (any_poisoned[0] == 0) return;   // prime modulus: nothing to redo (wave-uniform exit)
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += stride) {
        size_t idx = e;
        for (int l = 0; l < levels; l++) idx %= chains.m[l];
        if (!poisoned[idx]) continue;
        uint32_t v[N], r[N];
        load_elem<N>(v, x, e);
        bool ok = mont_inv_gcd<N>(r, v, P);
        if (!ok) {
#pragma unroll
            for (int l = 0; l < N; l++) r[l] = 0;
        }
        if (zero_flags) zero_flags[e] = ok ? 0 : 1;
        store_elem<N>(out, e, r);
    }
}

// ---- block-level modular sum (used by mat-vec) -----------------------------------
template <int N>
__device__ __forceinline__ void wave_reduce_add(uint32_t (&acc)[N], const FpParams<N>& P) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o[N], r[N];
#pragma unroll
        for (int l = 0; l < N; l++) o[l] = __shfl_down(acc[l], off, 64);
        mod_add<N>(r, acc, o, P);
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = r[l];
    }
}

// ---- dense mat-vec: y[i] = sum_j A[i][j] * x[j] -----------------------------------
// One workgroup per row; lane t takes columns t, t+256, ... (a wave-load of A is 64
// consecutive elements = 2 KiB contiguous; x is re-read by every row and stays
// L2-resident).  Lazy reduction: the plain 2N-limb products of T consecutive terms are
// summed in ONE product-scanning pass (T*N^2 multiply-adds, one set of column shifts),
// added into a per-lane 2N-limb accumulator kept below p*R (wide_acc_add), and the
// Montgomery reduction is paid once per lane per row instead of once per matrix
// element.  T terms may share a pass only when T*p <= R (their sum stays below p*R);
// the host picks T from the modulus' leading zero bits.  The result is the same
// canonical residue as the reference's fold of mul and add.
template <int N, int T>
__global__ __launch_bounds__(kBlock) void k_matvec(const uint32_t* __restrict__ A, const uint32_t* __restrict__ x,
                                                   uint32_t* __restrict__ y, size_t rows, size_t cols,
                                                   const uint32_t* __restrict__ run_if_set, const FpParams<N> P) {
    __shared__ uint32_t part[kBlock / 64][N];
    // as the fallback of k_matvec_mfma this runs only when the digitiser raised its flag
    if (run_if_set && *run_if_set == 0) return;
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const uint32_t* __restrict__ Arow = A + row * cols * N;
        uint32_t S[2 * N];
#pragma unroll
        for (int l = 0; l < 2 * N; l++) S[l] = 0;
        size_t j = threadIdx.x;
        if constexpr (T > 1) {
            for (; j + (size_t)(T - 1) * kBlock < cols; j += (size_t)T * kBlock) {
                uint32_t a[T][N], b[T][N], prod[2 * N];
#pragma unroll
                for (int t = 0; t < T; t++) {
                    load_elem<N>(a[t], Arow, j + (size_t)t * kBlock);
                    load_elem<N>(b[t], x, j + (size_t)t * kBlock);
                }
                mul_wide_sum<N, T>(prod, a, b);
                wide_acc_add<N>(S, prod, P);
            }
        }
        for (; j < cols; j += kBlock) {
            uint32_t a[1][N], b[1][N], prod[2 * N];
            load_elem<N>(a[0], Arow, j);
            load_elem<N>(b[0], x, j);
            mul_wide_sum<N, 1>(prod, a, b);
            wide_acc_add<N>(S, prod, P);
        }
        uint32_t acc[N];
        mont_redc_wide<N>(acc, S, P);
        wave_reduce_add<N>(acc, P);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) {
#pragma unroll
            for (int l = 0; l < N; l++) part[wave][l] = acc[l];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot[N], o[N], t[N];
#pragma unroll
            for (int l = 0; l < N; l++) tot[l] = part[0][l];
            for (int w = 1; w < kBlock / 64; w++) {
#pragma unroll
                for (int l = 0; l < N; l++) o[l] = part[w][l];
                mod_add<N>(t, tot, o, P);
#pragma unroll
                for (int l = 0; l < N; l++) tot[l] = t[l];
            }
            store_elem<N>(y, row, tot);
        }
        __syncthreads();
    }
}

// ---- dense mat-vec on unsaturated limbs: carry-free column sums carried ACROSS terms ----------------
// The same sum as k_matvec, computed as k_matmul computes its terms (fp_unsat.cuh, 29-bit limbs): a term a*b is L^2
// v_mad_u64_u32 into 2L - 1 independent 64-bit column sums that live across the terms of a row -- no carry
// instruction, no per-term reduction -- normalised every floor(63 / L) terms.  One row per WAVE: lane t takes
// columns t, t + 64, ... (a wave-load of A is 64 consecutive elements), the next element's loads are issued before
// the current element's multiplies, the 64 lanes' column sums are added with shuffles (plain integers: no modular
// step), and the exact integer V = sum of a*b is rebuilt and reduced once per row as in k_matmul:
// V = W0 + W1 R + W2 R^2  =>  V R^-1 = mont(W0, 1) + mont(W1, R) + mont(W2, R^2)  (mod p), the canonical residue,
// i.e. the reference's fold of mul and add, for any a, b < R and any number of columns below 2^32.
// Against k_matvec (N^2 multiply-adds + N^2 carry adds + a wide add per term): 81 + ~45 instead of ~160 instructions
// per 256-bit term, 196 + ~100 instead of ~330 per 384-bit term -- the wide fields stop being VALU-bound.
template <int N>
__global__ __launch_bounds__(kBlock) void k_matvec_cols(const uint32_t* __restrict__ A, const uint32_t* __restrict__ x,
                                                        uint32_t* __restrict__ y, size_t rows, size_t cols,
                                                        const uint32_t* __restrict__ run_if_set, const FpParams<N> P,
                                                        uint32_t parts, size_t per_part) {
    constexpr int L = UF<N>::L, NC = 2 * L + 2, TT = 63 / L;   // TT * L products of < 2^58 and a carry fit 64 bits
    constexpr uint32_t M = UF<N>::C::MASK;
    if (run_if_set && *run_if_set == 0) return;
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * (kBlock / 64);
    // parts > 1 (few rows, long ones): a wave takes columns [part * per_part, (part + 1) * per_part) of its row and
    // writes the canonical residue of that partial sum to y[row * parts + part] -- y is then the launcher's scratch,
    // and k_matvec_combine adds a row's partials (the sum is the same canonical residue however it is split)
    const size_t items = rows * parts;
    for (size_t item = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); item < items; item += waves) {
        const size_t row = parts > 1 ? item / parts : item;
        const size_t cbeg = parts > 1 ? (item % parts) * per_part : 0;
        const size_t cend = parts > 1 && cbeg + per_part < cols ? cbeg + per_part : cols;
        const uint32_t* __restrict__ Arow = A + row * cols * N;
        uint64_t c[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = 0;
        auto normalise = [&]() {
#pragma unroll
            for (int i = 0; i + 1 < NC; i++) {
                c[i + 1] += c[i] >> kUB;
                c[i] &= M;
            }
        };
        uint32_t va[N], vx[N];
        size_t j = cbeg + lane;
        bool have = j < cend;
        if (have) {
            load_elem<N>(va, Arow, j);
            load_elem<N>(vx, x, j);
        }
        int pending = 0;
        while (have) {
            uint32_t a[L], b[L];
            uload<N>(a, va);
            uload<N>(b, vx);
            const size_t jn = j + 64;
            const bool hn = jn < cend;
            if (hn) {   // in flight over this element's multiplies
                load_elem<N>(va, Arow, jn);
                load_elem<N>(vx, x, jn);
            }
#pragma unroll
            for (int i = 0; i < L; i++)
#pragma unroll
                for (int k = 0; k < L; k++) c[i + k] += (uint64_t)a[i] * b[k];
            if (++pending == TT) {
                normalise();
                pending = 0;
            }
            j = jn;
            have = hn;
        }
        normalise();   // every column below 2^29 (the top one holds what was carried out: small)
        // 64 lanes x < 2^30 per column: the wave's sums stay far below 2^64
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int i = 0; i < NC; i++) c[i] += __shfl_down((unsigned long long)c[i], off, 64);
        normalise();
        normalise();
        uint32_t limbs[NC], w[3 * N];
#pragma unroll
        for (int i = 0; i < NC; i++) limbs[i] = (uint32_t)c[i];
        from_limbs<3 * N, kUB, NC>(w, limbs);   // V < cols * R^2 < 2^(64 N + 32): words beyond that are zero
        uint32_t w0[N], w1[N], w2[N], one[N], r1[N], r2[N], t0[N], t1[N], t2[N], s[N], acc[N];
#pragma unroll
        for (int l = 0; l < N; l++) { w0[l] = w[l]; w1[l] = w[N + l]; w2[l] = w[2 * N + l]; one[l] = l == 0 ? 1u : 0u; }
        set_limbs<N>(r1, P.one);
        set_limbs<N>(r2, P.r2);
        mont_mul<N>(t0, w0, one, P);
        mont_mul<N>(t1, w1, r1, P);
        mont_mul<N>(t2, w2, r2, P);
        mod_add<N>(s, t0, t1, P);
        mod_add<N>(acc, s, t2, P);
        if (lane == 0) store_elem<N>(y, item, acc);
    }
}
// ---- dense mat-vec, one row per LANE: many short rows (tall, narrow matrices) -------------------------------
// k_matvec_cols pays a wave-wide sum and one reduction per ROW of 64 lanes; with 16 or 64 columns that is most of its
// instructions (2^21 rows x 64 columns: 0.73 TB/s).  Here a lane owns a row: the same carry-free column sums, no
// cross-lane step at all, and the per-row reduction runs once per lane.  x[j] is the same address in every lane (a
// scalar load; its limbs are wave-uniform); a lane walks its own row U elements at a time, i.e. whole 128-byte lines,
// and the 64 rows of a wave are one contiguous piece of the matrix.
template <int N>
__global__ __launch_bounds__(kBlock) void k_matvec_rows(const uint32_t* __restrict__ A, const uint32_t* __restrict__ x,
                                                        uint32_t* __restrict__ y, size_t rows, size_t cols, const FpParams<N> P,
                                                        size_t row_stride, size_t col_stride, uint32_t parts, size_t per_part) {
    // Element (row, j) of the matrix is A[row * row_stride + j * col_stride] (in elements): (cols, 1) for y = A x; (1, n)
    // for a VECTOR TIMES A MATRIX, c = a B with B k x n row-major -- then "row" is an output column, consecutive lanes
    // read consecutive elements of one row of B, and x[j] = a[j] is still a scalar load.  parts > 1 (few outputs over
    // many terms): item = part * rows + row takes terms [part * per_part, (part + 1) * per_part) and writes its partial
    // residue to y[row * parts + part] -- the launcher's scratch, summed per row by k_matvec_combine.
    constexpr int L = UF<N>::L, NC = 2 * L + 2, TT = 63 / L, U = N <= 8 ? 4 : 2;
    constexpr uint32_t M = UF<N>::C::MASK;
    const size_t stride = (size_t)gridDim.x * kBlock;
    const size_t items = rows * parts;
    for (size_t item = (size_t)blockIdx.x * kBlock + threadIdx.x; item < items; item += stride) {
        // (parts are the slow index: consecutive lanes stay on consecutive rows, i.e. on consecutive elements of B)
        const size_t row = parts > 1 ? item % rows : item;
        const size_t jb = parts > 1 ? (item / rows) * per_part : 0;
        const size_t je = parts > 1 && jb + per_part < cols ? jb + per_part : cols;
        const uint32_t* __restrict__ Arow = A + row * row_stride * N;
        uint64_t c[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) c[i] = 0;
        auto normalise = [&]() {
#pragma unroll
            for (int i = 0; i + 1 < NC; i++) {
                c[i + 1] += c[i] >> kUB;
                c[i] &= M;
            }
        };
        int pending = 0;
        auto term = [&](const uint32_t (&va)[N], size_t j) {
            uint32_t vx[N], a[L], b[L];
            load_elem<N>(vx, x, j);
            uload<N>(a, va);
            uload<N>(b, vx);
#pragma unroll
            for (int i = 0; i < L; i++)
#pragma unroll
                for (int k = 0; k < L; k++) c[i + k] += (uint64_t)a[i] * b[k];
            if (++pending == TT) {
                normalise();
                pending = 0;
            }
        };
        size_t j = jb;
        for (; j + U <= je; j += U) {
            uint32_t va[U][N];
#pragma unroll
            for (int u = 0; u < U; u++) load_elem<N>(va[u], Arow, (j + u) * col_stride);
#pragma unroll
            for (int u = 0; u < U; u++) term(va[u], j + u);
        }
        for (; j < je; j++) {
            uint32_t va[N];
            load_elem<N>(va, Arow, j * col_stride);
            term(va, j);
        }
        normalise();
        normalise();
        uint32_t limbs[NC], w[3 * N];
#pragma unroll
        for (int i = 0; i < NC; i++) limbs[i] = (uint32_t)c[i];
        from_limbs<3 * N, kUB, NC>(w, limbs);
        uint32_t w0[N], w1[N], w2[N], one[N], r1[N], r2[N], t0[N], t1[N], t2[N], s[N], acc[N];
#pragma unroll
        for (int l = 0; l < N; l++) { w0[l] = w[l]; w1[l] = w[N + l]; w2[l] = w[2 * N + l]; one[l] = l == 0 ? 1u : 0u; }
        set_limbs<N>(r1, P.one);
        set_limbs<N>(r2, P.r2);
        mont_mul<N>(t0, w0, one, P);
        mont_mul<N>(t1, w1, r1, P);
        mont_mul<N>(t2, w2, r2, P);
        mod_add<N>(s, t0, t1, P);
        mod_add<N>(acc, s, t2, P);
        store_elem<N>(y, parts > 1 ? row * parts + item / rows : item, acc);
    }
}

// y[row] = sum of the row's `parts` partial residues (one wave per row)
template <int N>
__global__ __launch_bounds__(kBlock) void k_matvec_combine(const uint32_t* __restrict__ partial, uint32_t* __restrict__ y,
                                                           size_t rows, uint32_t parts, const FpParams<N> P) {
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * (kBlock / 64);
    for (size_t row = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < rows; row += waves) {
        uint32_t acc[N], v[N], t[N];
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = 0;
        for (uint32_t i = lane; i < parts; i += 64) {
            load_elem<N>(v, partial, row * parts + i);
            mod_add<N>(t, acc, v, P);
#pragma unroll
            for (int l = 0; l < N; l++) acc[l] = t[l];
        }
        wave_reduce_add<N>(acc, P);
        if (lane == 0) store_elem<N>(y, row, acc);
    }
}

// ---- sparse mat-vec over SparseMatrix<T> (src/matrix.rs:46-61: `density`
// cells (column index, value) per row, rows stored back to back) -----------------------
// y[i] = sum_{j < density} mul(val[i*density + j], x[col[i*density + j]]): the same fold of mul and
// add as the dense product.  G = 2^log2g lanes share a row (chosen by the launcher, at most 64): values are read coalesced, x is gathered, products are accumulated
// lazily per lane (as k_matvec) and the G partial residues are summed with a butterfly.
// A column index >= cols contributes nothing (the host entry point rejects such input).
// A lane takes its cells U at a time: the U column indices and values are loaded first, then the U elements of
// x are gathered together, then the products are formed -- one dependent memory latency (index -> x) per U
// cells instead of one per cell, and no branch around a load (a cell that does not exist, or whose column is
// out of range, reads a valid address and multiplies by zero).
template <int N, int U>
__global__ __launch_bounds__(kBlock) void k_spmv(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ colidx,
                                                 const uint32_t* __restrict__ x, uint32_t* __restrict__ y, size_t rows,
                                                 size_t cols, uint32_t density, uint32_t log2g, const FpParams<N> P) {
    const uint32_t G = 1u << log2g;
    const size_t rows_per_pass = ((size_t)gridDim.x * kBlock) >> log2g;
    const size_t gid = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const uint32_t sub = (uint32_t)(gid & (G - 1));
    for (size_t row0 = 0; row0 < rows; row0 += rows_per_pass) {  // same trip count for every lane: shuffles below
        const size_t row = row0 + (gid >> log2g);
        uint32_t S[2 * N];
#pragma unroll
        for (int l = 0; l < 2 * N; l++) S[l] = 0;
        if (row < rows && cols > 0) {
            for (uint32_t j0 = sub; j0 < density; j0 += U * G) {
                uint32_t c[U], a[U][1][N], b[U][1][N];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t j = j0 + (uint32_t)u * G;
                    ok[u] = j < density;
                    const size_t cell = row * density + (ok[u] ? j : j0);   // j0 < density: a cell that exists
                    c[u] = colidx[cell];
                    load_elem<N>(a[u][0], vals, cell);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    ok[u] = ok[u] && c[u] < cols;
                    load_elem<N>(b[u][0], x, ok[u] ? c[u] : 0u);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!ok[u]) {
#pragma unroll
                        for (int l = 0; l < N; l++) a[u][0][l] = 0;
                    }
                    uint32_t prod[2 * N];
                    mul_wide_sum<N, 1>(prod, a[u], b[u]);
                    wide_acc_add<N>(S, prod, P);
                }
            }
        }
        uint32_t acc[N];
        mont_redc_wide<N>(acc, S, P);
        for (uint32_t off = G >> 1; off >= 1; off >>= 1) {
            uint32_t o[N], r[N];
#pragma unroll
            for (int l = 0; l < N; l++) o[l] = __shfl_xor(acc[l], (int)off, 64);
            mod_add<N>(r, acc, o, P);
#pragma unroll
            for (int l = 0; l < N; l++) acc[l] = r[l];
        }
        if (row < rows && sub == 0) store_elem<N>(y, row, acc);
    }
}

// ---- dense mat-mul: C = A (m x k) * B (k x n), LDS-tiled, multi-limb -----------------
// 16 x 16 outputs per workgroup, one per lane; K walked in 16-wide tiles staged in LDS as
// unsaturated 29-bit limbs (fp_unsat.cuh: L limbs per element, converted once per tile element).
// A term is L^2 v_mad_u64_u32 into 2L - 1 independent 64-bit COLUMN sums that live across terms:
// a limb product is < 2^58, so a column takes T * L of them before it must be normalised (carry
// = column >> 29 into the next column); no carry instruction, no shift and no reduction inside a
// term (81 multiplies per 256-bit term against 64 + 64 carry adds + a wide add before).  The exact
// integer V = sum of a*b is rebuilt once per output and reduced with three Montgomery
// multiplications, V = W0 + W1 R + W2 R^2  =>  V R^-1 = mont(W0, 1) + mont(W1, R) + mont(W2, R^2)
// (mod p): the canonical residue, i.e. the reference's fold of mul and add, for any a, b < R.
// (Single-limb fields take k_matmul_mfma_1limb / k_matmul_1limb instead.)
template <int L>
struct MMTerms {   // terms a column may absorb between normalisations: T * L + 1 <= 64, T | 16
    static constexpr int T = 63 / L >= 16 ? 16 : (63 / L >= 8 ? 8 : (63 / L >= 4 ? 4 : (63 / L >= 2 ? 2 : 1)));
};
template <int N>
__global__ __launch_bounds__(kMMTile* kMMTile) void k_matmul(const uint32_t* __restrict__ A,
                                                             const uint32_t* __restrict__ B,
                                                             uint32_t* __restrict__ Cm, size_t m, size_t k, size_t n,
                                                             const FpParams<N> P, uint32_t kparts, size_t kper,
                                                             const uint32_t* __restrict__ run_if_set) {
    // as the fallback of k_matmul_mfma_limbs this runs only when the digitiser raised its flag
    if (run_if_set && *run_if_set == 0) return;
    // kparts > 1 (a small output over a long k: too few tiles to fill the chip): blockIdx.z takes the terms
    // [z * kper, (z + 1) * kper) and writes the canonical residue of its partial sum to Cm[(row * n + col) * kparts + z]
    // -- Cm is then the launcher's scratch, and k_matvec_combine adds the parts of every output
    constexpr int L = UF<N>::L, NC = 2 * L + 2, T = MMTerms<L>::T;
    constexpr uint32_t M = UF<N>::C::MASK;
    __shared__ uint32_t sA[kMMTile][kMMTile][L];
    __shared__ uint32_t sB[kMMTile][kMMTile][L];
    const int tx = threadIdx.x % kMMTile, ty = threadIdx.x / kMMTile;
    const size_t row = (size_t)blockIdx.y * kMMTile + ty;
    const size_t col = (size_t)blockIdx.x * kMMTile + tx;
    uint64_t c[NC];   // column sums; c[2L-1 ..] only receive carries
#pragma unroll
    for (int i = 0; i < NC; i++) c[i] = 0;
    auto normalise = [&]() {
#pragma unroll
        for (int i = 0; i + 1 < NC; i++) {
            c[i + 1] += c[i] >> kUB;
            c[i] &= M;
        }
    };
    const size_t kb = kparts > 1 ? (size_t)blockIdx.z * kper : 0;
    const size_t ke = kparts > 1 && kb + kper < k ? kb + kper : k;
    for (size_t k0 = kb; k0 < ke; k0 += kMMTile) {
        uint32_t v[N], lv[L];
        if (row < m && k0 + tx < ke) load_elem<N>(v, A, row * k + k0 + tx);
        else {
#pragma unroll
            for (int l = 0; l < N; l++) v[l] = 0;
        }
        uload<N>(lv, v);
#pragma unroll
        for (int l = 0; l < L; l++) sA[ty][tx][l] = lv[l];
        if (k0 + ty < ke && col < n) load_elem<N>(v, B, (k0 + ty) * n + col);
        else {
#pragma unroll
            for (int l = 0; l < N; l++) v[l] = 0;
        }
        uload<N>(lv, v);
#pragma unroll
        for (int l = 0; l < L; l++) sB[ty][tx][l] = lv[l];
        __syncthreads();
#pragma unroll 1
        for (int kk = 0; kk < kMMTile; kk += T) {
#pragma unroll
            for (int t = 0; t < T; t++) {
                uint32_t a[L], b[L];
#pragma unroll
                for (int l = 0; l < L; l++) { a[l] = sA[ty][kk + t][l]; b[l] = sB[kk + t][tx][l]; }
#pragma unroll
                for (int i = 0; i < L; i++)
#pragma unroll
                    for (int j = 0; j < L; j++) c[i + j] += (uint64_t)a[i] * b[j];
            }
            normalise();
        }
        __syncthreads();
    }
    // the top column has absorbed every carry: spread it over the spare limbs, then all limbs are < 2^29
    normalise();
    normalise();
    uint32_t limbs[NC], w[3 * N];
#pragma unroll
    for (int i = 0; i < NC; i++) limbs[i] = (uint32_t)c[i];
    from_limbs<3 * N, kUB, NC>(w, limbs);   // V < k * R^2 < 2^(64 N + 64): words beyond that are zero
    uint32_t w0[N], w1[N], w2[N], one[N], r1[N], r2[N], t0[N], t1[N], t2[N], s[N], acc[N];
#pragma unroll
    for (int l = 0; l < N; l++) { w0[l] = w[l]; w1[l] = w[N + l]; w2[l] = w[2 * N + l]; one[l] = l == 0 ? 1u : 0u; }
    set_limbs<N>(r1, P.one);
    set_limbs<N>(r2, P.r2);
    mont_mul<N>(t0, w0, one, P);
    mont_mul<N>(t1, w1, r1, P);
    mont_mul<N>(t2, w2, r2, P);
    mod_add<N>(s, t0, t1, P);
    mod_add<N>(acc, s, t2, P);
    if (row < m && col < n) store_elem<N>(Cm, kparts > 1 ? (row * n + col) * kparts + blockIdx.z : row * n + col, acc);
}


// ---- batched pow with one exponent for the whole batch ---------------
// SemiringConfig::pow_u32 / FieldConfig::pow (crypto_bigint_monty.rs:163-172,239-249;
// helpers/crypto_bigint.rs:220-455, a 4-bit-window "almost Montgomery" ladder).  The result
// is the canonical residue x^e in Montgomery form whatever the ladder.  The exponent is
// wave-uniform (kernel argument), so every decision on its bits is scalar and the waves never
// diverge: a short or sparse exponent runs plain left-to-right square-and-multiply; a dense one a
// sliding window of three bits over the odd powers x, x^3, x^5, x^7 held in registers -- one
// multiplication per window (about bits / 4) instead of one per set bit (bits / 2), for one squaring
// and three multiplications of set-up.  x^0 = 1 (also 0^0), as the reference.
struct PowExp {
    uint32_t w[16];  // exponent, 32-bit little-endian limbs
    int top;         // index of its highest set bit, -1 if the exponent is zero
    int windowed;    // 1: sliding-window ladder (the host decides: more than kPowWindowMinOnes set bits)
};
constexpr int kPowWindowMinOnes = 12;
template <int N>
__global__ __launch_bounds__(kBlock) void k_pow(const uint32_t* __restrict__ a, uint32_t* __restrict__ out, size_t n,
                                                const PowExp e, const FpParams<N> P) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    PowLadder<N> ladder(P);   // fp_unsat.cuh: from N = 6 on, unsaturated limbs in the domain of R' = 2^(29 L)
    auto bit = [&](int b) -> uint32_t { return (e.w[b >> 5] >> (b & 31)) & 1u; };
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t w[N];
        if (e.top < 0) {
            set_limbs<N>(w, P.one);
        } else {
            load_elem<N>(w, a, i);
            if (e.top > 0) {   // exponent one hands the input back untouched
                ladder.start(w);
                if (!e.windowed) {
                    for (int b = e.top - 1; b >= 0; b--) {
                        ladder.square();
                        if (bit(b)) ladder.mul_base();
                    }
                } else {
                    typename PowLadder<N>::Val x1, x3, x5, x7, y;
                    ladder.get_acc(x1);
                    ladder.square();
                    ladder.get_acc(y);        // x^2
                    ladder.mul_val(x1);
                    ladder.get_acc(x3);
                    ladder.mul_val(y);
                    ladder.get_acc(x5);
                    ladder.mul_val(y);
                    ladder.get_acc(x7);
                    bool first = true;
                    for (int b = e.top; b >= 0;) {
                        if (!bit(b)) {        // never the first step: bit `top` is set
                            ladder.square();
                            b--;
                            continue;
                        }
                        int len = b >= 2 ? 3 : b + 1;          // the longest window of <= 3 bits that ends in a one
                        while (!bit(b - len + 1)) len--;
                        uint32_t val = 0;
                        for (int k = 0; k < len; k++) val = 2 * val + bit(b - k);
                        if (!first)
                            for (int k = 0; k < len; k++) ladder.square();
                        // the odd power of this window (a scalar choice: `val` is wave-uniform)
#pragma unroll
                        for (int l = 0; l < (int)(sizeof(y) / sizeof(y[0])); l++)
                            y[l] = val == 1 ? x1[l] : val == 3 ? x3[l] : val == 5 ? x5[l] : x7[l];
                        if (first) ladder.set_acc(y);
                        else ladder.mul_val(y);
                        first = false;
                        b -= len;
                    }
                }
                ladder.finish(w);
            }
        }
        store_elem<N>(out, i, w);
    }
}

// ---- sum / product folds (SemiringConfig::sum_refs / product_refs,
// src/semiring.rs:213-241: left folds from zero / one; the canonical result does not depend
// on the association, so this is a tree).  One partial per workgroup; run twice (slices ->
// partials, partials -> result).
template <int N, bool PROD>
__device__ __forceinline__ void fold2(uint32_t (&acc)[N], const uint32_t (&o)[N], const FpParams<N>& P) {
    uint32_t r[N];
    if constexpr (PROD) mont_mul<N>(r, acc, o, P);
    else mod_add<N>(r, acc, o, P);
#pragma unroll
    for (int l = 0; l < N; l++) acc[l] = r[l];
}
template <int N, bool PROD>
__global__ __launch_bounds__(kBlock) void k_fold(const uint32_t* __restrict__ v, size_t n,
                                                 uint32_t* __restrict__ partials, const FpParams<N> P) {
    __shared__ uint32_t part[kBlock / 64][N];
    uint32_t acc[N], o[N];
    if constexpr (PROD) set_limbs<N>(acc, P.one);
    else {
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = 0;
    }
    const size_t stride = (size_t)gridDim.x * kBlock;
    if constexpr (PROD) {
        // the streaming part of the product is a Chain<N> (fp_unsat.cuh), lazily reduced where possible
        Chain<N> run(P);
        run.set(acc);
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            load_elem<N>(o, v, i);
            run.mul(o);
        }
        run.get_canonical(acc);
    } else {
        for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            load_elem<N>(o, v, i);
            fold2<N, PROD>(acc, o, P);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
        for (int l = 0; l < N; l++) o[l] = __shfl_down(acc[l], off, 64);
        // lanes whose partner is past the wave read their own value back from __shfl_down;
        // folding it would double-count, so only lanes < off fold (the others' results are unused)
        if ((threadIdx.x & 63) < off) fold2<N, PROD>(acc, o, P);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < N; l++) part[wave][l] = acc[l];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; w++) {
#pragma unroll
            for (int l = 0; l < N; l++) o[l] = part[w][l];
            fold2<N, PROD>(acc, o, P);
        }
        store_elem<N>(partials, blockIdx.x, acc);
    }
}

// ---- dot product: sum_i mul(a[i], b[i]), the sum_refs fold of monty_mul
// (src/semiring.rs:213-226 over helpers::mul::monty_mul) -- one row of the mat-vec, spread over the
// whole chip.  Lazy reduction exactly as k_matvec: each lane adds its plain 2N-limb products into a
// wide accumulator kept below p*R and pays one Montgomery reduction at the end; one canonical
// partial per workgroup, summed by k_fold<N, false>.  Operands must be canonical (< p), as for the
// mat-vec: the canonical result then equals the reference's left fold.
template <int N>
__global__ __launch_bounds__(kBlock) void k_dot(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                size_t n, uint32_t* __restrict__ partials, const FpParams<N> P) {
    __shared__ uint32_t part[kBlock / 64][N];
    uint32_t S[2 * N];
#pragma unroll
    for (int l = 0; l < 2 * N; l++) S[l] = 0;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t x[1][N], y[1][N], prod[2 * N];
        load_elem<N>(x[0], a, i);
        load_elem<N>(y[0], b, i);
        mul_wide_sum<N, 1>(prod, x, y);
        wide_acc_add<N>(S, prod, P);
    }
    uint32_t acc[N];
    mont_redc_wide<N>(acc, S, P);
    wave_reduce_add<N>(acc, P);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < N; l++) part[wave][l] = acc[l];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t o[N], t[N];
        for (int w = 1; w < kBlock / 64; w++) {
#pragma unroll
            for (int l = 0; l < N; l++) o[l] = part[w][l];
            mod_add<N>(t, acc, o, P);
#pragma unroll
            for (int l = 0; l < N; l++) acc[l] = t[l];
        }
        store_elem<N>(partials, blockIdx.x, acc);
    }
}

// ---- single-limb (n64 = 1, N = 2) mat-mul: register-tiled, lazily reduced -------------
// BASELINE.json configs[4]: MontyField 64-bit prime (Goldilocks-size) 8192^2 mat-mul.
// Workgroup = 256 lanes computing a 64 x 32 tile of C; lane (ty, tx) owns 4 x 2 outputs.
// K is walked in 32-deep steps staged through LDS (A transposed to [k][row] so a lane's
// four rows are one 32-byte read; B as is), with the next step's global loads held in
// registers while the current step computes.  Every term a*b (64 x 64 -> 128 bits) is four
// v_mad_u64_u32 into three per-output 96-bit accumulators (weights 2^0, 2^32, 2^64) plus
// four carry adds: 8 VALU instructions per term, no reduction, no shifts.  One asm block
// per term keeps each carry two instructions away from its consumer (see fp_device.cuh).
// The 160-bit sums are reduced once per output: with S = s0 + s1 R + s2 R^2,
//   S R^-1 mod p = mont_mul(s0, 1) + mont_mul(s1, R mod p) + mont_mul(s2, R^2 mod p),
// valid for every odd p < 2^64 and any K < 2^32; canonical, hence bit-identical to the
// reference's fold of mul and add.
struct Acc3 {
    uint64_t p, q, r;     // low 64 bits of the three partial-product sums
    uint32_t ph, qh, rh;  // their carry counts
};
__device__ __forceinline__ void mac_1limb(Acc3& c, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
    uint64_t c0, c1, c2;
    asm("v_mad_u64_u32 %0, %6, %9, %11, %0\n\t"
        "v_mad_u64_u32 %1, %7, %9, %12, %1\n\t"
        "v_mad_u64_u32 %2, %8, %10, %12, %2\n\t"
        "v_addc_co_u32 %3, %6, 0, %3, %6\n\t"
        "v_mad_u64_u32 %1, %6, %10, %11, %1\n\t"
        "v_addc_co_u32 %4, %7, 0, %4, %7\n\t"
        "v_addc_co_u32 %5, %8, 0, %5, %8\n\t"
        "v_addc_co_u32 %4, %6, 0, %4, %6"
        : "+v"(c.p), "+v"(c.q), "+v"(c.r), "+v"(c.ph), "+v"(c.qh), "+v"(c.rh), "=&s"(c0), "=&s"(c1), "=&s"(c2)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}

constexpr int kMM1_BM = 64, kMM1_BN = 32, kMM1_BK = 32, kMM1_TM = 4, kMM1_TN = 2;

template <int N>  // N == 2 only; a template so that each limb-count translation unit may include it
__global__ __launch_bounds__(256) void k_matmul_1limb(const uint2* __restrict__ A, const uint2* __restrict__ B,
                                                      uint2* __restrict__ Cm, size_t m, size_t k, size_t n,
                                                      const uint32_t* __restrict__ run_if_set, const FpParams<N> P) {
    static_assert(N == 2, "single-limb kernel");
    // as the fallback of k_matmul_mfma_1limb this runs only when the digitiser raised its flag
    if (run_if_set && *run_if_set == 0) return;
    __shared__ uint2 sA[kMM1_BK][kMM1_BM];  // [k][row]
    __shared__ uint2 sB[kMM1_BK][kMM1_BN];  // [k][col]
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;            // 16 x 16 lanes
    const size_t row0 = (size_t)blockIdx.y * kMM1_BM, col0 = (size_t)blockIdx.x * kMM1_BN;
    // staging roles: A tile 64 rows x 32 k = 2048 elements, 8 per lane (row = tid/4, k = (tid%4)*8 ..+7)
    const int a_r = tid >> 2, a_k = (tid & 3) * 8;
    // B tile 32 k x 32 cols = 1024 elements, 4 per lane (k = tid/8, col = (tid%8)*4 ..+3)
    const int b_k = tid >> 3, b_c = (tid & 7) * 4;
    uint2 ra[8], rb[4];
    auto fetch = [&](size_t k0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            size_t r = row0 + a_r, kk = k0 + a_k + i;
            ra[i] = (r < m && kk < k) ? A[r * k + kk] : make_uint2(0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            size_t kk = k0 + b_k, c = col0 + b_c + i;
            rb[i] = (kk < k && c < n) ? B[kk * n + c] : make_uint2(0, 0);
        }
    };
    Acc3 acc[kMM1_TM][kMM1_TN];
#pragma unroll
    for (int i = 0; i < kMM1_TM; i++)
#pragma unroll
        for (int j = 0; j < kMM1_TN; j++) acc[i][j] = Acc3{0, 0, 0, 0, 0, 0};
    fetch(0);
    for (size_t k0 = 0; k0 < k; k0 += kMM1_BK) {
        __syncthreads();  // previous step's reads are done
#pragma unroll
        for (int i = 0; i < 8; i++) sA[a_k + i][a_r] = ra[i];
#pragma unroll
        for (int i = 0; i < 4; i++) sB[b_k][b_c + i] = rb[i];
        __syncthreads();
        if (k0 + kMM1_BK < k) fetch(k0 + kMM1_BK);  // in flight while this step computes
#pragma unroll 4
        for (int kk = 0; kk < kMM1_BK; kk++) {
            uint2 av[kMM1_TM], bv[kMM1_TN];
#pragma unroll
            for (int i = 0; i < kMM1_TM; i++) av[i] = sA[kk][ty * kMM1_TM + i];
#pragma unroll
            for (int j = 0; j < kMM1_TN; j++) bv[j] = sB[kk][tx * kMM1_TN + j];
#pragma unroll
            for (int i = 0; i < kMM1_TM; i++)
#pragma unroll
                for (int j = 0; j < kMM1_TN; j++) mac_1limb(acc[i][j], av[i].x, av[i].y, bv[j].x, bv[j].y);
        }
    }
    // reduce and store
    uint32_t onep[2] = {1u, 0u}, oneR[2], r2[2];
    set_limbs<2>(oneR, P.one);
    set_limbs<2>(r2, P.r2);
#pragma unroll
    for (int i = 0; i < kMM1_TM; i++)
#pragma unroll
        for (int j = 0; j < kMM1_TN; j++) {
            const size_t r = row0 + ty * kMM1_TM + i, c = col0 + tx * kMM1_TN + j;
            if (r >= m || c >= n) continue;
            const Acc3& a = acc[i][j];
            // S = P + Q 2^32 + R 2^64 as six 32-bit limbs
            uint32_t S[6];
            unsigned cy = 0;
            S[0] = (uint32_t)a.p;
            S[1] = __builtin_addc((uint32_t)(a.p >> 32), (uint32_t)a.q, cy, &cy);
            S[2] = __builtin_addc(a.ph, (uint32_t)(a.q >> 32), cy, &cy);
            S[3] = __builtin_addc(a.qh, 0u, cy, &cy);
            S[4] = cy;
            cy = 0;
            S[2] = __builtin_addc(S[2], (uint32_t)a.r, cy, &cy);
            S[3] = __builtin_addc(S[3], (uint32_t)(a.r >> 32), cy, &cy);
            S[4] = __builtin_addc(S[4], a.rh, cy, &cy);
            S[5] = cy;
            uint32_t s0[2] = {S[0], S[1]}, s1[2] = {S[2], S[3]}, s2[2] = {S[4], S[5]};
            uint32_t t0[2], t1[2], t2[2], u[2], v[2];
            mont_mul<2>(t0, s0, onep, P);
            mont_mul<2>(t1, s1, oneR, P);
            mont_mul<2>(t2, s2, r2, P);
            mod_add<2>(u, t0, t1, P);
            mod_add<2>(v, u, t2, P);
            Cm[r * n + c] = make_uint2(v[0], v[1]);
        }
}

}  // namespace fpgpu
