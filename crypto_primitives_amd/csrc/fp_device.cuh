// fp_device.cuh -- per-lane multi-limb prime-field arithmetic for gfx950 (CDNA4).
//
// One field element lives in N 32-bit limbs in VGPRs of ONE lane (N = 2 * n64, so
// R = 2^(32N) = 2^(64*n64) matches the reference's Montgomery radix and results
// are bit-identical).  The modulus and Montgomery constants arrive as a by-value
// kernel argument, i.e. in SGPRs.
//
// Instruction economics measured on MI355X (tools/ubench_valu.hip, profiles/):
// v_mad_u64_u32, v_addc_co_u32, v_mul_lo_u32 and v_lshl_add_u64 all issue at
// ~4.3 cycles per wave64 per SIMD (36.5 T lane-ops/s chip-wide); v_add_u32 /
// v_mov_b32 at ~2.3.  The multiplier is therefore built to minimise the COUNT of
// 4-cycle instructions: product scanning (FIPS) with one 96-bit column
// accumulator costs exactly one v_mad_u64_u32 + one v_addc_co_u32 per limb
// product, with no zero-extension moves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fpgpu {

template <int N>
struct FpParams {
    uint32_t m[N];    // modulus p
    uint32_t r2[N];   // R^2 mod p
    uint32_t one[N];  // R mod p
    uint32_t n0inv;   // -p^-1 mod 2^32 (low half of the reference's mod_neg_inv)
    uint32_t top_mask; // mask of significant bits in the top 64-bit word of p (fill_random)
    uint32_t top_mask_hi;
};

// ---- 96-bit column accumulator --------------------------------------------
struct Acc96 {
    uint64_t lo;  // bits 0..63   (even-aligned VGPR pair)
    uint32_t hi;  // bits 64..95
};

// acc += a * b   (a in VGPR, b in VGPR)
__device__ __forceinline__ void mac_vv(Acc96& acc, uint32_t a, uint32_t b) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc.lo), "+v"(acc.hi)
        : "v"(a), "v"(b)
        : "vcc");
}
// acc += a * s   (a in VGPR, s wave-uniform in an SGPR: a modulus limb)
__device__ __forceinline__ void mac_vs(Acc96& acc, uint32_t a, uint32_t s) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc.lo), "+v"(acc.hi)
        : "v"(a), "s"(s)
        : "vcc");
}
// acc += x (32-bit)
__device__ __forceinline__ void acc_add32(Acc96& acc, uint32_t x) {
    asm("v_mad_u64_u32 %0, vcc, %2, 1, %0\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(acc.lo), "+v"(acc.hi)
        : "v"(x)
        : "vcc");
}
// acc >>= 32
__device__ __forceinline__ void acc_shift(Acc96& acc) {
    acc.lo = (acc.lo >> 32) | ((uint64_t)acc.hi << 32);
    acc.hi = 0;
}

// ---- carry chains ------------------------------------------------------------
// r = a - m (m in SGPRs); returns borrow (1 if a < m).
template <int N>
__device__ __forceinline__ uint32_t sub_limbs_s(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t* m) {
    uint32_t bo = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a[i] - m[i] - bo;
        r[i] = (uint32_t)t;
        bo = (uint32_t)(t >> 63);
    }
    return bo;
}
template <int N>
__device__ __forceinline__ uint32_t sub_limbs(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
    uint32_t bo = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a[i] - b[i] - bo;
        r[i] = (uint32_t)t;
        bo = (uint32_t)(t >> 63);
    }
    return bo;
}
template <int N>
__device__ __forceinline__ uint32_t add_limbs(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N]) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)a[i] + b[i] + c;
        r[i] = (uint32_t)t;
        c = (uint32_t)(t >> 32);
    }
    return c;
}

template <int N>
__device__ __forceinline__ bool is_zero(const uint32_t (&a)[N]) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < N; i++) z |= a[i];
    return z == 0;
}

// ---- Montgomery multiplication -------------------------------------------------
// r = a * b * R^-1 mod p for any a, b < R; subtracts p once iff the (N+1)-limb
// result T = (a*b + U*p)/R has its overflow bit set or T >= p -- the same rule
// and the same integer T as helpers::mul::monty_mul (src/helpers/crypto_bigint.rs
// :74-107, CIOS :118-159), so the output is bit-identical for every input pair,
// including out-of-range ones.  Finely integrated product scanning (FIPS):
// column k accumulates sum_{i+j=k} a_i b_j + u_i m_j in one 96-bit accumulator.
template <int N>
__device__ __forceinline__ void mont_mul(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N],
                                         const FpParams<N>& P) {
    uint32_t u[N];
    uint32_t t[N];
    Acc96 acc{0, 0};
#pragma unroll
    for (int k = 0; k < N; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) mac_vv(acc, a[i], b[k - i]);
#pragma unroll
        for (int i = 0; i < k; i++) mac_vs(acc, u[i], P.m[k - i]);
        u[k] = (uint32_t)acc.lo * P.n0inv;
        mac_vs(acc, u[k], P.m[0]);
        acc_shift(acc);
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac_vv(acc, a[i], b[k - i]);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac_vs(acc, u[i], P.m[k - i]);
        t[k - N] = (uint32_t)acc.lo;
        acc_shift(acc);
    }
    uint32_t top = (uint32_t)acc.lo;  // overflow limb of T (0 or 1)
    uint32_t d[N];
    uint32_t bo = sub_limbs_s<N>(d, t, P.m);
    bool use_d = (top != 0) | (bo == 0);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = use_d ? d[i] : t[i];
}

// Montgomery reduction of a single N-limb value: r = (x + U*p) / R, NO final
// subtraction -- exactly monty_retrieve_inner (src/helpers/crypto_bigint.rs:13-37),
// which for canonical x < p yields the canonical integer x * R^-1 mod p.
template <int N>
__device__ __forceinline__ void mont_retrieve(uint32_t (&r)[N], const uint32_t (&x)[N], const FpParams<N>& P) {
    uint32_t u[N];
    Acc96 acc{0, 0};
#pragma unroll
    for (int k = 0; k < N; k++) {
        acc_add32(acc, x[k]);
#pragma unroll
        for (int i = 0; i < k; i++) mac_vs(acc, u[i], P.m[k - i]);
        u[k] = (uint32_t)acc.lo * P.n0inv;
        mac_vs(acc, u[k], P.m[0]);
        acc_shift(acc);
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac_vs(acc, u[i], P.m[k - i]);
        r[k - N] = (uint32_t)acc.lo;
        acc_shift(acc);
    }
}


// ---- lazy (unreduced) products for long dot products ---------------------------------
// prod[0..2N) = sum_{t<T} a[t] * b[t] as a plain 2N-limb integer (product scanning, no
// reduction).  The caller guarantees the sum fits 2N limbs (T * p^2 < R^2 always holds
// for the T <= 4 used here when p < R / T; see k_matvec).
template <int N, int T>
__device__ __forceinline__ void mul_wide_sum(uint32_t (&prod)[2 * N], const uint32_t (&a)[T][N],
                                             const uint32_t (&b)[T][N]) {
    Acc96 acc{0, 0};
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
        const int lo = k < N ? 0 : k - N + 1;
        const int hi = k < N ? k : N - 1;
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int i = lo; i <= hi; i++) mac_vv(acc, a[t][i], b[t][k - i]);
        prod[k] = (uint32_t)acc.lo;
        acc_shift(acc);
    }
    prod[2 * N - 1] = (uint32_t)acc.lo;
}

// S += prod with the invariant S < p*R kept (S = S_hi*R + S_lo, S_hi < p): the low halves
// add with a plain carry chain; the high halves add modulo p with that carry folded in.
// Valid whenever prod < p*R (prod_hi < p): any product of two canonical residues, or a sum
// of T such products when T*p <= R.
template <int N>
__device__ __forceinline__ void wide_acc_add(uint32_t (&S)[2 * N], const uint32_t (&prod)[2 * N],
                                             const FpParams<N>& P) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)S[i] + prod[i] + c;
        S[i] = (uint32_t)t;
        c = (uint32_t)(t >> 32);
    }
    uint32_t w[N], d[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)S[N + i] + prod[N + i] + c;
        w[i] = (uint32_t)t;
        c = (uint32_t)(t >> 32);
    }
    uint32_t bo = sub_limbs_s<N>(d, w, P.m);
    bool keep_w = c < bo;
#pragma unroll
    for (int i = 0; i < N; i++) S[N + i] = keep_w ? w[i] : d[i];
}

// r = S * R^-1 mod p for a 2N-limb S < p*R (Montgomery reduction + one conditional
// subtract).  sum_j mont_mul(a_j, b_j) mod p == mont_redc_wide(sum_j a_j*b_j) exactly.
template <int N>
__device__ __forceinline__ void mont_redc_wide(uint32_t (&r)[N], const uint32_t (&S)[2 * N], const FpParams<N>& P) {
    uint32_t u[N], t[N];
    Acc96 acc{0, 0};
#pragma unroll
    for (int k = 0; k < N; k++) {
        acc_add32(acc, S[k]);
#pragma unroll
        for (int i = 0; i < k; i++) mac_vs(acc, u[i], P.m[k - i]);
        u[k] = (uint32_t)acc.lo * P.n0inv;
        mac_vs(acc, u[k], P.m[0]);
        acc_shift(acc);
    }
#pragma unroll
    for (int k = N; k < 2 * N; k++) {
        acc_add32(acc, S[k]);
#pragma unroll
        for (int i = k - N + 1; i < N; i++) mac_vs(acc, u[i], P.m[k - i]);
        t[k - N] = (uint32_t)acc.lo;
        acc_shift(acc);
    }
    uint32_t top = (uint32_t)acc.lo;
    uint32_t d[N];
    uint32_t bo = sub_limbs_s<N>(d, t, P.m);
    bool use_d = (top != 0) | (bo == 0);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = use_d ? d[i] : t[i];
}

// ---- modular add / sub / neg (crypto-bigint add_mod / sub_mod / neg_mod) -------
// add: w = a + b; w' = w - p; keep w' unless the subtraction borrowed past the
// add's carry.  Bit-identical to crypto-bigint's masked form for all inputs.
template <int N>
__device__ __forceinline__ void mod_add(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N],
                                        const FpParams<N>& P) {
    uint32_t s[N], d[N];
    uint32_t c = add_limbs<N>(s, a, b);
    uint32_t bo = sub_limbs_s<N>(d, s, P.m);
    bool keep_s = c < bo;  // (carry - borrow) underflows -> add p back
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = keep_s ? s[i] : d[i];
}
template <int N>
__device__ __forceinline__ void mod_sub(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N],
                                        const FpParams<N>& P) {
    uint32_t d[N];
    uint32_t bo = sub_limbs<N>(d, a, b);
    uint32_t mask = 0u - bo;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)d[i] + (P.m[i] & mask) + c;
        r[i] = (uint32_t)t;
        c = (uint32_t)(t >> 32);
    }
}
template <int N>
__device__ __forceinline__ void mod_neg(uint32_t (&r)[N], const uint32_t (&a)[N], const FpParams<N>& P) {
    bool z = is_zero<N>(a);
    uint32_t bo = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t t = (uint64_t)P.m[i] - a[i] - bo;
        uint32_t v = (uint32_t)t;
        bo = (uint32_t)(t >> 63);
        r[i] = z ? 0u : v;
    }
}

// ---- element load / store (AoS: element i = N consecutive u32 at base + i*N) ----
template <int N>
__device__ __forceinline__ void load_elem(uint32_t (&x)[N], const uint32_t* __restrict__ base, size_t i) {
    if constexpr (N % 4 == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(base + i * N);
#pragma unroll
        for (int k = 0; k < N / 4; k++) {
            uint4 v = p[k];
            x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w;
        }
    } else {
        const uint2* p = reinterpret_cast<const uint2*>(base + i * N);
#pragma unroll
        for (int k = 0; k < N / 2; k++) {
            uint2 v = p[k];
            x[2 * k] = v.x; x[2 * k + 1] = v.y;
        }
    }
}
template <int N>
__device__ __forceinline__ void store_elem(uint32_t* __restrict__ base, size_t i, const uint32_t (&x)[N]) {
    if constexpr (N % 4 == 0) {
        uint4* p = reinterpret_cast<uint4*>(base + i * N);
#pragma unroll
        for (int k = 0; k < N / 4; k++) p[k] = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
    } else {
        uint2* p = reinterpret_cast<uint2*>(base + i * N);
#pragma unroll
        for (int k = 0; k < N / 2; k++) p[k] = make_uint2(x[2 * k], x[2 * k + 1]);
    }
}

template <int N>
__device__ __forceinline__ void set_limbs(uint32_t (&x)[N], const uint32_t* s) {
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = s[i];
}

}  // namespace fpgpu
