// Unsaturated-limb Montgomery arithmetic for the VALU-bound kernels.
//
// On gfx950 a carry add (v_addc_co_u32) costs the same 4 issue cycles as the multiply
// (v_mad_u64_u32) it follows, so a saturated 32-bit-limb product
// scanning multiplier spends half its time on carries.  Here an element is held as L
// limbs of B < 32 bits: limb products are < 2^(2B), and 2L of them plus a carry fit a
// 64-bit accumulator, so every column is a run of v_mad_u64_u32 with NO carry
// instruction; a column ends with one cheap 64-bit shift.
//
// Bit-exactness with the reference (monty_mul, src/helpers/crypto_bigint.rs:74-107,
// R = 2^(32*NW)): the word-by-word reduction here divides by R' = 2^(B*L) > R, so one
// operand is loaded pre-shifted left by SH = B*L - 32*NW bits:
//     (a * (b << SH)) / R'  ==  a * b / R   (mod p),
// and the result is < a*b/R + p < 2p, exactly the range the reference's final
// conditional subtraction handles; after it the canonical residue is the same number.
#pragma once
#include "fp_device.cuh"

namespace fpgpu {

template <int NW, int B>
struct UCfg {
    static constexpr int L = (32 * NW + B - 1) / B;  // limbs
    static constexpr int SH = B * L - 32 * NW;       // pre-shift of one operand
    static constexpr uint32_t MASK = (1u << B) - 1;
    static_assert(B < 32 && SH < B, "limb width");
    // column bound: 2L products < 2^(2B) plus a carry must stay below 2^64
    static_assert((2 * L + 1) <= (1ull << (64 - 2 * B)), "accumulator overflow");
};

// modulus in B-bit limbs (+ its negation limb by limb, for the conditional subtract)
template <int L>
struct UParams {
    uint32_t p[L];
    uint32_t n0inv;  // -p^-1 mod 2^B
};

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) {
    return __builtin_amdgcn_alignbit(hi, lo, s);  // ({hi,lo} >> s) low word, 0 <= s < 32
}

// limbs of (w << SHIFT), SHIFT < B
template <int NW, int B, int SHIFT, int L = UCfg<NW, B>::L>
__device__ __forceinline__ void to_limbs(uint32_t (&l)[L], const uint32_t (&w)[NW]) {
    constexpr uint32_t M = UCfg<NW, B>::MASK;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int o = i * B - SHIFT;  // bit offset of this limb in w
        uint32_t v;
        if (o < 0) {
            v = w[0] << (-o);
        } else {
            const int j = o >> 5, s = o & 31;
            const uint32_t lo = j < NW ? w[j] : 0u;
            const uint32_t hi = j + 1 < NW ? w[j + 1] : 0u;
            v = s ? alignbit(hi, lo, s) : lo;
        }
        l[i] = v & M;
    }
}

// words of the value held in L normalised limbs (every limb < 2^B); the value must be < 2^(32*NW)
template <int NW, int B, int L = UCfg<NW, B>::L>
__device__ __forceinline__ void from_limbs(uint32_t (&w)[NW], const uint32_t (&l)[L]) {
#pragma unroll
    for (int j = 0; j < NW; j++) {
        const int i0 = (32 * j) / B, s = 32 * j - i0 * B;  // word j starts s bits into limb i0
        uint32_t v = i0 < L ? l[i0] >> s : 0u;   // words beyond the limbs are zero
        if (i0 + 1 < L) v |= l[i0 + 1] << (B - s);
        if (2 * B - s < 32 && i0 + 2 < L) v |= l[i0 + 2] << (2 * B - s);
        w[j] = v;
    }
}

#include "fp_mad_groups.inc"
constexpr int kMadGroup = 13;

// acc += sum_{i<A} x[i]*y[i] + sum_{i<S} p[i]*q[i], chunk C of G
template <int A, int S, int C, int G>
struct MadChunks {
    static __device__ __forceinline__ void run(uint64_t& acc, const uint32_t* x, const uint32_t* y, const uint32_t* p,
                                               const uint32_t* q) {
        constexpr int T = A + S;
        constexpr int SZ = (T + G - 1) / G;
        constexpr int START = C * SZ;
        constexpr int LEN = START + SZ <= T ? SZ : T - START;
        if constexpr (LEN > 0) {
            constexpr int GV = START < A ? (START + LEN <= A ? LEN : A - START) : 0;
            constexpr int GS = LEN - GV;
            constexpr int V0 = START < A ? START : 0;
            constexpr int S0 = START < A ? 0 : START - A;
            MadG<GV, GS>::run(acc, x + V0, y + V0, p + S0, q + S0);
        }
        if constexpr (C + 1 < G) MadChunks<A, S, C + 1, G>::run(acc, x, y, p, q);
    }
};
template <int A, int S>
__device__ __forceinline__ void mad_column(uint64_t& acc, const uint32_t* x, const uint32_t* y, const uint32_t* p,
                                           const uint32_t* q) {
    if constexpr (A + S > 0) MadChunks<A, S, 0, (A + S + kMadGroup - 1) / kMadGroup>::run(acc, x, y, p, q);
}

__device__ __forceinline__ void acc_shr(uint64_t& acc, int b) { acc >>= b; }

// Column K of the product-scanning Montgomery multiplication.
template <int B, int L, int K>
struct UMontCol {
    static __device__ __forceinline__ void run(uint64_t& acc, const uint32_t (&a)[L], const uint32_t (&b)[L],
                                               uint32_t (&u)[L], uint32_t (&r)[L], const UParams<L>& P) {
        constexpr uint32_t M = (1u << B) - 1;
        if constexpr (K < L) {
            // a_0 b_K .. a_K b_0 and u_0 p_K .. u_{K-1} p_1, then u_K and u_K p_0
            uint32_t y[K + 1], q[K + 1];
#pragma unroll
            for (int i = 0; i <= K; i++) { y[i] = b[K - i]; q[i] = P.p[K - i]; }
            mad_column<K + 1, K>(acc, a, y, u, q);
            u[K] = ((uint32_t)acc * P.n0inv) & M;
            acc += (uint64_t)u[K] * P.p[0];
            acc_shr(acc, B);
        } else {
            constexpr int C = 2 * L - 1 - K, I0 = K - L + 1;
            uint32_t y[C], q[C];
#pragma unroll
            for (int i = 0; i < C; i++) { y[i] = b[L - 1 - i]; q[i] = P.p[L - 1 - i]; }
            mad_column<C, C>(acc, a + I0, y, u + I0, q);
            r[K - L] = (uint32_t)acc & M;
            acc_shr(acc, B);
        }
        if constexpr (K + 1 < 2 * L - 1) UMontCol<B, L, K + 1>::run(acc, a, b, u, r, P);
    }
};

// r = a * b / 2^(B*L) mod p, r < a*b/2^(B*L) + p, limbs of r normalised.
// a, b: limbs < 2^B (b is the pre-shifted operand when a saturated-R result is wanted).
template <int B, int L>
__device__ __forceinline__ void umont_mul(uint32_t (&r)[L], const uint32_t (&a)[L], const uint32_t (&b)[L],
                                          const UParams<L>& P) {
    uint32_t u[L];
    uint64_t acc = 0;
    UMontCol<B, L, 0>::run(acc, a, b, u, r, P);
    r[L - 1] = (uint32_t)acc;
}

// Column K of the product-scanning Montgomery SQUARING: a_i a_j with i < j taken once against the
// doubled limb d_j = 2 a_j (30 bits, product < 2^59), the diagonal a_i^2 as it is; L (L + 1) / 2
// products instead of L^2.  Column bound: ceil(L / 2) * 2^59 + 2^58 + L * 2^58 <= (2L + 1) * 2^58,
// as for the multiplication.
template <int B, int L, int K>
struct USqrCol {
    static __device__ __forceinline__ void run(uint64_t& acc, const uint32_t (&a)[L], const uint32_t (&d)[L],
                                               uint32_t (&u)[L], uint32_t (&r)[L], const UParams<L>& P) {
        constexpr uint32_t M = (1u << B) - 1;
        constexpr int I0 = K < L ? 0 : K - L + 1;             // first index of the column
        constexpr int NP = (K - 1) / 2 - I0 + 1 > 0 && K > 0 ? (K - 1) / 2 - I0 + 1 : 0;   // pairs i < j
        constexpr int A = NP + (K % 2 == 0 ? 1 : 0);
        uint32_t x[A], y[A];
#pragma unroll
        for (int t = 0; t < NP; t++) { x[t] = a[I0 + t]; y[t] = d[K - I0 - t]; }
        if constexpr (K % 2 == 0) { x[NP] = a[K / 2]; y[NP] = a[K / 2]; }
        if constexpr (K < L) {
            uint32_t q[K + 1];
#pragma unroll
            for (int i = 0; i <= K; i++) q[i] = P.p[K - i];
            mad_column<A, K>(acc, x, y, u, q);
            u[K] = ((uint32_t)acc * P.n0inv) & M;
            acc += (uint64_t)u[K] * P.p[0];
            acc_shr(acc, B);
        } else {
            constexpr int C = 2 * L - 1 - K;
            uint32_t q[C];
#pragma unroll
            for (int i = 0; i < C; i++) q[i] = P.p[L - 1 - i];
            mad_column<A, C>(acc, x, y, u + I0, q);
            r[K - L] = (uint32_t)acc & M;
            acc_shr(acc, B);
        }
        if constexpr (K + 1 < 2 * L - 1) USqrCol<B, L, K + 1>::run(acc, a, d, u, r, P);
    }
};
// r = a * a / 2^(B*L) mod p, r < a*a/2^(B*L) + p, limbs of r normalised.
template <int B, int L>
__device__ __forceinline__ void umont_sqr(uint32_t (&r)[L], const uint32_t (&a)[L], const UParams<L>& P) {
    uint32_t u[L], d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = a[i] << 1;
    uint64_t acc = 0;
    USqrCol<B, L, 0>::run(acc, a, d, u, r, P);
    r[L - 1] = (uint32_t)acc;
}

// r -= p if r >= p (r < 2p, limbs normalised)
template <int B, int L>
__device__ __forceinline__ void ucond_sub(uint32_t (&r)[L], const UParams<L>& P) {
    constexpr uint32_t M = (1u << B) - 1;
    uint32_t d[L];
    int32_t borrow = 0;  // 0 or -1
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int32_t t = (int32_t)r[i] - (int32_t)P.p[i] + borrow;
        borrow = t >> 31;
        d[i] = (uint32_t)t & M;
    }
#pragma unroll
    for (int i = 0; i < L; i++) r[i] = borrow ? r[i] : d[i];
}

// host-or-device helper: UParams from the saturated modulus words
template <int NW, int B, int L = UCfg<NW, B>::L>
__host__ __device__ inline void make_uparams(UParams<L>& U, const uint32_t* m) {
    constexpr uint32_t M = UCfg<NW, B>::MASK;
    for (int i = 0; i < L; i++) {
        const int o = i * B, j = o >> 5, s = o & 31;
        uint64_t v = j < NW ? m[j] : 0u;
        if (j + 1 < NW) v |= (uint64_t)m[j + 1] << 32;
        U.p[i] = (uint32_t)(v >> s) & M;
    }
    uint32_t inv = 1;  // Newton: p^-1 mod 2^32
    for (int i = 0; i < 5; i++) inv *= 2u - m[0] * inv;
    U.n0inv = (0u - inv) & M;
}

// ---- the configuration the library's VALU-bound kernels use -------------------------------
// 29-bit limbs for every compiled width: L = ceil(32 N / 29) limbs, 2L + 1 <= 37 products of
// < 2^58 fit the 64-bit column.  Narrow fields (N < 6) stay on the saturated multiplier, which
// needs fewer instructions there (N = 2: 10 multiplies + 8 carry adds against 21 multiplies).
constexpr int kUB = 29;
constexpr int kUnsatMinN = 6;
template <int N>
struct UF {
    using C = UCfg<N, kUB>;
    static constexpr int L = C::L, SH = C::SH;
};

// Wave-uniform: everything derives from the kernel argument, so the limbs of p stay in SGPRs.
template <int N>
__device__ __forceinline__ void uparams_from(UParams<UF<N>::L>& U, const FpParams<N>& P) {
    constexpr int L = UF<N>::L;
    constexpr uint32_t M = UF<N>::C::MASK;
#pragma unroll
    for (int i = 0; i < L; i++) {
        const int o = i * kUB, j = o >> 5, s = o & 31;
        uint64_t v = j < N ? P.m[j] : 0u;
        if (j + 1 < N) v |= (uint64_t)P.m[j + 1] << 32;
        U.p[i] = (uint32_t)(v >> s) & M;
    }
    U.n0inv = P.n0inv & M;  // -p^-1 mod 2^29 is the low part of -p^-1 mod 2^32
}

// Values may stay in [0, 2p) between multiplications ("lazy") when 4p <= R = 2^(32N): for
// a, b < 2p the product step returns a*b/R + p <= 2p.  Moduli without two spare bits are
// brought back below p after every multiplication instead.
template <int N>
__device__ __forceinline__ bool ulazy(const FpParams<N>& P) { return P.bitlen + 2 <= 32u * N; }

// Saturated words <-> limbs, umont product with the shift that matches the reference's R:
//   umul(r, a, bs) = a * b / 2^(32N) (mod p), < 2p for lazy inputs; bs = to_limbs<SH>(b).
template <int N>
__device__ __forceinline__ void uload(uint32_t (&l)[UF<N>::L], const uint32_t (&w)[N]) { to_limbs<N, kUB, 0>(l, w); }
template <int N>
__device__ __forceinline__ void uload_shifted(uint32_t (&l)[UF<N>::L], const uint32_t (&w)[N]) {
    to_limbs<N, kUB, UF<N>::SH>(l, w);
}
template <int N>
__device__ __forceinline__ void ustore(uint32_t (&w)[N], const uint32_t (&l)[UF<N>::L]) { from_limbs<N, kUB>(w, l); }
template <int N>
__device__ __forceinline__ void umul(uint32_t (&r)[UF<N>::L], const uint32_t (&a)[UF<N>::L],
                                     const uint32_t (&bs)[UF<N>::L], const UParams<UF<N>::L>& U) {
    umont_mul<kUB, UF<N>::L>(r, a, bs, U);
}
template <int N>
__device__ __forceinline__ void ucanon(uint32_t (&r)[UF<N>::L], const UParams<UF<N>::L>& U) {
    ucond_sub<kUB, UF<N>::L>(r, U);
}
// limbs of (x << SH) for normalised x < 2^(32N): the pre-shifted form of a value already in limbs
template <int N>
__device__ __forceinline__ void ushift(uint32_t (&r)[UF<N>::L], const uint32_t (&x)[UF<N>::L]) {
    constexpr int L = UF<N>::L, SH = UF<N>::SH;
    constexpr uint32_t M = UF<N>::C::MASK;
    if constexpr (SH == 0) {
#pragma unroll
        for (int i = 0; i < L; i++) r[i] = x[i];
    } else {
        r[0] = (x[0] << SH) & M;
#pragma unroll
        for (int i = 1; i < L; i++) r[i] = ((x[i] << SH) | (x[i - 1] >> (kUB - SH))) & M;
    }
}

// ---- one-shot products on words, for the elementwise kernels of wide fields ----------------
// r = a * b / R mod p, canonical, for a < R and b < p (then the result before the conditional
// subtraction is below 2p, as in the reference, and the canonical residue is the same number).
template <int N>
__device__ __forceinline__ void umul_words(uint32_t (&r)[N], const uint32_t (&a)[N], const uint32_t (&b)[N],
                                           const UParams<UF<N>::L>& U) {
    constexpr int L = UF<N>::L;
    uint32_t x[L], y[L], t[L];
    uload<N>(x, a);
    uload_shifted<N>(y, b);
    umul<N>(t, x, y, U);
    ucanon<N>(t, U);
    ustore<N>(r, t);
}
// r = a * a / R mod p, canonical, for a < p.  With an even pre-shift SH the square is symmetric:
// (a 2^(SH/2))^2 / R' = a^2 / R, L (L + 1) / 2 products (umont_sqr).
template <int N>
__device__ __forceinline__ void usqr_words(uint32_t (&r)[N], const uint32_t (&a)[N], const UParams<UF<N>::L>& U) {
    constexpr int L = UF<N>::L, SH = UF<N>::SH;
    uint32_t t[L];
    if constexpr (SH % 2 == 0) {
        uint32_t x[L];
        to_limbs<N, kUB, SH / 2>(x, a);
        umont_sqr<kUB, L>(t, x, U);
    } else {
        uint32_t x[L], y[L];
        uload<N>(x, a);
        uload_shifted<N>(y, a);
        umul<N>(t, x, y, U);
    }
    ucanon<N>(t, U);
    ustore<N>(r, t);
}
// ---- Chain<N>: the running value of a dependent multiplication chain ----------------------
// One interface over the two multipliers so that the chain kernels (mul chain, pow, batch
// inversion, product fold) are written once.  Operands come and go as saturated words, the form
// they have in memory; `get` returns the stored form of the running value, which is canonical on
// the saturated path and may be unreduced (below 2p, it fits the container) on the lazy
// unsaturated path; `get_canonical` always returns the residue in [0, p).  Word operands of
// mul / mul_to may be canonical or in that stored form.
template <int N, bool UNSAT = (N >= kUnsatMinN)>
struct Chain;

template <int N>
struct Chain<N, false> {
    uint32_t v[N];
    const FpParams<N>& P;
    __device__ __forceinline__ explicit Chain(const FpParams<N>& P_) : P(P_) {}
    __device__ __forceinline__ void set(const uint32_t (&w)[N]) {
#pragma unroll
        for (int l = 0; l < N; l++) v[l] = w[l];
    }
    __device__ __forceinline__ void get(uint32_t (&w)[N]) const {
#pragma unroll
        for (int l = 0; l < N; l++) w[l] = v[l];
    }
    __device__ __forceinline__ void get_canonical(uint32_t (&w)[N]) const { get(w); }
    __device__ __forceinline__ void mul(const uint32_t (&w)[N]) {
        uint32_t t[N];
        mont_mul<N>(t, v, w, P);
        set(t);
    }
    __device__ __forceinline__ void mul_to(uint32_t (&out)[N], const uint32_t (&w)[N]) const {
        mont_mul<N>(out, v, w, P);
    }
    __device__ __forceinline__ void square() {
        uint32_t t[N];
        mont_sqr<N>(t, v, P);
        set(t);
    }
    // a multiplier used many times, prepared once
    typedef uint32_t Op[N];
    static __device__ __forceinline__ void prep(Op& o, const uint32_t (&w)[N]) {
#pragma unroll
        for (int l = 0; l < N; l++) o[l] = w[l];
    }
    __device__ __forceinline__ void mul_op(const Op& o) { mul(o); }
    // -- the batch inversion's view (see Chain<N, true>): here R' = R and the raw form is the words themselves
    static constexpr int RAW = N;
    __device__ __forceinline__ void mulr(const uint32_t (&w)[N]) { mul(w); }
    __device__ __forceinline__ void get_raw(uint32_t (&t)[N]) const { get(t); }
    __device__ __forceinline__ void get_words(uint32_t (&w)[N]) const { get(w); }
    __device__ __forceinline__ void mulr_raw_to(uint32_t (&out)[N], const uint32_t (&t)[N]) const { mont_mul<N>(out, v, t, P); }
    __device__ __forceinline__ void mulr_to(uint32_t (&out)[N], const uint32_t (&w)[N]) const { mont_mul<N>(out, v, w, P); }
};

template <int N>
struct Chain<N, true> {
    static constexpr int L = UF<N>::L;
    uint32_t v[L];
    UParams<L> U;
    const uint32_t* pw;   // the modulus as saturated words (kernel argument: stays in SGPRs)
    bool lazy;
    __device__ __forceinline__ explicit Chain(const FpParams<N>& P) : pw(P.m) {
        uparams_from<N>(U, P);
        lazy = ulazy<N>(P);
    }
    __device__ __forceinline__ void set(const uint32_t (&w)[N]) { uload<N>(v, w); }
    __device__ __forceinline__ void get(uint32_t (&w)[N]) const { ustore<N>(w, v); }
    __device__ __forceinline__ void get_canonical(uint32_t (&w)[N]) const {
        uint32_t t[L];
#pragma unroll
        for (int l = 0; l < L; l++) t[l] = v[l];
        ucanon<N>(t, U);   // a no-op on a value already below p
        ustore<N>(w, t);
    }
    __device__ __forceinline__ void mul_limbs(const uint32_t (&s)[L]) {
        uint32_t t[L];
        umul<N>(t, v, s, U);
        if (!lazy) ucanon<N>(t, U);
#pragma unroll
        for (int l = 0; l < L; l++) v[l] = t[l];
    }
    __device__ __forceinline__ void mul(const uint32_t (&w)[N]) {
        uint32_t s[L];
        uload_shifted<N>(s, w);
        mul_limbs(s);
    }
    __device__ __forceinline__ void mul_to(uint32_t (&out)[N], const uint32_t (&w)[N]) const {
        uint32_t s[L], t[L];
        uload_shifted<N>(s, w);
        umul<N>(t, v, s, U);
        if (lazy) {
            // 2p < R: the unreduced product fits the words, and the conditional subtraction is
            // cheaper there (a borrow chain and a select per word instead of five operations per limb)
            uint32_t x[N], d[N];
            ustore<N>(x, t);
            const uint32_t below = sub_limbs_s<N>(d, x, pw);
#pragma unroll
            for (int l = 0; l < N; l++) out[l] = below ? x[l] : d[l];
        } else {
            ucanon<N>(t, U);
            ustore<N>(out, t);
        }
    }
    __device__ __forceinline__ void square() {
        uint32_t s[L];
        ushift<N>(s, v);
        mul_limbs(s);
    }
    typedef uint32_t Op[L];
    static __device__ __forceinline__ void prep(Op& o, const uint32_t (&w)[N]) { uload_shifted<N>(o, w); }
    __device__ __forceinline__ void mul_op(const Op& o) { mul_limbs(o); }

    // -- the batch inversion's view: products in the Montgomery domain of R' = 2^(29 L) ------------------
    // Montgomery's trick only needs SOME group operation on the representatives whose inverse the top level
    // can supply.  With a (x) b = a b / R' no operand is pre-shifted, every modulus of the container is lazy
    // (R' >= 4p: values stay in [0, 2p) with no conditional subtraction, tight moduli included), and a
    // running product can be stored and reloaded as its limbs ("raw" form) with no conversion at all.  The
    // chain totals differ from the R-domain ones by powers of R / R', but the top level returns R^2 / T for
    // whatever T it is given (MontyField::inv on the representative), and walking back down with (x) turns
    // that into R^2 / x for every element: by induction I_k = R^2 / t_k implies I_k (x) t_(k-1) = R^2 / x_k and
    // I_k (x) x_k = R^2 / t_(k-1) for t_k = t_(k-1) (x) x_k.  The outputs are the same canonical residues.
    static constexpr int RAW = L;
    static_assert(UF<N>::SH >= 2, "R' >= 4p needs two spare bits above the container");
    __device__ __forceinline__ void mulr_limbs(const uint32_t (&s)[L]) {
        uint32_t t[L];
        umul<N>(t, v, s, U);
#pragma unroll
        for (int l = 0; l < L; l++) v[l] = t[l];
    }
    __device__ __forceinline__ void mulr(const uint32_t (&w)[N]) {
        uint32_t s[L];
        uload<N>(s, w);
        mulr_limbs(s);
    }
    __device__ __forceinline__ void get_raw(uint32_t (&t)[L]) const {
#pragma unroll
        for (int l = 0; l < L; l++) t[l] = v[l];
    }
    // the running value as words that hold it: as it is where 2p <= R, else brought below p first
    __device__ __forceinline__ void get_words(uint32_t (&w)[N]) const {
        if (lazy) get(w);
        else get_canonical(w);
    }
    // canonical words of a value below 2p given as limbs
    __device__ __forceinline__ void finish(uint32_t (&out)[N], uint32_t (&t)[L]) const {
        if (lazy) {   // 2p < R: the conditional subtraction is cheaper on the words (see mul_to)
            uint32_t x[N], d[N];
            ustore<N>(x, t);
            const uint32_t below = sub_limbs_s<N>(d, x, pw);
#pragma unroll
            for (int l = 0; l < N; l++) out[l] = below ? x[l] : d[l];
        } else {
            ucanon<N>(t, U);
            ustore<N>(out, t);
        }
    }
    // out = canonical(running value (x) t), t in raw form
    __device__ __forceinline__ void mulr_raw_to(uint32_t (&out)[N], const uint32_t (&t)[L]) const {
        uint32_t r[L];
        umul<N>(r, v, t, U);
        finish(out, r);
    }
    // out = canonical(running value (x) w), w as words
    __device__ __forceinline__ void mulr_to(uint32_t (&out)[N], const uint32_t (&w)[N]) const {
        uint32_t s[L];
        uload<N>(s, w);
        mulr_raw_to(out, s);
    }
};

// ---- PowLadder<N>: square-and-multiply state for k_pow ------------------------------------
// The unsaturated ladder works in the Montgomery domain of R' = 2^(29 L) = R 2^SH, where nothing
// needs a pre-shift: squarings are symmetric (umont_sqr: L (L + 1) / 2 products) and, because
// R' >= 4p for every modulus of the container, values stay in [0, 2p) for the whole ladder, tight
// moduli included.  Entry: x~ = x 2^SH = umul(x, K) with K = R 2^SH mod p (SH modular doublings of
// R mod p, once per lane); exit: umont(acc, R mod p) = acc R / R' brings the result back to the
// reference's radix, then one conditional subtraction.
template <int N, bool UNSAT = (N >= kUnsatMinN)>
struct PowLadder;

template <int N>
struct PowLadder<N, false> {
    uint32_t acc[N], x[N];
    const FpParams<N>& P;
    __device__ __forceinline__ explicit PowLadder(const FpParams<N>& P_) : P(P_) {}
    __device__ __forceinline__ void start(const uint32_t (&w)[N]) {
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = x[l] = w[l];
    }
    __device__ __forceinline__ void square() {
        uint32_t t[N];
        mont_sqr<N>(t, acc, P);
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = t[l];
    }
    __device__ __forceinline__ void mul_base() {
        uint32_t t[N];
        mont_mul<N>(t, acc, x, P);
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = t[l];
    }
    __device__ __forceinline__ void finish(uint32_t (&w)[N]) const {
#pragma unroll
        for (int l = 0; l < N; l++) w[l] = acc[l];
    }
    // the running value in the ladder's own form (table entries of the windowed ladder)
    typedef uint32_t Val[N];
    __device__ __forceinline__ void get_acc(Val& o) const {
#pragma unroll
        for (int l = 0; l < N; l++) o[l] = acc[l];
    }
    __device__ __forceinline__ void set_acc(const Val& o) {
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = o[l];
    }
    __device__ __forceinline__ void mul_val(const Val& y) {
        uint32_t t[N];
        mont_mul<N>(t, acc, y, P);
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = t[l];
    }
};

template <int N>
struct PowLadder<N, true> {
    static constexpr int L = UF<N>::L;
    uint32_t acc[L], x[L], kin[L], kout[L];
    UParams<L> U;
    __device__ __forceinline__ explicit PowLadder(const FpParams<N>& P) {
        uparams_from<N>(U, P);
        uint32_t k[N], t[N];
        set_limbs<N>(k, P.one);
        for (int i = 0; i < UF<N>::SH; i++) {   // K = R 2^SH mod p
            mod_add<N>(t, k, k, P);
#pragma unroll
            for (int l = 0; l < N; l++) k[l] = t[l];
        }
        uload_shifted<N>(kin, k);
        set_limbs<N>(k, P.one);
        uload<N>(kout, k);
    }
    __device__ __forceinline__ void start(const uint32_t (&w)[N]) {
        uint32_t a[L];
        uload<N>(a, w);
        umul<N>(x, a, kin, U);   // x 2^SH, below 2p
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = x[l];
    }
    __device__ __forceinline__ void square() {
        uint32_t t[L];
        umont_sqr<kUB, L>(t, acc, U);
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = t[l];
    }
    __device__ __forceinline__ void mul_base() {
        uint32_t t[L];
        umont_mul<kUB, L>(t, acc, x, U);
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = t[l];
    }
    __device__ __forceinline__ void finish(uint32_t (&w)[N]) const {
        uint32_t t[L];
        umont_mul<kUB, L>(t, acc, kout, U);
        ucanon<N>(t, U);
        ustore<N>(w, t);
    }
    typedef uint32_t Val[L];
    __device__ __forceinline__ void get_acc(Val& o) const {
#pragma unroll
        for (int l = 0; l < L; l++) o[l] = acc[l];
    }
    __device__ __forceinline__ void set_acc(const Val& o) {
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = o[l];
    }
    __device__ __forceinline__ void mul_val(const Val& y) {
        uint32_t t[L];
        umont_mul<kUB, L>(t, acc, y, U);
#pragma unroll
        for (int l = 0; l < L; l++) acc[l] = t[l];
    }
};

}  // namespace fpgpu
