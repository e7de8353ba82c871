/*
 * fp_oracle.c -- CPU restatement of crypto-primitives' prime-field Montgomery
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object, and only as the checker / the timed CPU
 * baseline.  The product (libfpgpu.so and everything above it) never links,
 * imports or calls anything under oracle/.
 *
 * Each function cites the reference file:line (relative to the reference
 * repository root) it restates.  Limbs are 64-bit little-endian words, exactly
 * the reference's `Word` on a 64-bit target (src/helpers/crypto_bigint.rs:5-6),
 * so R = 2^(64*n).
 *
 * Pinning.  The reference is Rust and no Rust toolchain exists in the build
 * container, so the reference cannot be compiled or run here (oracle/_ref does
 * not exist; recorded in DESIGN.md).  This restatement is pinned by
 *   (1) every known-answer test the reference's own unit tests hold for this
 *       path (tests/test_oracle_kat.py lists them with file:line), and
 *   (2) independent Python-int golden vectors (tests/golden/, generator
 *       committed) for full-width random and edge inputs.
 * Arithmetic that lives in the third-party crate `crypto-bigint`
 * (Cargo.toml:22, ">= 0.7.1, < 0.8", no lockfile, sources absent) -- add_mod,
 * sub_mod, neg_mod, invert_odd_mod_vartime, MontyParams::new -- is restated
 * from its published algorithms; results on in-range inputs are canonical
 * residues and therefore method-independent.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t word;
typedef unsigned __int128 wide;

#define FPO_MAX_LIMBS 64

/* ---- src/helpers/crypto_bigint.rs:56-67  compute_mod_neg_inv ------------- */
word fpo_mod_neg_inv(word m0)
{
    word inv = 1;
    for (int i = 0; i < 6; i++)          /* LOG2_WORD_BITS = 6 (:50) */
        inv = inv * (2 - m0 * inv);
    return (word)0 - inv;
}

/* ---- src/helpers/crypto_bigint.rs:164-167  mul_add_carry ----------------- */
static inline void mul_add_carry(word a, word b, word c, word d, word *lo, word *hi)
{
    wide w = (wide)a * (wide)b + (wide)c + (wide)d;
    *lo = (word)w;
    *hi = (word)(w >> 64);
}

/* ---- src/helpers/crypto_bigint.rs:118-159  monty_mul_cios ----------------
 * `out` must be zero on entry (:115).  Returns the overflow word acc_hi. */
word fpo_monty_mul_cios(const word *a, const word *b, const word *modulus,
                        word mod_neg_inv, word *out, size_t n)
{
    word acc_hi = 0;
    for (size_t i = 0; i < n; i++) {
        word a_i = a[i];
        /* Step 1: acc += a_i * b   (:129-136) */
        word carry = 0, lo, hi;
        for (size_t j = 0; j < n; j++) {
            mul_add_carry(a_i, b[j], out[j], carry, &lo, &hi);
            out[j] = lo;
            carry = hi;
        }
        word new_acc_hi = acc_hi + carry;
        word meta_carry = new_acc_hi < acc_hi;
        acc_hi = new_acc_hi;

        /* Step 2: Montgomery reduction   (:138-154) */
        word u = out[0] * mod_neg_inv;
        mul_add_carry(u, modulus[0], out[0], 0, &lo, &hi);
        carry = hi;
        for (size_t j = 1; j < n; j++) {
            mul_add_carry(u, modulus[j], out[j], carry, &lo, &hi);
            out[j - 1] = lo;
            carry = hi;
        }
        word sum = acc_hi + carry;
        word c = sum < acc_hi;
        out[n - 1] = sum;
        acc_hi = meta_carry + c;
    }
    return acc_hi;
}

/* ---- src/helpers/crypto_bigint.rs:74-107  monty_mul ----------------------
 * a*b*R^-1 mod m; subtract m iff carry != 0 OR result >= m (:87-104).
 * mod_neg_inv is recomputed from modulus[0] exactly as the reference does (:82). */
void fpo_monty_mul(const word *a, const word *b, const word *modulus, word *result, size_t n)
{
    word tmp[FPO_MAX_LIMBS], diff[FPO_MAX_LIMBS];
    word mod_neg_inv = fpo_mod_neg_inv(modulus[0]);
    memset(tmp, 0, n * sizeof(word));
    word carry = fpo_monty_mul_cios(a, b, modulus, mod_neg_inv, tmp, n);

    word borrow = 0;
    for (size_t i = 0; i < n; i++) {
        word d = tmp[i] - modulus[i];
        word b1 = tmp[i] < modulus[i];
        word d2 = d - borrow;
        word b2 = d < borrow;
        diff[i] = d2;
        borrow = b1 | b2;
    }
    int use_diff = (carry != 0) | (borrow == 0);
    word mask = (word)0 - (word)use_diff;
    for (size_t i = 0; i < n; i++)
        result[i] = (diff[i] & mask) | (tmp[i] & ~mask);
}

/* ---- src/helpers/crypto_bigint.rs:172-217  monty_mul_limbs ---------------
 * crypto-bigint-style fused one-pass Montgomery multiply with 65-bit carries.
 * `out` must be zero on entry.  Returns meta_carry (0 or 1). */
word fpo_monty_mul_limbs(const word *x, const word *y, word *out,
                         const word *modulus, word mod_neg_inv, size_t n)
{
    wide meta_carry = 0;
    for (size_t i = 0; i < n; i++) {
        word xi = x[i];
        wide axy = (wide)xi * (wide)y[0] + (wide)out[0];
        word u = (word)axy * mod_neg_inv;
        wide um = (wide)u * (wide)modulus[0];
        wide a = um + axy;
        wide c = a < um;                               /* overflowing_add (:193) */
        wide carry = (c << 64) | (a >> 64);
        for (size_t j = 1; j < n; j++) {
            wide axy2 = (wide)xi * (wide)y[j] + (wide)out[j];
            wide umc = (wide)u * (wide)modulus[j] + carry;
            wide ab = axy2 + umc;
            wide c2 = ab < axy2;
            out[j - 1] = (word)ab;
            carry = (c2 << 64) | (ab >> 64);
        }
        carry += meta_carry;
        out[n - 1] = (word)carry;
        meta_carry = carry >> 64;
    }
    return (word)meta_carry;
}

/* ---- src/semiring/crypto_bigint_boxed_uint.rs:197-208 borrowing_sub_assign */
static word borrowing_sub_assign(word *self, const word *rhs, size_t n)
{
    word borrow = 0;                                  /* 0 or all-ones, as Limb::borrowing_sub */
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)self[i] - (wide)rhs[i] - (wide)(borrow & 1);
        self[i] = (word)t;
        borrow = (word)0 - (word)((t >> 64) & 1);
    }
    return borrow;
}

/* ---- src/semiring/crypto_bigint_boxed_uint.rs:210-226 sub_assign_mod_with_carry */
void fpo_sub_assign_mod_with_carry(word *self, word carry, const word *rhs, const word *p, size_t n)
{
    word borrow = borrowing_sub_assign(self, rhs, n);
    word mask = ~((word)0 - carry) & borrow;           /* carry.wrapping_neg().not() & borrow (:218) */
    if (mask != 0) {                                   /* conditional_add_assign(p) (:223-225) */
        word c = 0;
        for (size_t i = 0; i < n; i++) {
            wide t = (wide)self[i] + (wide)p[i] + (wide)c;
            self[i] = (word)t;
            c = (word)(t >> 64);
        }
    }
}

/* ---- src/field/crypto_bigint_boxed_monty.rs:182-196  BoxedMontyField::mul */
void fpo_boxed_mul(const word *x, const word *y, const word *modulus, word mod_neg_inv, word *out, size_t n)
{
    memset(out, 0, n * sizeof(word));
    word carry = fpo_monty_mul_limbs(x, y, out, modulus, mod_neg_inv, n);
    fpo_sub_assign_mod_with_carry(out, carry, modulus, modulus, n);
}

/* ---- src/helpers/crypto_bigint.rs:13-37  monty_retrieve_inner ------------
 * `out` must be zero on entry (crypto_bigint_monty.rs:376). */
void fpo_monty_retrieve_inner(const word *x, word *out, const word *modulus, word mod_neg_inv, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        word xi = x[i];
        word u = (out[0] + xi) * mod_neg_inv;
        /* carrying_mul_add(a=u, b=modulus[j], addend, carry): u*b + addend + carry -> (lo, hi) */
        wide t = (wide)u * (wide)modulus[0] + (wide)xi + (wide)out[0];
        out[0] = (word)(t >> 64);                      /* keeps .1 (the high word) (:27) */
        for (size_t j = 1; j < n; j++) {
            wide t2 = (wide)u * (wide)modulus[j] + (wide)out[j] + (wide)out[j - 1];
            out[j - 1] = (word)t2;
            out[j] = (word)(t2 >> 64);
        }
    }
}

/* ---- crypto-bigint Uint::add_mod (call sites: crypto_bigint_monty.rs:148-150,
 * const_monty.rs:320-323, boxed_monty.rs:170-172,229-230).  Published algorithm:
 * w = a + b; w -= p; if that underflowed past the add's carry, add p back. */
void fpo_add_mod(const word *a, const word *b, const word *p, word *out, size_t n)
{
    word carry = 0, borrow = 0;
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)a[i] + (wide)b[i] + (wide)carry;
        out[i] = (word)t;
        carry = (word)(t >> 64);
    }
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)out[i] - (wide)p[i] - (wide)borrow;
        out[i] = (word)t;
        borrow = (word)((t >> 64) & 1);
    }
    if (carry < borrow) {                              /* (carry - borrow) underflows -> mask */
        word c = 0;
        for (size_t i = 0; i < n; i++) {
            wide t = (wide)out[i] + (wide)p[i] + (wide)c;
            out[i] = (word)t;
            c = (word)(t >> 64);
        }
    }
}

/* ---- crypto-bigint Uint::sub_mod (call sites: crypto_bigint_monty.rs:154-156,
 * const_monty.rs:330-333, boxed_monty.rs:176-178). */
void fpo_sub_mod(const word *a, const word *b, const word *p, word *out, size_t n)
{
    word borrow = 0;
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)a[i] - (wide)b[i] - (wide)borrow;
        out[i] = (word)t;
        borrow = (word)((t >> 64) & 1);
    }
    if (borrow) {
        word c = 0;
        for (size_t i = 0; i < n; i++) {
            wide t = (wide)out[i] + (wide)p[i] + (wide)c;
            out[i] = (word)t;
            c = (word)(t >> 64);
        }
    }
}

/* ---- crypto-bigint Uint::neg_mod (call sites: crypto_bigint_monty.rs:199-201,
 * const_monty.rs:196-199, boxed_monty.rs:243-245): p - a, with 0 -> 0. */
void fpo_neg_mod(const word *a, const word *p, word *out, size_t n)
{
    word nz = 0;
    for (size_t i = 0; i < n; i++) nz |= a[i];
    if (!nz) { memset(out, 0, n * sizeof(word)); return; }
    word borrow = 0;
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)p[i] - (wide)a[i] - (wide)borrow;
        out[i] = (word)t;
        borrow = (word)((t >> 64) & 1);
    }
}

/* ---- plain multi-limb helpers (used only by params / inversion below) ---- */
static int cmp_n(const word *a, const word *b, size_t n)
{
    for (size_t i = n; i-- > 0;) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return 0;
}
static int is_zero_n(const word *a, size_t n)
{
    word z = 0;
    for (size_t i = 0; i < n; i++) z |= a[i];
    return z == 0;
}
static word add_n(word *r, const word *a, const word *b, size_t n)
{
    word c = 0;
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)a[i] + (wide)b[i] + (wide)c;
        r[i] = (word)t; c = (word)(t >> 64);
    }
    return c;
}
static word sub_n(word *r, const word *a, const word *b, size_t n)
{
    word bo = 0;
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)a[i] - (wide)b[i] - (wide)bo;
        r[i] = (word)t; bo = (word)((t >> 64) & 1);
    }
    return bo;
}
static void shr1_n(word *a, size_t n, word top)
{
    for (size_t i = 0; i + 1 < n; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
    a[n - 1] = (a[n - 1] >> 1) | (top << 63);
}
/* r = 2*a mod p, a < p */
static void dbl_mod(word *a, const word *p, size_t n)
{
    word top = a[n - 1] >> 63;
    for (size_t i = n; i-- > 1;) a[i] = (a[i] << 1) | (a[i - 1] >> 63);
    a[0] <<= 1;
    if (top || cmp_n(a, p, n) >= 0) sub_n(a, a, p, n);
}

/* ---- crypto-bigint MontyParams::new (call sites crypto_bigint_monty.rs:351,
 * boxed_monty.rs:411): one = R mod p, r2 = R^2 mod p, mod_neg_inv.
 * Returns 0 on success, 1 for an even (or zero) modulus, mirroring
 * FieldError::InvalidModulus (src/field.rs:345-349; crypto_bigint_monty.rs:347-350). */
int fpo_params(const word *modulus, size_t n, word *mod_neg_inv, word *one, word *r2)
{
    if (n == 0 || n > FPO_MAX_LIMBS || (modulus[0] & 1) == 0) return 1;
    *mod_neg_inv = fpo_mod_neg_inv(modulus[0]);
    word t[FPO_MAX_LIMBS];
    memset(t, 0, n * sizeof(word));
    /* 1 mod p (p may be 1 -> 0) */
    t[0] = 1;
    if (cmp_n(t, modulus, n) >= 0) sub_n(t, t, modulus, n);
    for (size_t i = 0; i < 64 * n; i++) dbl_mod(t, modulus, n);
    memcpy(one, t, n * sizeof(word));
    for (size_t i = 0; i < 64 * n; i++) dbl_mod(t, modulus, n);
    memcpy(r2, t, n * sizeof(word));
    return 0;
}

/* ---- crypto-bigint Uint::invert_odd_mod_vartime (call sites
 * crypto_bigint_monty.rs:230-231, boxed_monty.rs:275-278).  Restated as the
 * classic binary extended GCD for an odd modulus; the inverse of an invertible
 * residue is unique, so the method is immaterial.  Returns 1 and writes the
 * inverse when gcd(a, p) == 1, else returns 0 (the reference's `None`). */
int fpo_invert_odd_mod(const word *a_in, const word *p, word *out, size_t n)
{
    word a[FPO_MAX_LIMBS], u[FPO_MAX_LIMBS], v[FPO_MAX_LIMBS], x1[FPO_MAX_LIMBS], x2[FPO_MAX_LIMBS];
    memcpy(a, a_in, n * sizeof(word));
    /* reduce a mod p (inputs are expected < p, but stay total) */
    while (cmp_n(a, p, n) >= 0) sub_n(a, a, p, n);
    if (is_zero_n(a, n)) return 0;
    memcpy(u, a, n * sizeof(word));
    memcpy(v, p, n * sizeof(word));
    memset(x1, 0, n * sizeof(word)); x1[0] = 1;
    memset(x2, 0, n * sizeof(word));
    word one[FPO_MAX_LIMBS];
    memset(one, 0, n * sizeof(word)); one[0] = 1;
    if (cmp_n(x1, p, n) >= 0) sub_n(x1, x1, p, n);      /* p == 1 */
    while (cmp_n(u, one, n) != 0 && cmp_n(v, one, n) != 0) {
        if (is_zero_n(u, n) || is_zero_n(v, n)) return 0;
        while ((u[0] & 1) == 0) {
            shr1_n(u, n, 0);
            if (x1[0] & 1) { word c = add_n(x1, x1, p, n); shr1_n(x1, n, c); }
            else shr1_n(x1, n, 0);
        }
        while ((v[0] & 1) == 0) {
            shr1_n(v, n, 0);
            if (x2[0] & 1) { word c = add_n(x2, x2, p, n); shr1_n(x2, n, c); }
            else shr1_n(x2, n, 0);
        }
        if (cmp_n(u, v, n) >= 0) {
            sub_n(u, u, v, n);
            if (sub_n(x1, x1, x2, n)) add_n(x1, x1, p, n);
        } else {
            sub_n(v, v, u, n);
            if (sub_n(x2, x2, x1, n)) add_n(x2, x2, p, n);
        }
    }
    if (cmp_n(u, one, n) == 0) memcpy(out, x1, n * sizeof(word));
    else memcpy(out, x2, n * sizeof(word));
    return 1;
}

/* ---- src/field/crypto_bigint_monty.rs:223-237  MontyField::inv
 * (identical shape: boxed_monty.rs:273-284; const_monty.rs:277-283).
 * Input and output are Montgomery representatives.  Returns 1 = Some, 0 = None. */
int fpo_inv(const word *x, const word *modulus, const word *r2, word *out, size_t n)
{
    word inverted[FPO_MAX_LIMBS], x_inv[FPO_MAX_LIMBS];
    if (!fpo_invert_odd_mod(x, modulus, inverted, n)) return 0;
    fpo_monty_mul(inverted, r2, modulus, x_inv, n);
    fpo_monty_mul(x_inv, r2, modulus, out, n);
    return 1;
}

/* ---- src/field/crypto_bigint_monty.rs:319-340  project (to Montgomery form):
 * monty_mul(value, R^2); value may be any n-limb integer (>= p allowed). */
void fpo_project(const word *value, const word *modulus, const word *r2, word *out, size_t n)
{
    fpo_monty_mul(value, r2, modulus, out, n);
}

/* ---- src/field/crypto_bigint_monty.rs:369-385  lift (from Montgomery form) */
void fpo_lift(const word *x, const word *modulus, word mod_neg_inv, word *out, size_t n)
{
    memset(out, 0, n * sizeof(word));
    fpo_monty_retrieve_inner(x, out, modulus, mod_neg_inv, n);
}

/* ======================= batched drivers (AoS, n limbs per element) ========
 * Loops of the scalar reference ops over contiguous element arrays; these are
 * the "what you get by looping the reference's scalar ops" definitions of the
 * batch compositions.
 * op codes match include/fpgpu.h's kernels one-to-one for the tests. */
enum { FPO_MUL = 0, FPO_SQR = 1, FPO_ADD = 2, FPO_SUB = 3, FPO_NEG = 4,
       FPO_TO_MONT = 5, FPO_FROM_MONT = 6, FPO_MUL_BOXED = 7, FPO_DOUBLE = 8, FPO_DIV_BY_2 = 9 };

/* ConstMontyField::div_by_2 (src/field/crypto_bigint_const_monty.rs:81-84) hands the Montgomery
 * representative to crypto-bigint's ConstMontyForm::div_by_2 (third-party, >= 0.7.1 < 0.8, not under
 * /root/reference; published algorithm, modular/div_by_2.rs): (half, is_odd) = a.shr1_with_carry();
 * if_odd = half + (p >> 1) + 1 (wrapping); result = is_odd ? if_odd : half.  For a < p that is the
 * canonical x with x + x = a (mod p); the reference pins it with `half + half == even`
 * (crypto_bigint_const_monty.rs:1184-1187). */
static void fpo_div_by_2(const word *a, const word *p, word *out, size_t n)
{
    word half[FPO_MAX_LIMBS], hp[FPO_MAX_LIMBS];
    const int odd = (int)(a[0] & 1);
    for (size_t i = 0; i < n; i++) {
        half[i] = (a[i] >> 1) | (i + 1 < n ? a[i + 1] << 63 : 0);
        hp[i] = (p[i] >> 1) | (i + 1 < n ? p[i + 1] << 63 : 0);
    }
    if (!odd) { memcpy(out, half, n * sizeof(word)); return; }
    unsigned __int128 c = 1;                       /* + Uint::ONE */
    for (size_t i = 0; i < n; i++) {
        c += (unsigned __int128)half[i] + hp[i];
        out[i] = (word)c;                          /* wrapping_add */
        c >>= 64;
    }
}

int fpo_batch_op(int op, const word *modulus, size_t n, const word *a, const word *b,
                 word *out, size_t count, int threads)
{
    word inv, one[FPO_MAX_LIMBS], r2[FPO_MAX_LIMBS];
    if (fpo_params(modulus, n, &inv, one, r2)) return 1;
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
#endif
    for (ptrdiff_t i = 0; i < (ptrdiff_t)count; i++) {
        const word *ai = a + (size_t)i * n;
        const word *bi = b ? b + (size_t)i * n : NULL;
        word *oi = out + (size_t)i * n;
        word tmp[FPO_MAX_LIMBS];
        switch (op) {
        case FPO_MUL:       fpo_monty_mul(ai, bi, modulus, tmp, n); break;
        case FPO_SQR:       fpo_monty_mul(ai, ai, modulus, tmp, n); break;
        case FPO_ADD:       fpo_add_mod(ai, bi, modulus, tmp, n); break;
        case FPO_SUB:       fpo_sub_mod(ai, bi, modulus, tmp, n); break;
        case FPO_NEG:       fpo_neg_mod(ai, modulus, tmp, n); break;
        case FPO_TO_MONT:   fpo_project(ai, modulus, r2, tmp, n); break;
        case FPO_FROM_MONT: fpo_lift(ai, modulus, inv, tmp, n); break;
        case FPO_MUL_BOXED: fpo_boxed_mul(ai, bi, modulus, inv, tmp, n); break;
        /* ConstMontyField::double (crypto_bigint_const_monty.rs:86-89) -> ConstMontyForm::double = add_mod(a, a) */
        case FPO_DOUBLE:    fpo_add_mod(ai, ai, modulus, tmp, n); break;
        case FPO_DIV_BY_2:  fpo_div_by_2(ai, modulus, tmp, n); break;
        default: memset(tmp, 0, sizeof tmp); break;
        }
        memcpy(oi, tmp, n * sizeof(word));
    }
    return 0;
}

/* Per-element inversion loop = the reference's only way to invert a slice
 * (src/field/crypto_bigint_monty.rs:223-237 per element).  flags[i] = 1 where
 * inv returned None (x == 0 or not invertible); out[i] = 0 there. */
int fpo_batch_inv(const word *modulus, size_t n, const word *x, word *out, uint8_t *flags,
                  size_t count, int threads)
{
    word inv, one[FPO_MAX_LIMBS], r2[FPO_MAX_LIMBS];
    if (fpo_params(modulus, n, &inv, one, r2)) return 1;
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
#endif
    for (ptrdiff_t i = 0; i < (ptrdiff_t)count; i++) {
        word tmp[FPO_MAX_LIMBS];
        int ok = fpo_inv(x + (size_t)i * n, modulus, r2, tmp, n);
        if (!ok) memset(tmp, 0, n * sizeof(word));
        memcpy(out + (size_t)i * n, tmp, n * sizeof(word));
        if (flags) flags[i] = (uint8_t)!ok;
    }
    return 0;
}

/* ---- src/semiring.rs:213-226 sum_refs / :228-241 product_refs: sequential
 * left fold from zero / one. */
int fpo_sum(const word *modulus, size_t n, const word *v, size_t count, word *out)
{
    word acc[FPO_MAX_LIMBS], t[FPO_MAX_LIMBS];
    memset(acc, 0, n * sizeof(word));
    for (size_t i = 0; i < count; i++) {
        fpo_add_mod(acc, v + i * n, modulus, t, n);
        memcpy(acc, t, n * sizeof(word));
    }
    memcpy(out, acc, n * sizeof(word));
    return 0;
}
int fpo_product(const word *modulus, size_t n, const word *v, size_t count, word *out)
{
    word inv, one[FPO_MAX_LIMBS], r2[FPO_MAX_LIMBS], acc[FPO_MAX_LIMBS], t[FPO_MAX_LIMBS];
    if (fpo_params(modulus, n, &inv, one, r2)) return 1;
    memcpy(acc, one, n * sizeof(word));
    for (size_t i = 0; i < count; i++) {
        fpo_monty_mul(acc, v + i * n, modulus, t, n);
        memcpy(acc, t, n * sizeof(word));
    }
    memcpy(out, acc, n * sizeof(word));
    return 0;
}

/* Mat-vec over DenseRowMatrix (src/matrix.rs:135-145 row-major data,
 * :162-164 as_rows): y[i] = sum_refs_j mul(A[i][j], x[j])  (semiring.rs:213-226
 * folded over monty_mul), all in Montgomery form. */
int fpo_matvec(const word *modulus, size_t n, const word *A, size_t rows, size_t cols,
               const word *x, word *y, int threads)
{
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
#endif
    for (ptrdiff_t i = 0; i < (ptrdiff_t)rows; i++) {
        word acc[FPO_MAX_LIMBS], t[FPO_MAX_LIMBS], s[FPO_MAX_LIMBS];
        memset(acc, 0, n * sizeof(word));
        for (size_t j = 0; j < cols; j++) {
            fpo_monty_mul(A + ((size_t)i * cols + j) * n, x + j * n, modulus, t, n);
            fpo_add_mod(acc, t, modulus, s, n);
            memcpy(acc, s, n * sizeof(word));
        }
        memcpy(y + (size_t)i * n, acc, n * sizeof(word));
    }
    return 0;
}

/* Sparse mat-vec over SparseMatrix (src/matrix.rs:46-61: `density` cells (col, value) per row,
 * rows back to back): y[i] = sum_refs_j mul(val[i][j], x[col[i][j]]).  Returns 1 on an index out
 * of range (the reference would panic on x[col]). */
int fpo_spmv(const word *modulus, size_t n, const word *vals, const uint32_t *col, size_t rows, size_t cols,
             size_t density, const word *x, word *y)
{
    for (size_t i = 0; i < rows; i++) {
        word acc[FPO_MAX_LIMBS], t[FPO_MAX_LIMBS], s[FPO_MAX_LIMBS];
        memset(acc, 0, n * sizeof(word));
        for (size_t j = 0; j < density; j++) {
            size_t cell = i * density + j;
            if (col[cell] >= cols) return 1;
            fpo_monty_mul(vals + cell * n, x + (size_t)col[cell] * n, modulus, t, n);
            fpo_add_mod(acc, t, modulus, s, n);
            memcpy(acc, s, n * sizeof(word));
        }
        memcpy(y + i * n, acc, n * sizeof(word));
    }
    return 0;
}

/* Mat-mul C[m x nn] = A[m x k] * B[k x nn], row-major, same fold per output. */
int fpo_matmul(const word *modulus, size_t n, const word *A, const word *B, word *C,
               size_t m, size_t k, size_t nn, int threads)
{
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
#endif
    for (ptrdiff_t i = 0; i < (ptrdiff_t)m; i++) {
        for (size_t j = 0; j < nn; j++) {
            word acc[FPO_MAX_LIMBS], t[FPO_MAX_LIMBS], s[FPO_MAX_LIMBS];
            memset(acc, 0, n * sizeof(word));
            for (size_t l = 0; l < k; l++) {
                fpo_monty_mul(A + ((size_t)i * k + l) * n, B + (l * nn + j) * n, modulus, t, n);
                fpo_add_mod(acc, t, modulus, s, n);
                memcpy(acc, s, n * sizeof(word));
            }
            memcpy(C + ((size_t)i * nn + j) * n, acc, n * sizeof(word));
        }
    }
    return 0;
}

/* ---- benches/field_ops_bench_common.rs:27-39 "Mul assign": `iters`
 * dependent mul_assign(acc, elem) on one element; returns acc. */
void fpo_bench_mul_assign(const word *modulus, size_t n, const word *elem, size_t iters, word *out)
{
    word acc[FPO_MAX_LIMBS], t[FPO_MAX_LIMBS];
    memcpy(acc, elem, n * sizeof(word));
    for (size_t i = 0; i < iters; i++) {
        fpo_monty_mul(acc, elem, modulus, t, n);
        memcpy(acc, t, n * sizeof(word));
    }
    memcpy(out, acc, n * sizeof(word));
}

/* ---- benches/field_ops_bench_common.rs:15-107, the whole reference bench shape on ONE
 * element: `iters` (10000 there) repetitions of Mul ref by ref, Mul assign, Add ref to ref,
 * Add assign, Div ref by ref, Negation, Sum over 10, Product over 10.  Writes nanoseconds
 * per group call (i.e. per `iters` operations, as criterion reports them) to ns[0..7].
 * `sink` keeps the results alive (black_box). */
#include <time.h>
static double now_ns(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e9 + (double)ts.tv_nsec;
}
int fpo_bench_reference_shape(const word *modulus, size_t n, const word *elem, size_t iters, double *ns, word *sink)
{
    word inv, one[FPO_MAX_LIMBS], r2[FPO_MAX_LIMBS], acc[FPO_MAX_LIMBS], t[FPO_MAX_LIMBS], q[FPO_MAX_LIMBS];
    word v[10 * FPO_MAX_LIMBS];
    volatile word keep = 0;
    if (fpo_params(modulus, n, &inv, one, r2)) return 1;
    for (int i = 0; i < 10; i++) memcpy(v + (size_t)i * n, elem, n * sizeof(word));
    double t0 = now_ns();
    for (size_t i = 0; i < iters; i++) { fpo_monty_mul(elem, elem, modulus, t, n); keep ^= t[0]; }      /* :15-25 */
    ns[0] = now_ns() - t0; t0 = now_ns();
    memcpy(acc, elem, n * sizeof(word));
    for (size_t i = 0; i < iters; i++) { fpo_monty_mul(acc, elem, modulus, t, n); memcpy(acc, t, n * sizeof(word)); }  /* :27-39 */
    keep ^= acc[0];
    ns[1] = now_ns() - t0; t0 = now_ns();
    for (size_t i = 0; i < iters; i++) { fpo_add_mod(elem, elem, modulus, t, n); keep ^= t[0]; }          /* :41-51 */
    ns[2] = now_ns() - t0; t0 = now_ns();
    memcpy(acc, elem, n * sizeof(word));
    for (size_t i = 0; i < iters; i++) { fpo_add_mod(acc, elem, modulus, t, n); memcpy(acc, t, n * sizeof(word)); }    /* :53-65 */
    keep ^= acc[0];
    ns[3] = now_ns() - t0; t0 = now_ns();
    for (size_t i = 0; i < iters; i++) {                                                               /* :67-77 div = mul by inv */
        if (fpo_inv(elem, modulus, r2, q, n)) fpo_monty_mul(elem, q, modulus, t, n);
        keep ^= t[0];
    }
    ns[4] = now_ns() - t0; t0 = now_ns();
    for (size_t i = 0; i < iters; i++) { fpo_neg_mod(elem, modulus, t, n); keep ^= t[0]; }                /* :79-89 */
    ns[5] = now_ns() - t0; t0 = now_ns();
    for (size_t i = 0; i < iters; i++) {                                                               /* :93-99 */
        memset(acc, 0, n * sizeof(word));
        for (int j = 0; j < 10; j++) { fpo_add_mod(acc, v + (size_t)j * n, modulus, t, n); memcpy(acc, t, n * sizeof(word)); }
        keep ^= acc[0];
    }
    ns[6] = now_ns() - t0; t0 = now_ns();
    for (size_t i = 0; i < iters; i++) {                                                               /* :101-107 */
        memcpy(acc, one, n * sizeof(word));
        for (int j = 0; j < 10; j++) { fpo_monty_mul(acc, v + (size_t)j * n, modulus, t, n); memcpy(acc, t, n * sizeof(word)); }
        keep ^= acc[0];
    }
    ns[7] = now_ns() - t0;
    *sink = keep;
    return 0;
}

/* ======================= pow =====================
 * ---- src/helpers/crypto_bigint.rs:441-454 conditional_borrowing_sub_assign */
static int conditional_borrowing_sub_assign(word *lhs, const word *rhs, int choice, size_t n)
{
    word mask = choice ? ~(word)0 : 0, borrow = 0;
    for (size_t i = 0; i < n; i++) {
        wide t = (wide)lhs[i] - (wide)(rhs[i] & mask) - (wide)borrow;
        lhs[i] = (word)t;
        borrow = (word)((t >> 64) & 1);
    }
    return (int)borrow;
}
/* ---- :413-426 almost_montgomery_mul: out = x*y/R reduced to the bit length of the
 * modulus (may still exceed it). */
static void almost_montgomery_mul(const word *x, const word *y, word *out, const word *modulus,
                                  word mod_neg_inv, size_t n)
{
    memset(out, 0, n * sizeof(word));
    word overflow = fpo_monty_mul_limbs(x, y, out, modulus, mod_neg_inv, n);
    conditional_borrowing_sub_assign(out, modulus, (int)(overflow & 1), n);
}
/* ---- :436-439 almost_montgomery_reduce: two conditional subtractions */
static void almost_montgomery_reduce(word *z, const word *modulus, size_t n)
{
    conditional_borrowing_sub_assign(z, modulus, cmp_n(z, modulus, n) >= 0, n);
    conditional_borrowing_sub_assign(z, modulus, cmp_n(z, modulus, n) >= 0, n);
}
/* ---- :239-315 pow_montgomery_form_amm + :336-361 pow_bounded_exp: 4-bit fixed window,
 * "almost Montgomery" multiplications, fully reduced result in Montgomery form.
 * exponent: little-endian 64-bit limbs; exponent_bits: how many low bits count. */
void fpo_pow_bounded_exp(const word *x, const word *exponent, uint32_t exponent_bits, const word *one,
                         const word *modulus, word mod_neg_inv, word *out, size_t n)
{
    enum { WINDOW = 4, WINDOW_COUNT = 16, BITS_PER_WINDOW = 64 / 4 };
    if (exponent_bits == 0) { memcpy(out, one, n * sizeof(word)); return; }
    word powers[WINDOW_COUNT][FPO_MAX_LIMBS], prod[FPO_MAX_LIMBS], z[FPO_MAX_LIMBS], power[FPO_MAX_LIMBS];
    /* powers[i] = x^i (:258-270) */
    memcpy(powers[0], one, n * sizeof(word));
    memcpy(powers[1], x, n * sizeof(word));
    for (int i = 2; i < WINDOW_COUNT; i++) {
        almost_montgomery_mul(powers[i - 1], x, prod, modulus, mod_neg_inv, n);
        memcpy(powers[i], prod, n * sizeof(word));
    }
    /* pow_init (:428-434) */
    uint32_t starting_limb = (exponent_bits - 1) / 64;
    uint32_t starting_bit_in_limb = (exponent_bits - 1) % 64;
    uint32_t starting_window = starting_bit_in_limb / WINDOW;
    word starting_window_mask = ((word)1 << (starting_bit_in_limb % WINDOW + 1)) - 1;
    memcpy(z, one, n * sizeof(word));
    for (int limb_num = (int)starting_limb; limb_num >= 0; limb_num--) {
        word w = exponent[limb_num];
        uint32_t window_num = (uint32_t)limb_num == starting_limb ? starting_window + 1 : BITS_PER_WINDOW;
        while (window_num > 0) {
            window_num--;
            word idx = (w >> (window_num * WINDOW)) & 15;
            if ((uint32_t)limb_num == starting_limb && window_num == starting_window) {
                idx &= starting_window_mask;
            } else {
                for (int sq = 0; sq < WINDOW; sq++) {
                    almost_montgomery_mul(z, z, prod, modulus, mod_neg_inv, n);
                    memcpy(z, prod, n * sizeof(word));
                }
            }
            memcpy(power, powers[idx], n * sizeof(word));
            almost_montgomery_mul(z, power, prod, modulus, mod_neg_inv, n);
            memcpy(z, prod, n * sizeof(word));
        }
    }
    almost_montgomery_reduce(z, modulus, n);
    memcpy(out, z, n * sizeof(word));
}
/* Batch: one exponent for every element (MontyField::pow_u32 / ::pow,
 * src/field/crypto_bigint_monty.rs:163-172,239-249). */
int fpo_batch_pow(const word *modulus, size_t n, const word *x, const word *exponent, uint32_t exponent_bits,
                  word *out, size_t count, int threads)
{
    word inv, one[FPO_MAX_LIMBS], r2[FPO_MAX_LIMBS];
    if (fpo_params(modulus, n, &inv, one, r2)) return 1;
    (void)threads;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static) if (threads != 1)
#endif
    for (ptrdiff_t i = 0; i < (ptrdiff_t)count; i++) {
        word tmp[FPO_MAX_LIMBS];
        fpo_pow_bounded_exp(x + (size_t)i * n, exponent, exponent_bits, one, modulus, inv, tmp, n);
        memcpy(out + (size_t)i * n, tmp, n * sizeof(word));
    }
    return 0;
}

int fpo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
