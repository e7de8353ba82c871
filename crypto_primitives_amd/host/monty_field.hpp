// monty_field.hpp -- C++ host-side mirror of crypto-primitives' field interface for the
// Montgomery hot path, over the C ABI of libfpgpu (include/fpgpu.h).
//
// The reference is Rust; no Rust toolchain exists in the build image, so the host layer a
// Rust maintainer would write as `impl BatchFieldConfig for MontyField<LIMBS>`
//  is written here in C++ with the same names, argument meaning and error
// behaviour:
//
//   MontyField<LIMBS>            src/field/crypto_bigint_monty.rs:16-52,134-250,319-385
//   MontyFieldElement<LIMBS>     src/field/crypto_bigint_monty.rs:54-86
//   BoxedMontyField              src/field/crypto_bigint_boxed_monty.rs (runtime limb count)
//   ConstMontyField<Mod, LIMBS>  src/field/crypto_bigint_const_monty.rs (element style, operators)
//   DenseRowMatrix<T>            src/matrix.rs:135-164
//   SparseMatrix<T>              src/matrix.rs:46-61
//   FieldError::InvalidModulus   src/field.rs:345-349;  "Division by zero"  src/field.rs:155-157
//
// As in the reference, SCALAR methods (add, mul, inv, project, lift, ...) run on the host --
// a kernel launch per scalar op would be nonsense -- and the BATCH extension (mul_batch,
// inv_batch, matvec, ...) runs on the MI355X through libfpgpu, with no CPU fallback: batch
// calls throw FpgpuError when the library reports an error (e.g. no device).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fpgpu.h"

namespace crypto_primitives {

using u128 = unsigned __int128;

enum class FieldError { InvalidModulus };

struct FieldErrorException : std::runtime_error {
    FieldError kind;
    explicit FieldErrorException(FieldError k) : std::runtime_error("InvalidModulus"), kind(k) {}
};
struct FpgpuError : std::runtime_error {
    int code;
    FpgpuError(int c, const char* msg) : std::runtime_error(std::string("fpgpu: ") + msg), code(c) {}
};
inline void fpgpu_check(int rc) {
    if (rc != FPGPU_OK) throw FpgpuError(rc, fpgpu_last_error_string());
}

// ---- Uint<LIMBS>: little-endian 64-bit words (crypto_bigint_uint.rs:26-73) ---------------
template <size_t LIMBS>
struct Uint {
    std::array<uint64_t, LIMBS> words{};
    static constexpr size_t kLimbs = LIMBS;
    static Uint from_u64(uint64_t v) { Uint r; r.words[0] = v; return r; }
    static Uint from_u128(u128 v) {
        Uint r;
        r.words[0] = (uint64_t)v;
        if (LIMBS > 1) r.words[1] = (uint64_t)(v >> 64);
        return r;
    }
    static Uint from_be_hex(const std::string& hex) {
        Uint r;
        size_t bit = 0;
        for (size_t i = hex.size(); i-- > 0;) {
            char c = hex[i];
            if (c == '_') continue;
            uint64_t d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 99;
            if (d > 15) throw std::invalid_argument("bad hex digit");
            if (bit / 64 < LIMBS) r.words[bit / 64] |= d << (bit % 64);
            else if (d) throw std::invalid_argument("hex value does not fit");
            bit += 4;
        }
        return r;
    }
    const std::array<uint64_t, LIMBS>& as_words() const { return words; }
    static Uint from_words(const std::array<uint64_t, LIMBS>& w) { Uint r; r.words = w; return r; }
    bool is_zero() const { uint64_t z = 0; for (auto w : words) z |= w; return z == 0; }
    bool is_odd() const { return words[0] & 1; }
    bool operator==(const Uint& o) const { return words == o.words; }
    bool operator!=(const Uint& o) const { return !(*this == o); }
    bool operator<(const Uint& o) const {
        for (size_t i = LIMBS; i-- > 0;) if (words[i] != o.words[i]) return words[i] < o.words[i];
        return false;
    }
};

namespace detail {
// plain word-vector helpers (n words, little endian)
inline int cmp(const uint64_t* a, const uint64_t* b, size_t n) {
    for (size_t i = n; i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
}
inline uint64_t add(uint64_t* r, const uint64_t* a, const uint64_t* b, size_t n) {
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) { u128 t = (u128)a[i] + b[i] + c; r[i] = (uint64_t)t; c = (uint64_t)(t >> 64); }
    return c;
}
inline uint64_t sub(uint64_t* r, const uint64_t* a, const uint64_t* b, size_t n) {
    uint64_t bo = 0;
    for (size_t i = 0; i < n; i++) { u128 t = (u128)a[i] - b[i] - bo; r[i] = (uint64_t)t; bo = (uint64_t)(t >> 64) & 1; }
    return bo;
}
inline bool is_zero(const uint64_t* a, size_t n) { uint64_t z = 0; for (size_t i = 0; i < n; i++) z |= a[i]; return z == 0; }
inline void shr1(uint64_t* a, size_t n, uint64_t top) {
    for (size_t i = 0; i + 1 < n; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63);
    a[n - 1] = (a[n - 1] >> 1) | (top << 63);
}

// Host scalar Montgomery arithmetic for one modulus (n 64-bit words, R = 2^(64 n)).
struct HostMonty {
    size_t n = 0;
    std::vector<uint64_t> m, one, r2;
    uint64_t ninv = 0;  // -m^-1 mod 2^64

    // out = a*b*R^-1 mod m (operand scanning, 64-bit limbs; subtract m once iff overflow or >= m)
    void mul(uint64_t* out, const uint64_t* a, const uint64_t* b) const {
        std::vector<uint64_t> t(n + 2, 0);
        for (size_t i = 0; i < n; i++) {
            uint64_t c = 0;
            for (size_t j = 0; j < n; j++) { u128 x = (u128)a[i] * b[j] + t[j] + c; t[j] = (uint64_t)x; c = (uint64_t)(x >> 64); }
            u128 x = (u128)t[n] + c; t[n] = (uint64_t)x; t[n + 1] = (uint64_t)(x >> 64);
            uint64_t q = t[0] * ninv;
            x = (u128)q * m[0] + t[0]; c = (uint64_t)(x >> 64);
            for (size_t j = 1; j < n; j++) { x = (u128)q * m[j] + t[j] + c; t[j - 1] = (uint64_t)x; c = (uint64_t)(x >> 64); }
            x = (u128)t[n] + c; t[n - 1] = (uint64_t)x; t[n] = t[n + 1] + (uint64_t)(x >> 64);
        }
        std::vector<uint64_t> d(n);
        uint64_t bo = sub(d.data(), t.data(), m.data(), n);
        bool use_d = t[n] != 0 || bo == 0;
        std::memcpy(out, use_d ? d.data() : t.data(), n * 8);
    }
    void addm(uint64_t* out, const uint64_t* a, const uint64_t* b) const {
        std::vector<uint64_t> s(n), d(n);
        uint64_t c = add(s.data(), a, b, n);
        uint64_t bo = sub(d.data(), s.data(), m.data(), n);
        std::memcpy(out, (c < bo) ? s.data() : d.data(), n * 8);
    }
    void subm(uint64_t* out, const uint64_t* a, const uint64_t* b) const {
        std::vector<uint64_t> d(n);
        if (sub(d.data(), a, b, n)) add(d.data(), d.data(), m.data(), n);
        std::memcpy(out, d.data(), n * 8);
    }
    void negm(uint64_t* out, const uint64_t* a) const {
        if (is_zero(a, n)) { std::memset(out, 0, n * 8); return; }
        std::vector<uint64_t> d(n);
        sub(d.data(), m.data(), a, n);
        std::memcpy(out, d.data(), n * 8);
    }
    // lift: x*R^-1 mod m = mul(x, 1)
    void from_mont(uint64_t* out, const uint64_t* a) const {
        std::vector<uint64_t> e(n, 0); e[0] = 1;
        mul(out, a, e.data());
    }
    void to_mont(uint64_t* out, const uint64_t* a) const { mul(out, a, r2.data()); }
    // Montgomery-form inverse (x = XR -> X^-1 R), false when not invertible.
    bool inv(uint64_t* out, const uint64_t* x) const {
        std::vector<uint64_t> u(x, x + n), v(m), x1(n, 0), x2(n, 0), one_(n, 0);
        one_[0] = 1; x1[0] = 1;
        while (cmp(u.data(), m.data(), n) >= 0) sub(u.data(), u.data(), m.data(), n);
        if (is_zero(u.data(), n)) return false;
        while (cmp(u.data(), one_.data(), n) != 0 && cmp(v.data(), one_.data(), n) != 0) {
            if (is_zero(u.data(), n) || is_zero(v.data(), n)) return false;
            while (!(u[0] & 1)) {
                shr1(u.data(), n, 0);
                uint64_t c = (x1[0] & 1) ? add(x1.data(), x1.data(), m.data(), n) : 0;
                shr1(x1.data(), n, c);
            }
            while (!(v[0] & 1)) {
                shr1(v.data(), n, 0);
                uint64_t c = (x2[0] & 1) ? add(x2.data(), x2.data(), m.data(), n) : 0;
                shr1(x2.data(), n, c);
            }
            if (cmp(u.data(), v.data(), n) >= 0) {
                sub(u.data(), u.data(), v.data(), n);
                if (sub(x1.data(), x1.data(), x2.data(), n)) add(x1.data(), x1.data(), m.data(), n);
            } else {
                sub(v.data(), v.data(), u.data(), n);
                if (sub(x2.data(), x2.data(), x1.data(), n)) add(x2.data(), x2.data(), m.data(), n);
            }
        }
        const std::vector<uint64_t>& plain = cmp(u.data(), one_.data(), n) == 0 ? x1 : x2;
        std::vector<uint64_t> t(n);
        mul(t.data(), plain.data(), r2.data());
        mul(out, t.data(), r2.data());
        return true;
    }
};

struct CtxDeleter { void operator()(fpgpu_ctx* c) const { fpgpu_ctx_destroy(c); } };
}  // namespace detail

// ---- DenseRowMatrix<T> (src/matrix.rs:135-164) -------------------------------------------
template <typename T>
struct DenseRowMatrix {
    size_t num_rows = 0, num_cols = 0;
    std::vector<T> data;  // linearized rows
    static DenseRowMatrix from_rows(const std::vector<std::vector<T>>& rows) {
        DenseRowMatrix m;
        m.num_rows = rows.size();
        for (auto& r : rows) m.num_cols = std::max(m.num_cols, r.size());
        for (auto& r : rows) m.data.insert(m.data.end(), r.begin(), r.end());
        return m;
    }
    bool is_empty() const { return num_rows == 0 || num_cols == 0; }
    const T* row(size_t i) const { return data.data() + i * num_cols; }
};

// ---- SparseMatrix<T> (src/matrix.rs:46-61): a fixed number of cells per row ---------------
template <typename T>
struct SparseMatrix {
    size_t num_rows = 0, num_cols = 0, density = 0;
    std::vector<std::pair<size_t, T>> cells;  // (column, value), rows back to back
    bool is_empty() const { return num_rows == 0 || num_cols == 0; }
};

// ---- DeviceVec: the batch extension's device-side container -------------------------------
// The reference keeps vectors in host memory (Vec<T>, DenseRowMatrix<T>::data, one heap block per
// BoxedMontyFieldElement); a DeviceVec is the same elements living in HBM between batch calls
// (fpgpu_vec of include/fpgpu.h): upload once, chain operations with no host copy, download once.
// Placement over the device list of set_devices(): one device, sharded (contiguous ranges / whole
// rows per device) or replicated.
enum class Placement : int { Current = FPGPU_PLACE_CURRENT, Sharded = FPGPU_PLACE_SHARDED, Replicated = FPGPU_PLACE_REPLICATED };
inline void set_devices(const std::vector<int>& ids) { fpgpu_check(fpgpu_set_devices(ids.data(), (int)ids.size())); }

// A device list as a value (fpgpu_devset): vectors made with it do not depend on the process-wide list.
class DeviceSet {
   public:
    explicit DeviceSet(const std::vector<int>& ids) {
        fpgpu_devset* raw = nullptr;
        fpgpu_check(fpgpu_devset_create(ids.data(), (int)ids.size(), &raw));
        h_.reset(raw, [](fpgpu_devset* s) { fpgpu_devset_destroy(s); });
    }
    const fpgpu_devset* raw() const { return h_.get(); }

   private:
    std::shared_ptr<fpgpu_devset> h_;
};

class DeviceVec {
   public:
    DeviceVec() = default;
    DeviceVec(const fpgpu_ctx* ctx, size_t n, Placement where, size_t row_len = 0, const DeviceSet* on = nullptr) {
        fpgpu_vec* raw = nullptr;
        const fpgpu_devset* set = on ? on->raw() : nullptr;
        if (row_len) fpgpu_check(fpgpu_vec_alloc_rows_on(ctx, set, n / row_len, row_len, (int)where, &raw));
        else fpgpu_check(fpgpu_vec_alloc_on(ctx, set, n, (int)where, &raw));
        h_.reset(raw, [](fpgpu_vec* v) { fpgpu_vec_free(v); });
    }
    size_t size() const { return h_ ? fpgpu_vec_len(h_.get()) : 0; }
    int parts() const { return h_ ? fpgpu_vec_parts(h_.get()) : 0; }
    fpgpu_vec* raw() const { return h_.get(); }
    void upload(const uint64_t* words, size_t first, size_t n) { fpgpu_check(fpgpu_vec_upload(h_.get(), words, first, n)); }
    void download(uint64_t* words, size_t first, size_t n) const { fpgpu_check(fpgpu_vec_download(h_.get(), words, first, n)); }
    void sync() const { fpgpu_check(fpgpu_vec_sync(h_.get())); }
    void copy_from(const DeviceVec& src) { fpgpu_check(fpgpu_vec_copy(h_.get(), src.raw())); }

   private:
    std::shared_ptr<fpgpu_vec> h_;
};

namespace detail {
// The handle-taking batch operations, shared by MontyField<LIMBS> and BoxedMontyField.
struct DeviceOps {
    const fpgpu_ctx* ctx;
    void mul(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const { fpgpu_check(fpgpu_vec_mul(ctx, a.raw(), b.raw(), out.raw())); }
    void add(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const { fpgpu_check(fpgpu_vec_add(ctx, a.raw(), b.raw(), out.raw())); }
    void sub(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const { fpgpu_check(fpgpu_vec_sub(ctx, a.raw(), b.raw(), out.raw())); }
    void neg(const DeviceVec& a, DeviceVec& out) const { fpgpu_check(fpgpu_vec_neg(ctx, a.raw(), out.raw())); }
    void dbl(const DeviceVec& a, DeviceVec& out) const { fpgpu_check(fpgpu_vec_double(ctx, a.raw(), out.raw())); }
    void div_by_2(const DeviceVec& a, DeviceVec& out) const { fpgpu_check(fpgpu_vec_div_by_2(ctx, a.raw(), out.raw())); }
    void square(const DeviceVec& a, DeviceVec& out) const { fpgpu_check(fpgpu_vec_sqr(ctx, a.raw(), out.raw())); }
    void project(const DeviceVec& a, DeviceVec& out) const { fpgpu_check(fpgpu_vec_to_mont(ctx, a.raw(), out.raw())); }
    void lift(const DeviceVec& a, DeviceVec& out) const { fpgpu_check(fpgpu_vec_from_mont(ctx, a.raw(), out.raw())); }
    void pow(const DeviceVec& a, const uint64_t* e, uint32_t words, DeviceVec& out) const {
        fpgpu_check(fpgpu_vec_pow(ctx, a.raw(), e, words, out.raw()));
    }
    // flags[i] = 1 where the scalar inv returns None (out[i] = 0 there)
    std::vector<uint8_t> inv(const DeviceVec& x, DeviceVec& out) const {
        std::vector<uint8_t> none(x.size(), 0);
        fpgpu_check(fpgpu_vec_batch_inv(ctx, x.raw(), out.raw(), none.data(), nullptr));
        return none;
    }
    // a / b; flags[i] = 1 where checked_div returns None
    std::vector<uint8_t> div(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const {
        std::vector<uint8_t> none(a.size(), 0);
        fpgpu_check(fpgpu_vec_div(ctx, a.raw(), b.raw(), out.raw(), none.data(), nullptr));
        return none;
    }
    void matvec(const DeviceVec& A, size_t rows, size_t cols, const DeviceVec& x, DeviceVec& y) const {
        fpgpu_check(fpgpu_vec_matvec(ctx, A.raw(), rows, cols, x.raw(), y.raw()));
    }
    void matmul(const DeviceVec& A, const DeviceVec& B, DeviceVec& C, size_t m, size_t k, size_t n) const {
        fpgpu_check(fpgpu_vec_matmul(ctx, A.raw(), B.raw(), C.raw(), m, k, n));
    }
    // y = A x over SparseMatrix: values and column indices in two device containers with one row partition
    void spmv(const DeviceVec& vals, const fpgpu_idx* col_idx, size_t rows, size_t cols, size_t density, const DeviceVec& x,
              DeviceVec& y) const {
        fpgpu_check(fpgpu_vec_spmv(ctx, vals.raw(), col_idx, rows, cols, density, x.raw(), y.raw()));
    }
};

// Device-resident column indices of a SparseMatrix (fpgpu_idx), laid out like its values.
class DeviceIndices {
   public:
    DeviceIndices(size_t rows, size_t density, Placement where) {
        fpgpu_idx* raw = nullptr;
        fpgpu_check(fpgpu_idx_alloc_rows(rows, density, (int)where, &raw));
        h_.reset(raw, [](fpgpu_idx* i) { fpgpu_idx_free(i); });
    }
    void upload(const std::vector<uint32_t>& cols) { fpgpu_check(fpgpu_idx_upload(h_.get(), cols.data(), 0, cols.size())); }
    const fpgpu_idx* raw() const { return h_.get(); }

   private:
    std::shared_ptr<fpgpu_idx> h_;
};
}  // namespace detail
using detail::DeviceIndices;

// ---- MontyFieldElement<LIMBS> (repr(transparent) over Uint) ------------------------------
template <size_t LIMBS>
struct MontyFieldElement {
    Uint<LIMBS> v;  // Montgomery representative
    bool operator==(const MontyFieldElement& o) const { return v == o.v; }
    bool operator!=(const MontyFieldElement& o) const { return !(v == o.v); }
    bool operator<(const MontyFieldElement& o) const { return v < o.v; }  // compares Montgomery form, as the reference
    const Uint<LIMBS>& inner() const { return v; }
};
static_assert(sizeof(MontyFieldElement<4>) == 32, "element must be exactly LIMBS words for the C ABI");

// ---- MontyField<LIMBS>: runtime modulus, config style -------------------------------------
template <size_t LIMBS>
class MontyField {
   public:
    using Element = MontyFieldElement<LIMBS>;
    using Integer = Uint<LIMBS>;

    // BaseFieldConfig::new: rejects an even modulus with FieldError::InvalidModulus.
    static MontyField new_(const Integer& modulus) {
        fpgpu_ctx* raw = nullptr;
        int rc = fpgpu_ctx_create(modulus.words.data(), (uint32_t)LIMBS, &raw);
        if (rc == FPGPU_ERR_INVALID_MODULUS) throw FieldErrorException(FieldError::InvalidModulus);
        fpgpu_check(rc);
        MontyField f;
        f.ctx_.reset(raw, detail::CtxDeleter());
        f.h_.n = LIMBS;
        f.h_.m.assign(LIMBS, 0); f.h_.one.assign(LIMBS, 0); f.h_.r2.assign(LIMBS, 0);
        fpgpu_check(fpgpu_ctx_params(raw, f.h_.m.data(), f.h_.one.data(), f.h_.r2.data(), &f.h_.ninv));
        return f;
    }
    static std::optional<MontyField> try_new(const Integer& modulus) {
        try { return new_(modulus); } catch (const FieldErrorException&) { return std::nullopt; }
    }

    Integer modulus() const { Integer r; std::memcpy(r.words.data(), h_.m.data(), LIMBS * 8); return r; }
    Integer modulus_minus_one_div_two() const {
        Integer r = modulus();
        r.words[0] -= 1;  // modulus is odd
        detail::shr1(r.words.data(), LIMBS, 0);
        return r;
    }
    const fpgpu_ctx* gpu_ctx() const { return ctx_.get(); }

    // --- scalar SemiringConfig / RingConfig / FieldConfig (host) ---
    bool is_zero(const Element& x) const { return x.v.is_zero(); }
    Element zero() const { return Element{}; }
    Element one() const { Element e; std::memcpy(e.v.words.data(), h_.one.data(), LIMBS * 8); return e; }
    Element add(const Element& x, const Element& y) const { Element r; h_.addm(w(r), w(x), w(y)); return r; }
    Element sub(const Element& x, const Element& y) const { Element r; h_.subm(w(r), w(x), w(y)); return r; }
    Element mul(const Element& x, const Element& y) const { Element r; h_.mul(w(r), w(x), w(y)); return r; }
    Element neg(const Element& x) const { Element r; h_.negm(w(r), w(x)); return r; }
    void add_assign(Element& x, const Element& y) const { x = add(x, y); }
    void sub_assign(Element& x, const Element& y) const { x = sub(x, y); }
    void mul_assign(Element& x, const Element& y) const { x = mul(x, y); }
    std::optional<Element> inv(const Element& x) const {
        Element r;
        if (!h_.inv(w(r), w(x))) return std::nullopt;
        return r;
    }
    std::optional<Element> checked_div(const Element& x, const Element& y) const {
        auto i = inv(y);
        if (!i) return std::nullopt;
        return mul(x, *i);
    }
    Element div(const Element& x, const Element& y) const {
        auto q = checked_div(x, y);
        if (!q) throw std::domain_error("Division by zero");
        return *q;
    }
    void div_assign(Element& x, const Element& y) const { x = div(x, y); }
    Element pow_u32(const Element& x, uint32_t e) const {
        Element acc = one(), base = x;
        for (; e; e >>= 1) { if (e & 1) acc = mul(acc, base); base = mul(base, base); }
        return acc;
    }
    template <typename It> Element sum_refs(It first, It last) const {
        Element acc = zero();
        for (; first != last; ++first) acc = add(acc, *first);
        return acc;
    }
    template <typename It> Element product_refs(It first, It last) const {
        Element acc = one();
        for (; first != last; ++first) acc = mul(acc, *first);
        return acc;
    }
    // --- conversions: project (into Montgomery form) / lift (back to a reduced integer) ---
    Element project(const Integer& value) const { Element r; h_.to_mont(w(r), value.words.data()); return r; }
    Element project(uint64_t value) const { return project(Integer::from_u64(value)); }
    Element project_u128(u128 value) const { return project(Integer::from_u128(value)); }
    Element project(int64_t value) const {
        uint64_t mag = value < 0 ? (uint64_t)0 - (uint64_t)value : (uint64_t)value;
        Element e = project(Integer::from_u64(mag));
        return value < 0 ? neg(e) : e;
    }
    Element project(int value) const { return project((int64_t)value); }
    Element project(bool value) const { return value ? one() : zero(); }
    // A Uint wider than the modulus is reduced first (crypto_bigint_monty.rs:319-340).
    template <size_t L2> Element project_wide(const Uint<L2>& value) const {
        static_assert(L2 >= LIMBS, "use project for narrower values");
        std::vector<uint64_t> rem(L2 + 1, 0);  // binary long division by the modulus
        for (size_t bit = 64 * L2; bit-- > 0;) {
            uint64_t top = rem[LIMBS - 1] >> 63;
            for (size_t i = LIMBS; i-- > 1;) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 63);
            rem[0] = (rem[0] << 1) | ((value.words[bit / 64] >> (bit % 64)) & 1);
            if (top || detail::cmp(rem.data(), h_.m.data(), LIMBS) >= 0) detail::sub(rem.data(), rem.data(), h_.m.data(), LIMBS);
        }
        Integer v; std::memcpy(v.words.data(), rem.data(), LIMBS * 8);
        return project(v);
    }
    Integer lift(const Element& x) const { Integer r; h_.from_mont(r.words.data(), w(x)); return r; }
    bool is_odd(const Element& x) const { return lift(x).is_odd(); }
    bool is_even(const Element& x) const { return !is_odd(x); }

    // --- BATCH extension: runs on the MI355X through libfpgpu, no CPU fallback -------------
    void mul_batch(const Element* a, const Element* b, Element* out, size_t n) const { fpgpu_check(fpgpu_mul(ctx_.get(), p(a), p(b), p(out), n)); }
    void add_batch(const Element* a, const Element* b, Element* out, size_t n) const { fpgpu_check(fpgpu_add(ctx_.get(), p(a), p(b), p(out), n)); }
    void sub_batch(const Element* a, const Element* b, Element* out, size_t n) const { fpgpu_check(fpgpu_sub(ctx_.get(), p(a), p(b), p(out), n)); }
    void neg_batch(const Element* a, Element* out, size_t n) const { fpgpu_check(fpgpu_neg(ctx_.get(), p(a), p(out), n)); }
    void square_batch(const Element* a, Element* out, size_t n) const { fpgpu_check(fpgpu_sqr(ctx_.get(), p(a), p(out), n)); }
    // ConstMontyField::double / div_by_2 (crypto_bigint_const_monty.rs:81-89)
    void double_batch(const Element* a, Element* out, size_t n) const { fpgpu_check(fpgpu_double(ctx_.get(), p(a), p(out), n)); }
    void div_by_2_batch(const Element* a, Element* out, size_t n) const { fpgpu_check(fpgpu_div_by_2(ctx_.get(), p(a), p(out), n)); }
    void project_batch(const Integer* a, Element* out, size_t n) const { fpgpu_check(fpgpu_to_mont(ctx_.get(), a[0].words.data(), p(out), n)); }
    void lift_batch(const Element* a, Integer* out, size_t n) const { fpgpu_check(fpgpu_from_mont(ctx_.get(), p(a), out[0].words.data(), n)); }
    // pow with one exponent for the whole batch (FieldConfig::pow, crypto_bigint_monty.rs:239-249)
    void pow_batch(const Element* a, const Integer& exponent, Element* out, size_t n) const {
        fpgpu_check(fpgpu_pow(ctx_.get(), p(a), exponent.words.data(), (uint32_t)LIMBS, p(out), n));
    }
    void pow_u32_batch(const Element* a, uint32_t exponent, Element* out, size_t n) const {
        uint64_t e = exponent;
        fpgpu_check(fpgpu_pow(ctx_.get(), p(a), &e, 1, p(out), n));
    }
    // SemiringConfig::sum_refs / product_refs over a contiguous slice (src/semiring.rs:213-241)
    Element sum_batch(const Element* v, size_t n) const { Element r; fpgpu_check(fpgpu_sum(ctx_.get(), p(v), n, p(&r))); return r; }
    Element product_batch(const Element* v, size_t n) const { Element r; fpgpu_check(fpgpu_product(ctx_.get(), p(v), n, p(&r))); return r; }
    std::vector<Element> mul_batch(const std::vector<Element>& a, const std::vector<Element>& b) const {
        if (a.size() != b.size()) throw std::invalid_argument("length mismatch");
        std::vector<Element> out(a.size());
        if (!a.empty()) mul_batch(a.data(), b.data(), out.data(), a.size());
        return out;
    }
    // None where the scalar inv returns None.
    std::vector<std::optional<Element>> inv_batch(const std::vector<Element>& x) const {
        std::vector<Element> out(x.size());
        std::vector<uint8_t> none(x.size(), 0);
        if (!x.empty()) fpgpu_check(fpgpu_batch_inv(ctx_.get(), p(x.data()), p(out.data()), none.data(), x.size()));
        std::vector<std::optional<Element>> r(x.size());
        for (size_t i = 0; i < x.size(); i++) if (!none[i]) r[i] = out[i];
        return r;
    }
    // a[i] / b[i]; panics ("Division by zero") if any divisor has no inverse, like FieldConfig::div.
    std::vector<Element> div_batch(const std::vector<Element>& a, const std::vector<Element>& b) const {
        auto q = checked_div_batch(a, b);
        std::vector<Element> out(a.size());
        for (size_t i = 0; i < a.size(); i++) { if (!q[i]) throw std::domain_error("Division by zero"); out[i] = *q[i]; }
        return out;
    }
    // checked_div (src/field.rs:168-170) on slices: one device call (inversion + multiplication)
    std::vector<std::optional<Element>> checked_div_batch(const std::vector<Element>& a, const std::vector<Element>& b) const {
        if (a.size() != b.size()) throw std::invalid_argument("length mismatch");
        std::vector<Element> out(a.size());
        std::vector<uint8_t> none(a.size(), 0);
        if (!a.empty()) fpgpu_check(fpgpu_div(ctx_.get(), p(a.data()), p(b.data()), p(out.data()), none.data(), a.size()));
        std::vector<std::optional<Element>> r(a.size());
        for (size_t i = 0; i < a.size(); i++) if (!none[i]) r[i] = out[i];
        return r;
    }
    // y = A x : the sum_refs fold of mul over DenseRowMatrix::as_rows()
    std::vector<Element> matvec(const DenseRowMatrix<Element>& A, const std::vector<Element>& x) const {
        if (A.num_cols != x.size() || A.data.size() != A.num_rows * A.num_cols) throw std::invalid_argument("shape mismatch");
        std::vector<Element> y(A.num_rows);
        if (A.num_rows) fpgpu_check(fpgpu_matvec(ctx_.get(), p(A.data.data()), A.num_rows, A.num_cols, p(x.data()), p(y.data())));
        return y;
    }
    DenseRowMatrix<Element> matmul(const DenseRowMatrix<Element>& A, const DenseRowMatrix<Element>& B) const {
        if (A.num_cols != B.num_rows) throw std::invalid_argument("shape mismatch");
        DenseRowMatrix<Element> C;
        C.num_rows = A.num_rows; C.num_cols = B.num_cols; C.data.resize(C.num_rows * C.num_cols);
        if (!C.data.empty())
            fpgpu_check(fpgpu_matmul(ctx_.get(), p(A.data.data()), p(B.data.data()), p(C.data.data()), A.num_rows, A.num_cols, B.num_cols));
        return C;
    }
    // y = A x over SparseMatrix: the cells' halves travel as two arrays (a Rust tuple has no defined
    // layout either); an out-of-range column is the reference's index panic
    std::vector<Element> spmv(const SparseMatrix<Element>& A, const std::vector<Element>& x) const {
        if (A.num_cols != x.size() || A.cells.size() != A.num_rows * A.density) throw std::invalid_argument("shape mismatch");
        std::vector<Element> vals(A.cells.size()), y(A.num_rows);
        std::vector<uint32_t> cols(A.cells.size());
        for (size_t i = 0; i < A.cells.size(); i++) {
            if (A.cells[i].first >= A.num_cols) throw std::out_of_range("column index out of range");
            cols[i] = (uint32_t)A.cells[i].first;
            vals[i] = A.cells[i].second;
        }
        if (A.num_rows)
            fpgpu_check(fpgpu_spmv(ctx_.get(), p(vals.data()), cols.data(), A.num_rows, A.num_cols, A.density, p(x.data()), p(y.data())));
        return y;
    }

    // --- BATCH extension on device-resident vectors (fpgpu_vec): no host copy between calls -------
    // (`on`: a DeviceSet to lay the vector out over instead of the process-wide list of set_devices)
    DeviceVec alloc_device(size_t n, Placement where = Placement::Current, const DeviceSet* on = nullptr) const {
        return DeviceVec(ctx_.get(), n, where, 0, on);
    }
    DeviceVec alloc_device_rows(size_t rows, size_t cols, Placement where, const DeviceSet* on = nullptr) const {
        return DeviceVec(ctx_.get(), rows * cols, where, cols, on);
    }
    DeviceVec to_device(const std::vector<Element>& v, Placement where = Placement::Current, size_t row_len = 0,
                        const DeviceSet* on = nullptr) const {
        DeviceVec d(ctx_.get(), v.size(), where, row_len, on);
        if (!v.empty()) d.upload(p(v.data()), 0, v.size());
        return d;
    }
    DeviceVec to_device(const DenseRowMatrix<Element>& A, Placement where) const { return to_device(A.data, where, A.num_cols); }
    std::vector<Element> from_device(const DeviceVec& d) const {
        std::vector<Element> v(d.size());
        if (!v.empty()) d.download(p(v.data()), 0, v.size());
        return v;
    }
    detail::DeviceOps on_device() const { return detail::DeviceOps{ctx_.get()}; }
    void mul_batch(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const { on_device().mul(a, b, out); }
    void add_batch(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const { on_device().add(a, b, out); }
    void sub_batch(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const { on_device().sub(a, b, out); }
    void neg_batch(const DeviceVec& a, DeviceVec& out) const { on_device().neg(a, out); }
    void square_batch(const DeviceVec& a, DeviceVec& out) const { on_device().square(a, out); }
    void double_batch(const DeviceVec& a, DeviceVec& out) const { on_device().dbl(a, out); }
    void div_by_2_batch(const DeviceVec& a, DeviceVec& out) const { on_device().div_by_2(a, out); }
    void pow_batch(const DeviceVec& a, const Integer& e, DeviceVec& out) const { on_device().pow(a, e.words.data(), (uint32_t)LIMBS, out); }
    std::vector<uint8_t> inv_batch(const DeviceVec& x, DeviceVec& out) const { return on_device().inv(x, out); }
    std::vector<uint8_t> div_batch(const DeviceVec& a, const DeviceVec& b, DeviceVec& out) const { return on_device().div(a, b, out); }
    Element sum_batch(const DeviceVec& v) const { Element r; fpgpu_check(fpgpu_vec_sum(ctx_.get(), v.raw(), p(&r))); return r; }
    Element product_batch(const DeviceVec& v) const { Element r; fpgpu_check(fpgpu_vec_product(ctx_.get(), v.raw(), p(&r))); return r; }
    // sum_refs of mul over two slices (one row of the mat-vec)
    Element dot_batch(const DeviceVec& a, const DeviceVec& b) const { Element r; fpgpu_check(fpgpu_vec_dot(ctx_.get(), a.raw(), b.raw(), p(&r))); return r; }
    Element dot_batch(const std::vector<Element>& a, const std::vector<Element>& b) const {
        if (a.size() != b.size()) throw std::invalid_argument("length mismatch");
        Element r;
        fpgpu_check(fpgpu_dot(ctx_.get(), p(a.data()), p(b.data()), a.size(), p(&r)));
        return r;
    }
    // y = A x with A in row blocks over the device list, x whole on every device; a replicated y is all-gathered
    void matvec(const DeviceVec& A, size_t rows, size_t cols, const DeviceVec& x, DeviceVec& y) const { on_device().matvec(A, rows, cols, x, y); }
    void matmul(const DeviceVec& A, const DeviceVec& B, DeviceVec& C, size_t m, size_t k, size_t n) const { on_device().matmul(A, B, C, m, k, n); }
    // SparseMatrix on the device: its values in row blocks (a DeviceVec of rows x density) and its column indices
    std::pair<DeviceVec, DeviceIndices> to_device(const SparseMatrix<Element>& A, Placement where) const {
        std::vector<Element> vals(A.cells.size());
        std::vector<uint32_t> cols(A.cells.size());
        for (size_t i = 0; i < A.cells.size(); i++) { cols[i] = (uint32_t)A.cells[i].first; vals[i] = A.cells[i].second; }
        DeviceIndices idx(A.num_rows, A.density, where);
        idx.upload(cols);
        return {to_device(vals, where, A.density), idx};
    }
    void spmv(const std::pair<DeviceVec, DeviceIndices>& A, size_t rows, size_t cols, size_t density, const DeviceVec& x, DeviceVec& y) const {
        on_device().spmv(A.first, A.second.raw(), rows, cols, density, x, y);
    }

   private:
    static uint64_t* w(Element& e) { return e.v.words.data(); }
    static const uint64_t* w(const Element& e) { return e.v.words.data(); }
    static const uint64_t* p(const Element* e) { return reinterpret_cast<const uint64_t*>(e); }
    static uint64_t* p(Element* e) { return reinterpret_cast<uint64_t*>(e); }
    std::shared_ptr<fpgpu_ctx> ctx_;
    detail::HostMonty h_;
};

using F64 = MontyField<1>;
using F128 = MontyField<2>;
using F256 = MontyField<4>;
using F384 = MontyField<6>;
using F512 = MontyField<8>;

// ---- BoxedMontyField: limb count chosen at run time; every element owns its limbs ---------
struct BoxedMontyFieldElement {
    std::vector<uint64_t> limbs;  // Montgomery representative, bits_precision / 64 words
    bool operator==(const BoxedMontyFieldElement& o) const { return limbs == o.limbs; }
    uint32_t bits_precision() const { return (uint32_t)limbs.size() * 64; }
};

class BoxedMontyField {
   public:
    using Element = BoxedMontyFieldElement;
    static BoxedMontyField new_(const std::vector<uint64_t>& modulus_words) {
        fpgpu_ctx* raw = nullptr;
        int rc = fpgpu_ctx_create(modulus_words.data(), (uint32_t)modulus_words.size(), &raw);
        if (rc == FPGPU_ERR_INVALID_MODULUS) throw FieldErrorException(FieldError::InvalidModulus);
        fpgpu_check(rc);
        BoxedMontyField f;
        f.ctx_.reset(raw, detail::CtxDeleter());
        size_t n = modulus_words.size();
        f.h_.n = n; f.h_.m.assign(n, 0); f.h_.one.assign(n, 0); f.h_.r2.assign(n, 0);
        fpgpu_check(fpgpu_ctx_params(raw, f.h_.m.data(), f.h_.one.data(), f.h_.r2.data(), &f.h_.ninv));
        return f;
    }
    uint32_t bits_precision() const { return (uint32_t)h_.n * 64; }
    Element zero() const { return Element{std::vector<uint64_t>(h_.n, 0)}; }
    Element one() const { return Element{h_.one}; }
    Element add(const Element& x, const Element& y) const { Element r = zero(); h_.addm(r.limbs.data(), x.limbs.data(), y.limbs.data()); return r; }
    Element sub(const Element& x, const Element& y) const { Element r = zero(); h_.subm(r.limbs.data(), x.limbs.data(), y.limbs.data()); return r; }
    Element mul(const Element& x, const Element& y) const { Element r = zero(); h_.mul(r.limbs.data(), x.limbs.data(), y.limbs.data()); return r; }
    Element neg(const Element& x) const { Element r = zero(); h_.negm(r.limbs.data(), x.limbs.data()); return r; }
    std::optional<Element> inv(const Element& x) const {
        Element r = zero();
        if (!h_.inv(r.limbs.data(), x.limbs.data())) return std::nullopt;
        return r;
    }
    Element div(const Element& x, const Element& y) const {
        auto i = inv(y);
        if (!i) throw std::domain_error("Division by zero");
        return mul(x, *i);
    }
    Element project(uint64_t v) const {
        std::vector<uint64_t> t(h_.n, 0); t[0] = v;
        Element r = zero(); h_.to_mont(r.limbs.data(), t.data()); return r;
    }
    std::vector<uint64_t> lift(const Element& x) const { std::vector<uint64_t> r(h_.n); h_.from_mont(r.data(), x.limbs.data()); return r; }

    // Batch extension: gather every element's heap limbs into one contiguous array, run on the
    // GPU, scatter back.
    std::vector<Element> mul_batch(const std::vector<Element>& a, const std::vector<Element>& b) const {
        return binary(a, b, fpgpu_mul);
    }
    std::vector<Element> add_batch(const std::vector<Element>& a, const std::vector<Element>& b) const {
        return binary(a, b, fpgpu_add);
    }
    std::vector<Element> sub_batch(const std::vector<Element>& a, const std::vector<Element>& b) const {
        return binary(a, b, fpgpu_sub);
    }
    std::vector<std::optional<Element>> inv_batch(const std::vector<Element>& x) const {
        std::vector<uint64_t> gx = gather(x), go(gx.size());
        std::vector<uint8_t> none(x.size(), 0);
        if (!x.empty()) fpgpu_check(fpgpu_batch_inv(ctx_.get(), gx.data(), go.data(), none.data(), x.size()));
        std::vector<std::optional<Element>> r(x.size());
        for (size_t i = 0; i < x.size(); i++)
            if (!none[i]) r[i] = Element{std::vector<uint64_t>(go.begin() + i * h_.n, go.begin() + (i + 1) * h_.n)};
        return r;
    }

    // Device-resident form: the heap limbs are gathered ONCE into an fpgpu_vec; chains of batch
    // operations then run with no per-element allocation and no host copy
    // (the reference allocates per mul, crypto_bigint_boxed_monty.rs:182-196).
    DeviceVec to_device(const std::vector<Element>& v, Placement where = Placement::Current) const {
        DeviceVec d(ctx_.get(), v.size(), where);
        std::vector<uint64_t> g = gather(v);
        if (!v.empty()) d.upload(g.data(), 0, v.size());
        return d;
    }
    DeviceVec alloc_device(size_t n, Placement where = Placement::Current) const { return DeviceVec(ctx_.get(), n, where); }
    std::vector<Element> from_device(const DeviceVec& d) const {
        std::vector<uint64_t> g(d.size() * h_.n);
        if (d.size()) d.download(g.data(), 0, d.size());
        std::vector<Element> out(d.size());
        for (size_t i = 0; i < out.size(); i++) out[i].limbs.assign(g.begin() + i * h_.n, g.begin() + (i + 1) * h_.n);
        return out;
    }
    detail::DeviceOps on_device() const { return detail::DeviceOps{ctx_.get()}; }

   private:
    std::vector<uint64_t> gather(const std::vector<Element>& v) const {
        std::vector<uint64_t> g(v.size() * h_.n);
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i].limbs.size() != h_.n) throw std::invalid_argument("element precision mismatch");
            std::memcpy(g.data() + i * h_.n, v[i].limbs.data(), h_.n * 8);
        }
        return g;
    }
    template <typename Fn>
    std::vector<Element> binary(const std::vector<Element>& a, const std::vector<Element>& b, Fn fn) const {
        if (a.size() != b.size()) throw std::invalid_argument("length mismatch");
        std::vector<uint64_t> ga = gather(a), gb = gather(b), go(ga.size());
        if (!a.empty()) fpgpu_check(fn(ctx_.get(), ga.data(), gb.data(), go.data(), a.size()));
        std::vector<Element> out(a.size());
        for (size_t i = 0; i < a.size(); i++) out[i].limbs.assign(go.begin() + i * h_.n, go.begin() + (i + 1) * h_.n);
        return out;
    }
    std::shared_ptr<fpgpu_ctx> ctx_;
    detail::HostMonty h_;
};

// ---- ConstMontyField<Mod, LIMBS>: compile-time modulus, element style with operators ------
// `Mod` supplies `static const char* modulus_be_hex()` (const_monty_params! in the reference).
template <typename Mod, size_t LIMBS>
class ConstMontyField {
   public:
    using Config = MontyField<LIMBS>;
    static const Config& config() {
        static const Config cfg = Config::new_(Uint<LIMBS>::from_be_hex(Mod::modulus_be_hex()));
        return cfg;
    }
    ConstMontyField() = default;
    explicit ConstMontyField(const MontyFieldElement<LIMBS>& e) : e_(e) {}
    static ConstMontyField from(uint64_t v) { return ConstMontyField(config().project(v)); }
    static ConstMontyField from(int64_t v) { return ConstMontyField(config().project(v)); }
    static ConstMontyField from(int v) { return ConstMontyField(config().project((int64_t)v)); }
    static ConstMontyField from(const Uint<LIMBS>& v) { return ConstMontyField(config().project(v)); }
    static ConstMontyField zero() { return ConstMontyField(config().zero()); }
    static ConstMontyField one() { return ConstMontyField(config().one()); }
    Uint<LIMBS> retrieve() const { return config().lift(e_); }
    const Uint<LIMBS>& as_montgomery() const { return e_.v; }
    bool is_zero() const { return e_.v.is_zero(); }
    ConstMontyField operator+(const ConstMontyField& o) const { return ConstMontyField(config().add(e_, o.e_)); }
    ConstMontyField operator-(const ConstMontyField& o) const { return ConstMontyField(config().sub(e_, o.e_)); }
    ConstMontyField operator*(const ConstMontyField& o) const { return ConstMontyField(config().mul(e_, o.e_)); }
    ConstMontyField operator/(const ConstMontyField& o) const { return ConstMontyField(config().div(e_, o.e_)); }
    ConstMontyField operator-() const { return ConstMontyField(config().neg(e_)); }
    ConstMontyField& operator+=(const ConstMontyField& o) { return *this = *this + o; }
    ConstMontyField& operator-=(const ConstMontyField& o) { return *this = *this - o; }
    ConstMontyField& operator*=(const ConstMontyField& o) { return *this = *this * o; }
    ConstMontyField square() const { return *this * *this; }
    ConstMontyField dbl() const { return *this + *this; }   // `double` (crypto_bigint_const_monty.rs:86-89)
    // the x with x + x = self (crypto_bigint_const_monty.rs:81-84): halve the Montgomery representative mod p
    ConstMontyField div_by_2() const {
        const Uint<LIMBS> m = config().modulus();
        Uint<LIMBS> t = e_.v;
        uint64_t carry = 0;
        if (t.words[0] & 1) {
            for (size_t i = 0; i < LIMBS; i++) {
                unsigned __int128 s = (unsigned __int128)t.words[i] + m.words[i] + carry;
                t.words[i] = (uint64_t)s;
                carry = (uint64_t)(s >> 64);
            }
        }
        for (size_t i = 0; i < LIMBS; i++) t.words[i] = (t.words[i] >> 1) | ((i + 1 < LIMBS ? t.words[i + 1] : carry) << 63);
        MontyFieldElement<LIMBS> r;
        r.v = t;
        return ConstMontyField(r);
    }
    std::optional<ConstMontyField> inv() const {
        auto i = config().inv(e_);
        if (!i) return std::nullopt;
        return ConstMontyField(*i);
    }
    bool operator==(const ConstMontyField& o) const { return e_ == o.e_; }
    bool operator!=(const ConstMontyField& o) const { return !(e_ == o.e_); }
    // batch extension on slices of elements (repr-transparent over the Montgomery words)
    static void mul_batch(const ConstMontyField* a, const ConstMontyField* b, ConstMontyField* out, size_t n) {
        config().mul_batch(&a->e_, &b->e_, &out->e_, n);
    }
    static void add_batch(const ConstMontyField* a, const ConstMontyField* b, ConstMontyField* out, size_t n) {
        config().add_batch(&a->e_, &b->e_, &out->e_, n);
    }
    static void sub_batch(const ConstMontyField* a, const ConstMontyField* b, ConstMontyField* out, size_t n) {
        config().sub_batch(&a->e_, &b->e_, &out->e_, n);
    }
    static void square_batch(const ConstMontyField* a, ConstMontyField* out, size_t n) {
        config().square_batch(&a->e_, &out->e_, n);
    }
    static void double_batch(const ConstMontyField* a, ConstMontyField* out, size_t n) { config().double_batch(&a->e_, &out->e_, n); }
    static void div_by_2_batch(const ConstMontyField* a, ConstMontyField* out, size_t n) { config().div_by_2_batch(&a->e_, &out->e_, n); }

   private:
    MontyFieldElement<LIMBS> e_;
};

}  // namespace crypto_primitives
