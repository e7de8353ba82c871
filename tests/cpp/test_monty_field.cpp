// Tests of the C++ host mirror (crypto_primitives_amd/host/monty_field.hpp), shaped like the
// reference's own unit tests (file:line cited per test).  `cpu` mode covers the scalar host
// API and error behaviour (no GPU needed); `gpu` mode checks every BATCH method, which runs
// on the MI355X through the C ABI, against the scalar methods element by element.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../crypto_primitives_amd/host/monty_field.hpp"

using namespace crypto_primitives;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond)                                                            \
    do {                                                                       \
        g_checks++;                                                            \
        if (!(cond)) { g_fail++; std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static const char* SECP = "fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f";  // crypto_bigint_monty.rs:488-495
using F = F256;
static F make_f() { return F::new_(Uint<4>::from_be_hex(SECP)); }

static void test_new_with_cfg_correct() {  // crypto_bigint_monty.rs:497-520
    F f = make_f();
    CHECK(f.project(Uint<4>::from_be_hex("fffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc30")) == f.one());
    CHECK(f.project(f.modulus()) == f.zero());
    for (auto x : {f.zero(), f.one(), f.project((uint64_t)2), f.project((uint64_t)123456789)}) CHECK(f.project(f.lift(x)) == x);
}
static void test_basic_operations() {  // crypto_bigint_monty.rs:532-561
    F f = make_f();
    auto a = f.project(9);
    CHECK(f.add(a, f.neg(a)) == f.zero());
    a = f.project(10);
    auto b = f.project(5);
    CHECK(f.add(a, b) == f.project(15));
    CHECK(f.sub(a, b) == f.project(5));
    CHECK(f.mul(a, b) == f.project(50));
    auto num = f.project(11), den = f.project(5);
    CHECK(f.mul(f.div(num, den), den) == num);
}
static void test_basic_operations_overflow() {  // crypto_bigint_monty.rs:563-592
    F f = make_f();
    Uint<4> m1 = f.modulus();
    m1.words[0] -= 1;
    auto mm1 = f.project(m1);
    CHECK(f.neg(mm1) == f.one());
    CHECK(f.add(mm1, f.one()) == f.zero());
    CHECK(f.sub(f.zero(), f.one()) == mm1);
    CHECK(f.mul(mm1, f.project(2)) == f.sub(mm1, f.one()));
    CHECK(f.mul(mm1, mm1) == f.one());
    CHECK(f.div(f.one(), mm1) == mm1);
}
static void test_from_unsigned_and_signed() {  // crypto_bigint_monty.rs:613-653 (hand-calculated residues)
    F64 f = F64::new_(Uint<1>::from_u64(10064419296686275259ull));
    CHECK(f.project(0) == f.zero());
    CHECK(f.project(1) == f.one());
    CHECK(f.add(f.project(-1), f.one()) == f.zero());
    CHECK(f.add(f.project(-5), f.project(5)) == f.zero());
    CHECK(f.project((uint64_t)UINT64_MAX) == f.project(Uint<1>::from_be_hex("7453fff9266d8544")));
    CHECK(f.project((int64_t)INT64_MAX) == f.project(Uint<1>::from_be_hex("7fffffffffffffff")));
    CHECK(f.project((int64_t)INT64_MIN) == f.project(Uint<1>::from_be_hex("0bac0006d9927abb")));
    CHECK(f.add(f.project((int64_t)INT64_MIN), f.project((uint64_t)1 << 63)) == f.zero());
    // a Uint wider than the modulus is reduced first: 2^64 mod m (:680-686)
    CHECK(f.project_wide(Uint<2>::from_be_hex("00000000000000010000000000000000")) == f.project(Uint<1>::from_be_hex("7453fff9266d8545")));
}
static void test_assign_and_aggregate() {  // crypto_bigint_monty.rs:689-713, 799-822
    F f = make_f();
    auto a = f.project(5);
    f.add_assign(a, f.project(6));
    CHECK(a == f.project(11));
    a = f.project(20);
    f.sub_assign(a, f.project(7));
    CHECK(a == f.project(13));
    a = f.project(11);
    f.mul_assign(a, f.project(3));
    CHECK(a == f.project(33));
    std::vector<F::Element> v = {f.project(1), f.project(2), f.project(3)};
    CHECK(f.sum_refs(v.begin(), v.end()) == f.project(6));
    v = {f.project(2), f.project(3), f.project(4)};
    CHECK(f.product_refs(v.begin(), v.end()) == f.project(24));
    v.clear();
    CHECK(f.sum_refs(v.begin(), v.end()) == f.zero());
    CHECK(f.product_refs(v.begin(), v.end()) == f.one());
}
static void test_div_by_zero_and_inv() {  // crypto_bigint_monty.rs:715-723, 770-797
    F f = make_f();
    bool threw = false;
    try { (void)f.div(f.project(7), f.zero()); } catch (const std::domain_error& e) { threw = std::string(e.what()) == "Division by zero"; }
    CHECK(threw);
    auto a = f.project(5);
    CHECK(f.mul(a, *f.inv(a)) == f.one());
    CHECK(!f.inv(f.zero()).has_value());
    CHECK(!f.checked_div(f.project(10), f.zero()).has_value());
    CHECK(f.mul(*f.checked_div(f.project(10), f.project(5)), f.project(5)) == f.project(10));
}
static void test_pow_and_prime_field_methods() {  // crypto_bigint_monty.rs:725-768, 1006-1036
    F f = make_f();
    auto two = f.project(2);
    CHECK(f.pow_u32(two, 0) == f.one());
    CHECK(f.pow_u32(two, 10) == f.project(1024));
    CHECK(f.pow_u32(f.project(3), 4) == f.project(81));
    CHECK(f.pow_u32(f.zero(), 10) == f.zero());
    CHECK(f.modulus() == Uint<4>::from_be_hex(SECP));
    Uint<4> h = f.modulus_minus_one_div_two();
    CHECK(f.lift(f.add(f.project(h), f.project(h))) == [&] { Uint<4> m = f.modulus(); m.words[0] -= 1; return m; }());
}
static void test_new_rejects_even_modulus() {  // crypto_bigint_monty.rs:1050-1055
    bool threw = false;
    try { (void)F::new_(Uint<4>::from_u64(42)); } catch (const FieldErrorException& e) { threw = e.kind == FieldError::InvalidModulus; }
    CHECK(threw);
    CHECK(!F::try_new(Uint<4>::from_u64(42)).has_value());
}
struct Bn254 { static const char* modulus_be_hex() { return "30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001"; } };
static void test_const_monty_element_style() {  // crypto_bigint_const_monty.rs:955-981, 1168-1188
    using E = ConstMontyField<Bn254, 4>;
    E a = E::from(10), b = E::from(5);
    CHECK(a + b == E::from(15));
    CHECK(a - b == E::from(5));
    CHECK(a * b == E::from(50));
    CHECK((a / b) * b == a);
    CHECK(-a + a == E::zero());
    CHECK(a.square() == E::from(100));
    CHECK(a.dbl() == E::from(20));
    {   // field_methods (crypto_bigint_const_monty.rs:1184-1187) and an odd representative
        E even = E::from(20), half = even.div_by_2();
        CHECK(half + half == even);
        for (int v = 1; v < 40; v++) { E h = E::from(v).div_by_2(); CHECK(h + h == E::from(v)); }
    }
    CHECK(!E::zero().inv().has_value());
    CHECK(E::from(-1) + E::one() == E::zero());
    CHECK(E::from(7).retrieve() == Uint<4>::from_u64(7));
}
static void test_boxed_scalar() {  // crypto_bigint_boxed_monty.rs:584-612, 833-843
    std::vector<uint64_t> m = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                               0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    BoxedMontyField f = BoxedMontyField::new_(m);
    CHECK(f.bits_precision() == 384);
    auto a = f.project(10), b = f.project(5);
    CHECK(f.add(a, b) == f.project(15));
    CHECK(f.mul(a, b) == f.project(50));
    CHECK(f.mul(f.div(a, b), b) == a);
    CHECK(!f.inv(f.zero()).has_value());
    CHECK(f.lift(f.project(77))[0] == 77);
}

// ---- GPU: batch methods == scalar methods, element by element ---------------------------
template <size_t L>
static std::vector<MontyFieldElement<L>> random_elems(const MontyField<L>& f, size_t n, uint64_t seed) {
    std::mt19937_64 rng(seed);
    std::vector<MontyFieldElement<L>> v(n);
    for (auto& e : v) {
        Uint<L> x;
        for (auto& w : x.words) w = rng();
        e = f.project(f.lift(MontyFieldElement<L>{x}));  // any words -> canonical residue
    }
    return v;
}
static void test_gpu_batches() {
    F f = F::new_(Uint<4>::from_be_hex(Bn254::modulus_be_hex()));
    const size_t n = 3001;
    auto a = random_elems(f, n, 1), b = random_elems(f, n, 2);
    b[7] = f.zero();
    b[n - 1] = f.zero();
    std::vector<F::Element> out(n);
    f.mul_batch(a.data(), b.data(), out.data(), n);
    for (size_t i = 0; i < n; i++) CHECK(out[i] == f.mul(a[i], b[i]));
    f.add_batch(a.data(), b.data(), out.data(), n);
    for (size_t i = 0; i < n; i++) CHECK(out[i] == f.add(a[i], b[i]));
    f.sub_batch(a.data(), b.data(), out.data(), n);
    for (size_t i = 0; i < n; i++) CHECK(out[i] == f.sub(a[i], b[i]));
    f.neg_batch(a.data(), out.data(), n);
    for (size_t i = 0; i < n; i++) CHECK(out[i] == f.neg(a[i]));
    f.square_batch(a.data(), out.data(), n);
    for (size_t i = 0; i < n; i++) CHECK(out[i] == f.mul(a[i], a[i]));
    auto inv = f.inv_batch(b);
    for (size_t i = 0; i < n; i++) {
        auto e = f.inv(b[i]);
        CHECK(inv[i].has_value() == e.has_value());
        if (e && inv[i]) CHECK(*inv[i] == *e);
    }
    bool threw = false;
    try { (void)f.div_batch(a, b); } catch (const std::domain_error& e) { threw = std::string(e.what()) == "Division by zero"; }
    CHECK(threw);
    std::vector<Uint<4>> lifted(n);
    f.lift_batch(a.data(), lifted.data(), n);
    for (size_t i = 0; i < n; i++) CHECK(lifted[i] == f.lift(a[i]));
    f.project_batch(lifted.data(), out.data(), n);
    for (size_t i = 0; i < n; i++) CHECK(out[i] == a[i]);
    // pow (one exponent per batch) and the sum / product folds
    f.pow_u32_batch(a.data(), 65537u, out.data(), n);
    for (size_t i = 0; i < n; i += 97) CHECK(out[i] == f.pow_u32(a[i], 65537u));
    Uint<4> big = f.lift(a[3]);
    f.pow_batch(a.data(), big, out.data(), 64);
    for (size_t i = 0; i < 64; i += 9) {
        auto acc = f.one();
        for (int bit = 255; bit >= 0; bit--) {
            acc = f.mul(acc, acc);
            if ((big.words[bit / 64] >> (bit % 64)) & 1) acc = f.mul(acc, a[i]);
        }
        CHECK(out[i] == acc);
    }
    CHECK(f.sum_batch(a.data(), n) == f.sum_refs(a.begin(), a.end()));
    CHECK(f.product_batch(a.data(), n) == f.product_refs(a.begin(), a.end()));
    CHECK(f.sum_batch(a.data(), 0) == f.zero());
    CHECK(f.product_batch(a.data(), 0) == f.one());
    // mat-vec / mat-mul over DenseRowMatrix == folds of the scalar ops
    DenseRowMatrix<F::Element> A;
    A.num_rows = 9; A.num_cols = 300; A.data = random_elems(f, 9 * 300, 3);
    auto x = random_elems(f, 300, 4);
    auto y = f.matvec(A, x);
    for (size_t i = 0; i < A.num_rows; i++) {
        auto acc = f.zero();
        for (size_t j = 0; j < A.num_cols; j++) acc = f.add(acc, f.mul(A.row(i)[j], x[j]));
        CHECK(y[i] == acc);
    }
    DenseRowMatrix<F::Element> B;
    B.num_rows = 300; B.num_cols = 5; B.data = random_elems(f, 300 * 5, 5);
    auto C = f.matmul(A, B);
    for (size_t i = 0; i < 9; i++)
        for (size_t j = 0; j < 5; j++) {
            auto acc = f.zero();
            for (size_t l = 0; l < 300; l++) acc = f.add(acc, f.mul(A.row(i)[l], B.row(l)[j]));
            CHECK(C.row(i)[j] == acc);
        }
    // sparse: 40 rows, 7 cells per row over 300 columns (repeated columns allowed)
    SparseMatrix<F::Element> S;
    S.num_rows = 40; S.num_cols = 300; S.density = 7;
    {
        auto sv = random_elems(f, 40 * 7, 6);
        for (size_t i = 0; i < sv.size(); i++) S.cells.push_back({(i * 37 + i / 7) % 300, sv[i]});
    }
    auto ys = f.spmv(S, x);
    for (size_t i = 0; i < S.num_rows; i++) {
        auto acc = f.zero();
        for (size_t j = 0; j < S.density; j++) acc = f.add(acc, f.mul(S.cells[i * 7 + j].second, x[S.cells[i * 7 + j].first]));
        CHECK(ys[i] == acc);
    }
    // element style
    using E = ConstMontyField<Bn254, 4>;
    std::vector<E> ea(100), eb(100), eo(100);
    for (int i = 0; i < 100; i++) { ea[i] = E::from(1000003 * (i + 1)); eb[i] = E::from(-(int64_t)(77 * i + 1)); }
    E::mul_batch(ea.data(), eb.data(), eo.data(), 100);
    for (int i = 0; i < 100; i++) CHECK(eo[i] == ea[i] * eb[i]);
    // double / div_by_2 (crypto_bigint_const_monty.rs:81-89, checked there as half + half == even, :1184-1187)
    E::double_batch(ea.data(), eo.data(), 100);
    for (int i = 0; i < 100; i++) CHECK(eo[i] == ea[i].dbl());
    E::div_by_2_batch(eb.data(), eo.data(), 100);
    for (int i = 0; i < 100; i++) CHECK(eo[i] == eb[i].div_by_2() && eo[i] + eo[i] == eb[i]);
    // boxed: gather / scatter
    std::vector<uint64_t> m = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                               0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    BoxedMontyField bf = BoxedMontyField::new_(m);
    std::vector<BoxedMontyFieldElement> ba, bb;
    for (uint64_t i = 0; i < 500; i++) { ba.push_back(bf.project(0x9e3779b97f4a7c15ull * (i + 1))); bb.push_back(bf.project(i)); }
    auto bo = bf.mul_batch(ba, bb);
    for (size_t i = 0; i < 500; i++) CHECK(bo[i] == bf.mul(ba[i], bb[i]));
    auto binv = bf.inv_batch(bb);
    CHECK(!binv[0].has_value());
    for (size_t i = 1; i < 500; i++) CHECK(binv[i].has_value() && bf.mul(*binv[i], bb[i]) == bf.one());
}
// ---- GPU: device-resident vectors -- chains with no host copy, sharded over a device list -------
static void test_gpu_device_vectors() {
    auto f = F256::new_(Uint<4>::from_be_hex("30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001"));
    const size_t n = 5003;
    std::vector<F256::Element> a(n), b(n);
    for (size_t i = 0; i < n; i++) { a[i] = f.project((uint64_t)(i * 2654435761u + 12345)); b[i] = f.project((uint64_t)((i + 7) * 0x9E3779B97F4A7C15ull)); }
    a[11] = f.zero();
    for (int layout = 0; layout < 3; layout++) {
        if (layout == 1) set_devices({0});
        if (layout == 2) set_devices({0, 0, 0});
        const Placement where = layout == 0 ? Placement::Current : Placement::Sharded;
        DeviceVec da = f.to_device(a, where), db = f.to_device(b, where);
        DeviceVec t = f.alloc_device(n, where), u = f.alloc_device(n, where), v = f.alloc_device(n, where);
        f.mul_batch(da, db, t);          // t = a b
        f.add_batch(t, da, u);           // u = a b + a
        f.square_batch(u, t);            // t = u^2
        auto none = f.inv_batch(t, v);   // v = 1 / t   (None where t = 0)
        {   // checked_div on handles == inversion then multiplication
            DeviceVec q = f.alloc_device(n, where), q2 = f.alloc_device(n, where);
            auto none_q = f.div_batch(da, t, q);   // q = a / t
            f.mul_batch(da, v, q2);               // q2 = a * (1 / t)
            CHECK(f.from_device(q) == f.from_device(q2) && none_q == none);
        }
        f.sub_batch(v, db, u);           // u = v - b
        auto got = f.from_device(u);
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) {
            auto ti = f.mul(f.add(f.mul(a[i], b[i]), a[i]), f.add(f.mul(a[i], b[i]), a[i]));
            auto inv = f.inv(ti);
            if (none[i] != (inv ? 0 : 1)) bad++;
            auto want = f.sub(inv ? *inv : f.zero(), b[i]);
            if (!(got[i] == want)) bad++;
        }
        CHECK(bad == 0);
        CHECK(none[11] == 1);
        CHECK(f.sum_batch(da) == f.sum_refs(a.begin(), a.end()));
        CHECK(f.product_batch(db) == f.product_refs(b.begin(), b.end()));
        F256::Element dot = f.zero();
        for (size_t i = 0; i < n; i++) dot = f.add(dot, f.mul(a[i], b[i]));
        CHECK(f.dot_batch(da, db) == dot);
        CHECK(f.dot_batch(a, b) == dot);
        // row-sharded mat-vec, x replicated, y all-gathered (RCCL for the list of one, peer copies for the repeated GPU)
        const size_t rows = 37, cols = 64;
        DenseRowMatrix<F256::Element> A;
        A.num_rows = rows; A.num_cols = cols; A.data.assign(a.begin(), a.begin() + rows * cols);
        std::vector<F256::Element> x(b.begin(), b.begin() + cols);
        DeviceVec dA = f.to_device(A, where), dx = f.to_device(x, layout == 0 ? Placement::Current : Placement::Replicated);
        DeviceVec dy = f.alloc_device(rows, layout == 0 ? Placement::Current : Placement::Replicated);
        f.matvec(dA, rows, cols, dx, dy);
        CHECK(f.from_device(dy) == f.matvec(A, x));
        // SparseMatrix on handles: values and column indices in row blocks, the same x, y gathered
        SparseMatrix<F256::Element> S;
        S.num_rows = 41; S.num_cols = cols; S.density = 5;
        for (size_t i = 0; i < S.num_rows * S.density; i++) S.cells.push_back({(i * 29 + i / S.density) % cols, a[100 + i]});
        auto dS = f.to_device(S, where);
        DeviceVec dys = f.alloc_device(S.num_rows, layout == 0 ? Placement::Current : Placement::Replicated);
        f.spmv(dS, S.num_rows, S.num_cols, S.density, dx, dys);
        CHECK(f.from_device(dys) == f.spmv(S, x));
    }
    set_devices({});
    {   // a device list as a value: vectors laid out over it do not depend on the process-wide list
        DeviceSet two({0, 0});
        const size_t n = 5000;
        std::vector<F::Element> a, b;
        for (uint64_t i = 0; i < n; i++) { a.push_back(f.project((uint64_t)(i * 0x9E3779B97F4A7C15ull + 1))); b.push_back(f.project((uint64_t)(i * i + 7))); }
        DeviceVec da = f.to_device(a, Placement::Sharded, 0, &two), db = f.to_device(b, Placement::Sharded, 0, &two);
        DeviceVec t = f.alloc_device(n, Placement::Sharded, &two);
        CHECK(da.parts() == 2);
        f.on_device().mul(da, db, t);
        f.on_device().add(t, da, t);
        auto got = f.from_device(t);
        size_t bad = 0;
        for (size_t i = 0; i < n; i++) bad += !(got[i] == f.add(f.mul(a[i], b[i]), a[i]));
        CHECK(bad == 0);
    }
    // BoxedMontyField: limbs gathered once, chain on the device, scattered once
    auto bf = BoxedMontyField::new_({0xB9FEFFFFFFFFAAABull, 0x1EABFFFEB153FFFFull, 0x6730D2A0F6B0F624ull, 0x64774B84F38512BFull,
                                     0x4B1BA7B6434BACD7ull, 0x1A0111EA397FE69Aull});
    std::vector<BoxedMontyField::Element> ba, bb;
    for (uint64_t i = 0; i < 300; i++) { ba.push_back(bf.project(i * 77 + 3)); bb.push_back(bf.project(i * i + 1)); }
    DeviceVec xa = bf.to_device(ba), xb = bf.to_device(bb), xt = bf.alloc_device(300), xu = bf.alloc_device(300);
    bf.on_device().mul(xa, xb, xt);
    bf.on_device().add(xt, xb, xu);
    auto none = bf.on_device().inv(xu, xt);
    auto res = bf.from_device(xt);
    size_t bad = 0;
    for (size_t i = 0; i < 300; i++) {
        auto inv = bf.inv(bf.add(bf.mul(ba[i], bb[i]), bb[i]));
        if (!inv || none[i] || !(res[i] == *inv)) bad++;
    }
    CHECK(bad == 0);
}

static void test_no_device_is_an_error_not_a_fallback() {
    if (fpgpu_device_count() > 0) return;
    F f = make_f();
    std::vector<F::Element> a(4, f.one()), out(4);
    bool threw = false;
    try { f.mul_batch(a.data(), a.data(), out.data(), 4); } catch (const FpgpuError& e) { threw = e.code == FPGPU_ERR_NO_DEVICE; }
    CHECK(threw);
}

int main(int argc, char** argv) {
    std::string mode = argc > 1 ? argv[1] : "cpu";
    test_new_with_cfg_correct();
    test_basic_operations();
    test_basic_operations_overflow();
    test_from_unsigned_and_signed();
    test_assign_and_aggregate();
    test_div_by_zero_and_inv();
    test_pow_and_prime_field_methods();
    test_new_rejects_even_modulus();
    test_const_monty_element_style();
    test_boxed_scalar();
    test_no_device_is_an_error_not_a_fallback();
    if (mode == "gpu") { test_gpu_batches(); test_gpu_device_vectors(); }
    std::printf("%s: %d checks, %d failures\n", mode.c_str(), g_checks, g_fail);
    return g_fail ? 1 : 0;
}
