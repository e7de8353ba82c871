"""CPU: pins the oracle to every known-answer test the reference's own unit tests hold
for this path.  Each test names the reference test it restates
(file:line under the reference root).  The reference projects small integers into the
field with `project` / `From`, so the helpers below do the same through the oracle."""
import numpy as np
import pytest

from conftest import array_to_ints, ints_to_array
from oracle.fp_oracle import Oracle

SECP = 2**256 - 2**32 - 977            # crypto_bigint_monty.rs:488-495, const_monty.rs:790-796, boxed_monty.rs:504-513
P64 = 10064419296686275259             # 0x8bac0006d9927abb: crypto_bigint_monty.rs:619, const_monty.rs:845, boxed_monty.rs:637
PROP = 0x00DCA94D8A1ECCE3B6E8755D8999787D0524D8CA1EA755E7AF84FB646FA31F27  # const_monty.rs:1471-1476


class F:
    """Scalar view over the oracle's batch calls, shaped like the reference's config API."""

    def __init__(self, p, n64):
        self.o, self.p, self.n64 = Oracle(p, n64), p, n64

    def _a(self, v): return ints_to_array([v], self.n64)
    def _i(self, a): return array_to_ints(a)[0]
    def project(self, v):
        if v < 0:
            return self.neg(self.project(-v))          # impl_from_signed (crypto_bigint_monty.rs:276-288)
        return self._i(self.o.to_mont(self._a(v)))
    def lift(self, x): return self._i(self.o.from_mont(self._a(x)))
    def add(self, x, y): return self._i(self.o.add(self._a(x), self._a(y)))
    def sub(self, x, y): return self._i(self.o.sub(self._a(x), self._a(y)))
    def mul(self, x, y): return self._i(self.o.mul(self._a(x), self._a(y)))
    def mul_boxed(self, x, y): return self._i(self.o.mul_boxed(self._a(x), self._a(y)))
    def neg(self, x): return self._i(self.o.neg(self._a(x)))
    def inv(self, x):
        out, fl = self.o.batch_inv(self._a(x))
        return None if fl[0] else self._i(out)
    def div(self, x, y):
        i = self.inv(y)
        if i is None:
            raise ZeroDivisionError("Division by zero")  # src/field.rs:155-157
        return self.mul(x, i)
    def zero(self): return 0
    def one(self): return self._i(self.o.one.reshape(1, -1))


@pytest.fixture(params=[4])
def f(request):
    return F(SECP, request.param)


def test_new_with_cfg_correct(f):
    # crypto_bigint_monty.rs:497-520, boxed_monty.rs:516-540: modulus+1 -> one, modulus -> zero, lift/project round trip
    assert f.project(SECP + 1) == f.one()
    assert f.project(SECP) == f.zero()
    for x in (f.zero(), f.one(), f.project(2), f.project(123456789)):
        assert f.project(f.lift(x)) == x


def test_basic_operations(f):
    # crypto_bigint_monty.rs:532-561
    a = f.project(9)
    assert f.add(a, f.neg(a)) == f.zero()
    a, b = f.project(10), f.project(5)
    assert f.add(a, b) == f.project(15)
    assert f.sub(a, b) == f.project(5)
    assert f.mul(a, b) == f.project(50)
    assert f.mul_boxed(a, b) == f.project(50)
    num, den = f.project(11), f.project(5)
    assert f.mul(f.div(num, den), den) == num


def test_basic_operations_overflow(f):
    # crypto_bigint_monty.rs:563-592, const_monty.rs:955-981, boxed_monty.rs:584-612
    m1 = f.project(SECP - 1)
    assert f.neg(m1) == f.one()
    assert f.add(m1, f.one()) == f.zero()
    assert f.sub(f.zero(), f.one()) == m1
    assert f.mul(m1, f.project(2)) == f.sub(m1, f.one())
    assert f.mul(m1, m1) == f.one()
    assert f.mul_boxed(m1, m1) == f.one()
    assert f.div(f.one(), m1) == m1


def test_add_wrapping(f):
    # crypto_bigint_monty.rs:594-603
    assert f.add(f.project(-100), f.project(105)) == f.project(5)


def test_from_unsigned_and_signed():
    # crypto_bigint_monty.rs:613-653, const_monty.rs:843-874, boxed_monty.rs:634-674 (hand-calculated residues)
    f = F(P64, 1)
    assert f.project(0) == f.zero()
    assert f.project(1) == f.one()
    assert f.add(f.project(-1), f.one()) == f.zero()
    assert f.add(f.project(-5), f.project(5)) == f.zero()
    assert f.project(2**64 - 1) == f.project(0x7453FFF9266D8544)
    assert f.lift(f.project(2**64 - 1)) == 0x7453FFF9266D8544
    assert f.project(2**63 - 1) == f.project(0x7FFFFFFFFFFFFFFF)
    assert f.project(-(2**63)) == f.project(0x0BAC0006D9927ABB)
    assert f.add(f.project(-(2**63)), f.project(2**63)) == f.zero()


def test_from_uint_wide_input_is_reduced():
    # crypto_bigint_monty.rs:680-686: 2^64 mod m = 0x7453fff9266d8545 (the reference reduces the
    # wider Uint with `rem` first; value mod m is then projected)
    f = F(P64, 1)
    assert (2**64) % P64 == 0x7453FFF9266D8545
    assert f.project((2**64) % P64) == f.project(0x7453FFF9266D8545)
    assert f.add(f.project(-123), f.project(123)) == f.zero()


def test_conversions_2_pow_256_mod_p():
    # boxed_monty.rs:887-911: 2^256 mod p = 2^32 + 977
    f = F(SECP, 4)
    assert (2**256) % SECP == 2**32 + 977
    assert f.lift(f.project((2**256) % SECP)) == 2**32 + 977
    assert f.lift(f.one()) == 1


def test_assign_and_aggregate(f):
    # crypto_bigint_monty.rs:689-713 (assign ops == ops), :799-822 (sum / product folds, empty -> identity)
    assert f.add(f.project(5), f.project(6)) == f.project(11)
    assert f.sub(f.project(20), f.project(7)) == f.project(13)
    assert f.mul(f.project(11), f.project(3)) == f.project(33)
    vals = ints_to_array([f.project(1), f.project(2), f.project(3)], 4)
    assert array_to_ints(f.o.sum(vals).reshape(1, -1))[0] == f.project(6)
    vals = ints_to_array([f.project(2), f.project(3), f.project(4)], 4)
    assert array_to_ints(f.o.product(vals).reshape(1, -1))[0] == f.project(24)
    empty = np.zeros((0, 4), dtype=np.uint64)
    assert array_to_ints(f.o.sum(empty).reshape(1, -1))[0] == f.zero()
    assert array_to_ints(f.o.product(empty).reshape(1, -1))[0] == f.one()


def test_inv_and_division_by_zero(f):
    # crypto_bigint_monty.rs:770-797, :715-723; boxed_monty.rs:833-843, :702-710
    a = f.project(5)
    assert f.mul(a, f.inv(a)) == f.one()
    assert f.inv(f.zero()) is None
    with pytest.raises(ZeroDivisionError, match="Division by zero"):
        f.div(f.project(7), f.zero())


def test_negation_and_min_max():
    # const_monty.rs:1078-1084 (-0 = 0), :918-928
    f = F(SECP, 4)
    assert f.neg(f.zero()) == f.zero()
    assert f.lift(f.neg(f.one())) == SECP - 1


def test_modulus_minus_one_div_two():
    # const_monty.rs:1478-1484, crypto_bigint_monty.rs:1019-1029: constant only, (p-1)/2
    f = F(SECP, 4)
    h = (SECP - 1) // 2
    assert f.lift(f.add(f.project(h), f.project(h))) == SECP - 1


def test_even_modulus_rejected():
    # crypto_bigint_monty.rs:1050-1055
    with pytest.raises(ValueError, match="InvalidModulus"):
        Oracle(42, 4)


def test_algebraic_laws_u64_inputs():
    # the 14 proptest laws of const_monty.rs:1486-1584 on the same modulus (:1471-1476), inputs any::<u64>()
    import random
    rng = random.Random(7)
    f = F(PROP, 4)
    for _ in range(50):
        a, b, c = (f.project(rng.getrandbits(64)) for _ in range(3))
        assert f.add(a, b) == f.add(b, a)
        assert f.mul(a, b) == f.mul(b, a)
        assert f.add(f.add(a, b), c) == f.add(a, f.add(b, c))
        assert f.mul(f.mul(a, b), c) == f.mul(a, f.mul(b, c))
        assert f.mul(a, f.add(b, c)) == f.add(f.mul(a, b), f.mul(a, c))
        assert f.add(a, f.zero()) == a and f.mul(a, f.one()) == a
        assert f.add(a, f.neg(a)) == f.zero()
        assert f.sub(a, b) == f.add(a, f.neg(b))
        if a != 0:
            assert f.mul(a, f.inv(a)) == f.one()
        if b != 0:
            assert f.div(a, b) == f.mul(a, f.inv(b))


def test_mod_neg_inv_newton_matches_definition():
    # helpers/crypto_bigint.rs:56-67 against -m^-1 mod 2^64
    import ctypes
    from oracle.fp_oracle import lib
    for m0 in (1, 3, 0xFFFFFFFFFFFFFFFF, 0x43E1F593F0000001, 0xB9FEFFFFFFFFAAAB, 0xFFFFFFFF00000001):
        assert lib().fpo_mod_neg_inv(ctypes.c_uint64(m0)) == (-pow(m0, -1, 1 << 64)) % (1 << 64)


def _pow_both(f, x, e, n64):
    """pow_u32 walks 32 exponent bits, pow(&Uint) walks Uint::BITS (crypto_bigint_monty.rs:163-172,
    239-249); both must agree."""
    a = f.o.pow(f._a(x), e, exponent_bits=32)
    b = f.o.pow(f._a(x), e, exponent_bits=64 * n64)
    assert (a == b).all()
    return f._i(a)


def test_pow_operation(f):
    # crypto_bigint_monty.rs:726-760 (also boxed_monty.rs:713, const_monty.rs:1270)
    two, three = f.project(2), f.project(3)
    assert _pow_both(f, two, 0, 4) == f.one()
    assert _pow_both(f, two, 1, 4) == two
    assert _pow_both(f, two, 3, 4) == f.project(8)
    assert _pow_both(f, two, 10, 4) == f.project(1024)
    assert _pow_both(f, three, 4, 4) == f.project(81)


def test_pow_matches_definition(f):
    # crypto_bigint_monty.rs:934-971 compares with crypto-bigint's FixedMontyForm::pow (crate absent
    # here, pinned 0.7.0-rc in Cargo.toml); that value is base^exp mod p, so Python's pow pins it.
    for base, e in [(0, 0), (0, 7), (1, 1000), (2, 10), (3, 251), (123456789, 1), (123456789, 2**32 - 1)]:
        assert f.lift(_pow_both(f, f.project(base), e, 4)) == pow(base, e, SECP), (base, e)
    x = f.project(123456789)
    got = f._i(f.o.pow(f._a(x), 2**128 - 1))
    assert f.lift(got) == pow(123456789, 2**128 - 1, SECP)


def test_pow_amm_final_reduction():
    # crypto_bigint_monty.rs:978-1004: final AMM value in [p, 2p) needs exactly one subtraction
    g = F(P64, 1)
    for base, e, r in [(2, 8, 256), (2, 9, 512), (3, 5, 243)]:
        assert g._i(g.o.pow(g._a(g.project(base)), e, exponent_bits=32)) == g.project(r)
    for base, e in [(12345, 67), (2, 64), (123456789, 2**32 - 1)]:
        assert g.lift(g._i(g.o.pow(g._a(g.project(base)), e, exponent_bits=32))) == pow(base, e, P64)


def test_pow_random_against_definition():
    import random
    rnd = random.Random(5)
    for p, n64 in [(SECP, 4), (P64, 1), (PROP, 4)]:
        g = F(p, n64)
        xs = [rnd.randrange(p) for _ in range(40)]
        for e in [0, 1, 2, 15, 16, 17, 255, 2**31, p - 2, p - 1, rnd.getrandbits(64 * n64)]:
            out = array_to_ints(g.o.pow(g.o.to_mont(ints_to_array(xs, n64)), e))
            assert [g.lift(v) for v in out] == [pow(x, e, p) for x in xs]


def test_double_and_div_by_2_match_python_ints_and_reference_check():
    """ConstMontyField::double / div_by_2 (crypto_bigint_const_monty.rs:81-89) in the oracle: on the Montgomery
    representatives m they are 2m mod p and m/2 mod p; the reference's own check is `half + half == even` for 20
    and `double(a) == a + a` for 10 (:1177-1187)."""
    import random
    from oracle.fp_oracle import Oracle
    for p, n64 in ((0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 4), (2**64 - 2**32 + 1, 1),
                   (2**256 - 2**32 - 977, 4), (2**512 - 569, 8),
                   (0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6)):
        o = Oracle(p, n64)
        rnd = random.Random(n64)
        xs = [0, 1, 2, 3, p - 1, p - 2, p >> 1, (p >> 1) + 1] + [rnd.randrange(p) for _ in range(200)]
        a = ints_to_array(xs, n64)
        inv2 = pow(2, -1, p)
        assert array_to_ints(o.double(a)) == [2 * v % p for v in xs]
        half = o.div_by_2(a)
        assert array_to_ints(half) == [v * inv2 % p for v in xs]
        assert np.array_equal(o.add(half, half), a)
        ten, twenty = o.to_mont(ints_to_array([10], n64)), o.to_mont(ints_to_array([20], n64))
        assert np.array_equal(o.double(ten), o.add(ten, ten)) and np.array_equal(o.double(ten), twenty)
        assert np.array_equal(o.div_by_2(twenty), ten)
