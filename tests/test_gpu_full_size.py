"""GPU, BASELINE.json's full sizes: the oracle cannot cover 2^28 elements or a 32 GiB matrix in
seconds, so these check size-independent laws on the device-resident data (no host copy of the
bulk), plus the oracle on strided samples.  Everything is generated on the device by the SURVEY
section 8d generator (fpgpu_fill_random_dev)."""
import numpy as np
import pytest

from conftest import h2i, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as c
    assert c.load().fpgpu_device_count() >= 1, "no MI355X visible"
    return c


def _oracle(name):
    from oracle.fp_oracle import Oracle
    g = load_golden(name)
    return Oracle(h2i(g["modulus"]), g["n64"])


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def test_config1_elementwise_2pow28_bn254(cpa):
    """configs[1]: mul / sqr / add / sub over 2^28 HBM-resident BN254 Fr elements."""
    import torch
    f, o = cpa.MontyField(cpa.BN254_FR, 4), _oracle("bn254_fr")
    n = 1 << 28
    a, b = f.random_dev(n, 101), f.random_dev(n, 102)
    t, u = f.alloc(n), f.alloc(n)
    f.mul_dev(a, b, t)
    f.mul_dev(b, a, u)
    assert torch.equal(t, u)                                  # commutativity
    idx = torch.arange(0, n, 65521, device="cuda")
    assert np.array_equal(_host(t[idx]), o.mul(_host(a[idx]), _host(b[idx])))
    f.sqr_dev(a, t)
    f.mul_dev(a, a, u)
    assert torch.equal(t, u)                                  # sqr == mul(a, a)
    assert np.array_equal(_host(t[idx]), o.sqr(_host(a[idx])))
    f.add_dev(a, b, t)
    assert np.array_equal(_host(t[idx]), o.add(_host(a[idx]), _host(b[idx])))
    f.sub_dev(t, b, u)
    assert torch.equal(u, a)                                  # (a + b) - b == a
    f.sub_dev(a, b, t)
    f.sub_dev(b, a, u)
    f.add_dev(t, u, u)
    assert int(u.count_nonzero()) == 0                        # (a - b) + (b - a) == 0
    del a, b, t, u
    torch.cuda.empty_cache()


def test_config2_batch_inversion_2pow24_bls12_381(cpa):
    """configs[2]: BoxedMontyField-width (384-bit) batch inversion over 2^24 elements, zeros included."""
    import torch
    f, o = cpa.MontyField(cpa.BLS12_381_FQ, 6), _oracle("bls12_381_fq")
    n = 1 << 24
    x = f.random_dev(n, 103)
    zeros = torch.tensor([0, 1, 255, 256, 4097, n // 2, n - 1], device="cuda")
    x[zeros] = 0
    inv, back, chk = f.alloc(n), f.alloc(n), f.alloc(n)
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    ws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    f.inv_dev(x, inv, flags, ws)
    assert int(flags.sum()) == len(zeros) and bool((flags[zeros] == 1).all())   # None exactly at the zeros
    assert int(inv[zeros].count_nonzero()) == 0
    f.mul_dev(x, inv, chk)
    one = torch.from_numpy(f.one_words.view(np.int64)).cuda()
    nz = flags == 0
    assert torch.equal(chk[nz], one.expand(int(nz.sum()), 6))                   # x * inv(x) == one
    f.inv_dev(inv, back, flags, ws)
    assert torch.equal(back, x)                                                 # inv(inv(x)) == x (zeros map to zero)
    idx = torch.arange(3, n, 16381, device="cuda")
    want, _ = o.batch_inv(_host(x[idx]))
    assert np.array_equal(_host(inv[idx]), want)
    del x, inv, back, chk, ws
    torch.cuda.empty_cache()


def test_config3_matvec_2pow15_bn254(cpa):
    """configs[3]: 2^15 x 2^15 BN254 Fr mat-vec (32 GiB matrix): linearity, matrix-core path == VALU
    path, sampled rows against the oracle."""
    import torch
    f, o = cpa.MontyField(cpa.BN254_FR, 4), _oracle("bn254_fr")
    dim = 1 << 15
    A = f.random_dev(dim * dim, 104)
    x, z = f.random_dev(dim, 105), f.random_dev(dim, 106)
    ws = torch.empty(f.matvec_workspace_bytes(dim, dim), dtype=torch.uint8, device="cuda")
    yx, yz, ys, s, t = f.alloc(dim), f.alloc(dim), f.alloc(dim), f.alloc(dim), f.alloc(dim)
    f.matvec_dev(A, dim, dim, x, yx, workspace=ws)
    f.matvec_dev(A, dim, dim, z, yz, workspace=ws)
    f.add_dev(x, z, s)
    f.matvec_dev(A, dim, dim, s, ys, workspace=ws)
    f.add_dev(yx, yz, t)
    assert torch.equal(ys, t)                                  # A (x + z) == A x + A z
    f.matvec_dev(A, dim, dim, x, t)                            # no workspace: VALU kernel
    assert torch.equal(t, yx)
    hx = _host(x)
    for r in (0, 4099, dim - 1):
        row = _host(A[r * dim:(r + 1) * dim])
        assert np.array_equal(_host(yx[r:r + 1]), o.matvec(row, 1, dim, hx, threads=0))
    del A, ws
    torch.cuda.empty_cache()


def test_config4_matmul_8192_goldilocks(cpa):
    """configs[4]: 8192^3 single-limb mat-mul on the matrix cores: (A B) v == A (B v) for a random
    vector (an O(n^2) check of the O(n^3) product), sampled rows against the oracle."""
    import torch
    f, o = cpa.MontyField(cpa.GOLDILOCKS, 1), _oracle("goldilocks")
    dim = 8192
    A, B = f.random_dev(dim * dim, 107), f.random_dev(dim * dim, 108)
    C = f.alloc(dim * dim)
    ws = torch.empty(f.matmul_workspace_bytes(dim, dim, dim), dtype=torch.uint8, device="cuda")
    f.matmul_dev(A, B, C, dim, dim, dim, workspace=ws)
    v = f.random_dev(dim, 109)
    bv, abv, cv = f.alloc(dim), f.alloc(dim), f.alloc(dim)
    f.matvec_dev(B, dim, dim, v, bv)
    f.matvec_dev(A, dim, dim, bv, abv)
    f.matvec_dev(C, dim, dim, v, cv)
    # all operands are Montgomery representatives: (A B) v and A (B v) carry the same power of R^-1
    assert torch.equal(cv, abv)
    hB = _host(B)
    for r in (0, 4097, dim - 1):
        want = o.matmul(_host(A[r * dim:(r + 1) * dim]), hB, 1, dim, dim, threads=0)
        assert np.array_equal(_host(C[r * dim:(r + 1) * dim]).reshape(-1), want.reshape(-1))
    del A, B, C, ws
    torch.cuda.empty_cache()


def test_more_than_2pow32_elements_in_one_call(cpa):
    """A 288 GB part holds vectors of more than 2^32 elements: 2^32 + 12345 Goldilocks elements (32 GiB per
    vector) through the streaming kernels, the generator, the folds and the dot product -- every element index
    and byte offset beyond 32 bits.  Laws instead of an oracle pass over the bulk, the oracle on samples from
    both ends of the index range."""
    import torch
    f, o = cpa.MontyField(cpa.GOLDILOCKS, 1), _oracle("goldilocks")
    n = (1 << 32) + 12345
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * 8 * n + n + (16 << 30):
        pytest.skip("needs 120 GiB of free device memory")
    a, b, t = f.random_dev(n, 201), f.random_dev(n, 202), f.alloc(n)
    ws = torch.empty(f.fold_workspace_bytes(), dtype=torch.uint8, device="cuda")

    def total(v, w=None):
        out = f.alloc(1)
        (f.sum_dev(v, out, ws) if w is None else f.dot_dev(v, w, out, ws))
        return _host(out).reshape(1, 1)

    idx = torch.cat([torch.arange(0, 4096, device="cuda"), torch.arange(n - 4096, n, device="cuda"),
                     torch.arange(0, n, 1048573, device="cuda")])
    ha, hb = _host(a[idx]), _host(b[idx])
    f.mul_dev(a, b, t)
    assert np.array_equal(_host(t[idx]), o.mul(ha, hb))
    assert np.array_equal(total(a, b), total(t))                         # dot(a, b) == sum(a * b)
    f.neg_dev(a, t)
    assert np.array_equal(_host(t[idx]), o.neg(ha))
    s_a, s_na = total(a), total(t)
    assert np.array_equal(o.add(s_a, s_na), np.zeros((1, 1), dtype=np.uint64))   # sum(a) + sum(-a) == 0
    f.sub_dev(a, b, t)
    f.add_dev(t, b, t)
    assert torch.equal(t[-(1 << 20):], a[-(1 << 20):]) and torch.equal(t[:1 << 20], a[:1 << 20])   # (a - b) + b == a
    assert np.array_equal(total(t), s_a)
    # batch inversion over the same count: chain indices c + k M beyond 2^32, three levels of totals
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    iws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
    a[n - 1].zero_()
    a[1 << 32].zero_()
    f.inv_dev(a, t, flags, iws)
    f.mul_dev(a, t, b)                                                   # b = a * inv(a)
    one = int(torch.from_numpy(f.one_words.view(np.int64))[0])
    nz = int(flags.sum())
    assert nz >= 2 and bool(flags[n - 1]) and bool(flags[1 << 32])
    assert int((b.view(-1) == one).sum()) == n - nz and int((b.view(-1) == 0).sum()) == nz
    ha = _host(a[idx])
    winv, wflags = o.batch_inv(ha)
    assert np.array_equal(_host(t[idx]), winv) and np.array_equal(flags[idx].cpu().numpy(), wflags)
    del a, b, t, flags, iws
    torch.cuda.empty_cache()
