"""GPU: the shapes one rank gets when BASELINE.json's configs are split over 8 GPUs, run at full size
on the one GPU of the box.  The device list names GPU 0 eight times, so the library lays every
vector out exactly as it would over eight devices -- eight row blocks of 4096 x 32768 for the
mat-vec (which makes the matrix-core kernel split columns across workgroups: 32 row blocks per
part), eight 1024 x 8192 x 8192 mat-muls, eight batch inversions of 2^21 elements -- each on its own
stream, and gathers replicated outputs.  The concatenation must equal the unsharded result
bit for bit (the generator is layout-independent, so both runs see the same data)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EIGHT = [0] * 8


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as c
    assert c.load().fpgpu_device_count() >= 1, "no MI355X visible"
    return c


@pytest.fixture
def eight_entries(cpa):
    cpa.set_devices(EIGHT)
    yield
    cpa.set_devices([])


def test_matvec_config3_as_eight_row_blocks(cpa, eight_entries):
    f = cpa.MontyField(cpa.BN254_FR, 4)
    V = cpa.DeviceVector
    dim = 1 << 15
    A8 = V.alloc_rows(f, dim, dim, cpa.PLACE_SHARDED).fill_random(104)
    parts = A8.parts()
    assert len(parts) == 8 and all(c == (dim // 8) * dim for _, _, c, _, _ in parts)
    x8 = V.alloc(f, dim, cpa.PLACE_REPLICATED).fill_random(105)
    y8 = V.alloc(f, dim, cpa.PLACE_REPLICATED)
    f.matvec_vec(A8, dim, dim, x8, y8)
    got = y8.download()
    del A8
    A1 = V.alloc(f, dim * dim, 0).fill_random(104)
    x1 = V.alloc(f, dim, 0).fill_random(105)
    y1 = V.alloc(f, dim, 0)
    f.matvec_vec(A1, dim, dim, x1, y1)
    want = y1.download()
    assert np.array_equal(got, want)
    # and a sampled row against the oracle
    from oracle.fp_oracle import Oracle
    o = Oracle(cpa.BN254_FR, 4)
    r = 4096 * 5 + 17                                           # a row of the sixth block
    row = A1.download(first=r * dim, n=dim)
    assert np.array_equal(got[r:r + 1], o.matvec(row, 1, dim, x1.download(), threads=0))


def test_matmul_config4_as_eight_row_blocks(cpa, eight_entries):
    f = cpa.MontyField(cpa.GOLDILOCKS, 1)
    V = cpa.DeviceVector
    dim = 8192
    A8 = V.alloc_rows(f, dim, dim, cpa.PLACE_SHARDED).fill_random(107)
    B8 = V.alloc(f, dim * dim, cpa.PLACE_REPLICATED).fill_random(108)
    C8 = V.alloc_rows(f, dim, dim, cpa.PLACE_SHARDED)
    f.matmul_vec(A8, B8, C8, dim, dim, dim)
    got = C8.download()
    del A8, B8, C8
    A1 = V.alloc(f, dim * dim, 0).fill_random(107)
    B1 = V.alloc(f, dim * dim, 0).fill_random(108)
    C1 = V.alloc(f, dim * dim, 0)
    f.matmul_vec(A1, B1, C1, dim, dim, dim)
    assert np.array_equal(got, C1.download())


def test_batch_inversion_config2_as_eight_ranges(cpa, eight_entries):
    f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
    V = cpa.DeviceVector
    n = 1 << 24
    x8 = V.alloc(f, n, cpa.PLACE_SHARDED).fill_random(103)
    zeros = [0, 1, (1 << 21) - 1, 1 << 21, 5 * (1 << 21) + 77, n - 1]     # first / last element of a range included
    for z in zeros:
        x8.upload(np.zeros((1, 6), dtype=np.uint64), first=z)
    o8 = V.alloc(f, n, cpa.PLACE_SHARDED)
    flags8 = f.inv_vec(x8, o8)
    got = o8.download()
    x1 = V.alloc(f, n, 0).copy_from(x8)
    o1 = V.alloc(f, n, 0)
    flags1 = f.inv_vec(x1, o1)
    assert np.array_equal(flags8, flags1) and int(flags8.sum()) == len(zeros) and all(flags8[z] == 1 for z in zeros)
    assert np.array_equal(got, o1.download())
    # x * inv(x) == one wherever x != 0, on the sharded layout
    chk = V.alloc(f, n, cpa.PLACE_SHARDED)
    f.mul_vec(x8, o8, chk)
    h = chk.download()
    nz = flags8 == 0
    assert (h[nz] == f.one_words).all()
