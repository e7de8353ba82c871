"""GPU: fpgpu_batch_inv_static_dev (the entry point that allocates nothing: its workspace is a static
device array inside the library) against fpgpu_batch_inv_dev, bit for bit, on two calls in a row,
the second with a smaller input (the array keeps the first call's contents)."""
import numpy as np
import pytest

from conftest import h2i, ints_to_array, load_golden, synth_elements

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,sizes", [("bls12_381_fq", (300001, 4097)), ("bn254_fr", (70001, 65)),
                                        ("goldilocks", (100003, 1)), ("odd_composite_128", (9000, 777))])
def test_static_scratch_batch_inv_matches_workspace_variant(name, sizes):
    import torch
    import crypto_primitives_amd as cpa
    assert cpa.load().fpgpu_device_count() >= 1, "no MI355X visible"
    from oracle.fp_oracle import Oracle
    g = load_golden(name)
    p, n64 = h2i(g["modulus"]), g["n64"]
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    for call, n in enumerate(sizes):
        xs = synth_elements(p, n64, n, 500 + call)
        for i in range(0, n, 211):
            xs[i] = 0
        xs[-1] = 0
        if name == "odd_composite_128":
            xs[n // 2] = 3          # not invertible: poisons its group, the fallback kernel redoes it
        hx = ints_to_array(xs, n64)
        x = torch.from_numpy(hx.view(np.int64)).cuda()
        want, wflags = f.alloc(n), torch.empty(n, dtype=torch.uint8, device="cuda")
        ws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
        f.inv_dev(x, want, wflags, ws)
        got, gflags = f.alloc(n), torch.empty(n, dtype=torch.uint8, device="cuda")
        got.fill_(-1)
        gflags.fill_(7)
        f.inv_static_dev(x, got, gflags)
        torch.cuda.synchronize()
        assert torch.equal(gflags, wflags), (name, n)
        assert torch.equal(got, want), (name, n)
        # and both equal the oracle (the bar of the existing entry point's own tests)
        exp, eflags = o.batch_inv(hx)
        assert np.array_equal(got.cpu().numpy().view(np.uint64), exp) and np.array_equal(gflags.cpu().numpy(), eflags)


def test_static_scratch_rejects_oversized_batch():
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import _lib
    f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
    n = 1 << 27
    import torch
    x = torch.empty((16, 6), dtype=torch.int64, device="cuda")   # never read: the size check comes first
    with pytest.raises(cpa.FpgpuError) as e:
        import ctypes
        _lib.check(f._lib.fpgpu_batch_inv_static_dev(f._h, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(x.data_ptr() + 48),
                                                     None, n, None))
    assert e.value.code == _lib.ERR_SHAPE and "static array holds" in str(e.value)
