"""GPU: the *_dev entry points and the entry point that allocates nothing inside a HIP graph.

The callers fpgpu_batch_inv_static_dev exists for (include/fpgpu.h) capture their launches into a graph and
replay it; under capture any allocation fails the capture.  This captures a chain of elementwise kernels, a
dependent mul chain, the static-scratch batch inversion and a matrix-core mat-vec with a caller workspace
into one graph (torch.cuda.CUDAGraph = hipGraph on ROCm; the library launches on torch's capturing stream),
replays it on new inputs written into the same buffers, and compares with the oracle."""
import numpy as np
import pytest

from conftest import ints_to_array, synth_elements

pytestmark = pytest.mark.gpu


def test_dev_calls_and_static_inversion_replay_from_a_graph():
    import torch
    import crypto_primitives_amd as cpa
    from oracle.fp_oracle import Oracle
    assert cpa.load().fpgpu_device_count() >= 1, "no MI355X visible"
    p, n64, n = cpa.BN254_FR, 4, 20011
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rows, cols = 64, 128
    a, b = f.alloc(n), f.alloc(n)
    t, u, v, w = f.alloc(n), f.alloc(n), f.alloc(n), f.alloc(n)
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    A, x, y = f.alloc(rows * cols), f.alloc(cols), f.alloc(rows)
    ws = torch.empty(f.matvec_workspace_bytes(rows, cols), dtype=torch.uint8, device="cuda")

    def chain():
        f.mul_dev(a, b, t)               # t = a b
        f.add_dev(t, a, u)               # u = a b + a
        f.mul_chain_dev(u, b, v, 3)      # v = u b^3
        f.inv_static_dev(v, w, flags)    # w = 1 / v, None flags; workspace = the library's static array
        f.matvec_dev(A, rows, cols, x, y, ws)

    def fill(seed):
        ha = ints_to_array(synth_elements(p, n64, n, seed), n64)
        hb = ints_to_array(synth_elements(p, n64, n, seed + 1), n64)
        ha[seed % n] = 0
        hb[(seed * 7) % n] = 0
        hA = ints_to_array(synth_elements(p, n64, rows * cols, seed + 2), n64)
        hx = ints_to_array(synth_elements(p, n64, cols, seed + 3), n64)
        for dst, src in ((a, ha), (b, hb), (A, hA), (x, hx)):
            dst.copy_(torch.from_numpy(src.view(np.int64)).view(dst.shape))
        return ha, hb, hA, hx

    def check(ha, hb, hA, hx):
        want_u = o.add(o.mul(ha, hb), ha)
        want_v = want_u
        for _ in range(3):
            want_v = o.mul(want_v, hb)
        want_w, want_flags = o.batch_inv(want_v)
        got = lambda z: z.cpu().numpy().view(np.uint64).reshape(-1, n64)
        assert np.array_equal(got(u), want_u) and np.array_equal(got(v), want_v)
        assert np.array_equal(got(w), want_w) and np.array_equal(flags.cpu().numpy(), want_flags)
        assert np.array_equal(got(y), o.matvec(hA, rows, cols, hx))

    # eager once: also makes every one-time set-up (the static array's address on this device) happen outside the capture
    inputs = fill(11)
    chain()
    torch.cuda.synchronize()
    check(*inputs)

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain()
    for seed in (23, 37):
        inputs = fill(seed)
        for z in (t, u, v, w, y):
            z.zero_()
        g.replay()
        torch.cuda.synchronize()
        check(*inputs)
