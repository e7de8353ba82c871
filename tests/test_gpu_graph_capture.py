"""GPU: the *_dev entry points and the entry point that allocates nothing inside a HIP graph.

The callers fpgpu_batch_inv_static_dev exists for (include/fpgpu.h) capture their launches into a graph and
replay it; under capture any allocation fails the capture.  This captures a chain of elementwise kernels, a
dependent mul chain, the static-scratch batch inversion and a matrix-core mat-vec with a caller workspace
into one graph (torch.cuda.CUDAGraph = hipGraph on ROCm; the library launches on torch's capturing stream),
replays it on new inputs written into the same buffers, and compares with the oracle."""
import random
import time

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


# ---- round 3: every kind of device-pointer entry point, three fields, and what the replay buys a launch-bound loop ----
# An entry point records into a graph only if it does nothing but queue work on the stream it is given: no
# allocation, no synchronisation, no host-side state a replay would miss (kernel arguments are by value: the
# modulus, the exponent of `pow`, the plan of the inversion).


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as c
    assert c.load().fpgpu_device_count() >= 1, "no MI355X visible"
    return c


def _dev(torch, a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def _host(t, n64):
    return t.cpu().numpy().view(np.uint64).reshape(-1, n64)


@pytest.mark.parametrize("name,p,n64", [("bn254", None, 4), ("bls12_381", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6),
                                        ("goldilocks", 2**64 - 2**32 + 1, 1)])
def test_device_entry_points_record_into_a_hip_graph(cpa, name, p, n64):
    import torch
    from oracle.fp_oracle import Oracle
    p = p or cpa.BN254_FR
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    rnd = random.Random(n64)
    n, rows, cols, e = 20011, 96, 640, rnd.getrandbits(64 * n64)

    def fresh(k, zeros=()):
        v = [rnd.randrange(1, p) for _ in range(k)]
        for z in zeros:
            v[z] = 0
        return ints_to_array(v, n64)

    a, b, A, x = _dev(torch, fresh(n)), _dev(torch, fresh(n)), _dev(torch, fresh(rows * cols)), _dev(torch, fresh(cols))
    t1, t2, t3, t4, t5 = (f.alloc(n) for _ in range(5))
    y, s = f.alloc(rows), f.alloc(1)
    fl_static, fl_ws = torch.zeros(n, dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.uint8, device="cuda")
    ws_inv = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
    ws_mv = torch.empty(max(16, f.matvec_workspace_bytes(rows, cols)), dtype=torch.uint8, device="cuda")
    ws_fold = torch.empty(max(16, f.fold_workspace_bytes()), dtype=torch.uint8, device="cuda")

    def loop():
        f.mul_dev(a, b, t1)
        f.add_dev(t1, a, t2)
        f.sqr_dev(t2, t1)
        f.inv_static_dev(t1, t2, fl_static)            # the entry point that allocates nothing (static scratch)
        f.inv_dev(t1, t3, fl_ws, ws_inv)               # the same through a caller's workspace (limb-form prefixes)
        f.pow_dev(a, e, t4)
        f.mul_chain_dev(a, b, t5, 3)
        f.matvec_dev(A, rows, cols, x, y, workspace=ws_mv)   # matrix cores
        f.dot_dev(t4, t5, s, workspace=ws_fold)

    loop()                                            # outside capture first: code objects load on first use
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loop()
    for zeros in ((), (0, 5, n - 1)):                 # replay on new contents; the second time with zeros to flag
        ha, hb, hA, hx = fresh(n, zeros), fresh(n), fresh(rows * cols), fresh(cols)
        a.copy_(_dev(torch, ha)); b.copy_(_dev(torch, hb)); A.copy_(_dev(torch, hA)); x.copy_(_dev(torch, hx))
        for t in (t1, t2, t3, t4, t5, y, s):
            t.zero_()
        g.replay()
        torch.cuda.synchronize()
        w1 = o.mul(ha, hb)
        w2 = o.add(w1, ha)
        w1 = o.mul(w2, w2)
        winv, wflags = o.batch_inv(w1)
        assert np.array_equal(_host(t1, n64), w1)
        assert np.array_equal(_host(t2, n64), winv) and np.array_equal(fl_static.cpu().numpy(), wflags)
        assert np.array_equal(_host(t3, n64), winv) and np.array_equal(fl_ws.cpu().numpy(), wflags)
        assert int(wflags.sum()) == len(zeros)
        w4 = o.pow(ha, e, exponent_bits=64 * n64)
        w5 = o.mul(o.mul(o.mul(ha, hb), hb), hb)
        assert np.array_equal(_host(t4, n64), w4) and np.array_equal(_host(t5, n64), w5)
        assert np.array_equal(_host(y, n64), o.matvec(hA, rows, cols, hx))
        assert np.array_equal(_host(s, n64)[0], o.sum(o.mul(w4, w5)))


def test_graph_replay_of_a_launch_bound_chain(cpa):
    """Eighty-five dependent elementwise operations on 4096 elements: each launch is a few microseconds of work, so the
    loop is bound by launch overhead; the captured graph gives the same bits and its time is printed beside the
    direct loop's (no assertion on the ratio: it is the caller's gain, not a property of the kernels)."""
    import torch
    from oracle.fp_oracle import Oracle
    f, o = cpa.MontyField(cpa.BN254_FR, 4), Oracle(cpa.BN254_FR, 4)
    rnd = random.Random(5)
    n = 4096
    ha, hb = (ints_to_array([rnd.randrange(cpa.BN254_FR) for _ in range(n)], 4) for _ in range(2))
    a, b = _dev(torch, ha), _dev(torch, hb)
    u, v = f.alloc(n), f.alloc(n)

    def loop():
        f.mul_dev(a, b, u)
        for _ in range(21):
            f.add_dev(u, a, v)
            f.mul_dev(v, b, u)
            f.sqr_dev(u, v)
            f.sub_dev(v, a, u)                              # four launches per round, all through the C ABI
        return u

    loop()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loop()
    want = o.mul(ha, hb)
    for _ in range(21):
        want = o.sub(o.mul(o.mul(o.add(want, ha), hb), o.mul(o.add(want, ha), hb)), ha)
    times = {}
    for label, run in (("direct", loop), ("graph", g.replay)):
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        times[label] = (time.perf_counter() - t0) / 20 * 1e6
        assert np.array_equal(_host(u, 4), want), label
    print("85 launches on 4096 elements: direct %.0f us, graph replay %.0f us" % (times["direct"], times["graph"]))


@pytest.mark.parametrize("name,p,n64,shape", [("bn254: limb kernel, k split over work items", None, 4, (70, 300, 66)),
                                              ("bls12_381: limb kernel", 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB, 6, (65, 130, 64)),
                                              ("goldilocks: single-limb kernel", 2**64 - 2**32 + 1, 1, (70, 130, 66)),
                                              ("goldilocks: limb kernel with one limb (few tiles, long k)", 2**64 - 2**32 + 1, 1, (200, 2100, 210))])
def test_matrix_core_matmul_records_into_a_hip_graph(cpa, tuning, name, p, n64, shape):
    """fpgpu_matmul_dev with its workspace (digitisers, the matrix-core kernel, the combine of k parts, the gated VALU
    fallback, one hipMemsetAsync of the flag) is captured once and replayed on new operands; == oracle."""
    import torch
    from oracle.fp_oracle import Oracle
    from soak_inversion_plans import rand_elems
    p = p or cpa.BN254_FR
    f, o = cpa.MontyField(p, n64), Oracle(p, n64)
    if n64 > 1:
        tuning("FPGPU_MATMUL_MFMA", 1)            # (small shapes: the plan alone would leave them to the VALU kernel)
    m, k, n = shape
    rng = np.random.default_rng(n64)
    A, B, C = f.alloc(m * k), f.alloc(k * n), f.alloc(m * n)
    nb = f.matmul_workspace_bytes(m, k, n)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    hA, hB = rand_elems(rng, m * k, n64, p), rand_elems(rng, k * n, n64, p)
    A.copy_(_dev(torch, hA)); B.copy_(_dev(torch, hB))
    f.matmul_dev(A, B, C, m, k, n, workspace=ws)    # outside capture first: code objects load on first use
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        f.matmul_dev(A, B, C, m, k, n, workspace=ws)
    for _ in range(2):
        hA, hB = rand_elems(rng, m * k, n64, p), rand_elems(rng, k * n, n64, p)
        A.copy_(_dev(torch, hA)); B.copy_(_dev(torch, hB))
        C.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(_host(C, n64), o.matmul(hA, hB, m, k, n, threads=0).reshape(m * n, n64))
