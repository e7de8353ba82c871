"""CPU: what fpgpu_matmul_workspace_bytes reports for multi-limb fields (make_mfl_plan, csrc/fp_launch.hpp).  The
matrix-core product needs the digit planes of both operands (as many bytes as the padded operands themselves), a
scratch block per workgroup of its persistent grid and, where k is split over work items, the parts' residues; the
function must report that much wherever the launcher would take the path, nothing for shapes that stay on the VALU
kernel, and follow the FPGPU_MATMUL_MFMA hook.  No counterpart in the reference: scratch is the drop-in's own business."""
import pytest


def _planes(n64, m, k, n):
    r64 = lambda v: (v + 63) // 64 * 64
    return 8 * n64 * r64(k) * (r64(m) + r64(n))


@pytest.mark.parametrize("n64,p", [(2, 2**127 - 1), (4, None), (6, None), (8, 2**512 - 569)])
def test_multi_limb_matmul_workspace_follows_the_plan(tuning, n64, p):
    import crypto_primitives_amd as cpa
    p = p or {4: cpa.BN254_FR, 6: cpa.BLS12_381_FQ}[n64]
    f = cpa.MontyField(p, n64)
    scratch_per_wg = (2 * n64 - 1) * 6 * 4096 * 4
    # large products: the matrix cores, digit planes + at most 256 workgroups of scratch, no k parts from 192 tiles on
    for m, k, n in ((1024, 1024, 1024), (2048, 2048, 2048), (3000, 500, 3000), (1024, 8192, 1024), (4096, 70, 4096)):
        b = f.matmul_workspace_bytes(m, k, n)
        assert _planes(n64, m, k, n) + 256 <= b <= _planes(n64, m, k, n) + 256 * scratch_per_wg + 256, (m, k, n, b)
    # few tiles over a long k: k parts, their residues on top
    m, k, n = 512, 4096, 512
    b = f.matmul_workspace_bytes(m, k, n)
    assert b >= _planes(n64, m, k, n) + 4 * m * n * 8 * n64, (b,)
    # small products stay on the VALU kernel and ask for nothing
    for m, k, n in ((64, 512, 64), (17, 33, 19), (100, 20, 100)):
        assert f.matmul_workspace_bytes(m, k, n) == 0, (m, k, n)
    # the shapes with paths of their own keep them: one column (the mat-vec's size), a small output over a long k (k parts
    # of the VALU kernel), a few vectors times a matrix
    assert f.matmul_workspace_bytes(300, 5000, 1) == f.matvec_workspace_bytes(300, 5000)
    assert 0 < f.matmul_workspace_bytes(16, 1 << 20, 16) < _planes(n64, 16, 1 << 20, 16)
    assert 0 < f.matmul_workspace_bytes(1, 4096, 4096) < _planes(n64, 1, 4096, 4096)
    # the hook: 0 = never (nothing to ask for), 1 = every shape the plan can lay out
    tuning("FPGPU_MATMUL_MFMA", 0)
    assert f.matmul_workspace_bytes(2048, 2048, 2048) == 0
    tuning("FPGPU_MATMUL_MFMA", 1)
    assert f.matmul_workspace_bytes(64, 512, 64) >= _planes(n64, 64, 512, 64)
    assert f.matmul_workspace_bytes(17, 33, 19) >= _planes(n64, 17, 33, 19)


def test_single_limb_matmul_workspace():
    import crypto_primitives_amd as cpa
    f = cpa.MontyField(cpa.GOLDILOCKS, 1)
    assert f.matmul_workspace_bytes(8192, 8192, 8192) == 256 + 2 * 8192 * 8192 * 8
    # k beyond the int32 accumulators' 16320: the limb kernel with one limb (planes of both operands + scratch)
    b = f.matmul_workspace_bytes(1000, 20000, 1000)
    assert _planes(1, 1000, 20000, 1000) + 256 <= b <= _planes(1, 1000, 20000, 1000) + 256 + 256 * 6 * 4096 * 4
