"""CPU: the batch inversion's reported workspace sizes never shrink as the batch grows.  A caller sizes a workspace for
its largest batch and inverts shorter ones in it (the library's own chunk pipeline does, with its last chunk; so does
the static-scratch entry point) -- while the tree's layout does shrink where a longer level-0 chain takes over
(make_inv_plan, csrc/fpgpu_api.hip).  No counterpart in the reference: scratch space is the drop-in's own business."""
import random

import pytest


@pytest.mark.parametrize("n64,p", [(1, 2**64 - 2**32 + 1), (4, None), (6, None), (8, 2**512 - 569)])
def test_inversion_workspace_never_shrinks_with_n(n64, p):
    import crypto_primitives_amd as cpa
    p = p or {4: cpa.BN254_FR, 6: cpa.BLS12_381_FQ}[n64]
    f = cpa.MontyField(p, n64)
    pts = set()
    for base in (1, 2, 100, 65536, 131072, 131073, 196608, 2**20, 2**21, 84 * 65536, 6_000_000, 2**23, 2**24, 2**24 + 2**19,
                 48 * 8 * 65536, 2**25, 2**27, 2**28):
        pts.update(n for n in range(base - 3, base + 4) if n > 0)
    rnd = random.Random(n64)
    pts.update(rnd.randrange(1, 2**26) for _ in range(2000))
    for fast in (False, True):
        prev = 0
        for n in sorted(pts):
            b = f.inv_workspace_bytes(n, fast=fast)
            assert b >= prev, (n, b, prev, fast)
            prev = b
        assert f.inv_workspace_bytes(1 << 20, fast=True) >= f.inv_workspace_bytes(1 << 20)
