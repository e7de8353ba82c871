"""Times fpgpu_pow_dev (one exponent per batch, exponent p - 2) and checks x^(p-2) * x == 1."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd._lib import check
for name, p, n64, n in (("bn254_fr", cpa.BN254_FR, 4, 1 << 22), ("bls12_381_fq", cpa.BLS12_381_FQ, 6, 1 << 21)):
    f = cpa.MontyField(p, n64)
    x = f.random_dev(n, 5); out = f.alloc(n)
    e = p - 2
    ew = (ctypes.c_uint64 * n64)(*[(e >> (64 * i)) & (2**64 - 1) for i in range(n64)])
    f._lib.fpgpu_tuning_set(b"FPGPU_POW_WINDOW", int(os.environ.get("POW_WINDOW", "-1")) if os.environ.get("POW_WINDOW") else cpa._lib.TUNING_DEFAULT)
    def run():
        check(f._lib.fpgpu_pow_dev(f._h, f._chk_t(x), ew, n64, f._chk_t(out), n, f._stream()))
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); run(); run(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    chk = f.alloc(n); f.mul_dev(x, out, chk); torch.cuda.synchronize()
    one = torch.from_numpy(f.one_words.view(np.int64)).cuda()
    nz = (x != 0).any(dim=1)
    ok = bool(torch.equal(chk[nz], one.expand(int(nz.sum()), n64)))
    muls = (e.bit_length() - 1) + bin(e).count("1") - 1
    print({"window": os.environ.get("POW_WINDOW", "auto"), "field": name, "n": n, "ms": round(ms, 3), "Gpow_per_s": round(n / ms / 1e6, 3), "Gmul_equiv_per_s": round(n * muls / ms / 1e6, 1), "inverse_ok": ok}, flush=True)
