"""Elementwise kernels, folds and the dot product by field width (about 2 GiB per operand): TB/s of algorithmic bytes.
The bench quotes BN254; this shows what the other widths get."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
fields = [("goldilocks", 2**64 - 2**32 + 1, 1), ("p128", 2**127 - 1, 2), ("p192", 2**192 - 2**64 - 1, 3), ("bn254_fr", cpa.BN254_FR, 4),
          ("secp256k1", 2**256 - 2**32 - 977, 4), ("p320", 2**320 - 197, 5), ("bls12_381_fq", cpa.BLS12_381_FQ, 6), ("p448", 2**448 - 2**224 - 1, 7),
          ("p512", 2**512 - 569, 8)]
def timed(fn, reps=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, p, n64 in fields:
    f = cpa.MontyField(p, n64)
    n = (2 << 30) // (8 * n64)
    a = f.random_dev(n, 1); b = f.random_dev(n, 2); o = f.alloc(n); s = f.alloc(1)
    wsf = torch.empty(max(16, f.fold_workspace_bytes()), dtype=torch.uint8, device="cuda")
    eb = 8 * n64
    out = {"field": name, "n64": n64, "n": n}
    for label, fn, nbytes in (("mul", lambda: f.mul_dev(a, b, o), 3 * eb), ("add", lambda: f.add_dev(a, b, o), 3 * eb),
                              ("sqr", lambda: f.sqr_dev(a, o), 2 * eb), ("to_mont", lambda: f.to_mont_dev(a, o), 2 * eb),
                              ("neg", lambda: f.neg_dev(a, o), 2 * eb),
                              ("sum", lambda: f.sum_dev(a, s, workspace=wsf), eb), ("product", lambda: f.product_dev(a, s, workspace=wsf), eb),
                              ("dot", lambda: f.dot_dev(a, b, s, workspace=wsf), 2 * eb)):
        ms = timed(fn)
        out[label] = round(nbytes * n / ms / 1e9, 2)
    print(json.dumps(out), flush=True)
    del a, b, o, s
    torch.cuda.empty_cache()
