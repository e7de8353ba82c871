"""Sparse mat-vec with the gathered operand x in three kinds of device memory (2^20 rows x 32 cells over 2^20
columns, BN254): ordinary (cached in L2 by 128-byte lines), fine-grained and uncached (hipExtMallocWithFlags).
Question: does a 32-byte gather from memory the L2 does not cache cross the fabric as 32 / 64 bytes instead of a
128-byte line?  Results are checked against the ordinary run (bit-exact) before they are timed."""
import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import _lib

only = sys.argv[1] if len(sys.argv) > 1 else None
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipFree.argtypes = [ctypes.c_void_p]

f = cpa.MontyField(cpa.BN254_FR, 4)
lib = _lib.load()
rows, density, cols = 1 << 20, 32, 1 << 20
vals = f.random_dev(rows * density, 0x5A)
x = f.random_dev(cols, 0x5B)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
col = torch.randint(0, cols, (rows * density,), dtype=torch.int32, device="cuda", generator=g)
y_ref = f.alloc(rows)
f.spmv_dev(vals, col, rows, cols, density, x, y_ref)
torch.cuda.synchronize()

def run(xp, y):
    rc = lib.fpgpu_spmv_dev(f._h, ctypes.c_void_p(vals.data_ptr()), ctypes.c_void_p(col.data_ptr()), rows, cols, density,
                            ctypes.c_void_p(xp), ctypes.c_void_p(y.data_ptr()), f._stream())
    assert rc == 0, rc

for name, flags in (("ordinary", None), ("finegrained", 0x1), ("uncached", 0x3)):
    if only and name != only:
        continue
    if flags is None:
        xp, owned = x.data_ptr(), None
    else:
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), cols * 32, flags)
        if rc != 0:
            print(json.dumps({"x_memory": name, "error": f"hipExtMallocWithFlags -> {rc}"}), flush=True)
            continue
        assert hip.hipMemcpy(p, ctypes.c_void_p(x.data_ptr()), cols * 32, 3) == 0   # device to device
        xp, owned = p.value, p
    y = f.alloc(rows)
    run(xp, y)
    torch.cuda.synchronize()
    same = bool(torch.equal(y, y_ref))
    for _ in range(3): run(xp, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(xp, y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(json.dumps({"x_memory": name, "bit_exact": same, "ms": round(ms, 4),
                      "algorithmic_TBps": round((36 * rows * density + 32 * (rows + cols)) / ms / 1e9, 3)}), flush=True)
    if owned is not None:
        torch.cuda.synchronize()
        hip.hipFree(owned)
