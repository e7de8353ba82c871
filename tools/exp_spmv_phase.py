"""Would a column-blocked sparse matrix keep its gathers in L2?  Emulation with the shipped kernel: one lane per row
(FPGPU_SPMV_LOG2G=0), U cells per step, and a matrix whose cell j of EVERY row lies in column block
j * nblk / density -- so that step k of every lane gathers from the same block of x, as a blocked format would make it.
Nothing synchronises the waves; the question is whether they stay in step well enough for the block to stay in the
4 MiB of L2 an XCD has.  (The values are read at a 1 KiB stride per lane here, so the time is not the point: the L2
hit rate under `rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum` is.)  argv: nblk [random]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 16
unstructured = len(sys.argv) > 2 and sys.argv[2] == "random"
f = cpa.MontyField(cpa.BN254_FR, 4)
rows, density, cols = 1 << 20, 32, 1 << 20
vals = f.random_dev(rows * density, 0x5A)
x = f.random_dev(cols, 0x5B)
y = f.alloc(rows)
g = torch.Generator(device="cuda"); g.manual_seed(1234)
blk = cols // nblk
within = torch.randint(0, blk, (rows, density), dtype=torch.int64, device="cuda", generator=g)
which = (torch.arange(density, device="cuda", dtype=torch.int64) * nblk) // density
col = (within + which[None, :] * blk).to(torch.int32).reshape(-1).contiguous()
if unstructured:
    col = torch.randint(0, cols, (rows * density,), dtype=torch.int32, device="cuda", generator=g)
for _ in range(2): f.spmv_dev(vals, col, rows, cols, density, x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f.spmv_dev(vals, col, rows, cols, density, x, y)
e1.record(); torch.cuda.synchronize()
print(json.dumps({"nblk": nblk, "structured": not unstructured, "log2g": os.environ.get("FPGPU_SPMV_LOG2G"),
                  "unroll": os.environ.get("FPGPU_SPMV_UNROLL"), "ms": round(e0.elapsed_time(e1) / 5, 4)}), flush=True)
