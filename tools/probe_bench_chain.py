"""Runs bench.py's own sequence up to the drop-in chain in one process and times the chain's three host calls one by one."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import crypto_primitives_amd as cpa
import bench
f = cpa.MontyField(cpa.BN254_FR, 4)
n = 1 << 22
rng = np.random.default_rng(5)
ha = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); ha[:, 3] &= (1 << 60) - 1
hb = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); hb[:, 3] &= (1 << 60) - 1
def one_chain(tag):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); t = f.mul(ha, hb); t1 = time.perf_counter(); u = f.add(t, ha); t2 = time.perf_counter(); v = f.sqr(u); t3 = time.perf_counter()
        ts.append([round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1), round((t3 - t2) * 1e3, 1)])
    print(tag, ts, flush=True)
one_chain("fresh")
big = [f.random_dev(1 << 28, s) for s in (1, 2)]; out = f.alloc(1 << 28)
for _ in range(5): f.mul_dev(big[0], big[1], out)
torch.cuda.synchronize()
one_chain("after 24 GiB + mul kernels")
V = cpa.DeviceVector
vs = [V.alloc(f, n) for _ in range(5)]
one_chain("after handle allocations")
vs[0].upload(ha); vs[1].upload(hb); f.mul_vec(vs[0], vs[1], vs[2]); r = vs[2].download()
one_chain("after handle transfers")
def time_steps(fn, steps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); return 1.0, 1.0
bench.bench_spmv(f, 0, time_steps)
one_chain("after bench_spmv")
print(json.dumps(bench.bench_drop_in_chain(f)["host_buffer_calls"]))
one_chain("after bench_drop_in_chain")
