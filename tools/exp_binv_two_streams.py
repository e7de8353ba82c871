"""Experiment: batch inversion of 2^24 BLS12-381 elements as one call vs two / four parts on
separate streams (do the latency-bound top levels of one part hide behind the sweeps of another?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crypto_primitives_amd as cpa
f = cpa.MontyField(cpa.BLS12_381_FQ, 6)
n = 1 << int(os.environ.get("LOG2N", "24"))
x = f.random_dev(n, 7); out = f.alloc(n)
flags = torch.empty(n, dtype=torch.uint8, device="cuda")
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ws = torch.empty(f.inv_workspace_bytes(n), dtype=torch.uint8, device="cuda")
print({"parts": 1, "ms": round(timed(lambda: f.inv_dev(x, out, flags, ws)), 4)}, flush=True)
for parts in (2, 4):
    m = n // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    wss = [torch.empty(f.inv_workspace_bytes(m), dtype=torch.uint8, device="cuda") for _ in range(parts)]
    def run():
        cur = torch.cuda.current_stream()
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                f.inv_dev(x[i * m:(i + 1) * m], out[i * m:(i + 1) * m], flags[i * m:(i + 1) * m], wss[i])
        for s in streams:
            cur.wait_stream(s)
    print({"parts": parts, "ms": round(timed(run), 4)}, flush=True)
