"""Experiment: does batch inversion gain from running as two half-batches on two HIP streams, so that one half's
top kernel (exact inversions, one wave per SIMD, pure latency) overlaps the other half's sweeps?
BLS12-381 Fq, 2^24 elements (BASELINE configs[2]).  Prints one JSON line per variant."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import _lib  # noqa: E402

P_BLS = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    n = 1 << 24
    f = cpa.MontyField(P_BLS, 6)
    x = f.random_dev(n, 7)
    out, out2 = f.alloc(n), f.alloc(n)
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    ws = torch.empty(f.inv_workspace_bytes(n, fast=True), dtype=torch.uint8, device="cuda")
    med, best = timed(lambda: f.inv_dev(x, out, flags, ws))
    print(json.dumps({"variant": "one call", "ms_median": med, "ms_best": best}), flush=True)

    for parts, c0 in ((2, None), (2, 32), (4, 32), (2, 24), (3, 32)):
        if c0 is None:
            _lib.tuning_set("FPGPU_INV_C0")
        else:
            _lib.tuning_set("FPGPU_INV_C0", c0)
        h = (n + parts - 1) // parts
        streams = [torch.cuda.Stream() for _ in range(parts)]
        wss = [torch.empty(f.inv_workspace_bytes(h, fast=True), dtype=torch.uint8, device="cuda") for _ in range(parts)]

        def split():
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            for i, s in enumerate(streams):
                lo, hi = i * h, min(n, (i + 1) * h)
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    f.inv_dev(x[lo:hi], out2[lo:hi], flags[lo:hi], wss[i])
                    d = torch.cuda.Event()
                    d.record(s)
                cur.wait_event(d)

        med, best = timed(split)
        torch.cuda.synchronize()
        same = bool(torch.equal(out, out2))
        print(json.dumps({"variant": f"{parts} parts on {parts} streams", "c0": c0, "ms_median": med, "ms_best": best,
                          "same_result": same}), flush=True)
    _lib.tuning_set("FPGPU_INV_C0")


if __name__ == "__main__":
    main()
