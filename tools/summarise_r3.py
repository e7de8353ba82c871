#!/usr/bin/env python3
"""Summarises tools/collect_r3.sh (gpurun_out/r3) into profiles/r3_*: the bench line, kernel-trace
statistics, HBM traffic per launch (FETCH_SIZE / WRITE_SIZE with the gfx950 corrections of
MI355X_MICROARCH.md) and VALU rates.  Every traffic file records the commit and the digest of the kernel
sources it measured; bench.py quotes a figure only while the tree's digest is the same."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha256  # noqa: E402

SRC = os.path.join(ROOT, "gpurun_out", "r3")
TAG = "r3"
commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "crypto_primitives_amd/csrc"], text=True).strip())
sha = kernel_sources_sha256()
prov = {"commit": commit + ("+uncommitted kernel changes" if dirty else ""), "kernel_sources_sha256": sha}


def find(d, suffix):
    hits = glob.glob(os.path.join(SRC, d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def counters(d):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    path = find(d, "counter_collection.csv")
    if not path:
        return per
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return per


def avg_counter(d, cname, pred):
    vals = [v[cname] for name, disp in counters(d).items() if pred(name) for v in disp.values()]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def short(name):
    return name.split("(")[0].replace("void fpgpu::", "")


out = {}
# ---- bench line
bench = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
open(os.path.join(ROOT, "profiles", f"{TAG}_bench_latest.json"), "w").write(bench + "\n")
# ---- kernel stats
for d, name in (("trace_mul", "bench_kernel_stats"), ("trace_all", "bench_all_workloads_kernel_stats")):
    p = find(d, "kernel_stats.csv")
    if p:
        shutil.copy(p, os.path.join(ROOT, "profiles", f"{TAG}_{name}.csv"))

# ---- traffic per launch of the dominant kernel of each workload
spec = {
    "mul": ("k_elementwise<8, 0>", {"elements": 1 << 28}, 96 * (1 << 28)),
    "add": ("k_elementwise<8, 2>", {"elements": 1 << 28}, 96 * (1 << 28)),
    "matvec": ("k_matvec_mfma", {"rows": 1 << 15, "cols": 1 << 15}, 32 * ((1 << 30) + (1 << 16))),
    "matmul": ("k_matmul_mfma_1limb", {"dim": 8192}, 8 * 3 * 8192 * 8192),
}
for w, (kern, keys, alg) in spec.items():
    f, nf = avg_counter(f"pmc_{w}_FETCH_SIZE", "FETCH_SIZE", lambda n: kern in n)
    wr, nw = avg_counter(f"pmc_{w}_WRITE_SIZE", "WRITE_SIZE", lambda n: kern in n)
    if f is None or wr is None:
        print("no PMC data for", w)
        continue
    rd, wb = f * 1024 * 2, wr * 1024
    t = {"kernel": kern, **keys, "dispatches_averaged": [nf, nw], "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": wr,
         "corrections": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM "
                        "section); WRITE_SIZE as counted",
         "read_bytes": rd, "write_bytes": wb, "traffic_bytes_per_launch": rd + wb, "algorithmic_bytes_per_launch": alg,
         "traffic_over_algorithmic": (rd + wb) / alg,
         "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --workload {w} --steps 3 "
                   "--warmup 1 --no-extra --no-cpu (tools/collect_r3.sh, tools/summarise_r3.py)", **prov}
    json.dump(t, open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_traffic_{w}.json"), "w"), indent=1)
    out[w] = round(t["traffic_over_algorithmic"], 4)

# ---- batch inversion: all k_binv_* launches of one batch
fet, wri = counters("pmc_batch_inv_FETCH_SIZE"), counters("pmc_batch_inv_WRITE_SIZE")
if fet and wri:
    n = 1 << 24
    batches = max(1, sum(len(d) for k, d in fet.items() if "k_binv_fallback" in k))
    per_kernel = {}
    tot_r = tot_w = 0.0
    for k in sorted(set(fet) | set(wri)):
        if "binv" not in k:
            continue
        r = sum(v["FETCH_SIZE"] for v in fet.get(k, {}).values()) * 1024 * 2 / batches
        w_ = sum(v["WRITE_SIZE"] for v in wri.get(k, {}).values()) * 1024 / batches
        per_kernel[short(k)] = {"read_bytes_per_batch": r, "write_bytes_per_batch": w_, "bytes_per_element": (r + w_) / n}
        tot_r += r
        tot_w += w_
    t = {"kernel": "k_binv_* (all launches of one batch)", "elements": n, "batches_averaged": batches, "per_kernel": per_kernel,
         "read_bytes": tot_r, "write_bytes": tot_w, "traffic_bytes_per_launch": tot_r + tot_w,
         "traffic_bytes_per_element": (tot_r + tot_w) / n, "algorithmic_bytes_per_launch": 96 * n,
         "traffic_over_algorithmic": (tot_r + tot_w) / (96 * n),
         "corrections": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as counted",
         "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --workload batch_inv --steps 3 "
                   "--warmup 1 --no-extra --no-cpu (tools/collect_r3.sh, tools/summarise_r3.py)", **prov}
    json.dump(t, open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_traffic_batch_inv.json"), "w"), indent=1)
    out["batch_inv"] = round(t["traffic_over_algorithmic"], 4)

# ---- VALU rates
valu = counters("pmc_valu")
dur = collections.defaultdict(list)
p = find("trace_valu", "kernel_trace.csv")
if p:
    for r in csv.DictReader(open(p)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rates = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 "
                   "--warmup 1 --no-cpu --workload add; durations from a separate --kernel-trace run of the same command "
                   "(tools/collect_r3.sh)",
         "notes": "SQ_INSTS_VALU counts wave-level instructions (x64 lanes); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so "
                  "GUI_ACTIVE / 8 / duration is the shader clock the kernel ran at; nominal VALU ceiling 256 CU x 4 SIMD x 16 lanes x "
                  "2.4 GHz = 39.3 T lane-op/s for 4-cycle instructions (36.5 T measured)", **prov, "kernels": {}}
elems = {"k_binv_up<12, true, true>": 1 << 24, "k_binv_down<12, true, true>": 1 << 24, "k_binv_up<12, true, false>": 1 << 24, "k_binv_down<12, true, false>": 1 << 24, "k_mul_chain<8>": (1 << 22) * 256}
for name, disp in valu.items():
    if not any(k in name for k in ("k_mul_chain", "k_binv_", "k_matmul_mfma", "k_elementwise<8, 2>", "k_elementwise<8, 1>", "k_dot")):
        continue
    n = len(disp)
    avg = {c: sum(d[c] for d in disp.values()) / n for c in next(iter(disp.values()))}
    ds = dur.get(name, [])
    ns = sum(ds) / len(ds) if ds else None
    row = {"dispatches": n, "avg_duration_us": ns / 1e3 if ns else None, **avg}
    if ns:
        row["valu_lane_instr_per_s_T"] = avg["SQ_INSTS_VALU"] * 64 / (ns * 1e-9) / 1e12
        row["shader_clock_GHz"] = avg["GRBM_GUI_ACTIVE"] / 8 / ns if ns > 300e3 else None
    s = short(name)
    if s in elems:
        row["lane_instr_per_unit"] = avg["SQ_INSTS_VALU"] * 64 / elems[s]
    rates["kernels"][s] = row
json.dump(rates, open(os.path.join(ROOT, "profiles", f"{TAG}_pmc_valu_rates.json"), "w"), indent=1)

# ---- MFMA busy
m, nm = avg_counter("pmc_mfma", "SQ_VALU_MFMA_BUSY_CYCLES", lambda n: "k_matmul_mfma_1limb" in n)
g, _ = avg_counter("pmc_mfma", "GRBM_GUI_ACTIVE", lambda n: "k_matmul_mfma_1limb" in n)
if m:
    tr = os.path.join(ROOT, "profiles", f"{TAG}_pmc_traffic_matmul.json")
    if os.path.exists(tr):
        t = json.load(open(tr))
        t.update({"SQ_VALU_MFMA_BUSY_CYCLES": m, "GRBM_GUI_ACTIVE_sum_over_8_xcds": g})
        json.dump(t, open(tr, "w"), indent=1)
print(json.dumps({"traffic_over_algorithmic": out, **prov}, indent=1))
