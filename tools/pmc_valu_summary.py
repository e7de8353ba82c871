"""Summarises tools/collect_pmc_valu.sh (gpurun_out/pmcv) into profiles/: VALU instruction rates of
the VALU-bound kernels and the HBM traffic of the add kernel (same corrections as pmc_traffic_mul.json)."""
import collections, csv, json, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmcv"
tag = sys.argv[2] if len(sys.argv) > 2 else "r1"


def counters(path):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        per[r["Kernel_Name"]][r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return per


def durations(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


valu = counters(f"{src}/valu/v_counter_collection.csv")
dur = durations(f"{src}/trace/t_kernel_trace.csv")
out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- python3 bench.py "
                 "--steps 2 --warmup 1 --no-cpu --workload add; durations from a separate --kernel-trace run of the same "
                 "command (tools/collect_pmc_valu.sh)",
       "notes": "SQ_INSTS_VALU counts wave-level instructions (x64 lanes); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so "
                "GUI_ACTIVE / 8 / duration is the shader clock the kernel actually ran at; the nominal VALU ceiling is "
                "256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T lane-op/s for 4-cycle instructions",
       "kernels": {}}
for name, disp in valu.items():
    if not any(k in name for k in ("k_mul_chain", "k_binv_up", "k_binv_down", "k_binv_top", "k_matmul_mfma", "k_elementwise<8, 2>")):
        continue
    n = len(disp)
    avg = {c: sum(d[c] for d in disp.values()) / n for c in next(iter(disp.values()))}
    ds = dur.get(name, [])
    ns = sum(ds) / len(ds) if ds else None
    row = {"dispatches": n, "avg_duration_us": ns / 1e3 if ns else None, **{k: v for k, v in avg.items()}}
    if ns:
        row["valu_lane_instr_per_s_T"] = avg["SQ_INSTS_VALU"] * 64 / (ns * 1e-9) / 1e12
        # only meaningful for long kernels: the counter also ticks around a dispatch
        row["shader_clock_GHz"] = avg["GRBM_GUI_ACTIVE"] / 8 / ns if ns > 300e3 else None
    out["kernels"][name.split("(")[0].replace("void fpgpu::", "")] = row
json.dump(out, open(f"profiles/{tag}_pmc_valu_rates.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])

# HBM traffic of the add kernel
n = 1 << 28
kern = "k_elementwise<8, 2>"
def avg_counter(path, cname):
    per = counters(path)
    vals = [d[cname] for name, disp in per.items() if kern in name for d in disp.values()]
    return sum(vals) / len(vals), len(vals)
fetch, nf = avg_counter(f"{src}/fetch_add/f_counter_collection.csv", "FETCH_SIZE")
write, nw = avg_counter(f"{src}/write_add/w_counter_collection.csv", "WRITE_SIZE")
rd, wr = fetch * 1024 * 2, write * 1024
t = {"kernel": "k_elementwise<8,add>", "elements": n, "dispatches_averaged": [nf, nw], "FETCH_SIZE_KiB_raw": fetch,
     "WRITE_SIZE_KiB_raw": write, "corrections": "as pmc_traffic_mul.json (FETCH_SIZE x2 on gfx950; WRITE_SIZE as counted)",
     "read_bytes": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": 96 * n,
     "traffic_over_algorithmic": (rd + wr) / (96 * n),
     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --workload add --steps 3 "
               "--warmup 1 --no-extra --no-cpu (tools/collect_pmc_valu.sh, tools/pmc_valu_summary.py)"}
json.dump(t, open(f"profiles/{tag}_pmc_traffic_add.json", "w"), indent=1)
print(json.dumps(t, indent=1))
