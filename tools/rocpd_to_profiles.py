"""Turns the rocprofv3 (rocpd sqlite) outputs of tools/collect_profiles.sh into the
summaries kept under profiles/: per-kernel stats CSV and the PMC traffic JSON."""
import csv, json, sqlite3, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/collect"
tag = sys.argv[2] if len(sys.argv) > 2 else "r1"

import os


def stats(dbpath, outpath):
    db = sqlite3.connect(dbpath)
    with open(outpath, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPRs", "SGPRs"])
        for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            mn, mx, vg, sg = db.execute("select min(duration),max(duration),max(vgpr_count),max(sgpr_count) from kernels where name=?", (name,)).fetchone()
            w.writerow([name[:160], calls, int(tot * 1e3), int(avg * 1e3), round(pct, 3), mn, mx, vg, sg])


stats(f"{src}/trace/t_results.db", f"profiles/{tag}_bench_kernel_stats.csv")
if os.path.exists(f"{src}/trace_all/t_results.db"):
    stats(f"{src}/trace_all/t_results.db", f"profiles/{tag}_bench_all_workloads_kernel_stats.csv")


def counter(path, cname, kernel_like):
    d = sqlite3.connect(path)
    cur = d.execute("select * from counters_collection limit 1")
    cols = [c[0] for c in cur.description]
    rows = d.execute("select * from counters_collection").fetchall()
    ki, ci, vi = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name"), cols.index("counter_name"), cols.index("value")
    di = cols.index("dispatch_id")
    per = {}
    for r in rows:
        if kernel_like in r[ki] and r[ci] == cname:
            per[r[di]] = per.get(r[di], 0.0) + r[vi]
    vals = list(per.values())
    return sum(vals) / len(vals), len(vals)


n = 1 << 28
kern = "k_elementwise<8, 0>"
fetch, nf = counter(f"{src}/pmc_fetch/f_results.db", "FETCH_SIZE", kern)
write, nw = counter(f"{src}/pmc_write/w_results.db", "WRITE_SIZE", kern)
rd, wr = fetch * 1024 * 2, write * 1024
out = {
    "kernel": "k_elementwise<8,mul>", "elements": n, "dispatches_averaged": [nf, nw],
    "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
    "corrections": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> x2 "
                   "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-B-per-lane stores",
    "read_bytes": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": 96 * n, "traffic_over_algorithmic": (rd + wr) / (96 * n),
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 "
              "--warmup 1 --no-extra --no-cpu (tools/collect_profiles.sh, tools/rocpd_to_profiles.py)",
}
json.dump(out, open(f"profiles/{tag}_pmc_traffic_mul.json", "w"), indent=1)
print(json.dumps(out, indent=1))

# mat-vec (BASELINE.json configs[3]): traffic of the main kernel
if os.path.exists(f"{src}/pmc_fetch_mv/f_results.db"):
    dim = 1 << 15
    kern = "k_matvec_mfma"
    fetch, nf = counter(f"{src}/pmc_fetch_mv/f_results.db", "FETCH_SIZE", kern)
    write, nw = counter(f"{src}/pmc_write_mv/w_results.db", "WRITE_SIZE", kern)
    rd, wr = fetch * 1024 * 2, write * 1024
    alg = 32 * (dim * dim + dim + dim)
    out = {
        "kernel": "k_matvec_mfma<8,4>", "rows": dim, "cols": dim, "dispatches_averaged": [nf, nw],
        "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
        "corrections": "as pmc_traffic_mul.json (FETCH_SIZE x2 on gfx950; WRITE_SIZE as counted)",
        "read_bytes": rd, "write_bytes": wr, "traffic_bytes_per_launch": rd + wr,
        "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg,
        "note": "reads beyond the matrix are the Toeplitz operand (64 MiB per call, re-read by every workgroup, "
                "served by L2 / Infinity Cache where it hits); writes are the 64-bit diagonal-sum atomics",
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --workload matvec "
                  "--steps 3 --warmup 1 --no-extra --no-cpu",
    }
    json.dump(out, open(f"profiles/{tag}_pmc_traffic_matvec.json", "w"), indent=1)
    print(json.dumps(out, indent=1))
