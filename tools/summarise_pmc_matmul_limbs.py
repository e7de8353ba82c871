"""profiles/r3_pmc_matmul_limbs.json from the rocprofv3 passes over tools/time_matmul_limbs_one.py 2048 4 (BN254 Fr, 2048^3):
gpurun_out/mfl_pmc (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU), mfl_pmc_f (FETCH_SIZE), mfl_pmc_w
(WRITE_SIZE), mfl_trace4 (--kernel-trace --stats); each pass a run of its own.  Corrections as tools/summarise_r3.py:
FETCH_SIZE in KiB x 2 on gfx950, WRITE_SIZE in KiB; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_sources_sha256  # noqa: E402


def load(path):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(ROOT, path))):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


pm, pf, pw = (load(f"gpurun_out/{d}/p_counter_collection.csv") for d in ("mfl_pmc", "mfl_pmc_f", "mfl_pmc_w"))
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out/mfl_trace4/t_kernel_stats.csv")))}
dirty = subprocess.run(["git", "diff", "--quiet", "--", "crypto_primitives_amd/csrc"], cwd=ROOT).returncode != 0
res = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU / --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                 "(separate passes) and --kernel-trace --stats -- python3 tools/time_matmul_limbs_one.py 2048 4 (BN254 Fr, 2048^3, six launches; "
                 "averages per launch; tools/summarise_pmc_matmul_limbs.py)",
       "commit": subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], text=True, cwd=ROOT).strip() + ("+uncommitted kernel changes" if dirty else ""),
       "kernel_sources_sha256": kernel_sources_sha256(), "kernels": {}}
for k in pm:
    if "mfl" not in k and "matmul" not in k:
        continue
    short = k.split("(")[0].replace("void fpgpu::", "")
    avg = lambda d, c: sum(d[k][c]) / len(d[k][c]) if d[k].get(c) else None  # noqa: E731
    e = {"avg_duration_us": float(stats[k]["AverageNs"]) / 1e3 if k in stats else None}
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):
        e[c] = avg(pm, c)
    f, w = avg(pf, "FETCH_SIZE"), avg(pw, "WRITE_SIZE")
    e["hbm_read_bytes"] = f * 2 * 1024 if f is not None else None
    e["hbm_write_bytes"] = w * 1024 if w is not None else None
    if "k_matmul_mfma_limbs" in short:
        cycles = e["GRBM_GUI_ACTIVE"] / 8
        e["shader_clock_GHz"] = cycles / (e["avg_duration_us"] * 1e3)
        e["mfma_busy_fraction"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cycles)    # 1024 SIMDs
        e["mfma_count"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 16
        e["dma_request_bytes"] = 1024 * 16 * 32 * 65536                              # tiles x limb pairs x k-steps x (A + B chunk)
    res["kernels"][short] = e
json.dump(res, open(os.path.join(ROOT, "profiles", "r3_pmc_matmul_limbs.json"), "w"), indent=1)
k = res["kernels"]["k_matmul_mfma_limbs<8>"]
print(json.dumps({x: k[x] for x in ("avg_duration_us", "mfma_busy_fraction", "shader_clock_GHz", "hbm_read_bytes", "hbm_write_bytes")}))
