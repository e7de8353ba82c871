#!/bin/bash
# PMC passes for the single-limb mat-mul (configs[4]): HBM fetch / write bytes and L2 hit/miss per launch.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_mm
mkdir -p $OUT
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 bench.py --workload matmul --steps 2 --warmup 1 --no-extra --no-cpu > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w -- python3 bench.py --workload matmul --steps 2 --warmup 1 --no-extra --no-cpu > $OUT/write.log 2>&1
echo "write done"
timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/hit -o h -- python3 bench.py --workload matmul --steps 2 --warmup 1 --no-extra --no-cpu > $OUT/hit.log 2>&1
echo "hit/miss done"
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o m -- python3 bench.py --workload matmul --steps 2 --warmup 1 --no-extra --no-cpu > $OUT/mfma.log 2>&1
echo "mfma busy done"
python3 - <<'PY'
import sqlite3, glob
for p in sorted(glob.glob("gpurun_out/pmc_mm/*/*_results.db")):
    d = sqlite3.connect(p)
    cur = d.execute("select * from counters_collection limit 1"); cols = [c[0] for c in cur.description]
    ki = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
    ci, vi, di = cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    per = {}
    for r in d.execute("select * from counters_collection"):
        if "matmul_mfma" in r[ki]:
            per.setdefault(r[ci], {}).setdefault(r[di], 0.0)
            per[r[ci]][r[di]] += r[vi]
    for c, v in per.items():
        vals = list(v.values()); print(p, c, len(vals), sum(vals) / len(vals))
    try:
        rows = d.execute("select name, duration from kernels where name like '%matmul_mfma%'").fetchall()
        if rows: print(p, "duration_ns", len(rows), sum(r[1] for r in rows) / len(rows))
    except Exception as e:
        print(p, "no kernel durations:", e)
PY
