set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/binv1
timeout -k 10 120 python tools/exp_binv_two_streams.py > gpurun_out/binv1/two_streams.log 2>&1
for c in 16 32 64 128; do FPGPU_INV_C=$c timeout -k 10 120 python tools/exp_binv_two_streams.py 2>&1 | head -2 | tail -1 | sed "s/^/C=$c /" >> gpurun_out/binv1/two_streams.log; done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/binv1/trace -o t -- python3 bench.py --workload batch_inv --steps 10 --warmup 2 --no-extra --no-cpu > gpurun_out/binv1/trace.log 2>&1
cat gpurun_out/binv1/two_streams.log
python3 - <<'PY'
import csv,glob
for p in glob.glob("gpurun_out/binv1/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "binv" in r["Name"]: print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
