#!/bin/bash
# Per-kernel times of the batch-inversion kernels for tuning builds (tools/build_variant.sh).
# Usage (GPU box): tools/exp_binv_variants.sh name1 name2 ...   -> gpurun_out/expbinv_<name>/
set -e
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = base ]; then unset FPGPU_LIB; else export FPGPU_LIB=$PWD/build/libfpgpu_$v.so; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/expbinv_$v -o t -- python3 bench.py --workload batch_inv --steps 10 --warmup 2 --no-extra --no-cpu > gpurun_out/expbinv_$v.log 2>&1
  echo "== $v"
  python3 - "$v" <<'PY'
import csv, sys
for row in csv.reader(open('gpurun_out/expbinv_%s/t_kernel_stats.csv' % sys.argv[1])):
    if 'binv' in row[0]: print('  ', row[0][12:40].ljust(30), row[1], round(float(row[3]) / 1e3, 1), 'us')
PY
done
