#!/bin/bash
# Sweeps the streaming kernels' workgroup cap (FPGPU_GRID_CAP) for one bench workload.
# Usage (GPU box): tools/sweep_grid_cap.sh sqr
for cap in 16384 65536 262144 1048576 4194304; do
  export FPGPU_GRID_CAP=$cap
  timeout -k 10 200 python3 bench.py --workload $1 --steps 10 --warmup 2 --no-extra --no-cpu 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', $cap, round(d['roofline']['kernel_ms'],3), round(d['roofline']['achieved']))"
done
