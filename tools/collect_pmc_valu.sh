#!/bin/bash
# VALU-rate counters for the VALU-bound kernels (mul chain, batch inversion) and HBM traffic of the
# add kernel; each PMC pass is a run of its own.  Output: gpurun_out/pmcv/*, summarised by
# tools/pmc_valu_summary.py into profiles/.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmcv
mkdir -p $OUT
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/valu -o v -- python3 bench.py --steps 2 --warmup 1 --no-cpu --workload add > $OUT/valu.log 2>&1
echo "valu done"
timeout -s KILL 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 bench.py --steps 2 --warmup 1 --no-cpu --workload add > $OUT/trace.log 2>&1
echo "trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_add -o f -- python3 bench.py --workload add --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/fetch_add.log 2>&1
echo "fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_add -o w -- python3 bench.py --workload add --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/write_add.log 2>&1
echo "write done"
