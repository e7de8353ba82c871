#!/bin/bash
# PMC passes over the sparse mat-vec of bench.py (tools/time_spmv.py: 2^20 rows x 32 cells over 2^20 columns, BN254).
# One counter set per run, never combined with a trace domain; each run under its own time limit.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3_spmv
rm -rf $OUT
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -o p -- python3 tools/time_spmv.py > $OUT/tcc.log 2>&1
echo "tcc done"
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/tcp -o p -- python3 tools/time_spmv.py > $OUT/tcp.log 2>&1
echo "tcp done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 tools/time_spmv.py > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/time_spmv.py > $OUT/trace.log 2>&1
echo "trace done"
