#!/bin/bash
# Round-end measurement pass on the GPU box: bench line, rocprofv3 kernel traces of the same
# commands, and the PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) the traffic figures come from.
# Each GPU step has its own time limit; any failure ends the script.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/collect
mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu > $OUT/trace.log 2>&1
echo "trace (headline) done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_all -o t -- python3 bench.py --steps 5 --warmup 1 --no-cpu > $OUT/trace_all.log 2>&1
echo "trace (all workloads) done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/pmc_write.log 2>&1
echo "write done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_mv -o f -- python3 bench.py --workload matvec --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/pmc_fetch_mv.log 2>&1
echo "fetch (matvec) done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_mv -o w -- python3 bench.py --workload matvec --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/pmc_write_mv.log 2>&1
echo "write (matvec) done"
