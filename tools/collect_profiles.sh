#!/bin/bash
# Round-end measurement pass on the GPU box: bench line, rocprofv3 kernel trace of the same
# command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) the traffic figure comes from.
# Each GPU step has its own time limit; any failure ends the script.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/collect
mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 bench.py --steps 20 --warmup 3 --no-extra --no-cpu > $OUT/trace.log 2>&1
echo "trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/pmc_write.log 2>&1
echo "write done"
find $OUT -name "*.csv" | head -20
