#!/bin/bash
# Longer, freshly seeded run of tests/test_gpu_fuzz.py (FUZZ_EXAMPLES): random programs and random matrix products on
# random odd moduli of every width.  A heartbeat file shows progress while pytest is silent.
set -u
N=${1:-1000}
OUT=gpurun_out/soak
mkdir -p $OUT
( while true; do date >> $OUT/heartbeat; sleep 30; done ) &
HB=$!
FUZZ_EXAMPLES=$N timeout -k 10 1000 python3 -u -m pytest tests/test_gpu_fuzz.py -x -q -m gpu --timeout 900 --timeout-method thread -p no:cacheprovider > $OUT/fuzz_$N.log 2>&1
rc=$?
kill $HB
tail -5 $OUT/fuzz_$N.log
exit $rc
