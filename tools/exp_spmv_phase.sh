#!/bin/bash
# tools/exp_spmv_phase.py under the L2 hit / miss counters, one configuration per run (counters never with a trace domain).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/spmv_phase
rm -rf $OUT
mkdir -p $OUT
export FPGPU_SPMV_LOG2G=0
for cfg in "16 2 s" "32 2 s" "8 2 s" "16 4 s" "16 2 random"; do
  set -- $cfg
  export FPGPU_SPMV_UNROLL=$2
  tag="n$1_u$2_$3"
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/$tag -o p -- python3 tools/exp_spmv_phase.py $1 $3 > $OUT/$tag.log 2>&1
  grep '"nblk"' $OUT/$tag.log
done
