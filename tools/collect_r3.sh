#!/bin/bash
# Round-3 measurement pass on the GPU box: bench line, rocprofv3 kernel traces of the same commands,
# and the PMC passes (each counter set a run of its own, never combined with a trace domain).
# Every GPU step has its own time limit; the first failure ends the script.
# Output: gpurun_out/r3/*; summarised HERE (where git is) by tools/summarise_r3.py into profiles/.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/r3
rm -rf $OUT
mkdir -p $OUT
B="python3 bench.py"
timeout -k 10 500 $B > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_mul -o t -- $B --steps 20 --warmup 3 --no-extra --no-cpu > $OUT/trace_mul.log 2>&1
echo "trace (headline) done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_all -o t -- $B --steps 5 --warmup 1 --no-cpu --no-sharded > $OUT/trace_all.log 2>&1
echo "trace (all workloads) done"
for w in mul add matvec matmul batch_inv; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${w}_$c -o p -- $B --workload $w --steps 3 --warmup 1 --no-extra --no-cpu > $OUT/pmc_${w}_$c.log 2>&1
    echo "pmc $w $c done"
  done
done
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_valu -o p -- $B --steps 2 --warmup 1 --no-cpu --no-sharded --workload add > $OUT/pmc_valu.log 2>&1
echo "pmc valu done"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_valu -o t -- $B --steps 2 --warmup 1 --no-cpu --no-sharded --workload add > $OUT/trace_valu.log 2>&1
echo "trace for valu durations done"
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o p -- $B --workload matmul --steps 2 --warmup 1 --no-extra --no-cpu > $OUT/pmc_mfma.log 2>&1
echo "pmc mfma done"
