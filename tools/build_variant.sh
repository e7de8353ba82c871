#!/bin/bash
# Builds build/libfpgpu_<name>.so with extra -D flags (kernel tuning experiments; load it with
# FPGPU_LIB=build/libfpgpu_<name>.so).  Usage: build_variant.sh name "-DX -DY" [limb counts to rebuild, default "2"]
# The flags go to the listed limb-count translation units and to the API unit (which shares the plans).
set -e
cd "$(dirname "$0")/../crypto_primitives_amd"
make -s -j8
NAME=$1; FLAGS=$2; LIMBS=${3:-2}
HF="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -include ../tools/variants/matmul_timing.hpp"
OBJS=""
for n in 2 4 6 8 10 12 14 16; do
  if [[ " $LIMBS " == *" $n "* ]]; then
    hipcc $HF $FLAGS -DFPGPU_N=$n -c -o ../build/obj/fp_inst_${n}_$NAME.o csrc/fp_inst.hip &
    OBJS="$OBJS ../build/obj/fp_inst_${n}_$NAME.o"
  else
    OBJS="$OBJS ../build/obj/fp_inst_$n.o"
  fi
done
hipcc $HF $FLAGS -c -o ../build/obj/fpgpu_api_$NAME.o csrc/fpgpu_api.hip &
hipcc $HF $FLAGS -c -o ../build/obj/fpgpu_vec_$NAME.o csrc/fpgpu_vec.hip &
hipcc $HF $FLAGS -c -o ../build/obj/fpgpu_stage_$NAME.o csrc/fpgpu_stage.hip &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/libfpgpu_$NAME.so ../build/obj/fpgpu_api_$NAME.o ../build/obj/fpgpu_vec_$NAME.o ../build/obj/fpgpu_stage_$NAME.o $OBJS -ldl -lpthread
echo built build/libfpgpu_$NAME.so
