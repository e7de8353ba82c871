#!/bin/bash
# Builds build/libfpgpu_<name>.so with extra -D flags on the single-limb translation unit (kernel
# tuning experiments; load it with FPGPU_LIB=build/libfpgpu_<name>.so).  Usage: build_variant.sh name "-DX -DY"
set -e
cd "$(dirname "$0")/../crypto_primitives_amd"
make -s -j8
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -DFPGPU_N=2 -c -o ../build/obj/fp_inst_2_$1.o csrc/fp_inst.hip
OBJS=$(ls ../build/obj/fpgpu_api.o ../build/obj/fp_inst_{4,6,8,10,12,14,16}.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../build/libfpgpu_$1.so $OBJS ../build/obj/fp_inst_2_$1.o
echo built build/libfpgpu_$1.so
