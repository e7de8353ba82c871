#!/bin/bash
# Host-side sanitizer builds of libfpgpu (device code is NOT instrumented: GPU sanitizers are not available on this pool)
# and of the native callers, into build/<asan|tsan>/.  usage: tools/build_san.sh address|thread
# The kernels' translation units are taken as built (their host side is launch stubs); the API, handle and staging
# layers -- where the host-side pointer arithmetic and the locking live -- are instrumented.  Run on a GPU box:
#   ASAN_OPTIONS=protect_shadow_gap=0:detect_leaks=0 build/asan/stress_host_layers 16 2
#   (thread: builds, but under ThreadSanitizer the HIP runtime of this image finds no device -- nothing runs)
set -e
KIND=${1:-address}
case $KIND in address) OUT=build/asan;; thread) OUT=build/tsan;; *) echo "address or thread"; exit 2;; esac
cd "$(dirname "$0")/.."
make -s -C crypto_primitives_amd -j8 libfpgpu.so
mkdir -p $OUT
SAN="-Xarch_host -fsanitize=$KIND -Xarch_host -fno-omit-frame-pointer"
for tu in fpgpu_api fpgpu_vec fpgpu_stage; do
  hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $SAN -c -o $OUT/$tu.o crypto_primitives_amd/csrc/$tu.hip &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$KIND -fno-gpu-sanitize -shared-libsan -o $OUT/libfpgpu.so \
  $OUT/fpgpu_api.o $OUT/fpgpu_vec.o $OUT/fpgpu_stage.o $(ls build/obj/fp_inst_*.o | grep -E 'fp_inst_[0-9]+\.o$') -ldl -lpthread   # (not the variant objects of tools/build_variant.sh)
CLANG=/opt/rocm/lib/llvm/bin/clang
RT=$($CLANG -print-resource-dir)/lib/linux
LINK="-L$OUT -lfpgpu -Wl,-rpath,\$ORIGIN -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RT -Wl,--allow-shlib-undefined"
$CLANG++ -O1 -g -std=c++17 -fsanitize=$KIND -shared-libsan -fno-omit-frame-pointer -o $OUT/test_monty_field tests/cpp/test_monty_field.cpp $LINK
$CLANG++ -O1 -g -std=c++17 -fsanitize=$KIND -shared-libsan -fno-omit-frame-pointer -pthread -o $OUT/stress_host_layers tests/cpp/stress_host_layers.cpp $LINK
$CLANG -O1 -g -std=c99 -fsanitize=$KIND -shared-libsan -fno-omit-frame-pointer -o $OUT/abi_c99 tests/c/abi_c99.c $LINK
rm -f $OUT/*.o
ls -la $OUT
