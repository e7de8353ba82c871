// Issue-rate microbenchmark for the int8 MFMAs the byte-sliced mat-mul is built from:
// one wave per SIMD, 240 accumulator registers, operands resident in registers (no LDS).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_mfma_i8 tools/ubench_mfma_i8.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const v4i* in, int* out, int iters) {
    v4i a[16], b[2];
    for (int i = 0; i < 16; i++) a[i] = in[threadIdx.x + 256 * i];
    b[0] = in[threadIdx.x + 4096];
    b[1] = in[threadIdx.x + 4352];
    if constexpr (SHAPE == 16) {
        v4i acc[60];
        for (int i = 0; i < 60; i++) acc[i] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 60; i++)
                    acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(i + r) & 15], b[i & 1], acc[i], 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 60; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    } else {
        v16i acc[15];
        for (int i = 0; i < 15; i++)
            for (int e = 0; e < 16; e++) acc[i][e] = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 15; i++)
                    acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[(i + r) & 15], b[i & 1], acc[i], 0, 0, 0);
        }
        int s = 0;
        for (int i = 0; i < 15; i++)
            for (int e = 0; e < 16; e++) s += acc[i][e];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
}

template <int SHAPE>
void run(const char* name, const v4i* in, int* out, int ncu) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<SHAPE>, dim3(ncu), dim3(256), 0, 0, in, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<SHAPE>, dim3(ncu), dim3(256), 0, 0, in, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (SHAPE == 16 ? 240.0 : 120.0) * iters;  // MFMAs per wave
    const double macs = (SHAPE == 16 ? 16384.0 : 32768.0);
    printf("{\"mfma\": \"%s\", \"ms\": %.3f, \"ns_per_mfma_per_simd\": %.3f, \"cycles_at_2.4GHz\": %.2f, \"chip_int8_Tmac_s\": %.1f}\n", name, ms,
           ms * 1e6 / n, ms * 1e-3 * 2.4e9 / n, n * macs * ncu * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    v4i* in; int* out;
    CK(hipMalloc(&in, 4608 * sizeof(v4i)));
    CK(hipMemset(in, 0x11, 4608 * sizeof(v4i)));
    CK(hipMalloc(&out, p.multiProcessorCount * 256 * 4));
    for (int rep = 0; rep < 2; rep++) {
        run<16>("i32_16x16x64_i8", in, out, p.multiProcessorCount);
        run<32>("i32_32x32x32_i8", in, out, p.multiProcessorCount);
    }
    return 0;
}
