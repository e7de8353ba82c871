// HBM streaming microbenchmark for gfx950: which element layout / per-lane access
// width reaches the HBM roofline for the 3-array (a, b -> out) elementwise pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_hbm tools/ubench_hbm.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// (1) fully coalesced: lane i touches 16 B at base + 16 i.
__global__ __launch_bounds__(256) void k_x16(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, size_t n16) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += stride) {
    uint4 x = a[i], y = b[i];
    o[i] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
  }
}
// (2) AoS: lane i owns W consecutive uint4 (W = 2: 256-bit element, W = 3: 384-bit).
template <int W>
__global__ __launch_bounds__(256) void k_aos(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, size_t nelem) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nelem; i += stride) {
    uint4 x[W], y[W];
#pragma unroll
    for (int k = 0; k < W; k++) { x[k] = a[i * W + k]; y[k] = b[i * W + k]; }
#pragma unroll
    for (int k = 0; k < W; k++) o[i * W + k] = make_uint4(x[k].x ^ y[k].x, x[k].y ^ y[k].y, x[k].z ^ y[k].z, x[k].w ^ y[k].w);
  }
}
// (3) AoS with 2 elements per thread in flight (more bytes in flight per lane).
template <int W>
__global__ __launch_bounds__(256) void k_aos2(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, size_t nelem) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + stride < nelem; i += 2 * stride) {
    uint4 x[2][W], y[2][W];
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int k = 0; k < W; k++) { x[e][k] = a[(i + e * stride) * W + k]; y[e][k] = b[(i + e * stride) * W + k]; }
#pragma unroll
    for (int e = 0; e < 2; e++)
#pragma unroll
      for (int k = 0; k < W; k++) o[(i + e * stride) * W + k] = make_uint4(x[e][k].x ^ y[e][k].x, x[e][k].y ^ y[e][k].y, x[e][k].z ^ y[e][k].z, x[e][k].w ^ y[e][k].w);
  }
  for (; i < nelem; i += stride)
#pragma unroll
    for (int k = 0; k < W; k++) { uint4 x = a[i * W + k], y = b[i * W + k]; o[i * W + k] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w); }
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  size_t bytes = (size_t)2 << 30;  // per array
  if (argc > 1) bytes = (size_t)atoll(argv[1]) << 20;
  uint4 *a, *b, *o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(o, 0, bytes));
  size_t n16 = bytes / 16;
  double tot = 3.0 * bytes;
  for (int grid : {2048, 4096, 8192, 16384, 1 << 20}) {
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_x16, dim3(grid), dim3(256), 0, 0, a, b, o, n16); }, 5);
    printf("{\"hbm\":\"x16_coalesced\",\"grid\":%d,\"ms\":%.3f,\"GBps\":%.0f}\n", grid, ms, tot / ms * 1e-6);
    ms = timeit([&] { hipLaunchKernelGGL(k_aos<2>, dim3(grid), dim3(256), 0, 0, a, b, o, n16 / 2); }, 5);
    printf("{\"hbm\":\"aos_32B_per_lane\",\"grid\":%d,\"ms\":%.3f,\"GBps\":%.0f}\n", grid, ms, tot / ms * 1e-6);
    ms = timeit([&] { hipLaunchKernelGGL(k_aos2<2>, dim3(grid), dim3(256), 0, 0, a, b, o, n16 / 2); }, 5);
    printf("{\"hbm\":\"aos_32B_x2_inflight\",\"grid\":%d,\"ms\":%.3f,\"GBps\":%.0f}\n", grid, ms, tot / ms * 1e-6);
    ms = timeit([&] { hipLaunchKernelGGL(k_aos<3>, dim3(grid), dim3(256), 0, 0, a, b, o, n16 / 3); }, 5);
    printf("{\"hbm\":\"aos_48B_per_lane\",\"grid\":%d,\"ms\":%.3f,\"GBps\":%.0f}\n", grid, ms, 3.0 * (n16 / 3) * 48 / ms * 1e-6);
  }
  return 0;
}
