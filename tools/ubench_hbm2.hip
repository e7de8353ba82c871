// HBM streaming microbenchmark #2: read-only / copy / 2R+1W with and without
// nontemporal hints, several workgroup sizes.  AoS 32 B per lane throughout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0: read-only (xor-reduce, one store per thread at the end)   1: copy a->o   2: a,b -> o
template <int MODE, bool NTL, bool NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ o, size_t nelem) {
  size_t stride = (size_t)gridDim.x * BLOCK;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)BLOCK + threadIdx.x; i < nelem; i += stride) {
    u32x4 x0 = ld<NTL>(a + 2 * i), x1 = ld<NTL>(a + 2 * i + 1);
    if constexpr (MODE == 0) { acc ^= x0; acc ^= x1; }
    else if constexpr (MODE == 1) { st<NTS>(o + 2 * i, x0); st<NTS>(o + 2 * i + 1, x1); }
    else {
      u32x4 y0 = ld<NTL>(b + 2 * i), y1 = ld<NTL>(b + 2 * i + 1);
      st<NTS>(o + 2 * i, x0 ^ y0); st<NTS>(o + 2 * i + 1, x1 ^ y1);
    }
  }
  if constexpr (MODE == 0) o[blockIdx.x * (size_t)BLOCK + threadIdx.x] = acc;
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

template <int MODE, bool NTL, bool NTS, int BLOCK>
void run(const char* name, int grid, u32x4* a, u32x4* b, u32x4* o, size_t nelem, double bytes) {
  float ms = timeit([&] { hipLaunchKernelGGL((k<MODE, NTL, NTS, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, a, b, o, nelem); }, 5);
  printf("{\"hbm2\":\"%s\",\"block\":%d,\"grid\":%d,\"ntl\":%d,\"nts\":%d,\"ms\":%.3f,\"GBps\":%.0f}\n", name, BLOCK, grid, (int)NTL, (int)NTS, ms, bytes / ms * 1e-6);
}

int main() {
  size_t bytes = (size_t)4 << 30;  // per array
  u32x4 *a, *b, *o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(o, 0, bytes));
  size_t nelem = bytes / 32;
  for (int grid : {2048, 8192, 16384, 65536}) {
    run<0, false, false, 256>("read", grid, a, b, o, nelem, 1.0 * bytes);
    run<0, true, false, 256>("read", grid, a, b, o, nelem, 1.0 * bytes);
    run<1, false, false, 256>("copy", grid, a, b, o, nelem, 2.0 * bytes);
    run<1, true, true, 256>("copy", grid, a, b, o, nelem, 2.0 * bytes);
    run<2, false, false, 256>("2r1w", grid, a, b, o, nelem, 3.0 * bytes);
    run<2, true, false, 256>("2r1w", grid, a, b, o, nelem, 3.0 * bytes);
    run<2, false, true, 256>("2r1w", grid, a, b, o, nelem, 3.0 * bytes);
    run<2, true, true, 256>("2r1w", grid, a, b, o, nelem, 3.0 * bytes);
    run<2, false, false, 512>("2r1w", grid / 2, a, b, o, nelem, 3.0 * bytes);
    run<2, true, true, 512>("2r1w", grid / 2, a, b, o, nelem, 3.0 * bytes);
    run<2, false, false, 1024>("2r1w", grid / 4, a, b, o, nelem, 3.0 * bytes);
    run<2, true, true, 1024>("2r1w", grid / 4, a, b, o, nelem, 3.0 * bytes);
  }
  return 0;
}
