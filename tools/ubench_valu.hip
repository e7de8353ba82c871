// Register-only VALU microbenchmark for gfx950: measures the wave64 issue cost of
// the integer instructions the Montgomery kernels are built from, so the
// "VALU-integer roofline" in DESIGN.md is a measured number, not a datasheet guess
// (SURVEY.md section 7 "hard parts", section 8d "Peak rates").
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;

// Each test body issues 16 instructions of the kind under test per loop iteration.
#define R4(x) x x x x
#define R16(x) R4(R4(x))

template <int KIND>
__global__ __launch_bounds__(256) void k_bench(uint32_t* out, uint32_t seed, unsigned long long* cycles) {
  uint32_t a = seed + threadIdx.x, b = seed * 7u + threadIdx.x * 3u;
  uint64_t c0 = a, c1 = b, c2 = a ^ b, c3 = a + b, c4 = 5, c5 = 6, c6 = 7, c7 = 8;
  uint32_t d0 = a, d1 = b, d2 = 3, d3 = 4, d4 = 5, d5 = 6, d6 = 7, d7 = 8;
  double f0 = a, f1 = b, f2 = 1.5, f3 = 2.5, f4 = 3.5, f5 = 4.5, f6 = 5.5, f7 = 6.5, fa = 1.000001, fb = 0.5;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
    if constexpr (KIND == 0) {  // v_mad_u64_u32, 8 independent accumulators
      asm volatile(
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
          "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
          "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\t"
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\t"
          "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
          "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
          : "v"(a), "v"(b) : "vcc");
    } else if constexpr (KIND == 1) {  // v_mad_u64_u32 dependent chain (one accumulator)
      R4(asm volatile(
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
          "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
          : "+v"(c0) : "v"(a), "v"(b) : "vcc");)
    } else if constexpr (KIND == 2) {  // mad + addc (the product-scanning step), dependent
      R4(asm volatile(
          "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
          "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
          : "+v"(c0), "+v"(d0) : "v"(a), "v"(b) : "vcc");)
      // counts as 8 mads + 8 addc per R4 -> 16 instr, 8 "steps"
    } else if constexpr (KIND == 14) {  // mad ; s_nop 1 ; addc  (hazard-padded product-scanning step)
      R4(asm volatile(
          "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
          "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
          : "+v"(c0), "+v"(d0) : "v"(a), "v"(b) : "vcc");)
    } else if constexpr (KIND == 15) {  // software-pipelined carries: 3 rotating SGPR pairs, no nops
      unsigned long long s0, s1, s2;
      R4(asm volatile(
          "v_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_mad_u64_u32 %0, %3, %5, %6, %0\n\t"
          "v_mad_u64_u32 %0, %4, %5, %6, %0\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\t"
          "v_mad_u64_u32 %0, %2, %5, %6, %0\n\tv_addc_co_u32 %1, %3, 0, %1, %3\n\t"
          "v_addc_co_u32 %1, %4, 0, %1, %4\n\tv_addc_co_u32 %1, %2, 0, %1, %2\n\t"
          : "+v"(c0), "+v"(d0), "=&s"(s0), "=&s"(s1), "=&s"(s2) : "v"(a), "v"(b));)
      // 8 instr per R4 unit x2 ... counts as 16 instr per iteration? (4 mad + 4 addc) x 4 = 32
    } else if constexpr (KIND == 3) {  // v_mul_lo_u32 independent
      R4(asm volatile(
          "v_mul_lo_u32 %0, %0, %4\n\tv_mul_lo_u32 %1, %1, %4\n\tv_mul_lo_u32 %2, %2, %4\n\tv_mul_lo_u32 %3, %3, %4\n\t"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));)
    } else if constexpr (KIND == 4) {  // v_mul_hi_u32 independent
      R4(asm volatile(
          "v_mul_hi_u32 %0, %0, %4\n\tv_mul_hi_u32 %1, %1, %4\n\tv_mul_hi_u32 %2, %2, %4\n\tv_mul_hi_u32 %3, %3, %4\n\t"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));)
    } else if constexpr (KIND == 5) {  // add_co / addc chain
      R4(asm volatile(
          "v_add_co_u32 %0, vcc, %0, %4\n\tv_addc_co_u32 %1, vcc, %1, %4, vcc\n\t"
          "v_addc_co_u32 %2, vcc, %2, %4, vcc\n\tv_addc_co_u32 %3, vcc, %3, %4, vcc\n\t"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a) : "vcc");)
    } else if constexpr (KIND == 6) {  // v_lshl_add_u64 independent
      R4(asm volatile(
          "v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\t"
          "v_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c4));)
    } else if constexpr (KIND == 7) {  // v_mad_u32_u24 independent
      R4(asm volatile(
          "v_mad_u32_u24 %0, %0, %4, %5\n\tv_mad_u32_u24 %1, %1, %4, %5\n\t"
          "v_mad_u32_u24 %2, %2, %4, %5\n\tv_mad_u32_u24 %3, %3, %4, %5\n\t"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a), "v"(b));)
    } else if constexpr (KIND == 8) {  // v_fma_f64 independent (8 accumulators)
      R4(asm volatile(
          "v_fma_f64 %0, %4, %5, %0\n\tv_fma_f64 %1, %4, %5, %1\n\t"
          "v_fma_f64 %2, %4, %5, %2\n\tv_fma_f64 %3, %4, %5, %3\n\t"
          : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fa), "v"(fb));)
    } else if constexpr (KIND == 9) {  // v_add_u32 independent (baseline full-rate op)
      R4(asm volatile(
          "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));)
    } else if constexpr (KIND == 10) {  // v_mul_u32_u24 + v_mul_hi_u32_u24
      R4(asm volatile(
          "v_mul_u32_u24 %0, %0, %4\n\tv_mul_hi_u32_u24 %1, %1, %4\n\t"
          "v_mul_u32_u24 %2, %2, %4\n\tv_mul_hi_u32_u24 %3, %3, %4\n\t"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));)
    } else if constexpr (KIND == 11) {  // v_mad_u64_u32 with an SGPR multiplicand
      R4(asm volatile(
          "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
          "v_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "s"(seed) : "vcc");)
    } else if constexpr (KIND == 12) {  // v_mov_b32
      R4(asm volatile(
          "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4\n\t"
          : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));)
    } else if constexpr (KIND == 13) {  // v_mad_u64_u32 addend = inline 0 (pure 32x32->64 multiply)
      R4(asm volatile(
          "v_mad_u64_u32 %0, vcc, %4, %5, 0\n\tv_mad_u64_u32 %1, vcc, %4, %5, 0\n\t"
          "v_mad_u64_u32 %2, vcc, %4, %5, 0\n\tv_mad_u64_u32 %3, vcc, %4, %5, 0\n\t"
          : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b) : "vcc");)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint64_t s = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7;
  uint32_t r = (uint32_t)s ^ (uint32_t)(s >> 32) ^ d0 ^ d1 ^ d2 ^ d3 ^ d4 ^ d5 ^ d6 ^ d7;
  double fs = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
  r ^= (uint32_t)__double_as_longlong(fs);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int KIND>
void run(const char* name, int blocks_per_cu, int ncu, uint32_t* dout, unsigned long long* dcyc) {
  int blocks = blocks_per_cu * ncu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(256), 0, 0, dout, 12345u, dcyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(256), 0, 0, dout, 12345u, dcyc);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0; CK(hipMemcpy(&cyc, dcyc, sizeof cyc, hipMemcpyDeviceToHost));
  double inst = (double)blocks * 4 /*waves*/ * ITERS * 16.0;  // wave-instructions
  double lane_ops = inst * 64.0;
  // s_memtime ticks at 100 MHz reference on gfx9? report raw.
  printf("{\"ubench\":\"%s\",\"blocks_per_cu\":%d,\"ms\":%.4f,\"Gwaveinst_per_s\":%.2f,\"Tlaneops_per_s\":%.3f,"
         "\"cyc_per_waveinst_per_simd_at_2.4GHz\":%.2f,\"memtime_ticks_wave0\":%llu}\n",
         name, blocks_per_cu, ms, inst / ms * 1e-6, lane_ops / ms * 1e-9,
         (ms * 1e-3 * 2.4e9) / (inst / (ncu * 4.0)), cyc);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int ncu = p.multiProcessorCount;
  printf("{\"device\":\"%s\",\"cus\":%d,\"clock_khz\":%d,\"gcn\":\"%s\"}\n", p.name, ncu, p.clockRate, p.gcnArchName);
  uint32_t* dout; unsigned long long* dcyc;
  CK(hipMalloc(&dout, (size_t)ncu * 8 * 256 * 4));
  CK(hipMalloc(&dcyc, 8));
  for (int bpc : {1, 2, 4, 8}) {
    run<0>("mad_u64_u32_indep8", bpc, ncu, dout, dcyc);
    run<1>("mad_u64_u32_dep", bpc, ncu, dout, dcyc);
    run<2>("mad+addc_dep(16inst=8steps)", bpc, ncu, dout, dcyc);
    run<14>("mad+nop1+addc_dep(16inst+8nop)", bpc, ncu, dout, dcyc);
    run<15>("mad/addc_swpipelined_sgpr(32inst!)", bpc, ncu, dout, dcyc);
    run<11>("mad_u64_u32_sgpr_src", bpc, ncu, dout, dcyc);
    run<13>("mad_u64_u32_addend0", bpc, ncu, dout, dcyc);
    run<3>("mul_lo_u32", bpc, ncu, dout, dcyc);
    run<4>("mul_hi_u32", bpc, ncu, dout, dcyc);
    run<5>("add_co/addc_chain", bpc, ncu, dout, dcyc);
    run<6>("lshl_add_u64", bpc, ncu, dout, dcyc);
    run<7>("mad_u32_u24", bpc, ncu, dout, dcyc);
    run<10>("mul_u32_u24+mul_hi_u32_u24", bpc, ncu, dout, dcyc);
    run<8>("fma_f64", bpc, ncu, dout, dcyc);
    run<9>("add_u32", bpc, ncu, dout, dcyc);
    run<12>("mov_b32", bpc, ncu, dout, dcyc);
  }
  return 0;
}
