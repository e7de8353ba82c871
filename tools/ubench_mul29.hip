// Dependent Montgomery-multiply chain ("Mul assign" shape), saturated 32-bit limbs with
// carry adds (the library's mont_mul) against unsaturated 29/28-bit limbs with 64-bit
// column accumulators (fp_unsat.cuh).  Checks that both give the same bits, then times them.
// Build: hipcc --offload-arch=gfx950 -O3 -I crypto_primitives_amd/csrc -o ubench_mul29 tools/ubench_mul29.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fp_unsat.cuh"

using namespace fpgpu;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int N>
__global__ __launch_bounds__(256) void k_chain_sat(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n,
                                                   uint32_t iters, const FpParams<N> P) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t acc[N], y[N], r[N];
    load_elem<N>(acc, a, i);
    load_elem<N>(y, b, i);
    for (uint32_t it = 0; it < iters; it++) {
        mont_mul<N>(r, acc, y, P);
#pragma unroll
        for (int l = 0; l < N; l++) acc[l] = r[l];
    }
    store_elem<N>(out, i, acc);
}

template <int N, int B, bool LAZY>
__global__ __launch_bounds__(256) void k_chain_unsat(const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n,
                                                     uint32_t iters, const UParams<UCfg<N, B>::L> U) {
    constexpr int L = UCfg<N, B>::L;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t w[N], x[L], y[L], r[L];
    load_elem<N>(w, a, i);
    to_limbs<N, B, 0>(x, w);
    load_elem<N>(w, b, i);
    to_limbs<N, B, UCfg<N, B>::SH>(y, w);
    for (uint32_t it = 0; it < iters; it++) {
        umont_mul<B, L>(r, x, y, U);
        if constexpr (!LAZY) ucond_sub<B, L>(r, U);   // LAZY: values stay in [0, 2p), legal when 4p <= 2^(32N)
#pragma unroll
        for (int l = 0; l < L; l++) x[l] = r[l];
    }
    if constexpr (LAZY) ucond_sub<B, L>(x, U);
    from_limbs<N, B>(w, x);
    store_elem<N>(out, i, w);
}

static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int N, int B>
static void run(const char* name, const uint64_t* mod64, int top_bits_clear, size_t n, uint32_t iters) {
    FpParams<N> P;
    memset(&P, 0, sizeof(P));
    for (int i = 0; i < N; i++) P.m[i] = (uint32_t)(mod64[i / 2] >> (32 * (i & 1)));
    uint32_t inv = 1;
    for (int i = 0; i < 5; i++) inv *= 2u - P.m[0] * inv;
    P.n0inv = 0u - inv;
    UParams<UCfg<N, B>::L> U;
    make_uparams<N, B>(U, P.m);
    std::vector<uint32_t> ha(n * N), hb(n * N), o1(n * N), o2(n * N);
    uint64_t s = 42;
    for (size_t i = 0; i < n * N; i++) { ha[i] = (uint32_t)sm(s); hb[i] = (uint32_t)sm(s); }
    for (size_t i = 0; i < n; i++) {  // below p: clear enough top bits
        ha[i * N + N - 1] >>= top_bits_clear;
        hb[i * N + N - 1] >>= top_bits_clear;
    }
    uint32_t *a, *b, *out;
    CK(hipMalloc(&a, n * N * 4)); CK(hipMalloc(&b, n * N * 4)); CK(hipMalloc(&out, n * N * 4));
    CK(hipMemcpy(a, ha.data(), n * N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), n * N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid((unsigned)((n + 255) / 256));
    float ms_sat = 0, ms_un = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chain_sat<N>), grid, dim3(256), 0, 0, a, b, out, n, iters, P);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_sat, e0, e1));
    }
    CK(hipMemcpy(o1.data(), out, n * N * 4, hipMemcpyDeviceToHost));
    CK(hipMemset(out, 0, n * N * 4));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chain_unsat<N, B, false>), grid, dim3(256), 0, 0, a, b, out, n, iters, U);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_un, e0, e1));
    }
    CK(hipMemcpy(o2.data(), out, n * N * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n * N; i++) bad += o1[i] != o2[i];
    CK(hipMemset(out, 0, n * N * 4));
    float ms_lazy = 0;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_chain_unsat<N, B, true>), grid, dim3(256), 0, 0, a, b, out, n, iters, U);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_lazy, e0, e1));
    }
    CK(hipMemcpy(o2.data(), out, n * N * 4, hipMemcpyDeviceToHost));
    size_t bad_lazy = 0;
    for (size_t i = 0; i < n * N; i++) bad_lazy += o1[i] != o2[i];
    printf("{\"field\": \"%s\", \"B\": %d, \"lazy_ms\": %.3f, \"lazy_Gmul_s\": %.1f, \"lazy_mismatching_words\": %zu}\n", name, B, ms_lazy,
           n * (double)iters / ms_lazy / 1e6, bad_lazy);
    printf("{\"field\": \"%s\", \"limbs32\": %d, \"B\": %d, \"L\": %d, \"n\": %zu, \"iters\": %u, \"sat_ms\": %.3f, \"unsat_ms\": %.3f, "
           "\"sat_Gmul_s\": %.1f, \"unsat_Gmul_s\": %.1f, \"mismatching_words\": %zu}\n",
           name, N, B, UCfg<N, B>::L, n, iters, ms_sat, ms_un, n * (double)iters / ms_sat / 1e6, n * (double)iters / ms_un / 1e6, bad);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(out));
}

int main() {
    const uint64_t bn254[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    const uint64_t bls[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                             0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    run<8, 29>("bn254_fr", bn254, 3, 1 << 22, 256);
    run<12, 28>("bls12_381_fq", bls, 4, 1 << 21, 128);
    run<12, 29>("bls12_381_fq", bls, 4, 1 << 21, 128);
    return 0;
}
