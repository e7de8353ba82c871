// The row-tile read of the matrix-core mat-vec (every wave reads 64 bytes of each of 16 rows per instruction) against
// the ROW STRIDE: tools/time_matvec_stride.py shows the kernel at 5.1-5.3 TB/s for strides that are powers of two and
// at 3.5-4.3 TB/s for others (3 * 2^17, 384000, ...).  Is that the memory system's answer to the access pattern (then
// this bare read shows it too), and does starting every workgroup at a different column (skew) or a different
// workgroup height change it?
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_rowtile_stride tools/ubench_rowtile_stride.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void k_read(const uint8_t* A, size_t rowbytes, uint32_t ksplit, uint32_t skew, uint4* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t rowblock = blockIdx.x / ksplit, part = blockIdx.x % ksplit;
    const size_t row = (size_t)rowblock * (WAVES * 16) + wave * 16 + (lane & 15);
    const size_t steps = rowbytes / 64 / ksplit / DEPTH * DEPTH;
    const uint8_t* p = A + row * rowbytes + (size_t)part * steps * 64 + 16 * (lane >> 4);
    size_t s0 = skew ? ((size_t)rowblock * 977 * DEPTH) % steps : 0;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t it = 0; it < steps; it += DEPTH) {
        size_t s = it + s0;
        if (s >= steps) s -= steps;
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) v[d] = *reinterpret_cast<const uint4*>(p + (s + d) * 64);
#pragma unroll
        for (int d = 0; d < DEPTH; d++) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
    }
    out[(size_t)blockIdx.x * (WAVES * 64) + threadIdx.x] = acc;
}

template <int WAVES, int DEPTH>
void run(const uint8_t* A, size_t rows, size_t rowbytes, uint32_t ksplit, uint32_t skew, uint4* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid((unsigned)(rows / (WAVES * 16) * ksplit));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_read<WAVES, DEPTH>), grid, dim3(WAVES * 64), 0, 0, A, rowbytes, ksplit, skew, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const size_t steps = rowbytes / 64 / ksplit / DEPTH * DEPTH;
    printf("{\"row_stride\": %zu, \"rows\": %zu, \"waves\": %d, \"ksplit\": %u, \"skew\": %u, \"workgroups\": %u, \"ms\": %.3f, \"TBps\": %.3f}\n",
           rowbytes, rows, WAVES, ksplit, skew, grid.x, best, (double)rows * steps * 64 * ksplit / best / 1e9);
    fflush(stdout);
}

int main() {
    const size_t total = (size_t)8 << 30;
    uint8_t* A; uint4* out;
    CK(hipMalloc(&A, total + (1 << 20)));
    CK(hipMemset(A, 1, total));
    CK(hipMalloc(&out, (size_t)(1 << 22) * sizeof(uint4)));
    const size_t strides[] = {524288, 393216, 384000, 262144, 196608, 786432, 524256, 524352, 1048576, 1048576 + 4096, 393216 + 4096};
    for (size_t rb : strides) {
        const size_t rows = total / rb / 256 * 256;
        const uint32_t ks = rows / 128 >= 256 ? 1 : 2;
        run<8, 8>(A, rows, rb, ks, 0, out);
        run<8, 8>(A, rows, rb, ks, 1, out);
        run<4, 8>(A, rows, rb, ks, 0, out);
        run<16, 8>(A, rows, rb, 2 * ks, 0, out);
    }
    return 0;
}
