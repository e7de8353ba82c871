// Can HBM be streamed at full rate when every wave reads 16 matrix rows at once (the operand
// shape an MFMA mat-vec needs: lane (r = l & 15, g = l >> 4) loads 16 bytes of row r at
// 64 t + 16 g) instead of one row linearly?  Rows are `cols * 32` bytes apart (1 MiB at C4).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_rowtile_read tools/ubench_rowtile_read.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// WAVES waves per workgroup, each owns 16 rows; the workgroup walks `ksteps` 64-byte steps of its
// column part; DEPTH loads in flight per lane.
template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void k_read(const uint8_t* A, size_t rowbytes, uint32_t ksplit, uint32_t skew,
                                                      uint4* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t rowblock = blockIdx.x / ksplit, part = blockIdx.x % ksplit;
    const size_t row = (size_t)rowblock * (WAVES * 16) + wave * 16 + (lane & 15);
    const size_t partbytes = rowbytes / ksplit;
    const uint8_t* p = A + row * rowbytes + part * partbytes + 16 * (lane >> 4);
    const size_t steps = partbytes / 64;
    // optional skew: start each workgroup at a different place of its part (wraps around)
    size_t s0 = skew ? ((size_t)rowblock * 977) % steps : 0;
    s0 -= s0 % DEPTH;
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
    out[blockIdx.x * (WAVES * 64) + threadIdx.x] = acc;
}

// reference: the same bytes read linearly (one workgroup per row, lanes contiguous)
__global__ __launch_bounds__(256) void k_linear(const uint8_t* A, size_t rowbytes, uint4* out) {
    const uint8_t* p = A + (size_t)blockIdx.x * rowbytes + threadIdx.x * 32;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t off = 0; off < rowbytes; off += 256 * 32) {
        uint4 a = *reinterpret_cast<const uint4*>(p + off), b = *reinterpret_cast<const uint4*>(p + off + 16);
        acc.x ^= a.x ^ b.x; acc.y ^= a.y ^ b.y; acc.z ^= a.z ^ b.z; acc.w ^= a.w ^ b.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
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
    printf("{\"pattern\": \"rowtile\", \"waves\": %d, \"depth\": %d, \"ksplit\": %u, \"skew\": %u, \"workgroups\": %u, \"ms\": %.3f, \"TBps\": %.3f}\n",
           WAVES, DEPTH, ksplit, skew, grid.x, best, rows * rowbytes / best / 1e9);
}

int main() {
    const size_t rows = 32768, rowbytes = 32768 * 32;
    uint8_t* A; uint4* out;
    CK(hipMalloc(&A, rows * rowbytes));
    CK(hipMemset(A, 1, rows * rowbytes));
    CK(hipMalloc(&out, rows * 256 * sizeof(uint4)));  // k_linear writes one uint4 per thread: rows x 256 of them (the largest launch)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_linear, dim3(rows), dim3(256), 0, 0, A, rowbytes, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"pattern\": \"linear\", \"ms\": %.3f, \"TBps\": %.3f}\n", ms, rows * rowbytes / ms / 1e9);
    }
    for (uint32_t skew = 0; skew < 2; skew++) {
        run<8, 8>(A, rows, rowbytes, 1, skew, out);
        run<8, 8>(A, rows, rowbytes, 2, skew, out);
        run<8, 8>(A, rows, rowbytes, 4, skew, out);
        run<4, 8>(A, rows, rowbytes, 2, skew, out);
        run<4, 16>(A, rows, rowbytes, 2, skew, out);
        run<16, 4>(A, rows, rowbytes, 4, skew, out);
        run<16, 8>(A, rows, rowbytes, 8, skew, out);
    }
    return 0;
}
