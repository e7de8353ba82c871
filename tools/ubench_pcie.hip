// Host-buffer transfer options for the fpgpu_* host entry points (PCIe-inclusive path):
// pageable hipMemcpy, pin-in-place (hipHostRegister) + async copy, and a kernel that streams registered
// host memory directly.  Build: hipcc -O3 --offload-arch=gfx950 -o build/ubench_pcie tools/ubench_pcie.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_add(const uint4* a, const uint4* b, uint4* o, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 x = a[i], y = b[i];
        o[i] = make_uint4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    }
}

int main() {
    const size_t bytes = 256u << 20;
    char *a = (char*)aligned_alloc(4096, bytes), *b = (char*)aligned_alloc(4096, bytes), *o = (char*)aligned_alloc(4096, bytes);
    memset(a, 1, bytes); memset(b, 2, bytes); memset(o, 0, bytes);
    void *da, *db, *dout;
    double t = now();
    CK(hipMalloc(&da, bytes)); CK(hipMalloc(&db, bytes)); CK(hipMalloc(&dout, bytes));
    printf("3 x hipMalloc(256 MiB): %.2f ms\n", (now() - t) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
        t = now(); CK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice)); double t1 = now() - t;
        t = now(); CK(hipMemcpy(o, dout, bytes, hipMemcpyDeviceToHost)); double t2 = now() - t;
        printf("pageable H2D %.1f GB/s, D2H %.1f GB/s\n", bytes / t1 / 1e9, bytes / t2 / 1e9);
    }
    // hipHostRegister (pin in place) returns "invalid argument" for an ordinary user on the GPU boxes, so
    // the alternatives are pageable copies (above) and a pinned staging buffer filled by the CPU
    for (int rep = 0; rep < 3; rep++) {
        void *x, *y, *z;
        t = now(); CK(hipMalloc(&x, bytes / 2)); CK(hipMalloc(&y, bytes / 2)); CK(hipMalloc(&z, bytes / 2)); double tm = now() - t;
        t = now(); CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(z)); double tf = now() - t;
        printf("3 x hipMalloc(128 MiB) %.2f ms, 3 x hipFree %.2f ms\n", tm * 1e3, tf * 1e3);
    }
    char* pin;
    t = now(); CK(hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault)); printf("hipHostMalloc(256 MiB) %.2f ms\n", (now() - t) * 1e3);
    for (int rep = 0; rep < 2; rep++) {
        t = now(); memcpy(pin, a, bytes); double tc = now() - t;
        t = now(); CK(hipMemcpy(da, pin, bytes, hipMemcpyHostToDevice)); double th = now() - t;
        t = now(); CK(hipMemcpy(pin, dout, bytes, hipMemcpyDeviceToHost)); double td = now() - t;
        printf("CPU memcpy into pinned %.1f GB/s; pinned H2D %.1f GB/s, D2H %.1f GB/s\n", bytes / tc / 1e9, bytes / th / 1e9, bytes / td / 1e9);
    }
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    t = now();
    CK(hipMemcpyAsync(da, pin, bytes / 2, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(pin + bytes / 2, dout, bytes / 2, hipMemcpyDeviceToHost, s2));
    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
    printf("pinned H2D + D2H concurrently (128 MiB each): %.1f GB/s each way\n", bytes / 2 / (now() - t) / 1e9);
    for (int rep = 0; rep < 2; rep++) {
        t = now();
        hipLaunchKernelGGL(k_add, dim3(2048), dim3(256), 0, s1, (const uint4*)pin, (const uint4*)(pin + bytes / 4), (uint4*)(pin + bytes / 2), bytes / 4 / 16);
        CK(hipStreamSynchronize(s1));
        double tk = now() - t;
        printf("zero-copy kernel on pinned memory (2 reads + 1 write of 64 MiB): %.2f ms = %.1f GB/s in, %.1f GB/s out\n", tk * 1e3,
               2.0 * bytes / 4 / tk / 1e9, bytes / 4 / tk / 1e9);
    }
    return 0;
}
