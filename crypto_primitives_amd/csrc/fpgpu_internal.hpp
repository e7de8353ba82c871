// fpgpu_internal.hpp -- what the translation units of the C ABI share: the thread-local error
// message, status-code helpers and the dispatch over compiled limb counts.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>

#include "../../include/fpgpu.h"
#include "fp_launch.hpp"

namespace fpgpu {

extern thread_local std::string g_last_error;
int fail(int code, const std::string& msg);       // records msg for fpgpu_last_error_string(), returns code
int hip_fail(hipError_t e, const char* what);     // FPGPU_ERR_HIP with the runtime's message

// Host <-> device copies through the library's pinned staging buffers and copy threads
// (fpgpu_stage.hip); both return when the host buffer is the caller's again / holds the data.
int upload_staged(void* dev, const void* host, size_t bytes, hipStream_t stream);
int download_staged(void* host, const void* dev, size_t bytes, hipStream_t stream);
// The same for several ranges at once, each with its own device and stream (the parts of a sharded or
// replicated vector): ranges on different devices are in flight together, every device over its own link.
struct CopySeg {
    int device;
    void* dev;
    void* host;   // read by an upload, written by a download
    size_t bytes;
    hipStream_t stream;
};
int upload_staged_multi(const CopySeg* segs, size_t count);
int download_staged_multi(const CopySeg* segs, size_t count);
// out[i] = op(a[i], b[i]) on host arrays of n elements of esz bytes (b may be NULL): `call` queues the
// operation for `count` elements in device memory on `stream`.  Chunked, full duplex, no allocation per
// call above 4 MiB; synchronous.
using ChunkCall = std::function<int(void* da, void* db, void* dout, size_t count, hipStream_t stream)>;
int host_stream_elementwise(const void* a, const void* b, void* out, size_t n, size_t esz, const ChunkCall& call);
// The same pipeline for any map over independent element ranges: operands a (and b), result out, optionally
// one flag byte per element to the host, optionally a device workspace per chunk (workspace_for(count) bytes).
// `call` queues the chunk's kernels on the stream it is given.  lane_compute: the chunk's kernels are
// latency-bound (a batch inversion), so they run on the chunk's own stream and overlap with other chunks'.
struct StreamJob {
    const void* a = nullptr;
    const void* b = nullptr;
    void* out = nullptr;
    uint8_t* flags = nullptr;      // host, may be NULL even when the kernels need a device flag array
    bool want_flags = false;       // the kernels write one flag byte per element
    size_t n = 0, esz = 0;
    bool lane_compute = false;
    std::function<size_t(size_t count)> workspace_for;
    std::function<int(void* da, void* db, void* dout, uint8_t* dflags, void* ws, size_t ws_bytes, size_t count, hipStream_t s)> call;
};
bool host_stream_usable(size_t bytes);
int host_stream_map(const StreamJob& job);   // FPGPU_ERR_UNSUPPORTED: not available, nothing was done

}  // namespace fpgpu

#define HIP_TRY(expr)                                            \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return ::fpgpu::hip_fail(_e, #expr); \
    } while (0)
#define CHECK(expr)                      \
    do {                                 \
        int _rc = (expr);                \
        if (_rc != FPGPU_OK) return _rc; \
    } while (0)

// Dispatch on the compiled limb counts.
#define DISPATCH_N64(n64, ...)                                                      \
    switch (n64) {                                                                  \
        case 1: { constexpr int N = 2; __VA_ARGS__; } break;                        \
        case 2: { constexpr int N = 4; __VA_ARGS__; } break;                        \
        case 3: { constexpr int N = 6; __VA_ARGS__; } break;                        \
        case 4: { constexpr int N = 8; __VA_ARGS__; } break;                        \
        case 5: { constexpr int N = 10; __VA_ARGS__; } break;                       \
        case 6: { constexpr int N = 12; __VA_ARGS__; } break;                       \
        case 7: { constexpr int N = 14; __VA_ARGS__; } break;                       \
        case 8: { constexpr int N = 16; __VA_ARGS__; } break;                       \
        default: return ::fpgpu::fail(FPGPU_ERR_UNSUPPORTED, "unsupported limb count"); \
    }
