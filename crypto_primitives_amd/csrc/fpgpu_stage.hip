// fpgpu_stage.hip -- host <-> device copies for the host-buffer entry points and the handle API.
//
// The reference's slices are ordinary (pageable) host memory: `&[MontyFieldElement]`, `Vec<T>`.  A plain
// hipMemcpy from such memory is staged by the runtime through pinned buffers on ONE thread and reaches
// ~30 GB/s to the device and ~18 GB/s back on the MI355X boxes (tools/ubench_pcie.hip), and the caller's
// memory cannot be pinned in place there (hipHostRegister is refused for ordinary users).  Here the
// staging is the library's: two pinned buffers per direction, filled / drained by a few host threads
// while the DMA engine moves the previous chunk, so a transfer runs at the slower of "several cores of
// memcpy" and the PCIe link instead of at one core of memcpy.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "fpgpu_internal.hpp"

namespace fpgpu {

namespace {

// ---- a few persistent copy threads -------------------------------------------------------------
class CopyPool {
   public:
    explicit CopyPool(int workers) {
        for (int i = 0; i < workers; i++) threads_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    // dst <- src, split over the workers and the calling thread
    void copy(void* dst, const void* src, size_t bytes) {
        const size_t parts = threads_.size() + 1;
        if (parts == 1 || bytes < ((size_t)1 << 20)) {
            std::memcpy(dst, src, bytes);
            return;
        }
        const size_t piece = ((bytes / parts) + 4095) & ~(size_t)4095;
        {
            std::lock_guard<std::mutex> l(mu_);
            for (size_t i = 1; i < parts; i++) {
                const size_t off = i * piece;
                if (off >= bytes) break;
                jobs_.push_back({(char*)dst + off, (const char*)src + off, std::min(piece, bytes - off)});
                pending_++;
            }
        }
        cv_.notify_all();
        std::memcpy(dst, src, std::min(piece, bytes));
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return pending_ == 0; });
    }

   private:
    struct Job { char* dst; const char* src; size_t bytes; };
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return stop_ || !jobs_.empty(); });
                if (stop_ && jobs_.empty()) return;
                j = jobs_.back();
                jobs_.pop_back();
            }
            std::memcpy(j.dst, j.src, j.bytes);
            {
                std::lock_guard<std::mutex> l(mu_);
                pending_--;
            }
            done_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::vector<Job> jobs_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    size_t pending_ = 0;
    bool stop_ = false;
};

constexpr size_t kChunk = (size_t)8 << 20;   // bytes per staging buffer
constexpr size_t kMinStaged = (size_t)4 << 20;   // below this a plain hipMemcpy is as fast

struct Stage {
    std::mutex mu;   // one staged transfer at a time (the pinned buffers are shared)
    char* pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    CopyPool* pool = nullptr;
    bool ready = false;
};
Stage g_stage;

int stage_init() {
    if (g_stage.ready) return FPGPU_OK;
    for (int i = 0; i < 2; i++) {
        HIP_TRY(hipHostMalloc((void**)&g_stage.pin[i], kChunk, hipHostMallocPortable));
        HIP_TRY(hipEventCreateWithFlags(&g_stage.ev[i], hipEventDisableTiming));
    }
    long t = tuning().copy_threads;
    if (t <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        t = hw >= 16 ? 6 : (hw >= 8 ? 4 : (hw >= 4 ? 2 : 1));   // measured on the MI355X boxes: 1 thread 0.18 G elem/s
                                                                // for fpgpu_mul, 2: 0.31, 4: 0.37, 8: 0.39, 12: 0.36
    }
    g_stage.pool = new CopyPool((int)std::min<long>(t, 16) - 1);
    g_stage.ready = true;
    return FPGPU_OK;
}

}  // namespace

// Events are created on whatever device is current at first use; HIP events may only be recorded on
// streams of their own device, so the two helpers below re-create them when the device differs.
static int g_stage_device = -1;
static int stage_for_current_device() {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    CHECK(stage_init());
    if (g_stage_device != dev) {
        for (int i = 0; i < 2; i++) {
            if (g_stage.ev[i]) (void)hipEventDestroy(g_stage.ev[i]);
            HIP_TRY(hipEventCreateWithFlags(&g_stage.ev[i], hipEventDisableTiming));
        }
        g_stage_device = dev;
    }
    return FPGPU_OK;
}

// dev <- host (pageable), on `stream` of the current device; returns when `host` is the caller's again
// (the last chunks may still be in flight on `stream`: later work on that stream is ordered after them).
int upload_staged(void* dev, const void* host, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return FPGPU_OK;
    if (bytes < kMinStaged || tuning().copy_threads == 1) {
        HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return FPGPU_OK;
    }
    std::lock_guard<std::mutex> lock(g_stage.mu);
    CHECK(stage_for_current_device());
    size_t k = 0;
    for (size_t off = 0; off < bytes; off += kChunk, k++) {
        const int s = (int)(k & 1);
        const size_t sz = std::min(kChunk, bytes - off);
        if (k >= 2) HIP_TRY(hipEventSynchronize(g_stage.ev[s]));   // the DMA out of this buffer is through
        g_stage.pool->copy(g_stage.pin[s], (const char*)host + off, sz);
        HIP_TRY(hipMemcpyAsync((char*)dev + off, g_stage.pin[s], sz, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(g_stage.ev[s], stream));
    }
    // the pinned buffers are reused by the next transfer: wait until the DMA engine has read them
    HIP_TRY(hipEventSynchronize(g_stage.ev[0]));
    if (k > 1) HIP_TRY(hipEventSynchronize(g_stage.ev[1]));
    return FPGPU_OK;
}

// host (pageable) <- dev, after everything queued on `stream`; returns when `host` holds the data.
int download_staged(void* host, const void* dev, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return FPGPU_OK;
    if (bytes < kMinStaged || tuning().copy_threads == 1) {
        HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return FPGPU_OK;
    }
    std::lock_guard<std::mutex> lock(g_stage.mu);
    CHECK(stage_for_current_device());
    const size_t chunks = (bytes + kChunk - 1) / kChunk;
    auto issue = [&](size_t k) -> int {
        const size_t off = k * kChunk, sz = std::min(kChunk, bytes - off);
        HIP_TRY(hipMemcpyAsync(g_stage.pin[k & 1], (const char*)dev + off, sz, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipEventRecord(g_stage.ev[k & 1], stream));
        return FPGPU_OK;
    };
    CHECK(issue(0));
    for (size_t k = 0; k < chunks; k++) {
        if (k + 1 < chunks) CHECK(issue(k + 1));   // the other buffer: drained in the previous iteration
        HIP_TRY(hipEventSynchronize(g_stage.ev[k & 1]));
        const size_t off = k * kChunk, sz = std::min(kChunk, bytes - off);
        g_stage.pool->copy((char*)host + off, g_stage.pin[k & 1], sz);
    }
    return FPGPU_OK;
}

}  // namespace fpgpu
