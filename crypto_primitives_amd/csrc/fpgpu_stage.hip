// fpgpu_stage.hip -- host <-> device copies for the host-buffer entry points and the handle API.
//
// The reference's slices are ordinary (pageable) host memory: `&[MontyFieldElement]`, `Vec<T>`.  A plain
// hipMemcpy from such memory is staged by the runtime through pinned buffers on ONE thread and reaches
// ~30 GB/s to the device and ~18 GB/s back on the MI355X boxes (tools/ubench_pcie.hip), and the caller's
// memory cannot be pinned in place there (hipHostRegister is refused for ordinary users).  Here the
// staging is the library's, organised as a few independent LANES: lane t of T owns chunks t, t + T,
// t + 2T, ... of the transfer, a host thread, and two pinned + two device buffers.  A lane copies its
// chunk into pinned memory, queues the DMA and goes on with its next chunk while the DMA engine works;
// no lane waits for another, so T cores of memcpy feed the link, and copies in opposite directions
// overlap on the two DMA directions of PCIe.  Measured on the MI355X boxes (tools/time_xfer.py): copies
// TO the device must all be queued on ONE stream -- with a stream per lane, a hipMemcpyAsync now and
// then blocks its caller for the length of the whole transfer (6 ms for 128 MiB) and the rate falls from
// 43 to 18-25 GB/s -- while copies FROM the device are fastest on a stream per lane (40-45 GB/s against
// 30 on one); 4-6 MiB per DMA to the device (2 MiB: 38 GB/s, 6 MiB: 43-45).
//
// The elementwise host calls (fpgpu_mul, fpgpu_add, ... fpgpu_pow) never hold a whole operand on the
// device: a lane uploads a chunk of a and b, runs the kernel on it, downloads the chunk of the result and
// copies it out while its next chunk is in flight -- full duplex, no hipMalloc per call, and host arrays
// larger than HBM work.
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

// ---- persistent host threads: run(parts, fn) calls fn(0) here and fn(1..parts-1) on workers ----------
class WorkerPool {
   public:
    explicit WorkerPool(int workers) {
        for (int i = 0; i < workers; i++) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> l(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    int width() const { return (int)threads_.size() + 1; }
    void run(int parts, const std::function<void(int)>& fn) {   // parts <= width(), checked by the caller
        if (parts <= 1) {
            fn(0);
            return;
        }
        {
            std::lock_guard<std::mutex> l(mu_);
            fn_ = &fn;
            next_ = 1;
            parts_ = parts;
            pending_ = parts - 1;
        }
        cv_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

   private:
    void loop() {
        for (;;) {
            int part;
            const std::function<void(int)>* fn;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return stop_ || (fn_ && next_ < parts_); });
                if (stop_) return;
                part = next_++;
                fn = fn_;
            }
            (*fn)(part);
            {
                std::lock_guard<std::mutex> l(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, parts_ = 0, pending_ = 0;
    bool stop_ = false;
};

constexpr size_t kLaneChunk = (size_t)2 << 20;   // bytes per operand per pipeline slot
constexpr size_t kSlotBytes = 3 * kLaneChunk;    // a | b | out
constexpr size_t kUpChunk = kSlotBytes;           // plain uploads: one DMA per slot
constexpr size_t kDownChunk = kLaneChunk;        // plain downloads: small chunks, every lane's stream busy early
constexpr size_t kMinStaged = (size_t)4 << 20;   // below this a plain hipMemcpy is as fast
constexpr int kMaxLanes = 16;

struct Lane {
    hipStream_t stream = nullptr;                  // copies from the device
    hipEvent_t ev[2] = {nullptr, nullptr};         // slot s: its last DMA is through
    hipEvent_t uploaded[2] = {nullptr, nullptr};   // slot s: its operands are on the device
    hipEvent_t computed[2] = {nullptr, nullptr};   // slot s: its kernel is through
    char* dev[2] = {nullptr, nullptr};   // kSlotBytes each, on this device (elementwise pipeline only)
};
struct DeviceLanes {
    int device = -1;
    hipEvent_t order = nullptr;   // "everything queued on the caller's stream so far"
    hipStream_t up = nullptr;     // every copy to the device -- see the header
    hipStream_t compute = nullptr;   // the pipeline's kernels (on `up` they would hold back the next copy)
    std::mutex issue;             // one lane at a time queues on `up` (keeps a slot's commands together)
    std::vector<Lane> lanes;
};
struct Stage {
    std::mutex mu;   // one staged transfer at a time: the pinned buffers and the lanes are shared
    WorkerPool* pool = nullptr;
    char* pin[kMaxLanes][2] = {};   // kSlotBytes each, portable: every device's lane t uses pin[t]
    int pinned_lanes = 0;
    std::vector<DeviceLanes*> devices;
};
Stage g_stage;   // never torn down: the threads, buffers and streams live until the process ends

// how many lanes a transfer uses: FPGPU_COPY_THREADS, else by the host's core count (measured on the
// MI355X boxes, fpgpu_mul over 2^22 BN254 elements -- see DESIGN.md section 6)
int lanes_wanted() {
    long t = tuning().copy_threads;
    if (t <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        t = hw >= 8 ? 4 : (hw >= 4 ? 2 : 1);   // fpgpu_mul, 2^22 elements, steady state: 2 lanes 9.4 ms, 3: 7.2, 4: 6.5, 5: 7.1, 6: 7.4, 8: 7.6
    }
    return (int)std::min<long>(t, kMaxLanes);
}

// the current device's lanes, grown to `lanes` (caller holds g_stage.mu)
int acquire(int lanes, DeviceLanes** out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DeviceLanes* d = nullptr;
    for (auto* c : g_stage.devices)
        if (c->device == dev) d = c;
    if (!d) {
        d = new DeviceLanes();
        d->device = dev;
        g_stage.devices.push_back(d);   // kept even if a creation below fails: the next call completes it
    }
    if (!d->order) HIP_TRY(hipEventCreateWithFlags(&d->order, hipEventDisableTiming));
    if (!d->up) HIP_TRY(hipStreamCreateWithFlags(&d->up, hipStreamNonBlocking));
    if (!d->compute) HIP_TRY(hipStreamCreateWithFlags(&d->compute, hipStreamNonBlocking));
    for (; g_stage.pinned_lanes < lanes; g_stage.pinned_lanes++)
        for (int s = 0; s < 2; s++)
            HIP_TRY(hipHostMalloc((void**)&g_stage.pin[g_stage.pinned_lanes][s], kSlotBytes, hipHostMallocPortable));
    while ((int)d->lanes.size() < lanes) {
        Lane l;
        HIP_TRY(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        for (int s = 0; s < 2; s++) {
            HIP_TRY(hipEventCreateWithFlags(&l.ev[s], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&l.uploaded[s], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&l.computed[s], hipEventDisableTiming));
        }
        d->lanes.push_back(l);
    }
    if (!g_stage.pool || g_stage.pool->width() < lanes) {
        delete g_stage.pool;
        g_stage.pool = new WorkerPool(lanes - 1);
    }
    *out = d;
    return FPGPU_OK;
}

int ensure_device_slots(DeviceLanes* d, int lanes) {
    for (int t = 0; t < lanes; t++)
        for (int s = 0; s < 2; s++)
            if (!d->lanes[t].dev[s]) HIP_TRY(hipMalloc((void**)&d->lanes[t].dev[s], kSlotBytes));
    return FPGPU_OK;
}

// A lane's failure reaches the calling thread (the message of fail() is thread-local).
struct LaneStatus {
    std::mutex mu;
    int code = FPGPU_OK;
    std::string msg;
    void note(int rc) {
        if (rc == FPGPU_OK) return;
        std::lock_guard<std::mutex> l(mu);
        if (code == FPGPU_OK) {
            code = rc;
            msg = g_last_error;
        }
    }
    int result() { return code == FPGPU_OK ? FPGPU_OK : fail(code, msg); }
};

// Runs body(lane index) on `lanes` threads, each with `device` current; a failing lane drains its
// stream before it returns so that no DMA is left running on the shared buffers.
template <typename Body>
int run_lanes(DeviceLanes* d, int lanes, Body&& body) {
    if (g_stage.pool->width() < lanes || (int)d->lanes.size() < lanes)   // every lane owns chunks: none may be missing
        return fail(FPGPU_ERR_HIP, "staging: fewer copy threads than lanes");
    LaneStatus st;
    g_stage.pool->run(lanes, [&](int t) {
        int rc = FPGPU_OK;
        hipError_t e = t == 0 ? hipSuccess : hipSetDevice(d->device);   // part 0 is the calling thread
        if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice");
        if (rc == FPGPU_OK) rc = body(t);
        if (rc != FPGPU_OK) {
            (void)hipStreamSynchronize(d->up);
            (void)hipStreamSynchronize(d->compute);
            (void)hipStreamSynchronize(d->lanes[t].stream);
        }
        st.note(rc);
    });
    return st.result();
}

inline size_t lane_chunks(size_t chunks, int t, int T) { return chunks > (size_t)t ? (chunks - t + T - 1) / T : 0; }

int lane_upload(DeviceLanes* d, int t, int T, char* dev, const char* host, size_t bytes) {
    Lane& L = d->lanes[t];
    const size_t mine = lane_chunks((bytes + kUpChunk - 1) / kUpChunk, t, T);
    for (size_t j = 0; j < mine; j++) {
        const int s = (int)(j & 1);
        const size_t off = (t + j * T) * kUpChunk, sz = std::min(kUpChunk, bytes - off);
        if (j >= 2) HIP_TRY(hipEventSynchronize(L.ev[s]));   // the DMA out of this buffer is through
        std::memcpy(g_stage.pin[t][s], host + off, sz);
        std::lock_guard<std::mutex> issue(d->issue);
        HIP_TRY(hipMemcpyAsync(dev + off, g_stage.pin[t][s], sz, hipMemcpyHostToDevice, d->up));
        HIP_TRY(hipEventRecord(L.ev[s], d->up));
    }
    for (size_t j = mine > 2 ? mine - 2 : 0; j < mine; j++) HIP_TRY(hipEventSynchronize(L.ev[j & 1]));
    return FPGPU_OK;
}

int lane_download(DeviceLanes* d, int t, int T, char* host, const char* dev, size_t bytes) {
    Lane& L = d->lanes[t];
    HIP_TRY(hipStreamWaitEvent(L.stream, d->order, 0));
    const size_t mine = lane_chunks((bytes + kDownChunk - 1) / kDownChunk, t, T);
    auto issue = [&](size_t j) -> int {
        const int s = (int)(j & 1);
        const size_t off = (t + j * T) * kDownChunk, sz = std::min(kDownChunk, bytes - off);
        HIP_TRY(hipMemcpyAsync(g_stage.pin[t][s], dev + off, sz, hipMemcpyDeviceToHost, L.stream));
        HIP_TRY(hipEventRecord(L.ev[s], L.stream));
        return FPGPU_OK;
    };
    if (mine) CHECK(issue(0));
    for (size_t j = 0; j < mine; j++) {
        const int s = (int)(j & 1);
        if (j + 1 < mine) CHECK(issue(j + 1));   // the other buffer: drained in the previous round
        HIP_TRY(hipEventSynchronize(L.ev[s]));
        const size_t off = (t + j * T) * kDownChunk, sz = std::min(kDownChunk, bytes - off);
        std::memcpy(host + off, g_stage.pin[t][s], sz);
    }
    return FPGPU_OK;
}

// Slot layout of the chunk pipeline (device and pinned alike), every region 256-byte aligned:
//   a | b (if any) | out | flags (if any) | workspace (device only)
struct SlotPlan {
    size_t per;                      // elements per chunk
    size_t b_off, out_off, flags_off, ws_off, ws_bytes;
};
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
SlotPlan plan_slot(const StreamJob& job, size_t per) {
    SlotPlan p{};
    p.per = per;
    size_t off = up256(per * job.esz);
    p.b_off = off;
    if (job.b) off += up256(per * job.esz);
    p.out_off = off;
    off += up256(per * job.esz);
    p.flags_off = off;
    if (job.want_flags) off += up256(per);
    p.ws_off = off;
    p.ws_bytes = job.workspace_for ? job.workspace_for(per) : 0;
    return p;
}
inline size_t slot_bytes(const SlotPlan& p) { return p.ws_off + up256(p.ws_bytes); }

// chunk j of lane t: up (a | b in one DMA) on the upload stream; the kernels on the shared compute stream, or
// on the lane's own when a chunk's kernels are latency-bound (several chunks then run side by side); down
// (out | flags in one DMA) on the lane's own; two rounds later the lane copies the result out of the pinned
// buffer and reuses the slot.
int lane_map(DeviceLanes* d, int t, int T, const StreamJob& job, const SlotPlan& sp) {
    Lane& L = d->lanes[t];
    const size_t esz = job.esz, per = sp.per, n = job.n;
    const size_t mine = lane_chunks((n + per - 1) / per, t, T);
    for (size_t j = 0; j < mine + 2; j++) {
        const int s = (int)(j & 1);
        char* pin = g_stage.pin[t][s];
        if (j >= 2) {
            const size_t first = (t + (j - 2) * T) * per, cnt = std::min(per, n - first);
            HIP_TRY(hipEventSynchronize(L.ev[s]));
            std::memcpy((char*)job.out + first * esz, pin + sp.out_off, cnt * esz);
            if (job.flags) std::memcpy(job.flags + first, pin + sp.flags_off, cnt);
        }
        if (j < mine) {
            const size_t first = (t + j * T) * per, cnt = std::min(per, n - first), bytes = cnt * esz;
            std::memcpy(pin, (const char*)job.a + first * esz, bytes);
            if (job.b) std::memcpy(pin + sp.b_off, (const char*)job.b + first * esz, bytes);
            char* dev = L.dev[s];
            hipStream_t cs = job.lane_compute ? L.stream : d->compute;
            {
                std::lock_guard<std::mutex> issue(d->issue);
                HIP_TRY(hipMemcpyAsync(dev, pin, job.b ? sp.b_off + bytes : bytes, hipMemcpyHostToDevice, d->up));
                HIP_TRY(hipEventRecord(L.uploaded[s], d->up));
                HIP_TRY(hipStreamWaitEvent(cs, L.uploaded[s], 0));
                CHECK(job.call(dev, job.b ? dev + sp.b_off : nullptr, dev + sp.out_off,
                               job.want_flags ? (uint8_t*)(dev + sp.flags_off) : nullptr, dev + sp.ws_off, sp.ws_bytes, cnt, cs));
                if (!job.lane_compute) HIP_TRY(hipEventRecord(L.computed[s], d->compute));
            }
            if (!job.lane_compute) HIP_TRY(hipStreamWaitEvent(L.stream, L.computed[s], 0));
            const size_t down = job.want_flags ? (sp.flags_off - sp.out_off) + cnt : bytes;   // out | flags in one DMA
            HIP_TRY(hipMemcpyAsync(pin + sp.out_off, dev + sp.out_off, down, hipMemcpyDeviceToHost, L.stream));
            HIP_TRY(hipEventRecord(L.ev[s], L.stream));
        }
    }
    return FPGPU_OK;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
        if (e != hipSuccess) {
            p = nullptr;
            return hip_fail(e, "hipMalloc");
        }
        return FPGPU_OK;
    }
};

}  // namespace

// The lanes need pinned host memory, streams and threads; where the process may not have them (a locked-
// memory limit, no more streams) the transfers still work, through the runtime's own staging: the failure
// is remembered and every later transfer takes the plain path.
static bool g_lanes_unavailable = false;
static DeviceLanes* lanes_or_null(int lanes, bool device_slots) {
    if (g_lanes_unavailable) return nullptr;
    DeviceLanes* d = nullptr;
    if (acquire(lanes, &d) == FPGPU_OK && (!device_slots || ensure_device_slots(d, lanes) == FPGPU_OK)) return d;
    (void)hipGetLastError();
    g_lanes_unavailable = true;
    return nullptr;
}
static int plain_copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, kind, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return FPGPU_OK;
}

// dev <- host (pageable), ordered after everything queued on `stream` of the current device; returns when
// the data is in device memory (so later work on any stream sees it and `host` is the caller's again).
int upload_staged(void* dev, const void* host, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return FPGPU_OK;
    const int lanes = lanes_wanted();
    if (bytes < kMinStaged || lanes == 1) return plain_copy(dev, host, bytes, hipMemcpyHostToDevice, stream);
    std::lock_guard<std::mutex> lock(g_stage.mu);
    DeviceLanes* d = lanes_or_null(lanes, false);
    if (!d) return plain_copy(dev, host, bytes, hipMemcpyHostToDevice, stream);
    HIP_TRY(hipEventRecord(d->order, stream));
    HIP_TRY(hipStreamWaitEvent(d->up, d->order, 0));
    return run_lanes(d, lanes, [&](int t) { return lane_upload(d, t, lanes, (char*)dev, (const char*)host, bytes); });
}

// host (pageable) <- dev, after everything queued on `stream`; returns when `host` holds the data.
int download_staged(void* host, const void* dev, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return FPGPU_OK;
    const int lanes = lanes_wanted();
    if (bytes < kMinStaged || lanes == 1) return plain_copy(host, dev, bytes, hipMemcpyDeviceToHost, stream);
    std::lock_guard<std::mutex> lock(g_stage.mu);
    DeviceLanes* d = lanes_or_null(lanes, false);
    if (!d) return plain_copy(host, dev, bytes, hipMemcpyDeviceToHost, stream);
    HIP_TRY(hipEventRecord(d->order, stream));
    return run_lanes(d, lanes, [&](int t) { return lane_download(d, t, lanes, (char*)host, (const char*)dev, bytes); });
}

// out[i] = op(a[i], b[i]) for host arrays: `call(da, db, dout, count, stream)` queues the operation on
// `count` elements that lie in device memory (db == nullptr when b is).  Small batches, or one copy
// thread: whole operands on the device through plain copies; otherwise the chunk pipeline above.
static int whole_operands_elementwise(const void* a, const void* b, void* out, size_t n, size_t bytes, const ChunkCall& call) {
    DevBuf da, db, dout;
    CHECK(da.alloc(bytes));
    if (b) CHECK(db.alloc(bytes));
    CHECK(dout.alloc(bytes));
    HIP_TRY(hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, nullptr));
    if (b) HIP_TRY(hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, nullptr));
    CHECK(call(da.p, db.p, dout.p, n, nullptr));
    HIP_TRY(hipMemcpyAsync(out, dout.p, bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    return FPGPU_OK;
}

// Whether a host-array call of this size takes the chunk pipeline (else: whole operands on the device).
bool host_stream_usable(size_t bytes) { return bytes >= kMinStaged && lanes_wanted() > 1 && !g_lanes_unavailable; }

// The chunk pipeline for any per-element map of host arrays (see StreamJob): returns FPGPU_ERR_UNSUPPORTED
// without touching anything when the lanes cannot be had -- the caller then takes its whole-operand path.
int host_stream_map(const StreamJob& job) {
    if (job.n == 0) return FPGPU_OK;
    const int lanes = lanes_wanted();
    if (lanes == 1) return FPGPU_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_stage.mu);
    DeviceLanes* d = lanes_or_null(lanes, true);
    if (!d) return FPGPU_ERR_UNSUPPORTED;
    // the largest chunk (a multiple of 32 elements) whose slot fits
    size_t per = (kSlotBytes / (job.esz * (job.b ? 3 : 2) + 1)) & ~(size_t)31;
    while (per > 32 && slot_bytes(plan_slot(job, per)) > kSlotBytes) per -= std::max<size_t>(32, (per / 16) & ~(size_t)31);
    if (per < 32 || slot_bytes(plan_slot(job, per)) > kSlotBytes) return FPGPU_ERR_UNSUPPORTED;
    const SlotPlan sp = plan_slot(job, per);
    return run_lanes(d, lanes, [&](int t) { return lane_map(d, t, lanes, job, sp); });
}

int host_stream_elementwise(const void* a, const void* b, void* out, size_t n, size_t esz, const ChunkCall& call) {
    if (n == 0) return FPGPU_OK;
    const size_t bytes = n * esz;
    if (host_stream_usable(bytes)) {
        StreamJob job;
        job.a = a;
        job.b = b;
        job.out = out;
        job.n = n;
        job.esz = esz;
        job.call = [&](void* da, void* db, void* dout, uint8_t*, void*, size_t, size_t cnt, hipStream_t s) {
            return call(da, db, dout, cnt, s);
        };
        const int rc = host_stream_map(job);
        if (rc != FPGPU_ERR_UNSUPPORTED) return rc;
    }
    return whole_operands_elementwise(a, b, out, n, bytes, call);
}

}  // namespace fpgpu
