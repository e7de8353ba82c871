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
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "fpgpu_internal.hpp"

namespace fpgpu {

namespace {

// ---- host threads: a queue of independent tasks, grown on demand, never torn down ------------------
class TaskPool {
   public:
    // grows the pool towards `n` threads; returns how many there are (a thread that cannot be created --
    // EAGAIN, RLIMIT_NPROC -- is simply not there: callers size their lane sets by the return value)
    int ensure(int n) {
        std::lock_guard<std::mutex> l(mu_);
        while ((int)threads_.size() < n) {
            try {
                threads_.emplace_back([this] { loop(); });
            } catch (const std::exception&) {
                break;
            }
        }
        return (int)threads_.size();
    }
    void submit(std::function<void()> fn) {
        {
            std::lock_guard<std::mutex> l(mu_);
            q_.push_back(std::move(fn));
        }
        cv_.notify_one();
    }

   private:
    void loop() {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return !q_.empty(); });
                fn = std::move(q_.front());
                q_.pop_front();
            }
            fn();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
};
struct Join {
    std::mutex mu;
    std::condition_variable cv;
    int pending = 0;
    void done() {
        std::lock_guard<std::mutex> l(mu);
        if (--pending == 0) cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> l(mu);
        cv.wait(l, [this] { return pending == 0; });
    }
};

constexpr size_t kLaneChunk = (size_t)2 << 20;   // bytes per operand per pipeline slot
constexpr size_t kSlotBytes = 3 * kLaneChunk;    // a | b | out
constexpr size_t kUpChunk = kSlotBytes;           // plain uploads: one DMA per slot
constexpr size_t kDownChunk = kLaneChunk;        // plain downloads: small chunks, every lane's stream busy early
constexpr size_t kMinStaged = (size_t)4 << 20;   // below this a plain hipMemcpy is as fast
constexpr int kMaxLanes = 16;

// A lane is the unit a transfer owns while it runs: two pinned buffers (portable: usable with every
// device) and, per device it has worked with, a stream, events and the device halves of its pipeline
// slots.  Lanes are handed out from one process-wide table, so independent host threads -- the
// reference's configs are Send + Sync and shared across rayon workers (src/lib.rs:62-65) -- run their
// staged transfers side by side, each on its own lanes; a transfer that finds every lane taken waits for
// the first one to come back.
struct LaneDev {
    int device = -1;
    hipStream_t stream = nullptr;                  // copies from the device; lane-local kernels
    hipEvent_t ev[2] = {nullptr, nullptr};         // slot s: its last DMA is through
    hipEvent_t uploaded[2] = {nullptr, nullptr};   // slot s: its operands are on the device
    hipEvent_t computed[2] = {nullptr, nullptr};   // slot s: its kernel is through
    hipEvent_t order = nullptr;                    // "everything queued on the caller's stream so far"
    char* dev[2] = {nullptr, nullptr};             // kSlotBytes each (elementwise pipeline only)
};
struct Lane {
    char* pin[2] = {nullptr, nullptr};   // kSlotBytes each
    std::vector<LaneDev*> devs;          // touched only by the transfer that holds the lane
    bool busy = false;
    bool dead = false;                   // its pinned buffers could not be had
};
struct DeviceShared {
    int device = -1;
    hipStream_t up = nullptr;        // every copy to the device -- see the header
    hipStream_t compute = nullptr;   // the pipeline's kernels (on `up` they would hold back the next copy)
    std::mutex issue;                // one lane at a time queues on `up` (keeps a slot's commands together)
};
struct Stage {
    std::mutex mu;                   // the lane table and the device table; never held across a transfer
    std::condition_variable freed;
    Lane lanes[kMaxLanes];
    int waiters = 0;
    bool host_unavailable = false;   // no pinned memory to be had: plain copies from now on
    std::vector<DeviceShared*> devices;
    TaskPool* pool = nullptr;
};
Stage g_stage;   // never torn down: the threads, buffers and streams live until the process ends

// how many lanes a transfer to ONE device asks for: FPGPU_COPY_THREADS, else by the host's core count
// (measured on the MI355X boxes, fpgpu_mul over 2^22 BN254 elements -- see DESIGN.md section 6)
int lanes_wanted() {
    long t = tuning().copy_threads;
    if (t <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        t = hw >= 8 ? 4 : (hw >= 4 ? 2 : 1);   // fpgpu_mul, 2^22 elements, steady state: 2 lanes 9.4 ms, 3: 7.2, 4: 6.5, 5: 7.1, 6: 7.4, 8: 7.6
    }
    return (int)std::min<long>(t, kMaxLanes);
}

struct DeviceGuard {  // restores the caller's current device
    int prev = -1;
    DeviceGuard() {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// The lanes one transfer holds; given back when it goes out of scope.
struct LaneSet {
    std::vector<Lane*> lanes;
    ~LaneSet() {
        if (lanes.empty()) return;
        {
            std::lock_guard<std::mutex> l(g_stage.mu);
            for (Lane* L : lanes) L->busy = false;
        }
        g_stage.freed.notify_all();
    }
    int width() const { return (int)lanes.size(); }
};

// Takes up to `want` lanes (at least one) while no more than `want` are busy in the whole process: more
// copy threads than that only contend for the link and the memory bus (fpgpu_mul over 2^22 elements with
// 2 / 3 / 4 / 5 / 6 / 8 lanes: 9.4 / 7.2 / 6.5 / 7.1 / 7.4 / 7.6 ms), so sixteen host threads calling at once
// share the four lanes one caller would use -- side by side when each gets one, one after the other when
// the first took them all -- instead of starting sixty-four memcpy threads.  A transfer that spans g devices
// asks for (and is capped at) g times as many.  Waits while the cap is reached, and leaves a share for the
// transfers waiting beside it.  FPGPU_ERR_UNSUPPORTED: staging is not available in this process (no pinned
// memory) -- the caller takes the plain path.
int acquire_lanes(int want, LaneSet* out) {
    out->lanes.reserve(kMaxLanes);   // before any lane is marked busy: a failed allocation must not strand one
    std::unique_lock<std::mutex> l(g_stage.mu);
    if (!g_stage.pool) g_stage.pool = new (std::nothrow) TaskPool();
    if (!g_stage.pool) return FPGPU_ERR_UNSUPPORTED;
    auto free_lanes = [want] {
        int f = 0, busy = 0;
        for (Lane& L : g_stage.lanes) {
            f += !L.busy && !L.dead;
            busy += L.busy;
        }
        return std::min(f, std::max(0, want - busy));
    };
    for (;;) {
        if (g_stage.host_unavailable) return FPGPU_ERR_UNSUPPORTED;
        g_stage.waiters++;
        g_stage.freed.wait(l, [&] { return free_lanes() > 0 || g_stage.host_unavailable; });
        g_stage.waiters--;
        if (g_stage.host_unavailable) return FPGPU_ERR_UNSUPPORTED;
        int take = std::min(want, std::max(1, free_lanes() / (g_stage.waiters + 1)));
        // lanes beyond the caller's own need a thread each; without one (thread creation refused) the caller's does it all
        int busy = 0;
        for (Lane& L : g_stage.lanes) busy += L.busy;
        if (g_stage.pool->ensure(std::min(kMaxLanes - 1, busy + take - 1)) == 0) take = 1;
        for (Lane& L : g_stage.lanes) {
            if ((int)out->lanes.size() == take) break;
            if (L.busy || L.dead) continue;
            for (int s = 0; s < 2 && !L.dead; s++)
                if (!L.pin[s] && hipHostMalloc((void**)&L.pin[s], kSlotBytes, hipHostMallocPortable) != hipSuccess) {
                    (void)hipGetLastError();
                    L.pin[s] = nullptr;
                    L.dead = true;   // a locked-memory limit: the table ends at the lanes that have their buffers
                }
            if (L.dead) {
                for (Lane& M : g_stage.lanes)
                    if (!M.pin[0] || !M.pin[1]) M.dead = true;
                break;
            }
            L.busy = true;
            out->lanes.push_back(&L);
        }
        if (!out->lanes.empty()) return FPGPU_OK;
        bool any = false;
        for (Lane& L : g_stage.lanes) any = any || !L.dead;
        if (!any) {   // not even one lane's buffers: a host-side limit, so for good
            g_stage.host_unavailable = true;
            g_stage.freed.notify_all();
            return FPGPU_ERR_UNSUPPORTED;
        }
        // every lane that exists is taken: wait for one
    }
}

int device_shared(int device, DeviceShared** out) {   // `device` must be current
    std::lock_guard<std::mutex> l(g_stage.mu);
    DeviceShared* d = nullptr;
    for (auto* c : g_stage.devices)
        if (c->device == device) d = c;
    if (!d) {
        d = new (std::nothrow) DeviceShared();
        if (!d) return fail(FPGPU_ERR_HIP, "out of host memory");
        d->device = device;
        g_stage.devices.push_back(d);   // kept even if a creation below fails: the next call completes it
    }
    if (!d->up) HIP_TRY(hipStreamCreateWithFlags(&d->up, hipStreamNonBlocking));
    if (!d->compute) HIP_TRY(hipStreamCreateWithFlags(&d->compute, hipStreamNonBlocking));
    *out = d;
    return FPGPU_OK;
}

int lane_dev(Lane* L, int device, LaneDev** out) {   // `device` must be current; the caller holds the lane
    LaneDev* d = nullptr;
    for (LaneDev* c : L->devs)
        if (c->device == device) d = c;
    if (!d) {
        d = new (std::nothrow) LaneDev();
        if (!d) return fail(FPGPU_ERR_HIP, "out of host memory");
        d->device = device;
        L->devs.push_back(d);
    }
    if (!d->stream) HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    if (!d->order) HIP_TRY(hipEventCreateWithFlags(&d->order, hipEventDisableTiming));
    for (int s = 0; s < 2; s++) {
        if (!d->ev[s]) HIP_TRY(hipEventCreateWithFlags(&d->ev[s], hipEventDisableTiming));
        if (!d->uploaded[s]) HIP_TRY(hipEventCreateWithFlags(&d->uploaded[s], hipEventDisableTiming));
        if (!d->computed[s]) HIP_TRY(hipEventCreateWithFlags(&d->computed[s], hipEventDisableTiming));
    }
    *out = d;
    return FPGPU_OK;
}
LaneDev* lane_dev_made(Lane* L, int device) {   // after lane_dev succeeded for (L, device)
    for (LaneDev* c : L->devs)
        if (c->device == device) return c;
    return nullptr;
}
DeviceShared* device_shared_made(int device) {
    std::lock_guard<std::mutex> l(g_stage.mu);
    for (auto* c : g_stage.devices)
        if (c->device == device) return c;
    return nullptr;
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

// Runs body(t) for every lane t of the set: lane 0 on the calling thread, the others on pool threads.
// A body sets its own current device.
template <typename Body>
int run_lanes(const LaneSet& set, Body&& body) {
    const int T = set.width();
    LaneStatus st;
    Join join;
    join.pending = T - 1;
    for (int t = 1; t < T; t++) {
        try {
            g_stage.pool->submit([&, t] {
                st.note(body(t));
                join.done();
            });
        } catch (const std::exception&) {   // no memory for the task: this thread does the lane's share itself
            st.note(body(t));
            join.done();
        }
    }
    st.note(body(0));
    join.wait();
    return st.result();
}

// after a failed transfer: nothing may still be reading or writing the lanes' buffers when they go back
void drain(const LaneSet& set, const std::vector<int>& devices) {
    for (int dev : devices) {
        if (hipSetDevice(dev) != hipSuccess) continue;
        if (DeviceShared* ds = device_shared_made(dev)) {
            if (ds->up) (void)hipStreamSynchronize(ds->up);
            if (ds->compute) (void)hipStreamSynchronize(ds->compute);
        }
        for (Lane* L : set.lanes)
            if (LaneDev* ld = lane_dev_made(L, dev))
                if (ld->stream) (void)hipStreamSynchronize(ld->stream);
    }
    (void)hipGetLastError();
}

// ---- plain transfers over any number of devices ----------------------------------------------------------
struct Chunk {
    size_t seg, off, size;
};
// chunks of every segment, interleaved segment by segment so that every device's link is busy from the start
std::vector<Chunk> interleave(const CopySeg* segs, size_t count, size_t chunk) {
    std::vector<Chunk> out;
    size_t most = 0;
    for (size_t i = 0; i < count; i++) most = std::max(most, (segs[i].bytes + chunk - 1) / chunk);
    for (size_t k = 0; k < most; k++)
        for (size_t i = 0; i < count; i++)
            if (k * chunk < segs[i].bytes) out.push_back(Chunk{i, k * chunk, std::min(chunk, segs[i].bytes - k * chunk)});
    return out;
}

int lane_upload(Lane* L, int t, int T, const std::vector<Chunk>& chunks, const CopySeg* segs) {
    hipEvent_t last[2] = {nullptr, nullptr};
    int last_dev[2] = {-1, -1};
    size_t j = 0;
    for (size_t c = t; c < chunks.size(); c += T, j++) {
        const int s = (int)(j & 1);
        const Chunk& ch = chunks[c];
        const CopySeg& seg = segs[ch.seg];
        if (last[s]) {   // the DMA out of this buffer is through
            HIP_TRY(hipSetDevice(last_dev[s]));
            HIP_TRY(hipEventSynchronize(last[s]));
        }
        std::memcpy(L->pin[s], (const char*)seg.host + ch.off, ch.size);
        HIP_TRY(hipSetDevice(seg.device));
        LaneDev* ld = lane_dev_made(L, seg.device);
        DeviceShared* ds = device_shared_made(seg.device);
        {
            std::lock_guard<std::mutex> issue(ds->issue);
            HIP_TRY(hipMemcpyAsync((char*)seg.dev + ch.off, L->pin[s], ch.size, hipMemcpyHostToDevice, ds->up));
            HIP_TRY(hipEventRecord(ld->ev[s], ds->up));
        }
        last[s] = ld->ev[s];
        last_dev[s] = seg.device;
    }
    for (int s = 0; s < 2; s++)
        if (last[s]) {
            HIP_TRY(hipSetDevice(last_dev[s]));
            HIP_TRY(hipEventSynchronize(last[s]));
        }
    return FPGPU_OK;
}

int lane_download(Lane* L, int t, int T, const std::vector<Chunk>& chunks, const CopySeg* segs) {
    const size_t mine = chunks.size() > (size_t)t ? (chunks.size() - t + T - 1) / T : 0;
    hipEvent_t evs[2] = {nullptr, nullptr};
    int devs[2] = {-1, -1};
    auto issue = [&](size_t j) -> int {
        const int s = (int)(j & 1);
        const Chunk& ch = chunks[t + j * T];
        const CopySeg& seg = segs[ch.seg];
        HIP_TRY(hipSetDevice(seg.device));
        LaneDev* ld = lane_dev_made(L, seg.device);
        HIP_TRY(hipMemcpyAsync(L->pin[s], (const char*)seg.dev + ch.off, ch.size, hipMemcpyDeviceToHost, ld->stream));
        HIP_TRY(hipEventRecord(ld->ev[s], ld->stream));
        evs[s] = ld->ev[s];
        devs[s] = seg.device;
        return FPGPU_OK;
    };
    if (mine) CHECK(issue(0));
    for (size_t j = 0; j < mine; j++) {
        const int s = (int)(j & 1);
        if (j + 1 < mine) CHECK(issue(j + 1));   // the other buffer: drained in the previous round
        HIP_TRY(hipSetDevice(devs[s]));
        HIP_TRY(hipEventSynchronize(evs[s]));
        const Chunk& ch = chunks[t + j * T];
        std::memcpy((char*)segs[ch.seg].host + ch.off, L->pin[s], ch.size);
    }
    return FPGPU_OK;
}

int plain_multi(const CopySeg* segs, size_t count, bool up) {
    for (size_t i = 0; i < count; i++) {
        if (!segs[i].bytes) continue;
        HIP_TRY(hipSetDevice(segs[i].device));
        if (up) HIP_TRY(hipMemcpyAsync(segs[i].dev, segs[i].host, segs[i].bytes, hipMemcpyHostToDevice, segs[i].stream));
        else HIP_TRY(hipMemcpyAsync(segs[i].host, segs[i].dev, segs[i].bytes, hipMemcpyDeviceToHost, segs[i].stream));
    }
    for (size_t i = 0; i < count; i++) {
        if (!segs[i].bytes) continue;
        HIP_TRY(hipSetDevice(segs[i].device));
        HIP_TRY(hipStreamSynchronize(segs[i].stream));
    }
    return FPGPU_OK;
}

// Makes the lanes' and the devices' objects for every device of the transfer and orders the transfer behind
// what each segment's stream holds so far.  FPGPU_ERR_UNSUPPORTED (nothing queued yet) when an object cannot
// be made: the caller takes the plain path for this transfer.
int prepare(const LaneSet& set, const CopySeg* segs, size_t count, bool up, std::vector<int>* devices) {
    for (size_t i = 0; i < count; i++)
        if (segs[i].bytes && std::find(devices->begin(), devices->end(), segs[i].device) == devices->end())
            devices->push_back(segs[i].device);
    for (int dev : *devices) {
        DeviceShared* ds = nullptr;
        if (hipSetDevice(dev) != hipSuccess || device_shared(dev, &ds) != FPGPU_OK) return FPGPU_ERR_UNSUPPORTED;
        for (Lane* L : set.lanes) {
            LaneDev* ld = nullptr;
            if (lane_dev(L, dev, &ld) != FPGPU_OK) return FPGPU_ERR_UNSUPPORTED;
        }
    }
    for (size_t i = 0; i < count; i++) {
        if (!segs[i].bytes) continue;
        HIP_TRY(hipSetDevice(segs[i].device));
        LaneDev* l0 = lane_dev_made(set.lanes[0], segs[i].device);
        HIP_TRY(hipEventRecord(l0->order, segs[i].stream));
        if (up) {
            HIP_TRY(hipStreamWaitEvent(device_shared_made(segs[i].device)->up, l0->order, 0));
        } else {
            for (Lane* L : set.lanes) HIP_TRY(hipStreamWaitEvent(lane_dev_made(L, segs[i].device)->stream, l0->order, 0));
        }
    }
    return FPGPU_OK;
}

int staged_multi_impl(const CopySeg* segs, size_t count, bool up);
int staged_multi(const CopySeg* segs, size_t count, bool up) {   // nothing may unwind through the C ABI
    try {
        return staged_multi_impl(segs, count, up);
    } catch (const std::exception&) {
        return fail(FPGPU_ERR_HIP, "out of host memory in a staged transfer");
    }
}
int staged_multi_impl(const CopySeg* segs, size_t count, bool up) {
    size_t total = 0;
    std::vector<int> distinct;
    for (size_t i = 0; i < count; i++) {
        total += segs[i].bytes;
        if (segs[i].bytes && std::find(distinct.begin(), distinct.end(), segs[i].device) == distinct.end())
            distinct.push_back(segs[i].device);
    }
    if (total == 0) return FPGPU_OK;
    DeviceGuard guard;
    const int per = lanes_wanted();
    if (total < kMinStaged || per == 1) return plain_multi(segs, count, up);
    // every device has its own link to the host: a transfer that spans g devices may use g times the lanes
    LaneSet set;
    if (acquire_lanes((int)std::min<size_t>(kMaxLanes, (size_t)per * distinct.size()), &set) != FPGPU_OK)
        return plain_multi(segs, count, up);
    std::vector<int> devices;
    const int prc = prepare(set, segs, count, up, &devices);
    if (prc == FPGPU_ERR_UNSUPPORTED) {
        (void)hipGetLastError();
        return plain_multi(segs, count, up);
    }
    int rc = prc;
    if (rc == FPGPU_OK) {
        const std::vector<Chunk> chunks = interleave(segs, count, up ? kUpChunk : kDownChunk);
        const int T = set.width();
        rc = run_lanes(set, [&](int t) {
            return up ? lane_upload(set.lanes[t], t, T, chunks, segs) : lane_download(set.lanes[t], t, T, chunks, segs);
        });
    }
    if (rc != FPGPU_OK) {
        const std::string msg = g_last_error;
        drain(set, devices);
        return fail(rc, msg);
    }
    return FPGPU_OK;
}

// ---- the chunk pipeline of the elementwise host calls ------------------------------------------------------
// Slot layout (device and pinned alike), every region 256-byte aligned:
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
inline size_t lane_chunks(size_t chunks, int t, int T) { return chunks > (size_t)t ? (chunks - t + T - 1) / T : 0; }

// chunk j of lane t: up (a | b in one DMA) on the upload stream; the kernels on the shared compute stream, or
// on the lane's own when a chunk's kernels are latency-bound (several chunks then run side by side); down
// (out | flags in one DMA) on the lane's own; two rounds later the lane copies the result out of the pinned
// buffer and reuses the slot.
int lane_map(Lane* L, LaneDev* ld, DeviceShared* d, int t, int T, const StreamJob& job, const SlotPlan& sp) {
    HIP_TRY(hipSetDevice(d->device));
    const size_t esz = job.esz, per = sp.per, n = job.n;
    const size_t mine = lane_chunks((n + per - 1) / per, t, T);
    for (size_t j = 0; j < mine + 2; j++) {
        const int s = (int)(j & 1);
        char* pin = L->pin[s];
        if (j >= 2) {
            const size_t first = (t + (j - 2) * T) * per, cnt = std::min(per, n - first);
            HIP_TRY(hipEventSynchronize(ld->ev[s]));
            std::memcpy((char*)job.out + first * esz, pin + sp.out_off, cnt * esz);
            if (job.flags) std::memcpy(job.flags + first, pin + sp.flags_off, cnt);
        }
        if (j < mine) {
            const size_t first = (t + j * T) * per, cnt = std::min(per, n - first), bytes = cnt * esz;
            std::memcpy(pin, (const char*)job.a + first * esz, bytes);
            if (job.b) std::memcpy(pin + sp.b_off, (const char*)job.b + first * esz, bytes);
            char* dev = ld->dev[s];
            hipStream_t cs = job.lane_compute ? ld->stream : d->compute;
            {
                std::lock_guard<std::mutex> issue(d->issue);
                HIP_TRY(hipMemcpyAsync(dev, pin, job.b ? sp.b_off + bytes : bytes, hipMemcpyHostToDevice, d->up));
                HIP_TRY(hipEventRecord(ld->uploaded[s], d->up));
                HIP_TRY(hipStreamWaitEvent(cs, ld->uploaded[s], 0));
                CHECK(job.call(dev, job.b ? dev + sp.b_off : nullptr, dev + sp.out_off,
                               job.want_flags ? (uint8_t*)(dev + sp.flags_off) : nullptr, dev + sp.ws_off, sp.ws_bytes, cnt, cs));
                if (!job.lane_compute) HIP_TRY(hipEventRecord(ld->computed[s], d->compute));
            }
            if (!job.lane_compute) HIP_TRY(hipStreamWaitEvent(ld->stream, ld->computed[s], 0));
            const size_t down = job.want_flags ? (sp.flags_off - sp.out_off) + cnt : bytes;   // out | flags in one DMA
            HIP_TRY(hipMemcpyAsync(pin + sp.out_off, dev + sp.out_off, down, hipMemcpyDeviceToHost, ld->stream));
            HIP_TRY(hipEventRecord(ld->ev[s], ld->stream));
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

// dev <- host (pageable) for every segment, each ordered after everything queued on its `stream`; returns when
// the data is in device memory (later work on any stream sees it; the host ranges are the caller's again).
// Segments on different devices are in flight together.
int upload_staged_multi(const CopySeg* segs, size_t count) { return staged_multi(segs, count, true); }
// host (pageable) <- dev for every segment, after everything queued on its `stream`; returns when the host
// ranges hold the data.
int download_staged_multi(const CopySeg* segs, size_t count) { return staged_multi(segs, count, false); }

int upload_staged(void* dev, const void* host, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return FPGPU_OK;
    CopySeg seg{0, dev, const_cast<void*>(host), bytes, stream};
    HIP_TRY(hipGetDevice(&seg.device));
    return staged_multi(&seg, 1, true);
}
int download_staged(void* host, const void* dev, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return FPGPU_OK;
    CopySeg seg{0, const_cast<void*>(dev), host, bytes, stream};
    HIP_TRY(hipGetDevice(&seg.device));
    return staged_multi(&seg, 1, false);
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
bool host_stream_usable(size_t bytes) {
    if (bytes < kMinStaged || lanes_wanted() <= 1) return false;
    std::lock_guard<std::mutex> l(g_stage.mu);
    return !g_stage.host_unavailable;
}

// The chunk pipeline for any per-element map of host arrays (see StreamJob): returns FPGPU_ERR_UNSUPPORTED
// without touching anything when the lanes cannot be had -- the caller then takes its whole-operand path.
// (A device that cannot hold a lane's slots -- out of memory -- only costs THIS call the pipeline.)
static int host_stream_map_impl(const StreamJob& job);
int host_stream_map(const StreamJob& job) {   // nothing may unwind through the C ABI
    try {
        return host_stream_map_impl(job);
    } catch (const std::exception&) {
        return fail(FPGPU_ERR_HIP, "out of host memory in the chunk pipeline");
    }
}
static int host_stream_map_impl(const StreamJob& job) {
    if (job.n == 0) return FPGPU_OK;
    const int want = lanes_wanted();
    if (want == 1) return FPGPU_ERR_UNSUPPORTED;
    // the largest chunk (a multiple of 32 elements) whose slot fits
    size_t per = (kSlotBytes / (job.esz * (job.b ? 3 : 2) + 1)) & ~(size_t)31;
    while (per > 32 && slot_bytes(plan_slot(job, per)) > kSlotBytes) per -= std::max<size_t>(32, (per / 16) & ~(size_t)31);
    if (per < 32 || slot_bytes(plan_slot(job, per)) > kSlotBytes) return FPGPU_ERR_UNSUPPORTED;
    const SlotPlan sp = plan_slot(job, per);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    LaneSet set;
    if (acquire_lanes(want, &set) != FPGPU_OK) return FPGPU_ERR_UNSUPPORTED;
    DeviceShared* ds = nullptr;
    std::vector<LaneDev*> lds;
    bool ok = device_shared(dev, &ds) == FPGPU_OK;
    for (Lane* L : set.lanes) {
        LaneDev* ld = nullptr;
        ok = ok && lane_dev(L, dev, &ld) == FPGPU_OK;
        for (int s = 0; s < 2 && ok; s++)
            if (!ld->dev[s] && hipMalloc((void**)&ld->dev[s], kSlotBytes) != hipSuccess) {
                ld->dev[s] = nullptr;
                ok = false;
            }
        lds.push_back(ld);
    }
    if (!ok) {
        (void)hipGetLastError();
        return FPGPU_ERR_UNSUPPORTED;
    }
    const int T = set.width();
    const int rc = run_lanes(set, [&](int t) { return lane_map(set.lanes[t], lds[t], ds, t, T, job, sp); });
    if (rc != FPGPU_OK) {
        const std::string msg = g_last_error;
        drain(set, std::vector<int>{dev});
        (void)hipSetDevice(dev);
        return fail(rc, msg);
    }
    return FPGPU_OK;
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
