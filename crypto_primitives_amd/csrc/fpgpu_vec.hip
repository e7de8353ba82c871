// fpgpu_vec.hip -- device-resident vectors (fpgpu_vec) and multi-device sharding behind the C ABI.
//
// The reference's containers are host memory: `Vec<T>`, `DenseRowMatrix<T>::data` (src/matrix.rs:135-145),
// and for BoxedMontyField one heap allocation PER ELEMENT (src/field/crypto_bigint_boxed_monty.rs:62-64).
// A batch call on host slices pays PCIe for every operand of every call; a caller that chains
// operations wants its vectors to live in HBM.  A caller in Rust (or C) has no HIP bindings, so the
// boundary itself offers the device memory: an opaque handle with alloc / upload / download and a
// handle-taking form of every batch operation.  The same handles carry the multi-GPU layout of
// contiguous element ranges per device for elementwise work and batch
// inversion (no collective), row blocks for the matrix products -- the reference's own parallel
// axis, rayon over `par_chunks_exact(num_cols)` (src/matrix.rs:170-184) -- with x / B replicated and
// an all-gather of the output row blocks (RCCL over xGMI) when a replicated result is asked for.
//
// One host thread drives every device of a call: each entry of a device list owns a stream, so the loops
// below only enqueue work; nothing here blocks except upload / download / sync and the calls that
// return a value to the host.
//
// Threads.  The reference's configs are Send + Sync and shared across rayon workers (src/lib.rs:62-65),
// so any number of host threads may call in at once.  There is no library-wide lock on the call paths:
//   g_reg_mu     the registry (slots, the default device list): held while a list or a vector is made;
//   Slot::mu     held while ONE call enqueues the kernels that use the entry's scratch block (inversion,
//                folds, matrix-core products), so two calls never interleave their use of it;
//   g_coll_mu    the event-ordered peer copies and RCCL group calls of an all-gather / fpgpu_vec_copy:
//                one such protocol at a time.
// Elementwise operations take no lock at all.  Order: g_coll_mu before Slot::mu; g_reg_mu is never held
// while another is taken.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "fpgpu_internal.hpp"

using namespace fpgpu;

namespace {

// ---- one entry of a device list ------------------------------------------------------------
struct Slot {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ready = nullptr;  // "my part is written" (copy transport of the all-gather, fpgpu_vec_copy)
    hipEvent_t done = nullptr;   // "I have finished reading my peers' parts"
    void* scratch = nullptr;     // grow-only workspace of the operations that need one
    size_t scratch_bytes = 0;
    std::mutex mu;               // see the header: one call at a time enqueues work that uses `scratch`
};

std::mutex g_reg_mu;
std::mutex g_coll_mu;
std::vector<std::unique_ptr<Slot>> g_slots;  // never destroyed: vectors keep pointers to their slots

int rccl_destroy_comms(std::vector<ncclComm_t>& comms);

// A device list: what fpgpu_set_devices installs as the default, or what fpgpu_devset_create returns.
// Vectors hold a reference, so a list outlives fpgpu_set_devices / fpgpu_devset_destroy for as long as a
// vector laid out over it exists.  The RCCL communicators of a list are made at its first all-gather.
struct DevSet {
    std::vector<Slot*> list;
    bool distinct = true;
    std::vector<ncclComm_t> comms;   // one per entry while comm_ready (guarded by g_coll_mu)
    bool comm_ready = false;
    ~DevSet() {
        if (!comm_ready) return;
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        for (Slot* s : list)   // collectives still queued on an entry's stream must be through before its communicator goes
            if (hipSetDevice(s->device) == hipSuccess) (void)hipStreamSynchronize(s->stream);
        (void)rccl_destroy_comms(comms);
        if (prev >= 0) (void)hipSetDevice(prev);
        (void)hipGetLastError();
    }
};
// the default list (fpgpu_set_devices); null = not set (the current device alone).  The holder itself is
// never destroyed, so no communicator is torn down during static destruction at process exit.
std::shared_ptr<DevSet>& default_set() {
    static auto* holder = new std::shared_ptr<DevSet>();
    return *holder;
}
std::vector<std::shared_ptr<DevSet>>& single_sets() {   // "the current device alone", one per device, made on demand
    static auto* v = new std::vector<std::shared_ptr<DevSet>>();
    return *v;
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

int device_count(int* n) {
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess || *n <= 0) {
        (void)hipGetLastError();
        return fail(FPGPU_ERR_NO_DEVICE, "no HIP device visible: libfpgpu has no CPU fallback");
    }
    return FPGPU_OK;
}

int make_slot(int device, Slot** out) {   // g_reg_mu held
    auto s = std::make_unique<Slot>();
    s->device = device;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    *out = s.get();
    g_slots.push_back(std::move(s));
    return FPGPU_OK;
}

// The k-th slot of `device` (made if there are fewer): the k-th mention of a device in any list takes the
// device's k-th slot, so two lists that name a device share its stream and their work stays ordered.
int nth_slot(int device, int k, Slot** out) {   // g_reg_mu held
    int seen = 0;
    for (auto& t : g_slots)
        if (t->device == device && seen++ == k) { *out = t.get(); return FPGPU_OK; }
    Slot* s = nullptr;
    for (; seen <= k; seen++) CHECK(make_slot(device, &s));
    *out = s;
    return FPGPU_OK;
}

int current_device(int* dev) {
    int n = 0;
    CHECK(device_count(&n));
    HIP_TRY(hipGetDevice(dev));
    return FPGPU_OK;
}

int build_set(const int* ids, int count, std::shared_ptr<DevSet>* out) {   // g_reg_mu held; ids validated
    auto set = std::make_shared<DevSet>();
    DeviceGuard guard;
    for (int i = 0; i < count; i++) {
        int k = 0;
        for (int j = 0; j < i; j++) k += ids[j] == ids[i];
        set->distinct = set->distinct && k == 0;
        Slot* s = nullptr;
        CHECK(nth_slot(ids[i], k, &s));
        set->list.push_back(s);
    }
    // direct peer copies (fpgpu_vec_copy, the copy transport of the all-gather) between the list's devices:
    // best effort -- without peer access hipMemcpyPeerAsync still works, staged by the runtime
    for (Slot* a : set->list)
        for (Slot* b : set->list) {
            int can = 0;
            if (a->device == b->device || hipSetDevice(a->device) != hipSuccess) continue;
            if (hipDeviceCanAccessPeer(&can, a->device, b->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(b->device, 0);
            (void)hipGetLastError();   // "already enabled" included
        }
    *out = set;
    return FPGPU_OK;
}

// The list a placement refers to: `given` (fpgpu_devset), else what fpgpu_set_devices set, else the
// current device alone.
int effective_set(const std::shared_ptr<DevSet>* given, std::shared_ptr<DevSet>* out) {   // g_reg_mu held
    if (given && *given) { *out = *given; return FPGPU_OK; }
    if (default_set()) { *out = default_set(); return FPGPU_OK; }
    int dev = 0;
    CHECK(current_device(&dev));
    auto& singles = single_sets();
    if ((int)singles.size() <= dev) singles.resize(dev + 1);
    if (!singles[dev]) CHECK(build_set(&dev, 1, &singles[dev]));
    *out = singles[dev];
    return FPGPU_OK;
}

// The slot that work placed on ONE device uses: the first entry of the list in effect on that device, else
// the device's first slot.
int slot_for_device(const std::shared_ptr<DevSet>* given, int device, Slot** out) {   // g_reg_mu held
    const std::shared_ptr<DevSet>& set = (given && *given) ? *given : default_set();
    if (set)
        for (Slot* s : set->list)
            if (s->device == device) { *out = s; return FPGPU_OK; }
    DeviceGuard guard;
    return nth_slot(device, 0, out);
}

int ensure_scratch(Slot* s, size_t bytes) {  // the slot's device must be current, s->mu held
    if (bytes <= s->scratch_bytes) return FPGPU_OK;
    if (s->scratch) {
        HIP_TRY(hipStreamSynchronize(s->stream));  // earlier work may still be using the old block
        HIP_TRY(hipFree(s->scratch));
        s->scratch = nullptr;
        s->scratch_bytes = 0;
    }
    const size_t want = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    HIP_TRY(hipMalloc(&s->scratch, want));
    s->scratch_bytes = want;
    return FPGPU_OK;
}

// ---- RCCL, loaded on first use (the single-GPU paths need no communication library) --------
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int rccl_fail(ncclResult_t r, const char* what) {
    return fail(FPGPU_ERR_RCCL, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
#define RCCL_TRY(expr)                                      \
    do {                                                    \
        ncclResult_t _r = (expr);                           \
        if (_r != ncclSuccess) return rccl_fail(_r, #expr); \
    } while (0)

int load_rccl() {
    if (g_rccl.handle) return FPGPU_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(FPGPU_ERR_RCCL, std::string("cannot load librccl.so.1: ") + dlerror());
    Rccl r;
    r.handle = h;
#define SYM(field, name)                                                                          \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                \
    if (!r.field) return fail(FPGPU_ERR_RCCL, std::string("librccl.so.1 lacks ") + name);
    SYM(CommInitAll, "ncclCommInitAll")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(AllGather, "ncclAllGather")
    SYM(Broadcast, "ncclBroadcast")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    g_rccl = r;
    return FPGPU_OK;
}

int rccl_destroy_comms(std::vector<ncclComm_t>& comms) {
    for (ncclComm_t& c : comms)
        if (c && g_rccl.CommDestroy) {
            (void)g_rccl.CommDestroy(c);
            c = nullptr;
        }
    return FPGPU_OK;
}

int ensure_comms(DevSet* set) {  // g_coll_mu held; one communicator per list entry (ncclCommInitAll: one process, many devices)
    if (set->comm_ready) return FPGPU_OK;
    CHECK(load_rccl());
    std::vector<int> devs;
    for (Slot* s : set->list) devs.push_back(s->device);
    set->comms.assign(devs.size(), nullptr);
    RCCL_TRY(g_rccl.CommInitAll(set->comms.data(), (int)devs.size(), devs.data()));
    set->comm_ready = true;
    return FPGPU_OK;
}

}  // namespace

// ---- the handle ---------------------------------------------------------------------------------
enum VecKind { VEC_SINGLE = 0, VEC_SHARDED = 1, VEC_REPLICATED = 2 };
struct VecPart {
    Slot* slot;
    size_t first, count;  // elements [first, first + count) of the vector
    char* ptr;            // device memory of this part (the whole vector for a replica)
};
struct fpgpu_vec {
    uint32_t n64 = 0;      // words per element; 0 for a vector of 32-bit indices (fpgpu_idx)
    size_t n = 0;
    int kind = VEC_SINGLE;
    std::vector<VecPart> parts;
    std::shared_ptr<DevSet> set;   // the list a sharded / replicated vector is laid out over (null for VEC_SINGLE)
    size_t esz() const { return n64 ? (size_t)n64 * 8 : 4; }
};
// A device list as a handle (fpgpu_devset_create).
struct fpgpu_devset {
    std::shared_ptr<DevSet> set;
};
// Column indices of a SparseMatrix: the same container with 4-byte elements.
struct fpgpu_idx {
    fpgpu_vec v;
};

namespace {

int check_vec(const fpgpu_ctx* ctx, const fpgpu_vec* v, const char* name) {
    if (!ctx) return fail(FPGPU_ERR_SHAPE, "null context");
    if (!v) return fail(FPGPU_ERR_SHAPE, std::string("null vector: ") + name);
    if (v->n64 != ctx->n64) return fail(FPGPU_ERR_SHAPE, std::string("vector of another element width: ") + name);
    return FPGPU_OK;
}

int vec_alloc_n64(const fpgpu_devset* on, uint32_t n64, size_t n, size_t unit, int placement, fpgpu_vec** out);
int vec_alloc(const fpgpu_ctx* ctx, const fpgpu_devset* on, size_t n, size_t unit, int placement, fpgpu_vec** out) {
    if (!ctx || !out) return fail(FPGPU_ERR_SHAPE, "null pointer");
    return vec_alloc_n64(on, ctx->n64, n, unit, placement, out);
}
int vec_alloc_n64(const fpgpu_devset* on, uint32_t n64, size_t n, size_t unit, int placement, fpgpu_vec** out) {
    if (!out) return fail(FPGPU_ERR_SHAPE, "null pointer");
    *out = nullptr;
    if (n > ((size_t)1 << 40)) return fail(FPGPU_ERR_SHAPE, "element count too large");
    if (unit == 0) unit = 1;
    if (n % unit) return fail(FPGPU_ERR_SHAPE, "vec_alloc: length is not a whole number of rows");
    int ndev = 0;
    CHECK(device_count(&ndev));
    DeviceGuard guard;
    auto v = std::make_unique<fpgpu_vec>();
    v->n64 = n64;
    v->n = n;
    {
        std::lock_guard<std::mutex> lock(g_reg_mu);
        const std::shared_ptr<DevSet>* given = on ? &on->set : nullptr;
        if (placement >= 0 || placement == FPGPU_PLACE_CURRENT) {
            int dev = placement;
            if (placement == FPGPU_PLACE_CURRENT) CHECK(current_device(&dev));
            if (dev >= ndev) return fail(FPGPU_ERR_SHAPE, "vec_alloc: no such device");
            Slot* s = nullptr;
            CHECK(slot_for_device(given, dev, &s));
            v->kind = VEC_SINGLE;
            v->parts.push_back(VecPart{s, 0, n, nullptr});
        } else if (placement == FPGPU_PLACE_SHARDED || placement == FPGPU_PLACE_REPLICATED) {
            CHECK(effective_set(given, &v->set));
            const std::vector<Slot*>& list = v->set->list;
            const size_t g = list.size(), units = n / unit, base = units / g, rem = units % g;
            v->kind = placement == FPGPU_PLACE_SHARDED ? VEC_SHARDED : VEC_REPLICATED;
            for (size_t i = 0; i < g; i++) {
                if (v->kind == VEC_REPLICATED) v->parts.push_back(VecPart{list[i], 0, n, nullptr});
                else  // contiguous, balanced: the first `rem` devices take one unit more
                    v->parts.push_back(VecPart{list[i], (i * base + std::min(i, rem)) * unit, (base + (i < rem ? 1 : 0)) * unit, nullptr});
            }
        } else {
            return fail(FPGPU_ERR_SHAPE, "vec_alloc: unknown placement");
        }
    }
    for (VecPart& p : v->parts) {
        hipError_t e = hipSetDevice(p.slot->device);
        void* mem = nullptr;
        if (e == hipSuccess) e = hipMalloc(&mem, std::max<size_t>(p.count * v->esz(), 256));
        if (e != hipSuccess) {
            for (VecPart& q : v->parts)
                if (q.ptr) { (void)hipSetDevice(q.slot->device); (void)hipFree(q.ptr); }
            return hip_fail(e, "hipMalloc");
        }
        p.ptr = (char*)mem;
    }
    *out = v.release();
    return FPGPU_OK;
}

// Device pointer to the elements of operand `v` that line up with part `pi` of `outv`:
// a replica on the same list entry serves any range; otherwise the partitions must be the same.
int operand_ptr(const fpgpu_vec* v, const fpgpu_vec* outv, size_t pi, const char* name, const char** p) {
    const VecPart& o = outv->parts[pi];
    if (v->n != outv->n) return fail(FPGPU_ERR_SHAPE, std::string("operand length mismatch: ") + name);
    if (v->kind == VEC_REPLICATED) {
        for (const VecPart& q : v->parts)
            if (q.slot == o.slot) { *p = q.ptr + o.first * v->esz(); return FPGPU_OK; }
        return fail(FPGPU_ERR_SHAPE, std::string("replicated operand has no copy on the output's device: ") + name);
    }
    if (v->parts.size() == outv->parts.size() && v->parts[pi].slot == o.slot && v->parts[pi].first == o.first &&
        v->parts[pi].count == o.count) {
        *p = v->parts[pi].ptr;
        return FPGPU_OK;
    }
    return fail(FPGPU_ERR_SHAPE, std::string("operand is partitioned differently from the output: ") + name);
}

// The copy of `v` that holds ALL of it on list entry `s` (a replica, or a single-part vector there).
int whole_on_slot(const fpgpu_vec* v, const Slot* s, const char* name, const char** p) {
    for (const VecPart& q : v->parts)
        if (q.slot == s && q.first == 0 && q.count == v->n) { *p = q.ptr; return FPGPU_OK; }
    return fail(FPGPU_ERR_SHAPE, std::string("operand must be whole (replicated) on every device that computes: ") + name);
}

using BinaryDev = int (*)(const fpgpu_ctx*, const void*, const void*, void*, size_t, void*);
using UnaryDev = int (*)(const fpgpu_ctx*, const void*, void*, size_t, void*);

int vec_binary(BinaryDev fn, const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* b, fpgpu_vec* out) {
    CHECK(check_vec(ctx, a, "a"));
    CHECK(check_vec(ctx, b, "b"));
    CHECK(check_vec(ctx, out, "out"));
    DeviceGuard guard;
    for (size_t pi = 0; pi < out->parts.size(); pi++) {
        const VecPart& o = out->parts[pi];
        const char *pa, *pb;
        CHECK(operand_ptr(a, out, pi, "a", &pa));
        CHECK(operand_ptr(b, out, pi, "b", &pb));
        HIP_TRY(hipSetDevice(o.slot->device));
        CHECK(fn(ctx, pa, pb, o.ptr, o.count, o.slot->stream));
    }
    return FPGPU_OK;
}
int vec_unary(UnaryDev fn, const fpgpu_ctx* ctx, const fpgpu_vec* a, fpgpu_vec* out) {
    CHECK(check_vec(ctx, a, "a"));
    CHECK(check_vec(ctx, out, "out"));
    DeviceGuard guard;
    for (size_t pi = 0; pi < out->parts.size(); pi++) {
        const VecPart& o = out->parts[pi];
        const char* pa;
        CHECK(operand_ptr(a, out, pi, "a", &pa));
        HIP_TRY(hipSetDevice(o.slot->device));
        CHECK(fn(ctx, pa, o.ptr, o.count, o.slot->stream));
    }
    return FPGPU_OK;
}

int copy_d2d(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t stream) {
    if (dst_dev == src_dev) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    else HIP_TRY(hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, stream));
    return FPGPU_OK;
}

// ---- all-gather of the parts of a replicated vector ---------------------------------------------
// Every list entry i has written elements [first_i, first_i + count_i) of ITS OWN copy of `y`; afterwards
// every copy holds every range.  Transport: RCCL (ncclAllGather in place when the ranges are equal,
// else one ncclBroadcast per range, all in one group) when the list's devices are distinct; direct
// peer copies ordered by events when the list names a device twice (a single-GPU test layout RCCL
// refuses) or when the tuning hook asks for it.
struct Range { size_t first, count; };

int allgather_copy(fpgpu_vec* y, const std::vector<Range>& r) {
    const size_t g = y->parts.size(), esz = y->esz();
    for (size_t i = 0; i < g; i++) {
        HIP_TRY(hipSetDevice(y->parts[i].slot->device));
        HIP_TRY(hipEventRecord(y->parts[i].slot->ready, y->parts[i].slot->stream));
    }
    for (size_t i = 0; i < g; i++) {  // i pulls every other range into its copy
        Slot* si = y->parts[i].slot;
        HIP_TRY(hipSetDevice(si->device));
        for (size_t j = 0; j < g; j++) {
            if (j == i || r[j].count == 0) continue;
            Slot* sj = y->parts[j].slot;
            HIP_TRY(hipStreamWaitEvent(si->stream, sj->ready, 0));
            CHECK(copy_d2d(y->parts[i].ptr + r[j].first * esz, si->device, y->parts[j].ptr + r[j].first * esz, sj->device,
                           r[j].count * esz, si->stream));
        }
        HIP_TRY(hipEventRecord(si->done, si->stream));
    }
    for (size_t i = 0; i < g; i++)  // nobody overwrites a range that a peer may still be reading
        for (size_t j = 0; j < g; j++)
            if (j != i) {
                HIP_TRY(hipSetDevice(y->parts[i].slot->device));
                HIP_TRY(hipStreamWaitEvent(y->parts[i].slot->stream, y->parts[j].slot->done, 0));
            }
    return FPGPU_OK;
}

int allgather_rccl(fpgpu_vec* y, const std::vector<Range>& r) {
    CHECK(ensure_comms(y->set.get()));
    const std::vector<ncclComm_t>& comms = y->set->comms;
    const size_t g = y->parts.size(), esz = y->esz(), w = y->n64;
    bool equal = true;
    for (size_t i = 0; i < g; i++) equal = equal && r[i].count == r[0].count && r[i].first == i * r[0].count;
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t i = 0; i < g; i++) {
        Slot* si = y->parts[i].slot;
        char* base = y->parts[i].ptr;
        if (equal) {
            if (r[0].count)
                RCCL_TRY(g_rccl.AllGather(base + r[i].first * esz, base, r[0].count * w, ncclUint64, comms[i], si->stream));
        } else {
            for (size_t root = 0; root < g; root++)
                if (r[root].count)
                    RCCL_TRY(g_rccl.Broadcast(base + r[root].first * esz, base + r[root].first * esz, r[root].count * w,
                                              ncclUint64, (int)root, comms[i], si->stream));
        }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return FPGPU_OK;
}

int allgather(fpgpu_vec* y, const std::vector<Range>& r) {
    if (y->kind != VEC_REPLICATED || !y->set || y->parts.size() != y->set->list.size() || r.size() != y->parts.size())
        return fail(FPGPU_ERR_SHAPE, "all-gather: the output is not replicated over a device list");
    if (y->parts.size() == 1) return FPGPU_OK;   // one copy, which already holds everything
    std::lock_guard<std::mutex> lock(g_coll_mu);
    if (!y->set->distinct || tuning().allgather_copy > 0) return allgather_copy(y, r);
    return allgather_rccl(y, r);
}

// Row-block products share this shape: `A` holds whole rows (row length `row_len` elements) on each of
// its parts; output row r has `out_len` elements.  Calls body(part index, rows of the part, first row,
// output pointer) per part, then gathers the output if it is replicated while A is sharded.
template <typename Body>
int rowblock_product(const fpgpu_vec* A, size_t rows, size_t row_len, fpgpu_vec* out, size_t out_len, const char* what,
                     Body&& body) {
    if (A->n != rows * row_len) return fail(FPGPU_ERR_SHAPE, std::string(what) + ": A does not hold rows x cols elements");
    if (out->n != rows * out_len) return fail(FPGPU_ERR_SHAPE, std::string(what) + ": output length mismatch");
    const bool gather = out->kind == VEC_REPLICATED && A->kind != VEC_REPLICATED;
    if (!gather && out->parts.size() != A->parts.size())
        return fail(FPGPU_ERR_SHAPE, std::string(what) + ": output is partitioned differently from A's rows");
    // what every copy of a gathered output contributes: one range per OUTPUT part, empty for a copy whose
    // device holds no rows of A (A on fewer devices than the output's list)
    std::vector<Range> ranges(out->parts.size(), Range{0, 0});
    DeviceGuard guard;
    for (size_t pi = 0; pi < A->parts.size(); pi++) {
        const VecPart& a = A->parts[pi];
        if (row_len && (a.first % row_len || a.count % row_len))
            return fail(FPGPU_ERR_SHAPE, std::string(what) + ": a part of A does not hold whole rows (use fpgpu_vec_alloc_rows)");
        // rows of this part: from A's partition; a matrix with no columns has no elements to go by, so
        // then from the output's partition (or everything on the first part when the output is gathered)
        size_t prow0 = row_len ? a.first / row_len : 0, prows = row_len ? a.count / row_len : 0;
        if (!row_len) {
            if (gather) { prow0 = pi == 0 ? 0 : rows; prows = pi == 0 ? rows : 0; }
            else if (out_len) { prow0 = out->parts[pi].first / out_len; prows = out->parts[pi].count / out_len; }
        }
        char* dst = nullptr;
        if (gather) {
            size_t oi = out->parts.size();
            for (size_t j = 0; j < out->parts.size(); j++)
                if (out->parts[j].slot == a.slot) { oi = j; break; }
            if (oi == out->parts.size())
                return fail(FPGPU_ERR_SHAPE, std::string(what) + ": replicated output has no copy on a device that holds rows of A");
            dst = out->parts[oi].ptr + prow0 * out_len * out->esz();
            ranges[oi] = Range{prow0 * out_len, prows * out_len};
        } else {
            const VecPart& o = out->parts[pi];
            if (o.slot != a.slot || o.first != prow0 * out_len || o.count != prows * out_len)
                return fail(FPGPU_ERR_SHAPE, std::string(what) + ": output is partitioned differently from A's rows");
            dst = o.ptr;
        }
        HIP_TRY(hipSetDevice(a.slot->device));
        std::lock_guard<std::mutex> lock(a.slot->mu);   // the body may use the entry's scratch block
        CHECK(body(pi, prows, prow0, dst));
    }
    if (gather) CHECK(allgather(out, ranges));
    return FPGPU_OK;
}

// result of a fold / dot product: per-part partials, combined on the first part's device
template <typename PartCall>
int vec_reduce(const fpgpu_ctx* ctx, const fpgpu_vec* v, bool product, uint64_t* out_host, PartCall&& call) {
    if (!out_host) return fail(FPGPU_ERR_SHAPE, "null host pointer: out");
    DeviceGuard guard;
    const size_t esz = v->esz(), fold_ws = fpgpu_fold_workspace_bytes(ctx);
    const size_t nparts = v->kind == VEC_REPLICATED ? 1 : v->parts.size();  // a replica holds everything
    std::vector<uint64_t> partials(nparts * v->n64);
    for (size_t pi = 0; pi < nparts; pi++) {
        const VecPart& p = v->parts[pi];
        HIP_TRY(hipSetDevice(p.slot->device));
        // the result leaves the scratch block in stream order (the copy below), so the lock ends with the enqueue
        std::lock_guard<std::mutex> lock(p.slot->mu);
        CHECK(ensure_scratch(p.slot, fold_ws + 256));
        char* res = (char*)p.slot->scratch + fold_ws;
        CHECK(call(pi, p, res, p.slot->scratch, fold_ws));
        HIP_TRY(hipMemcpyAsync(partials.data() + pi * v->n64, res, esz, hipMemcpyDeviceToHost, p.slot->stream));
    }
    for (size_t pi = 0; pi < nparts; pi++) {
        HIP_TRY(hipSetDevice(v->parts[pi].slot->device));
        HIP_TRY(hipStreamSynchronize(v->parts[pi].slot->stream));
    }
    if (nparts == 1) {
        std::memcpy(out_host, partials.data(), esz);
        return FPGPU_OK;
    }
    // fold the per-device partials on the first device (g elements: one small launch)
    Slot* s = v->parts[0].slot;
    HIP_TRY(hipSetDevice(s->device));
    std::lock_guard<std::mutex> lock(s->mu);
    CHECK(ensure_scratch(s, fold_ws + 256 + nparts * esz));
    char* res = (char*)s->scratch + fold_ws;
    char* in = res + 256;
    HIP_TRY(hipMemcpyAsync(in, partials.data(), nparts * esz, hipMemcpyHostToDevice, s->stream));
    if (product) CHECK(fpgpu_product_dev(ctx, in, nparts, res, s->scratch, fold_ws, s->stream));
    else CHECK(fpgpu_sum_dev(ctx, in, nparts, res, s->scratch, fold_ws, s->stream));
    HIP_TRY(hipMemcpyAsync(out_host, res, esz, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return FPGPU_OK;
}

}  // namespace

extern "C" {

// ---- device lists --------------------------------------------------------------------------------
static int check_ids(const int* ids, int count, const char* who) {
    if (count < 0 || count > 64 || (count > 0 && !ids)) return fail(FPGPU_ERR_SHAPE, std::string(who) + ": bad device list");
    int ndev = 0;
    if (count > 0) CHECK(device_count(&ndev));
    for (int i = 0; i < count; i++)
        if (ids[i] < 0 || ids[i] >= ndev) return fail(FPGPU_ERR_SHAPE, std::string(who) + ": no such device");
    return FPGPU_OK;
}

int fpgpu_set_devices(const int* ids, int count) {
    CHECK(check_ids(ids, count, "set_devices"));
    std::shared_ptr<DevSet> fresh, old;
    {
        std::lock_guard<std::mutex> lock(g_reg_mu);
        if (count > 0) CHECK(build_set(ids, count, &fresh));
        old = default_set();
        default_set() = fresh;
    }
    // `old` goes here, outside the registry lock (its communicators, unless a vector still holds the list)
    std::lock_guard<std::mutex> coll(g_coll_mu);
    old.reset();
    return FPGPU_OK;
}

int fpgpu_get_devices(int* ids, int capacity) {
    std::lock_guard<std::mutex> lock(g_reg_mu);
    const std::shared_ptr<DevSet>& set = default_set();
    if (!set) return 0;
    for (int i = 0; i < (int)set->list.size() && i < capacity && ids; i++) ids[i] = set->list[i]->device;
    return (int)set->list.size();
}

int fpgpu_devset_create(const int* ids, int count, fpgpu_devset** out) {
    if (!out) return fail(FPGPU_ERR_SHAPE, "null pointer");
    *out = nullptr;
    if (count <= 0) return fail(FPGPU_ERR_SHAPE, "devset_create: empty device list");
    CHECK(check_ids(ids, count, "devset_create"));
    auto h = std::unique_ptr<fpgpu_devset>(new (std::nothrow) fpgpu_devset());
    if (!h) return fail(FPGPU_ERR_HIP, "out of host memory");
    std::lock_guard<std::mutex> lock(g_reg_mu);
    CHECK(build_set(ids, count, &h->set));
    *out = h.release();
    return FPGPU_OK;
}
void fpgpu_devset_destroy(fpgpu_devset* set) {
    if (!set) return;
    std::lock_guard<std::mutex> coll(g_coll_mu);   // the last reference tears the communicators down
    delete set;
}
int fpgpu_devset_devices(const fpgpu_devset* set, int* ids, int capacity) {
    if (!set || !set->set) return 0;
    for (int i = 0; i < (int)set->set->list.size() && i < capacity && ids; i++) ids[i] = set->set->list[i]->device;
    return (int)set->set->list.size();
}

// ---- vectors ----------------------------------------------------------------------------------------
int fpgpu_vec_alloc(const fpgpu_ctx* ctx, size_t n, int placement, fpgpu_vec** out) {
    return vec_alloc(ctx, nullptr, n, 1, placement, out);
}
int fpgpu_vec_alloc_rows(const fpgpu_ctx* ctx, size_t rows, size_t cols, int placement, fpgpu_vec** out) {
    if (cols && rows > ((size_t)1 << 40) / cols) return fail(FPGPU_ERR_SHAPE, "matrix too large");
    return vec_alloc(ctx, nullptr, rows * cols, cols, placement, out);
}
int fpgpu_vec_alloc_on(const fpgpu_ctx* ctx, const fpgpu_devset* set, size_t n, int placement, fpgpu_vec** out) {
    return vec_alloc(ctx, set, n, 1, placement, out);
}
int fpgpu_vec_alloc_rows_on(const fpgpu_ctx* ctx, const fpgpu_devset* set, size_t rows, size_t cols, int placement,
                            fpgpu_vec** out) {
    if (cols && rows > ((size_t)1 << 40) / cols) return fail(FPGPU_ERR_SHAPE, "matrix too large");
    return vec_alloc(ctx, set, rows * cols, cols, placement, out);
}

void fpgpu_vec_free(fpgpu_vec* v) {
    if (!v) return;
    DeviceGuard guard;
    for (VecPart& p : v->parts)
        if (p.ptr) {
            (void)hipSetDevice(p.slot->device);
            (void)hipStreamSynchronize(p.slot->stream);
            (void)hipFree(p.ptr);
        }
    std::lock_guard<std::mutex> coll(g_coll_mu);   // the last reference to a list tears its communicators down
    delete v;
}

size_t fpgpu_vec_len(const fpgpu_vec* v) { return v ? v->n : 0; }
int fpgpu_vec_parts(const fpgpu_vec* v) { return v ? (int)v->parts.size() : 0; }
int fpgpu_vec_part(const fpgpu_vec* v, int i, int* device, size_t* first, size_t* count, void** dev_ptr, void** stream) {
    if (!v || i < 0 || i >= (int)v->parts.size()) return fail(FPGPU_ERR_SHAPE, "vec_part: no such part");
    const VecPart& p = v->parts[i];
    if (device) *device = p.slot->device;
    if (first) *first = p.first;
    if (count) *count = p.count;
    if (dev_ptr) *dev_ptr = p.ptr;
    if (stream) *stream = p.slot->stream;
    return FPGPU_OK;
}

static int vec_upload_bytes(fpgpu_vec* v, const char* host, size_t first, size_t n);
int fpgpu_vec_upload(fpgpu_vec* v, const uint64_t* host, size_t first, size_t n) {
    if (v && v->n64 == 0) return fail(FPGPU_ERR_SHAPE, "vec_upload: not a vector of field elements");
    return vec_upload_bytes(v, (const char*)host, first, n);
}
static int vec_upload_bytes(fpgpu_vec* v, const char* host, size_t first, size_t n) {
    if (!v) return fail(FPGPU_ERR_SHAPE, "null vector");
    if (first > v->n || n > v->n - first) return fail(FPGPU_ERR_SHAPE, "vec_upload: range beyond the vector");
    if (n == 0) return FPGPU_OK;
    if (!host) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    const size_t esz = v->esz();
    // every part that overlaps the range (every copy of a replicated vector), all devices in flight together:
    // each part's copies are ordered behind what its stream holds, and the data is in device memory on return
    std::vector<CopySeg> segs;
    segs.reserve(v->parts.size());   // (std::bad_alloc here would be the process's last gasp anyway; the transfer itself catches)
    for (VecPart& p : v->parts) {
        const size_t lo = std::max(first, p.first), hi = std::min(first + n, p.first + p.count);
        if (lo >= hi) continue;
        segs.push_back(CopySeg{p.slot->device, p.ptr + (lo - p.first) * esz, const_cast<char*>(host) + (lo - first) * esz,
                               (hi - lo) * esz, p.slot->stream});
    }
    return upload_staged_multi(segs.data(), segs.size());
}

static int vec_download_bytes(const fpgpu_vec* v, char* host, size_t first, size_t n);
int fpgpu_vec_download(const fpgpu_vec* v, uint64_t* host, size_t first, size_t n) {
    if (v && v->n64 == 0) return fail(FPGPU_ERR_SHAPE, "vec_download: not a vector of field elements");
    return vec_download_bytes(v, (char*)host, first, n);
}
static int vec_download_bytes(const fpgpu_vec* v, char* host, size_t first, size_t n) {
    if (!v) return fail(FPGPU_ERR_SHAPE, "null vector");
    if (first > v->n || n > v->n - first) return fail(FPGPU_ERR_SHAPE, "vec_download: range beyond the vector");
    if (n == 0) return FPGPU_OK;
    if (!host) return fail(FPGPU_ERR_SHAPE, "null host pointer");
    const size_t esz = v->esz(), nparts = v->kind == VEC_REPLICATED ? 1 : v->parts.size();
    std::vector<CopySeg> segs;
    for (size_t pi = 0; pi < nparts; pi++) {
        const VecPart& p = v->parts[pi];
        const size_t lo = std::max(first, p.first), hi = std::min(first + n, p.first + p.count);
        if (lo >= hi) continue;
        segs.push_back(CopySeg{p.slot->device, p.ptr + (lo - p.first) * esz, host + (lo - first) * esz, (hi - lo) * esz,
                               p.slot->stream});
    }
    return download_staged_multi(segs.data(), segs.size());
}

int fpgpu_vec_sync(const fpgpu_vec* v) {
    if (!v) return fail(FPGPU_ERR_SHAPE, "null vector");
    DeviceGuard guard;
    for (const VecPart& p : v->parts) {
        HIP_TRY(hipSetDevice(p.slot->device));
        HIP_TRY(hipStreamSynchronize(p.slot->stream));
    }
    return FPGPU_OK;
}

// dst <- src for any two layouts of the same length (scatter, gather, re-shard, replicate): every part
// of dst pulls the ranges it needs from the parts of src that hold them, device to device.
int fpgpu_vec_copy(fpgpu_vec* dst, const fpgpu_vec* src) {
    if (!dst || !src) return fail(FPGPU_ERR_SHAPE, "null vector");
    if (dst->n != src->n || dst->n64 != src->n64) return fail(FPGPU_ERR_SHAPE, "vec_copy: length or element width mismatch");
    if (dst == src) return FPGPU_OK;
    std::lock_guard<std::mutex> lock(g_coll_mu);   // an event-ordered protocol over the entries' shared events
    DeviceGuard guard;
    const size_t esz = dst->esz();
    std::vector<Slot*> srcs, dsts;
    for (const VecPart& s : src->parts)
        if (std::find(srcs.begin(), srcs.end(), s.slot) == srcs.end()) srcs.push_back(s.slot);
    for (Slot* s : srcs) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipEventRecord(s->ready, s->stream));
    }
    for (VecPart& d : dst->parts) {
        HIP_TRY(hipSetDevice(d.slot->device));
        // a replicated source serves everything from its copy on this entry, if it has one
        const VecPart* local = nullptr;
        if (src->kind == VEC_REPLICATED)
            for (const VecPart& s : src->parts)
                if (s.slot == d.slot) local = &s;
        const size_t nsrc = src->kind == VEC_REPLICATED ? 1 : src->parts.size();
        for (size_t si = 0; si < nsrc; si++) {
            const VecPart& s = local ? *local : src->parts[si];
            const size_t lo = std::max(d.first, s.first), hi = std::min(d.first + d.count, s.first + s.count);
            if (lo >= hi) continue;
            if (s.slot != d.slot) HIP_TRY(hipStreamWaitEvent(d.slot->stream, s.slot->ready, 0));
            CHECK(copy_d2d(d.ptr + (lo - d.first) * esz, d.slot->device, s.ptr + (lo - s.first) * esz, s.slot->device,
                           (hi - lo) * esz, d.slot->stream));
        }
        if (std::find(dsts.begin(), dsts.end(), d.slot) == dsts.end()) dsts.push_back(d.slot);
    }
    for (Slot* d : dsts) {  // the sources may be overwritten only after every reader is through
        HIP_TRY(hipSetDevice(d->device));
        HIP_TRY(hipEventRecord(d->done, d->stream));
    }
    for (Slot* s : srcs)
        for (Slot* d : dsts)
            if (s != d) {
                HIP_TRY(hipSetDevice(s->device));
                HIP_TRY(hipStreamWaitEvent(s->stream, d->done, 0));
            }
    return FPGPU_OK;
}

int fpgpu_vec_fill_random(const fpgpu_ctx* ctx, fpgpu_vec* v, uint64_t seed) {
    CHECK(check_vec(ctx, v, "v"));
    DeviceGuard guard;
    for (VecPart& p : v->parts) {
        if (p.count == 0) continue;
        HIP_TRY(hipSetDevice(p.slot->device));
        DISPATCH_N64(ctx->n64, HIP_TRY(Launch<N>::fill_random(ctx, p.ptr, p.count, seed, p.first, p.slot->stream)));
        HIP_TRY(hipGetLastError());
    }
    return FPGPU_OK;
}

// ---- elementwise operations on handles ---------------------------------------------------------------
int fpgpu_vec_mul(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* b, fpgpu_vec* out) {
    return vec_binary(fpgpu_mul_dev, ctx, a, b, out);
}
int fpgpu_vec_add(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* b, fpgpu_vec* out) {
    return vec_binary(fpgpu_add_dev, ctx, a, b, out);
}
int fpgpu_vec_sub(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* b, fpgpu_vec* out) {
    return vec_binary(fpgpu_sub_dev, ctx, a, b, out);
}
int fpgpu_vec_sqr(const fpgpu_ctx* ctx, const fpgpu_vec* a, fpgpu_vec* out) { return vec_unary(fpgpu_sqr_dev, ctx, a, out); }
int fpgpu_vec_neg(const fpgpu_ctx* ctx, const fpgpu_vec* a, fpgpu_vec* out) { return vec_unary(fpgpu_neg_dev, ctx, a, out); }
int fpgpu_vec_double(const fpgpu_ctx* ctx, const fpgpu_vec* a, fpgpu_vec* out) { return vec_unary(fpgpu_double_dev, ctx, a, out); }
int fpgpu_vec_div_by_2(const fpgpu_ctx* ctx, const fpgpu_vec* a, fpgpu_vec* out) { return vec_unary(fpgpu_div_by_2_dev, ctx, a, out); }
int fpgpu_vec_to_mont(const fpgpu_ctx* ctx, const fpgpu_vec* a, fpgpu_vec* out) {
    return vec_unary(fpgpu_to_mont_dev, ctx, a, out);
}
int fpgpu_vec_from_mont(const fpgpu_ctx* ctx, const fpgpu_vec* a, fpgpu_vec* out) {
    return vec_unary(fpgpu_from_mont_dev, ctx, a, out);
}

int fpgpu_vec_mul_chain(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* b, fpgpu_vec* out, uint32_t iters) {
    CHECK(check_vec(ctx, a, "a"));
    CHECK(check_vec(ctx, b, "b"));
    CHECK(check_vec(ctx, out, "out"));
    DeviceGuard guard;
    for (size_t pi = 0; pi < out->parts.size(); pi++) {
        const VecPart& o = out->parts[pi];
        const char *pa, *pb;
        CHECK(operand_ptr(a, out, pi, "a", &pa));
        CHECK(operand_ptr(b, out, pi, "b", &pb));
        HIP_TRY(hipSetDevice(o.slot->device));
        CHECK(fpgpu_mul_chain_dev(ctx, pa, pb, o.ptr, o.count, iters, o.slot->stream));
    }
    return FPGPU_OK;
}

int fpgpu_vec_pow(const fpgpu_ctx* ctx, const fpgpu_vec* a, const uint64_t* exponent_le, uint32_t exponent_words,
                  fpgpu_vec* out) {
    CHECK(check_vec(ctx, a, "a"));
    CHECK(check_vec(ctx, out, "out"));
    DeviceGuard guard;
    for (size_t pi = 0; pi < out->parts.size(); pi++) {
        const VecPart& o = out->parts[pi];
        const char* pa;
        CHECK(operand_ptr(a, out, pi, "a", &pa));
        HIP_TRY(hipSetDevice(o.slot->device));
        CHECK(fpgpu_pow_dev(ctx, pa, exponent_le, exponent_words, o.ptr, o.count, o.slot->stream));
    }
    return FPGPU_OK;
}

// ---- batch inversion: every part runs Montgomery's trick on its own range, no collective ----------------
// out = 1 / x, or a / x when `a` is given (FieldConfig::checked_div = mul(a, inv(x)?), src/field.rs:168-170)
static int vec_inv_or_div(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* x, fpgpu_vec* out,
                          uint8_t* none_flags_host, size_t* none_count);
int fpgpu_vec_batch_inv(const fpgpu_ctx* ctx, const fpgpu_vec* x, fpgpu_vec* out, uint8_t* none_flags_host,
                        size_t* none_count) {
    return vec_inv_or_div(ctx, nullptr, x, out, none_flags_host, none_count);
}
int fpgpu_vec_div(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* b, fpgpu_vec* out, uint8_t* none_flags_host,
                  size_t* none_count) {
    CHECK(check_vec(ctx, a, "a"));
    if (a == out) return fail(FPGPU_ERR_SHAPE, "div: out must not alias a");
    return vec_inv_or_div(ctx, a, b, out, none_flags_host, none_count);
}
static int vec_inv_or_div(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* x, fpgpu_vec* out,
                          uint8_t* none_flags_host, size_t* none_count) {
    CHECK(check_vec(ctx, x, "x"));
    CHECK(check_vec(ctx, out, "out"));
    if (x == out) return fail(FPGPU_ERR_SHAPE, "batch_inv: out must not alias x");
    DeviceGuard guard;
    const bool want_flags = none_flags_host || none_count;
    std::vector<uint8_t> tmp;
    uint8_t* flags_host = none_flags_host;
    if (want_flags && !flags_host) {
        tmp.resize(out->n);
        flags_host = tmp.data();
    }
    for (size_t pi = 0; pi < out->parts.size(); pi++) {
        const VecPart& o = out->parts[pi];
        const char *px, *pa = nullptr;
        CHECK(operand_ptr(x, out, pi, "x", &px));
        if (a) CHECK(operand_ptr(a, out, pi, "a", &pa));
        if (o.count == 0) continue;
        HIP_TRY(hipSetDevice(o.slot->device));
        const size_t wsb = (fpgpu_batch_inv_fast_workspace_bytes(ctx, o.count) + 255) & ~(size_t)255;
        std::lock_guard<std::mutex> lock(o.slot->mu);   // workspace and flags live in the entry's scratch block
        CHECK(ensure_scratch(o.slot, wsb + o.count));
        uint8_t* dflags = (uint8_t*)o.slot->scratch + wsb;
        CHECK(fpgpu_batch_inv_dev(ctx, px, o.ptr, dflags, o.count, o.slot->scratch, wsb, o.slot->stream));
        if (a) CHECK(fpgpu_mul_dev(ctx, pa, o.ptr, o.ptr, o.count, o.slot->stream));   // 0 where there is no inverse
        if (want_flags)  // in stream order: the scratch is not reused before this copy has run
            HIP_TRY(hipMemcpyAsync(flags_host + o.first, dflags, o.count, hipMemcpyDeviceToHost, o.slot->stream));
    }
    if (want_flags) {
        for (const VecPart& o : out->parts) {
            HIP_TRY(hipSetDevice(o.slot->device));
            HIP_TRY(hipStreamSynchronize(o.slot->stream));
        }
        if (none_count) {
            size_t c = 0;
            for (size_t i = 0; i < out->n; i++) c += flags_host[i] != 0;
            *none_count = c;
        }
    }
    return FPGPU_OK;
}

// ---- folds and the dot product: a value comes back to the host ----------------------------------------
int fpgpu_vec_sum(const fpgpu_ctx* ctx, const fpgpu_vec* v, uint64_t* out_host) {
    CHECK(check_vec(ctx, v, "v"));
    return vec_reduce(ctx, v, false, out_host, [&](size_t, const VecPart& p, char* res, void* ws, size_t wsb) {
        return fpgpu_sum_dev(ctx, p.ptr, p.count, res, ws, wsb, p.slot->stream);
    });
}
int fpgpu_vec_product(const fpgpu_ctx* ctx, const fpgpu_vec* v, uint64_t* out_host) {
    CHECK(check_vec(ctx, v, "v"));
    return vec_reduce(ctx, v, true, out_host, [&](size_t, const VecPart& p, char* res, void* ws, size_t wsb) {
        return fpgpu_product_dev(ctx, p.ptr, p.count, res, ws, wsb, p.slot->stream);
    });
}
int fpgpu_vec_dot(const fpgpu_ctx* ctx, const fpgpu_vec* a, const fpgpu_vec* b, uint64_t* out_host) {
    CHECK(check_vec(ctx, a, "a"));
    CHECK(check_vec(ctx, b, "b"));
    if (a->n != b->n) return fail(FPGPU_ERR_SHAPE, "dot: operand length mismatch");
    return vec_reduce(ctx, a, false, out_host, [&](size_t pi, const VecPart& p, char* res, void* ws, size_t wsb) {
        const char* pb;
        CHECK(operand_ptr(b, a, pi, "b", &pb));
        return fpgpu_dot_dev(ctx, p.ptr, pb, p.count, res, ws, wsb, p.slot->stream);
    });
}

// ---- matrix products over row blocks -----------------------------------------------------------------
int fpgpu_vec_matvec(const fpgpu_ctx* ctx, const fpgpu_vec* A, size_t rows, size_t cols, const fpgpu_vec* x,
                     fpgpu_vec* y) {
    CHECK(check_vec(ctx, A, "A"));
    CHECK(check_vec(ctx, x, "x"));
    CHECK(check_vec(ctx, y, "y"));
    if (x->n != cols) return fail(FPGPU_ERR_SHAPE, "matvec: x does not hold cols elements");
    return rowblock_product(A, rows, cols, y, 1, "matvec", [&](size_t pi, size_t prows, size_t, char* dst) {
        const VecPart& a = A->parts[pi];
        if (prows == 0) return (int)FPGPU_OK;
        const char* px;
        CHECK(whole_on_slot(x, a.slot, "x", &px));
        size_t wsb = fpgpu_matvec_workspace_bytes(ctx, prows, cols);
        if (wsb && ensure_scratch(a.slot, wsb) != FPGPU_OK) {   // the workspace is optional: without it the VALU kernels run
            wsb = 0;
            (void)hipGetLastError();
        }
        return fpgpu_matvec_dev(ctx, a.ptr, prows, cols, px, dst, wsb ? a.slot->scratch : nullptr, wsb, a.slot->stream);
    });
}

int fpgpu_vec_matmul(const fpgpu_ctx* ctx, const fpgpu_vec* A, const fpgpu_vec* B, fpgpu_vec* C, size_t m, size_t k,
                     size_t n) {
    CHECK(check_vec(ctx, A, "A"));
    CHECK(check_vec(ctx, B, "B"));
    CHECK(check_vec(ctx, C, "C"));
    if (B->n != k * n) return fail(FPGPU_ERR_SHAPE, "matmul: B does not hold k x n elements");
    return rowblock_product(A, m, k, C, n, "matmul", [&](size_t pi, size_t prows, size_t, char* dst) {
        const VecPart& a = A->parts[pi];
        if (prows == 0) return (int)FPGPU_OK;
        const char* pb;
        CHECK(whole_on_slot(B, a.slot, "B", &pb));
        size_t wsb = fpgpu_matmul_workspace_bytes(ctx, prows, k, n);
        if (wsb && ensure_scratch(a.slot, wsb) != FPGPU_OK) {   // optional, and it can be gigabytes (digit planes): VALU kernels otherwise
            wsb = 0;
            (void)hipGetLastError();
        }
        return fpgpu_matmul_dev(ctx, a.ptr, pb, dst, prows, k, n, wsb ? a.slot->scratch : nullptr, wsb, a.slot->stream);
    });
}

// ---- sparse mat-vec over SparseMatrix<T> (src/matrix.rs:46-61) on handles ------------------------------
// The cells' two halves live in two containers with the same row partition: `vals` (rows * density
// elements, fpgpu_vec_alloc_rows(rows, density)) and `col_idx` (rows * density 32-bit column indices).
int fpgpu_idx_alloc_rows_on(const fpgpu_devset* on, size_t rows, size_t density, int placement, fpgpu_idx** out);
int fpgpu_idx_alloc_rows(size_t rows, size_t density, int placement, fpgpu_idx** out) {
    return fpgpu_idx_alloc_rows_on(nullptr, rows, density, placement, out);
}
int fpgpu_idx_alloc_rows_on(const fpgpu_devset* on, size_t rows, size_t density, int placement, fpgpu_idx** out) {
    if (!out) return fail(FPGPU_ERR_SHAPE, "null pointer");
    *out = nullptr;
    if (density && rows > ((size_t)1 << 40) / density) return fail(FPGPU_ERR_SHAPE, "sparse matrix too large");
    fpgpu_vec* v = nullptr;
    CHECK(vec_alloc_n64(on, 0, rows * density, density, placement, &v));
    auto idx = new (std::nothrow) fpgpu_idx();
    if (!idx) {
        fpgpu_vec_free(v);
        return fail(FPGPU_ERR_HIP, "out of host memory");
    }
    idx->v = std::move(*v);   // the parts (device memory) move into the index container
    delete v;
    *out = idx;
    return FPGPU_OK;
}
void fpgpu_idx_free(fpgpu_idx* idx) {
    if (!idx) return;
    auto v = new fpgpu_vec(std::move(idx->v));
    delete idx;
    fpgpu_vec_free(v);
}
int fpgpu_idx_upload(fpgpu_idx* idx, const uint32_t* host, size_t first, size_t n) {
    if (!idx) return fail(FPGPU_ERR_SHAPE, "null index vector");
    return vec_upload_bytes(&idx->v, (const char*)host, first, n);
}

int fpgpu_vec_spmv(const fpgpu_ctx* ctx, const fpgpu_vec* vals, const fpgpu_idx* col_idx, size_t rows, size_t cols,
                   size_t density, const fpgpu_vec* x, fpgpu_vec* y) {
    CHECK(check_vec(ctx, vals, "vals"));
    CHECK(check_vec(ctx, x, "x"));
    CHECK(check_vec(ctx, y, "y"));
    if (!col_idx) return fail(FPGPU_ERR_SHAPE, "null index vector");
    if (x->n != cols) return fail(FPGPU_ERR_SHAPE, "spmv: x does not hold cols elements");
    if (col_idx->v.n != vals->n) return fail(FPGPU_ERR_SHAPE, "spmv: values and column indices differ in length");
    return rowblock_product(vals, rows, density, y, 1, "spmv", [&](size_t pi, size_t prows, size_t, char* dst) {
        const VecPart& a = vals->parts[pi];
        if (prows == 0) return (int)FPGPU_OK;
        const char *pc, *px;
        CHECK(operand_ptr(&col_idx->v, vals, pi, "col_idx", &pc));
        CHECK(whole_on_slot(x, a.slot, "x", &px));
        return fpgpu_spmv_dev(ctx, a.ptr, (const uint32_t*)pc, prows, cols, density, px, dst, a.slot->stream);
    });
}

}  // extern "C"
