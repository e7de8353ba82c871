// Stress driver for the HOST side of libfpgpu (handle layer, device sets, staging lanes) -- run by
// tests/test_host_mirror.py against the shipped library and by tools/build_san.sh against an AddressSanitizer build
// of the host layers, on a GPU box.
// Several host threads at once: host-buffer calls large enough for the staging lanes, vectors sharded over a
// device list that names GPU 0 several times (ragged parts), partial uploads / downloads across part boundaries,
// matrix products with replicated outputs (the all-gather path), sparse products on handles, batch inversion through
// the slot's scratch, device sets created and destroyed while their vectors live on.  Every sharded result is compared
// with the same operation on one device; the sanitizer watches the pointer arithmetic and the locking.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fpgpu.h"

static std::atomic<int> g_fail{0}, g_checks{0};
#define CHECK_OK(expr)                                                                                   \
    do {                                                                                                 \
        int rc_ = (expr);                                                                                \
        g_checks++;                                                                                      \
        if (rc_ != FPGPU_OK) {                                                                           \
            g_fail++;                                                                                    \
            std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #expr, rc_, fpgpu_last_error_string()); \
        }                                                                                                \
    } while (0)
#define CHECK_TRUE(cond)                                                        \
    do {                                                                        \
        g_checks++;                                                             \
        if (!(cond)) {                                                          \
            g_fail++;                                                           \
            std::fprintf(stderr, "%s:%d: %s is false\n", __FILE__, __LINE__, #cond); \
        }                                                                       \
    } while (0)

// BN254 Fr
static const uint64_t kP[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

static std::vector<uint64_t> rand_elems(size_t n, uint64_t seed) {
    std::mt19937_64 g(seed);
    std::vector<uint64_t> v(n * 4);
    for (size_t i = 0; i < n; i++) {
        for (int j = 0; j < 4; j++) v[4 * i + j] = g();
        v[4 * i + 3] &= 0x1fffffffffffffffull;   // < 2^253 < p
    }
    return v;
}

static void worker(const fpgpu_ctx* ctx, int t, const fpgpu_devset* shared_set) {
    const size_t sizes[] = {1, 100, 4097, 300007, (size_t)1 << 18, ((size_t)1 << 18) + 5};
    const size_t n = sizes[t % 6];
    const auto a = rand_elems(n, 1000 + t), b = rand_elems(n, 2000 + t);
    std::vector<uint64_t> ref(n * 4), got(n * 4), tmp(n * 4);
    // host-buffer call (>= 4 MiB per operand at the two largest sizes: the staging lanes)
    CHECK_OK(fpgpu_mul(ctx, a.data(), b.data(), ref.data(), n));
    // the same product on handles sharded over this thread's own device set (three entries on GPU 0)
    const int ids[3] = {0, 0, 0};
    fpgpu_devset* own = nullptr;
    CHECK_OK(fpgpu_devset_create(ids, 3, &own));
    const fpgpu_devset* set = (t & 1) ? shared_set : own;
    fpgpu_vec *va = nullptr, *vb = nullptr, *vo = nullptr, *vr = nullptr;
    CHECK_OK(fpgpu_vec_alloc_on(ctx, set, n, FPGPU_PLACE_SHARDED, &va));
    CHECK_OK(fpgpu_vec_alloc_on(ctx, set, n, FPGPU_PLACE_SHARDED, &vb));
    CHECK_OK(fpgpu_vec_alloc_on(ctx, set, n, FPGPU_PLACE_SHARDED, &vo));
    CHECK_OK(fpgpu_vec_alloc_on(ctx, set, n, FPGPU_PLACE_REPLICATED, &vr));
    fpgpu_devset_destroy(own);   // vectors keep their list alive
    // uploads in two pieces that straddle the part boundaries
    const size_t cut = n / 2 + (n > 2 ? 1 : 0);
    CHECK_OK(fpgpu_vec_upload(va, a.data(), 0, cut));
    if (cut < n) CHECK_OK(fpgpu_vec_upload(va, a.data() + 4 * cut, cut, n - cut));
    CHECK_OK(fpgpu_vec_upload(vb, b.data(), 0, n));
    CHECK_OK(fpgpu_vec_mul(ctx, va, vb, vo));
    CHECK_OK(fpgpu_vec_download(vo, got.data(), 0, n));
    CHECK_TRUE(got == ref);
    // replicate, then read an inner range back
    CHECK_OK(fpgpu_vec_copy(vr, vo));
    if (n > 10) {
        CHECK_OK(fpgpu_vec_download(vr, tmp.data(), 3, n - 7));
        CHECK_TRUE(std::memcmp(tmp.data(), ref.data() + 4 * 3, (n - 7) * 32) == 0);
    }
    // out-of-range transfers are refused, not performed
    CHECK_TRUE(fpgpu_vec_upload(va, a.data(), n, 1) == FPGPU_ERR_SHAPE);
    CHECK_TRUE(fpgpu_vec_download(va, tmp.data(), 1, n) == FPGPU_ERR_SHAPE);
    // inversion through the slot's scratch: x * inv(x) = one
    std::vector<uint8_t> none(n);
    size_t none_count = 99;
    CHECK_OK(fpgpu_vec_batch_inv(ctx, vo, va, none.data(), &none_count));
    CHECK_TRUE(none_count == 0);
    CHECK_OK(fpgpu_vec_mul(ctx, vo, va, vb));
    CHECK_OK(fpgpu_vec_download(vb, got.data(), 0, n));
    uint64_t mod[4], one[4], r2[4], ninv;
    CHECK_OK(fpgpu_ctx_params(ctx, mod, one, r2, &ninv));
    bool all_one = true;
    for (size_t i = 0; i < n; i++) all_one = all_one && std::memcmp(&got[4 * i], one, 32) == 0;
    CHECK_TRUE(all_one);
    // folds
    uint64_t s1[4], s2[4];
    CHECK_OK(fpgpu_vec_sum(ctx, vo, s1));
    CHECK_OK(fpgpu_sum(ctx, ref.data(), n, s2));
    CHECK_TRUE(std::memcmp(s1, s2, 32) == 0);
    fpgpu_vec_free(va);
    fpgpu_vec_free(vb);
    fpgpu_vec_free(vo);
    fpgpu_vec_free(vr);

    // matrix products: rows sharded, output replicated (the gather) and sharded; ragged row counts
    const size_t rows = 37 + (size_t)t * 11, cols = 129 + (size_t)t * 7, density = 5;
    const auto A = rand_elems(rows * cols, 3000 + t), x = rand_elems(cols, 4000 + t);
    std::vector<uint64_t> yref(rows * 4), y(rows * 4);
    CHECK_OK(fpgpu_matvec(ctx, A.data(), rows, cols, x.data(), yref.data()));
    fpgpu_vec *vA = nullptr, *vx = nullptr, *vy = nullptr, *vys = nullptr;
    CHECK_OK(fpgpu_vec_alloc_rows_on(ctx, shared_set, rows, cols, FPGPU_PLACE_SHARDED, &vA));
    CHECK_OK(fpgpu_vec_alloc_on(ctx, shared_set, cols, FPGPU_PLACE_REPLICATED, &vx));
    CHECK_OK(fpgpu_vec_alloc_on(ctx, shared_set, rows, FPGPU_PLACE_REPLICATED, &vy));
    CHECK_OK(fpgpu_vec_alloc_rows_on(ctx, shared_set, rows, 1, FPGPU_PLACE_SHARDED, &vys));
    CHECK_OK(fpgpu_vec_upload(vA, A.data(), 0, rows * cols));
    CHECK_OK(fpgpu_vec_upload(vx, x.data(), 0, cols));
    CHECK_OK(fpgpu_vec_matvec(ctx, vA, rows, cols, vx, vy));
    CHECK_OK(fpgpu_vec_download(vy, y.data(), 0, rows));
    CHECK_TRUE(y == yref);
    CHECK_OK(fpgpu_vec_matvec(ctx, vA, rows, cols, vx, vys));
    CHECK_OK(fpgpu_vec_download(vys, y.data(), 0, rows));
    CHECK_TRUE(y == yref);
    // sparse product on handles
    std::mt19937_64 g(5000 + t);
    std::vector<uint32_t> ci(rows * density);
    for (auto& c : ci) c = (uint32_t)(g() % cols);
    const auto vals = rand_elems(rows * density, 6000 + t);
    CHECK_OK(fpgpu_spmv(ctx, vals.data(), ci.data(), rows, cols, density, x.data(), yref.data()));
    fpgpu_vec* vv = nullptr;
    fpgpu_idx* vi = nullptr;
    CHECK_OK(fpgpu_vec_alloc_rows_on(ctx, shared_set, rows, density, FPGPU_PLACE_SHARDED, &vv));
    CHECK_OK(fpgpu_idx_alloc_rows_on(shared_set, rows, density, FPGPU_PLACE_SHARDED, &vi));
    CHECK_OK(fpgpu_vec_upload(vv, vals.data(), 0, rows * density));
    CHECK_OK(fpgpu_idx_upload(vi, ci.data(), 0, rows * density));
    CHECK_OK(fpgpu_vec_spmv(ctx, vv, vi, rows, cols, density, vx, vy));
    CHECK_OK(fpgpu_vec_download(vy, y.data(), 0, rows));
    CHECK_TRUE(y == yref);
    fpgpu_idx_free(vi);
    fpgpu_vec_free(vv);
    fpgpu_vec_free(vA);
    fpgpu_vec_free(vx);
    fpgpu_vec_free(vy);
    fpgpu_vec_free(vys);
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? std::atoi(argv[1]) : 8;
    const int rounds = argc > 2 ? std::atoi(argv[2]) : 2;
    if (fpgpu_device_count() < 1) {
        std::printf("no device: nothing to stress\n");
        return 0;
    }
    fpgpu_ctx* ctx = nullptr;
    CHECK_OK(fpgpu_ctx_create(kP, 4, &ctx));
    const int ids[5] = {0, 0, 0, 0, 0};
    fpgpu_devset* shared = nullptr;
    CHECK_OK(fpgpu_devset_create(ids, 5, &shared));
    for (int r = 0; r < rounds; r++) {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker, ctx, t + r * threads, shared);
        // the process-wide default list changes underneath: vectors laid out over handles do not care
        const int two[2] = {0, 0};
        CHECK_OK(fpgpu_set_devices(two, 2));
        CHECK_OK(fpgpu_set_devices(nullptr, 0));
        for (auto& th : pool) th.join();
    }
    fpgpu_devset_destroy(shared);
    fpgpu_ctx_destroy(ctx);
    std::printf("stress: %d threads x %d rounds, %d checks, %d failures\n", threads, rounds, g_checks.load(), g_fail.load());
    std::fflush(stdout);
    // leave without the runtime's exit-time finalizers: under the sanitizer runtime of this ROCm release they trip an
    // internal check of its own device allocator ("dev_runtime_unloaded_") after everything here has been freed
    std::_Exit(g_fail ? 1 : 0);
}
