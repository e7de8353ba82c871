/* include/fpgpu.h from plain C99 (what cgo, Rust's bindgen or any other FFI generator sees): compiled with
 * -std=c99 -pedantic, linked against libfpgpu.so.
 *   abi_c99 cpu   context and error behaviour; no GPU call
 *   abi_c99 gpu   host-buffer calls, device-vector handles and a chain, checked with field laws
 *                 (a*1 = a, a+0 = a, a*inv(a) = 1, a/b*b = a, half+half = a) -- the oracle comparisons are
 *                 the Python tests' job; this file proves that the boundary is usable from C as declared. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fpgpu.h"

static int checks = 0, failures = 0;
#define CHECK(cond)                                                     \
    do {                                                                \
        checks++;                                                       \
        if (!(cond)) {                                                  \
            failures++;                                                 \
            printf("FAIL %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, fpgpu_last_error_string()); \
        }                                                               \
    } while (0)

/* BN254 Fr, little-endian words */
static const uint64_t P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

static void cpu_part(void)
{
    fpgpu_ctx *ctx = NULL;
    uint64_t m[4], one[4], r2[4], inv = 0, even[4] = {2, 0, 0, 1};
    CHECK(fpgpu_ctx_create(P, 4, &ctx) == FPGPU_OK && ctx != NULL);
    CHECK(fpgpu_ctx_n64(ctx) == 4);
    CHECK(fpgpu_ctx_params(ctx, m, one, r2, &inv) == FPGPU_OK);
    CHECK(memcmp(m, P, sizeof m) == 0);
    CHECK(inv == 0xc2e1f593efffffffull);                     /* -p^-1 mod 2^64 */
    CHECK(one[0] == 0xac96341c4ffffffbull && one[3] == 0x0e0a77c19a07df2full);   /* R mod p */
    CHECK(r2[0] == 0x1bb8e645ae216da7ull && r2[3] == 0x0216d0b17f4e44a5ull);     /* R^2 mod p */
    fpgpu_ctx_destroy(ctx);
    ctx = NULL;
    CHECK(fpgpu_ctx_create(even, 4, &ctx) == FPGPU_ERR_INVALID_MODULUS && ctx == NULL);
    CHECK(fpgpu_ctx_create(P, 9, &ctx) == FPGPU_ERR_UNSUPPORTED);
    CHECK(fpgpu_last_error_string() != NULL && fpgpu_version() != NULL);
    CHECK(fpgpu_static_scratch_bytes() == FPGPU_STATIC_SCRATCH_BYTES);
    if (fpgpu_device_count() <= 0) {                         /* no device: an error, never a CPU fallback */
        uint64_t a[4] = {1, 0, 0, 0}, o[4];
        CHECK(fpgpu_ctx_create(P, 4, &ctx) == FPGPU_OK);
        CHECK(fpgpu_mul(ctx, a, a, o, 1) == FPGPU_ERR_NO_DEVICE);
        CHECK(fpgpu_mul(ctx, a, a, o, 0) == FPGPU_OK);
        fpgpu_ctx_destroy(ctx);
    }
}

static void gpu_part(void)
{
    enum { N = 20011 };
    fpgpu_ctx *ctx = NULL;
    fpgpu_vec *va = NULL, *vb = NULL, *vt = NULL, *vu = NULL;
    uint64_t one[4], *a, *b, *t, *u, seed = 7;
    uint8_t *flags;
    size_t i, none = 0;
    int same;
    CHECK(fpgpu_device_count() >= 1);
    CHECK(fpgpu_ctx_create(P, 4, &ctx) == FPGPU_OK);
    CHECK(fpgpu_ctx_params(ctx, NULL, one, NULL, NULL) == FPGPU_OK);
    a = malloc(N * 32); b = malloc(N * 32); t = malloc(N * 32); u = malloc(N * 32); flags = malloc(N);
    for (i = 0; i < N; i++) {          /* values below 2^253 < p are canonical representatives */
        int l;
        for (l = 0; l < 4; l++) { a[4 * i + l] = splitmix(&seed); b[4 * i + l] = splitmix(&seed); }
        a[4 * i + 3] >>= 3; b[4 * i + 3] >>= 3;
    }
    memset(b + 4 * 5, 0, 32);          /* a zero divisor */
    /* host-buffer calls */
    for (i = 0; i < N; i++) memcpy(t + 4 * i, one, 32);
    CHECK(fpgpu_mul(ctx, a, t, u, N) == FPGPU_OK && memcmp(u, a, N * 32) == 0);               /* a * 1 = a */
    memset(t, 0, N * 32);
    CHECK(fpgpu_add(ctx, a, t, u, N) == FPGPU_OK && memcmp(u, a, N * 32) == 0);               /* a + 0 = a */
    CHECK(fpgpu_div_by_2(ctx, a, t, N) == FPGPU_OK && fpgpu_add(ctx, t, t, u, N) == FPGPU_OK && memcmp(u, a, N * 32) == 0);
    CHECK(fpgpu_double(ctx, a, t, N) == FPGPU_OK && fpgpu_add(ctx, a, a, u, N) == FPGPU_OK && memcmp(u, t, N * 32) == 0);
    CHECK(fpgpu_batch_inv(ctx, b, t, flags, N) == FPGPU_OK && fpgpu_mul(ctx, b, t, u, N) == FPGPU_OK);
    same = 1;
    for (i = 0; i < N; i++) {
        if (i == 5) same = same && flags[i] == 1 && (u[4 * i] | u[4 * i + 1] | u[4 * i + 2] | u[4 * i + 3]) == 0;
        else same = same && flags[i] == 0 && memcmp(u + 4 * i, one, 32) == 0;                 /* b * inv(b) = 1 */
    }
    CHECK(same);
    /* device-vector handles: q = a / b, r = q * b = a wherever b != 0; nothing but a, b and r cross PCIe */
    CHECK(fpgpu_vec_alloc(ctx, N, FPGPU_PLACE_CURRENT, &va) == FPGPU_OK && fpgpu_vec_alloc(ctx, N, FPGPU_PLACE_CURRENT, &vb) == FPGPU_OK);
    CHECK(fpgpu_vec_alloc(ctx, N, FPGPU_PLACE_CURRENT, &vt) == FPGPU_OK && fpgpu_vec_alloc(ctx, N, FPGPU_PLACE_CURRENT, &vu) == FPGPU_OK);
    CHECK(fpgpu_vec_len(va) == N && fpgpu_vec_parts(va) == 1);
    CHECK(fpgpu_vec_upload(va, a, 0, N) == FPGPU_OK && fpgpu_vec_upload(vb, b, 0, N) == FPGPU_OK);
    CHECK(fpgpu_vec_div(ctx, va, vb, vt, flags, &none) == FPGPU_OK && none == 1 && flags[5] == 1);
    CHECK(fpgpu_vec_mul(ctx, vt, vb, vu) == FPGPU_OK && fpgpu_vec_download(vu, u, 0, N) == FPGPU_OK);
    same = 1;
    for (i = 0; i < N; i++)
        if (i != 5) same = same && memcmp(u + 4 * i, a + 4 * i, 32) == 0;
    CHECK(same);
    CHECK(fpgpu_vec_sum(ctx, va, t) == FPGPU_OK && fpgpu_sum(ctx, a, N, u) == FPGPU_OK && memcmp(t, u, 32) == 0);
    CHECK(fpgpu_vec_mul(ctx, va, vb, va) == FPGPU_OK);                       /* in place is fine for elementwise */
    CHECK(fpgpu_vec_div(ctx, va, vb, va, NULL, NULL) == FPGPU_ERR_SHAPE);     /* but not for a division */
    fpgpu_vec_free(va); fpgpu_vec_free(vb); fpgpu_vec_free(vt); fpgpu_vec_free(vu);
    /* a device list as a handle (the process-wide list stays unset): two entries on GPU 0, sharded operands, and the
     * inversion's two workspace sizes through the *_dev call on one part */
    {
        fpgpu_devset *set = NULL;
        int ids[2] = {0, 0}, got[2] = {-1, -1}, dev = -1;
        size_t first = 0, count = 0;
        void *ptr = NULL, *stream = NULL;
        CHECK(fpgpu_devset_create(ids, 2, &set) == FPGPU_OK && fpgpu_devset_devices(set, got, 2) == 2 && got[0] == 0 && got[1] == 0);
        CHECK(fpgpu_get_devices(NULL, 0) == 0);
        CHECK(fpgpu_vec_alloc_on(ctx, set, N, FPGPU_PLACE_SHARDED, &va) == FPGPU_OK && fpgpu_vec_parts(va) == 2);
        CHECK(fpgpu_vec_alloc_on(ctx, set, N, FPGPU_PLACE_SHARDED, &vb) == FPGPU_OK);
        CHECK(fpgpu_vec_alloc_on(ctx, set, N, FPGPU_PLACE_SHARDED, &vt) == FPGPU_OK);
        fpgpu_devset_destroy(set);                                            /* the vectors keep their list */
        CHECK(fpgpu_vec_upload(va, a, 0, N) == FPGPU_OK && fpgpu_vec_upload(vb, b, 0, N) == FPGPU_OK);
        CHECK(fpgpu_vec_batch_inv(ctx, vb, vt, flags, &none) == FPGPU_OK && none == 1 && flags[5] == 1);
        CHECK(fpgpu_vec_mul(ctx, vt, vb, vt) == FPGPU_OK && fpgpu_vec_download(vt, u, 0, N) == FPGPU_OK);
        same = 1;
        for (i = 0; i < N; i++)
            if (i != 5) same = same && memcmp(u + 4 * i, one, 32) == 0;
        CHECK(same);
        CHECK(fpgpu_vec_part(va, 1, &dev, &first, &count, &ptr, &stream) == FPGPU_OK && dev == 0 && first + count == N && ptr != NULL);
        CHECK(fpgpu_batch_inv_fast_workspace_bytes(ctx, N) >= fpgpu_batch_inv_workspace_bytes(ctx, N));
        fpgpu_vec_free(va); fpgpu_vec_free(vb); fpgpu_vec_free(vt);
    }
    fpgpu_ctx_destroy(ctx);
    free(a); free(b); free(t); free(u); free(flags);
}

int main(int argc, char **argv)
{
    cpu_part();
    if (argc > 1 && strcmp(argv[1], "gpu") == 0) gpu_part();
    printf("%s: %d checks, %d failures\n", argc > 1 ? argv[1] : "cpu", checks, failures);
    return failures ? 1 : 0;
}
