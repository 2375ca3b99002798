// gnnpe_pool.hip -- output pool: which piece of the device's address space the emit kernel's output buffers live in.
//
// What was measured (scripts/vmm_probe*.hip, scripts/pool_probe.py): the rate at which a kernel streams
// into a multi-GiB buffer on MI355X differs by up to 25 % from one buffer to the next (5.0 against 6.3-6.5 TB/s for a plain
// 16-byte streaming write; 3.2 against 3.7-3.9 ms for the emit kernel at config 3), whatever the write pattern
// (grid-stride, one moving window, per-block streams, wave pieces all rank the buffers alike), reproducibly for a given
// buffer, and not predictable from its address: with physical memory from hipMemCreate, the SAME chunks mapped at
// another virtual window change class and two different sets of chunks mapped in turn at one window run at the same rate
// -- yet a single 12 GB handle walked through twelve windows of one reservation runs at one rate in all of them, fast in
// one process and slow in the next.  What a caller can rely on is only this: a buffer keeps its class for its lifetime,
// and independent allocations draw their class independently (one fast one in three, bench records of rounds 2 and 3).
//
// So the pool draws: `candidates` independent allocations of the whole output ([pde rows | id rows], one hipMalloc each,
// all alive at once so that they cannot be handed the same memory again), each timed with the real consumer -- the emit
// kernel itself when the context holds a count, a 16-byte non-temporal streaming write otherwise -- the fastest kept,
// the others freed before the call returns.  Round 2 did this inside bench.py only; the product (gnnpe_main,
// offline.py, bench.py) now gets its output buffers here.  Transient memory: candidates x the output size, bounded by
// what hipMemGetInfo reports free; a few milliseconds per candidate, once per pool.
// The reference has nothing to mirror here (its outputs are host vectors, main.cpp:87-96); this is the device-side
// home of `all_paths` / `pde` between the emit kernel and its consumers.
#include "gnnpe_common.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <vector>

struct gnnpe_pool {
    gnnpe_ctx *ctx = nullptr;
    char *base = nullptr;  // the kept allocation
    uint64_t bytes = 0, rows_cap = 0, pde_off = 0, ids_off = 0;
    uint32_t L = 0, D = 0;
    std::vector<float> probe_ms;
    int kept = -1;
    bool probed_with_kernel = false;
};

namespace gnnpe {

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_pool_stream(f4 *__restrict__ dst, uint64_t n)
{
    const f4 v = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(v, &dst[i]);
}

// The same draw for a library-owned buffer (the index image, 22 GB at config 3, written once front to back by the leaf
// kernel): `candidates` allocations alive at once, a streaming write timed into each, the fastest kept.  *out = nullptr
// and an error code when not even one allocation fits.
int draw_device_buffer(gnnpe_ctx *c, uint64_t bytes, uint32_t candidates, void **out)
{
    *out = nullptr;
    size_t free_b = 0, tot_b = 0;
    GNNPE_HIP_TRY(hipMemGetInfo(&free_b, &tot_b));
    const uint32_t K = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(candidates, (uint64_t)free_b / 4 * 3 / std::max<uint64_t>(bytes, 1)));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<void *> cand;
    std::vector<float> ms_of;
    for (uint32_t k = 0; k < K; k++) {
        void *q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        cand.push_back(q);
        float best = 0.f;
        if (K > 1) {
            best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k_pool_stream, dim3(kMaxGrid), dim3(kBlock), 0, c->stream, (f4 *)q, bytes / 16);
                if (hipStreamSynchronize(c->stream) != hipSuccess) break;
                const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0) best = std::min(best, ms);
            }
        }
        ms_of.push_back(best);
        if (c->sw.debug) fprintf(stderr, "[draw] %llu bytes, candidate %u at %p: %.3f ms\n", (unsigned long long)bytes, k, q, best);
    }
    if (cand.empty()) {
        set_error("hipMalloc(%llu) failed", (unsigned long long)bytes);
        return GNNPE_ERR_HIP;
    }
    size_t kept = 0;
    for (size_t k = 1; k < cand.size(); k++)
        if (ms_of[k] < ms_of[kept]) kept = k;
    for (size_t k = 0; k < cand.size(); k++)
        if (k != kept) (void)hipFree(cand[k]);
    *out = cand[kept];
    return GNNPE_OK;
}

void pool_free(gnnpe_pool *p)
{
    if (!p) return;
    if (p->base) {
        gnnpe_forget_emit_pref(p->ctx, p->base, p->bytes);  // the next allocation at this address is another buffer
        (void)hipFree(p->base);
    }
    delete p;
}

}  // namespace gnnpe

using namespace gnnpe;

extern "C" {

int gnnpe_output_pool_create(gnnpe_ctx *c, uint64_t rows_cap, uint32_t L, uint32_t D, uint32_t candidates, gnnpe_pool **out)
{
    GNNPE_REQUIRE(c && out && L >= 1 && L <= 16 && D <= 512, GNNPE_ERR_ARG, "gnnpe_output_pool_create: bad argument");
    const bool calibrate = !(candidates & GNNPE_POOL_NO_CALIBRATION);
    candidates &= ~GNNPE_POOL_NO_CALIBRATION;
    GNNPE_REQUIRE(candidates >= 1 && candidates <= 64, GNNPE_ERR_ARG, "gnnpe_output_pool_create: 1..64 candidate allocations");
    *out = nullptr;
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    GNNPE_REQUIRE(rows_cap <= (1ull << 40), GNNPE_ERR_ARG, "gnnpe_output_pool_create: %llu rows", (unsigned long long)rows_cap);
    const uint64_t MiB2 = 2ull << 20;
    rows_cap = std::max<uint64_t>(rows_cap, 1);
    // [ pde rows | id rows ], each on a 2 MiB boundary
    const uint64_t pde_bytes = (rows_cap * D * 8 + MiB2 - 1) / MiB2 * MiB2, ids_bytes = (rows_cap * L * 4 + MiB2 - 1) / MiB2 * MiB2;
    const uint64_t bytes = pde_bytes + ids_bytes;
    // below half a GiB a buffer's class is not measurable (the probe would be launch latency): take what comes.
    // (GNNPE_TESTING=pool_min_probe_bytes=<n>: testing aid, lowers that bound so that small test graphs exercise the draw)
    const uint64_t min_probe = c->sw.pool_min_probe_bytes;
    uint32_t K = 1;
    if (candidates > 1 && bytes >= min_probe) {
        // every candidate is alive until the choice is made: as many as fit beside a quarter of the free memory
        size_t free_b = 0, tot_b = 0;
        GNNPE_HIP_TRY(hipMemGetInfo(&free_b, &tot_b));
        const uint64_t fit = (uint64_t)free_b / 4 * 3 / bytes;
        K = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(candidates, fit));
    }

    // the probe: the emit kernel into the candidate when the context holds an l = L-1 count that fits, a streaming write otherwise
    int rc = resolve_total(c);
    const bool with_kernel = rc == GNNPE_OK && c->counted && c->l + 1 == L && c->total_paths > 0 &&
                             c->total_paths <= rows_cap && (D == 0 || (c->have_vde && D == L * c->e));
    const bool debug = c->sw.debug;
    // timed on the host around stream synchronisations: the probes are milliseconds long, the launch latency inside the
    // bracket is microseconds and the same for every candidate
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<char *> cand;
    std::vector<float> ms_of;
    rc = GNNPE_OK;
    for (uint32_t k = 0; k < K && rc == GNNPE_OK; k++) {
        void *q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) {
            (void)hipGetLastError();
            if (cand.empty()) {
                set_error("gnnpe_output_pool_create: hipMalloc(%llu) failed", (unsigned long long)bytes);
                rc = GNNPE_ERR_HIP;
            }
            break;  // fewer candidates than asked for is not an error
        }
        char *at = (char *)q;
        cand.push_back(at);
        float best = 0.f;
        if (K > 1) {
            best = 1e30f;
            for (int rep = 0; rep < 3 && rc == GNNPE_OK; rep++) {  // first pass touches the pages (untimed), then best of two
                const auto t0 = std::chrono::steady_clock::now();
                if (with_kernel)
                    rc = gnnpe_fill_paths_device(c, 0, c->total_paths, at + pde_bytes, D ? at : nullptr, nullptr);
                else
                    hipLaunchKernelGGL(k_pool_stream, dim3(kMaxGrid), dim3(kBlock), 0, c->stream, (f4 *)at, bytes / 16);
                if (rc == GNNPE_OK && hipStreamSynchronize(c->stream) != hipSuccess) {
                    set_error("gnnpe_output_pool_create: probe launch failed: %s", hipGetErrorString(hipGetLastError()));
                    rc = GNNPE_ERR_HIP;
                }
                const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0) best = std::min(best, ms);
            }
        }
        ms_of.push_back(best);
        if (debug) fprintf(stderr, "[pool] candidate %u at %p: %.3f ms (%s)\n", k, (void *)at, best, with_kernel ? "emit kernel" : "stream");
    }
    (void)hipStreamSynchronize(c->stream);
    int kept = 0;
    for (size_t k = 1; k < ms_of.size(); k++)
        if (ms_of[k] < ms_of[kept]) kept = (int)k;
    for (size_t k = 0; k < cand.size(); k++)
        if (rc != GNNPE_OK || (int)k != kept) (void)hipFree(cand[k]);
    if (rc != GNNPE_OK) return rc;
    gnnpe_pool *p = new gnnpe_pool();
    p->ctx = c;
    p->base = cand[kept];
    p->bytes = bytes;
    p->rows_cap = rows_cap;
    p->pde_off = 0;
    p->ids_off = pde_bytes;
    p->L = L;
    p->D = D;
    p->probe_ms = ms_of;
    p->kept = kept;
    p->probed_with_kernel = with_kernel && K > 1;
    c->pools.push_back(p);
    *out = p;
    // which emit shape is faster into the buffer that was kept (three shapes timed, nine launches: worth it for a buffer that is
    // filled more than a few times; callers that fill it once per chunk pass GNNPE_POOL_NO_CALIBRATION)
    if (calibrate && with_kernel && bytes >= min_probe && c->l == 2 &&
        (rc = gnnpe_emit_calibrate_device(c, p->rows_cap, p->base + p->ids_off, D ? p->base + p->pde_off : nullptr, nullptr, nullptr)) != GNNPE_OK)
        (void)hipGetLastError();  // a failed calibration leaves the default shape; the pool is usable
    return GNNPE_OK;
}

int gnnpe_output_pool_acquire(gnnpe_pool *p, void **dev_ids, void **dev_pde, uint64_t *rows_cap)
{
    GNNPE_REQUIRE(p && p->base, GNNPE_ERR_ARG, "gnnpe_output_pool_acquire: no pool");
    if (dev_ids) *dev_ids = p->base + p->ids_off;
    if (dev_pde) *dev_pde = p->D ? p->base + p->pde_off : nullptr;
    if (rows_cap) *rows_cap = p->rows_cap;
    return GNNPE_OK;
}

int gnnpe_output_pool_report(gnnpe_pool *p, uint32_t cap, float *probe_ms, uint32_t *n_candidates, uint32_t *kept, int *probed_with_emit_kernel)
{
    GNNPE_REQUIRE(p, GNNPE_ERR_ARG, "gnnpe_output_pool_report: no pool");
    const uint32_t n = (uint32_t)p->probe_ms.size();
    for (uint32_t k = 0; probe_ms && k < std::min(cap, n); k++) probe_ms[k] = p->probe_ms[k];
    if (n_candidates) *n_candidates = n;
    if (kept) *kept = (uint32_t)p->kept;
    if (probed_with_emit_kernel) *probed_with_emit_kernel = p->probed_with_kernel ? 1 : 0;
    return GNNPE_OK;
}

void gnnpe_output_pool_destroy(gnnpe_pool *p)
{
    if (!p) return;
    // (a pool must not outlive its context: gnnpe_destroy frees the pools still registered with it)
    gnnpe_ctx *c = p->ctx;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->pools.erase(std::remove(c->pools.begin(), c->pools.end(), p), c->pools.end());
    pool_free(p);
}

}  // extern "C"
