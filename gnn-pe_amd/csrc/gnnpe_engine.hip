// gnnpe_engine.hip -- C-ABI (include/gnnpe_hip.h) over the gfx950 kernels in gnnpe_kernels.hip.h.
//
// Host-side orchestration only: buffer ownership, launch configuration, the rocPRIM/hipCUB scans
// and selections that glue the hand-written kernels together.  No CPU fallback exists: if HIP is
// not usable every entry point returns an error.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <map>
#include <mutex>
#include <random>
#include <vector>

#include <chrono>

#include "gnnpe_common.h"
#include "gnnpe_kernels.hip.h"
#include "gnnpe_fill_pairwave.hip.h"
#include "gnnpe_fill_ranked.hip.h"
#include "gnnpe_fill_tiles.hip.h"
#ifdef GNNPE_DIAG
#include "gnnpe_fill_tickets.hip.h"  // emit shape 3: measured, never chosen -- diagnostic builds only
#endif
#include "gnnpe_fill_deep.hip.h"

namespace gnnpe {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct CastU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};

// exclusive sum of cnt uint32 -> uint64 (n items) on the context stream
static int scan_u32_to_u64(gnnpe_ctx *c, const uint32_t *in, uint64_t *out, uint64_t n)
{
    hipcub::TransformInputIterator<uint64_t, CastU64, const uint32_t *> it(in, CastU64());
    size_t tb = 0;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, out, (int64_t)n, c->stream));
    int rc = c->cub_tmp.reserve(tb);
    if (rc) return rc;
    tb = c->cub_tmp.bytes;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, it, out, (int64_t)n, c->stream));
    return GNNPE_OK;
}

static int scan_u32(gnnpe_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n)
{
    size_t tb = 0;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, out, (int64_t)n, c->stream));
    int rc = c->cub_tmp.reserve(tb);
    if (rc) return rc;
    tb = c->cub_tmp.bytes;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, in, out, (int64_t)n, c->stream));
    return GNNPE_OK;
}

static int read_back_u64(gnnpe_ctx *c, const void *dev, size_t bytes, uint64_t *host_word)
{
    *c->h_pinned = 0;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->h_pinned, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    *host_word = *c->h_pinned;
    return GNNPE_OK;
}

static int ensure_rank_arrays(gnnpe_ctx *c)
{
    GNNPE_REQUIRE(c->have_graph, GNNPE_ERR_ARG, "no graph loaded (gnnpe_load_csr / gnnpe_load_rows first)");
    GNNPE_REQUIRE(c->have_order, GNNPE_ERR_ARG, "no processing order (gnnpe_set_order first)");
    return GNNPE_OK;
}

// The count leaves its total on the device (eoff[n_edges]); gnnpe_count_paths_enqueue does not fetch it.  Everything
// that needs the number on the host (range checks, the index build) resolves it here: one 8-byte read-back.
int resolve_total(gnnpe_ctx *c)
{
    if (!c->counted || c->total_known) return GNNPE_OK;
    uint64_t w = 0;
    int rc = read_back_u64(c, c->eoff.as<uint64_t>() + c->n_edges, 8, &w);
    if (rc) return rc;
    c->total_paths = w;
    c->total_known = true;
    return GNNPE_OK;
}

// blocks of a kernel that fit one CU with `dyn_lds` bytes of dynamic LDS (occupancy query, cached per kernel ADDRESS and
// size: every instantiation of a kernel template has the same function-pointer type)
static int blocks_per_cu_at(const void *kernel, size_t dyn_lds)
{
    static std::mutex mu;
    static std::map<std::pair<const void *, size_t>, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({kernel, dyn_lds});
    if (it != cache.end()) return it->second;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kBlock, dyn_lds) != hipSuccess || nb < 1) nb = 4;
    cache[{kernel, dyn_lds}] = nb;
    return nb;
}
template <class K> static int blocks_per_cu(K kernel) { return blocks_per_cu_at(reinterpret_cast<const void *>(kernel), 0); }

// Holding a kernel to `want` workgroups per CU: the dynamic LDS (bytes nobody touches) that does it, from the kernel's own
// static LDS and the device's LDS per CU, checked with the occupancy query.  {0, natural occupancy} when the kernel does not
// reach `want` anyway.
struct OccPlan {
    size_t pad;
    int per_cu;
};
template <class K> static OccPlan occupancy_plan(K kernel_fn, int want, int device)
{
    const void *kernel = reinterpret_cast<const void *>(kernel_fn);
    const int natural = blocks_per_cu_at(kernel, 0);
    if (want <= 0 || natural <= want) return {0, natural};
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, OccPlan> cache;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find({kernel, want});
        if (it != cache.end()) return it->second;
    }
    hipFuncAttributes fa;
    int lds_cu = 0;
    OccPlan plan = {0, natural};
    if (hipFuncGetAttributes(&fa, kernel) == hipSuccess &&
        hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess && lds_cu > 0) {
        // a block size half way between "fits want + 1 times" and "fits want times", less the kernel's own -- NOT the largest
        // that fits `want` times: at 54 112 B per workgroup the occupancy query still answers three per CU and the chip runs
        // two (4.5 ms instead of 3.3: profiles/r05_emit_ab.txt section 5) -- then checked against the occupancy query
        long pad = ((long)lds_cu / want + (long)lds_cu / (want + 1)) / 2 - (long)fa.sharedSizeBytes;
        pad = std::max<long>(0, std::min<long>(pad, 64 * 1024 - (long)fa.sharedSizeBytes - 256));
        pad &= ~255L;
        int got = blocks_per_cu_at(kernel, (size_t)pad);
        while (got > want && pad + 512 < 64 * 1024 - (long)fa.sharedSizeBytes) {
            pad += 512;
            got = blocks_per_cu_at(kernel, (size_t)pad);
        }
        while (got < want && pad >= 512) {
            pad -= 512;
            got = blocks_per_cu_at(kernel, (size_t)pad);
        }
        plan = {(size_t)pad, got};
    }
    std::lock_guard<std::mutex> lock(mu);
    cache[{kernel, want}] = plan;
    return plan;
}

}  // namespace gnnpe

// (diagnostic builds read the environment again at every count and fill, so that the A/B scripts under scripts/ can switch shapes
// and knobs between two launches of one process: GNNPE_DIAG_REREAD below)
gnnpe::Switches gnnpe::read_switches()
{
    Switches w;
    if (const char *ev = getenv("GNNPE_EMIT"))
        w.emit = !strcmp(ev, "tickets") ? 3 : !strcmp(ev, "tiles") ? 2 : !strcmp(ev, "starts") ? 1 : !strcmp(ev, "starts_low") ? 4 : 0;
    if (const char *ev = getenv("GNNPE_DEEP_COUNT")) w.deep_merge = !strcmp(ev, "merge");
    if (const char *ev = getenv("GNNPE_DEEP_EMIT")) w.deep_emit = !strcmp(ev, "slices") ? 1 : !strcmp(ev, "units") ? 2 : 0;
    if (const char *ev = getenv("GNNPE_AUX_WIDE")) w.aux_wide = atoi(ev) != 0;
    if (const char *ev = getenv("GNNPE_DEBUG")) w.debug = atoi(ev) != 0;
    if (const char *ev = getenv("GNNPE_TESTING")) {  // k=v,k=v
        const std::string all(ev);
        size_t at = 0;
        while (at < all.size()) {
            const size_t end = std::min(all.find(',', at), all.size()), eq = all.find('=', at);
            if (eq != std::string::npos && eq < end) {
                const std::string k = all.substr(at, eq - at);
                const uint64_t v = strtoull(all.c_str() + eq + 1, nullptr, 10);
                if (k == "pool_min_probe_bytes") w.pool_min_probe_bytes = v;
                else if (k == "index_keep_bytes") w.index_keep_bytes = v;
                else if (k == "index_max_units") w.index_max_units = v;
            }
            at = end + 1;
        }
    }
    return w;
}

// a buffer's measured emit shape is forgotten when the buffer goes away (another allocation may get its address)
void gnnpe_forget_emit_pref(gnnpe_ctx *c, const void *lo, size_t bytes)
{
    const char *a = static_cast<const char *>(lo);
    for (size_t k = 0; k < c->emit_prefs.size(); k++) {
        const char *q = static_cast<const char *>(c->emit_prefs[k].key);
        if (q >= a && q < a + std::max<size_t>(bytes, 1)) c->emit_prefs.erase(c->emit_prefs.begin() + (long)k--);
    }
}
static void forget_emit_pref(gnnpe_ctx *c, const void *key) { gnnpe_forget_emit_pref(c, key, 1); }

using namespace gnnpe;

extern "C" {

int gnnpe_abi_version(void) { return GNNPE_ABI_VERSION; }
const char *gnnpe_last_error(void) { return g_err; }
const char *gnnpe_fill_kernel_name(void) { return "k_fill_ranked"; }

gnnpe_ctx *gnnpe_create(int device_id)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        set_error("no HIP device available (%s); this engine has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return nullptr;
    }
    if (device_id < 0 || device_id >= ndev) {
        set_error("device %d out of range (have %d)", device_id, ndev);
        return nullptr;
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        set_error("hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
        return nullptr;
    }
    gnnpe_ctx *c = new gnnpe_ctx();
    c->device = device_id;
    c->sw = read_switches();
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->h_pinned, 64 * sizeof(uint64_t))) != hipSuccess) {
        set_error("context setup: %s", hipGetErrorString(e));
        delete c;
        return nullptr;
    }
    c->stream = c->own_stream;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cus > 0) c->num_cus = cus;
    if (c->small.reserve(4096) != GNNPE_OK) {  // every small read-back slot lives here: one allocation for the context's life
        delete c;
        return nullptr;
    }
    return c;
}

void gnnpe_destroy(gnnpe_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (gnnpe_pool *p : c->pools) pool_free(p);  // output pools the caller did not destroy
    c->pools.clear();
    // every DevBuf member releases itself in ~gnnpe_ctx
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int gnnpe_set_stream(gnnpe_ctx *c, void *hip_stream)
{
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return GNNPE_OK;
}

int gnnpe_sync(gnnpe_ctx *c)
{
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    return GNNPE_OK;
}

int gnnpe_rows_held(gnnpe_ctx *c, uint64_t *n_rows, uint64_t *n_entries, uint32_t *n_hub_rows)
{
    GNNPE_REQUIRE(c && c->have_graph, GNNPE_ERR_ARG, "gnnpe_rows_held: no graph");
    if (n_rows) *n_rows = c->n_held;
    if (n_entries) *n_entries = c->nbr_used;
    if (n_hub_rows) *n_hub_rows = c->n_hub;
    return GNNPE_OK;
}

int gnnpe_get_stream(gnnpe_ctx *c, void **hip_stream)
{
    GNNPE_REQUIRE(c && hip_stream, GNNPE_ERR_ARG, "null argument");
    *hip_stream = (void *)c->stream;
    return GNNPE_OK;
}

int gnnpe_copy_device(gnnpe_ctx *c, void *dev_dst, const void *dev_src, uint64_t bytes)
{
    GNNPE_REQUIRE(c && (bytes == 0 || (dev_dst && dev_src)), GNNPE_ERR_ARG, "null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    // hipMemcpyDefault: source and destination may live on different devices of this process (peer copy over xGMI)
    if (bytes) GNNPE_HIP_TRY(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDefault, c->stream));
    return GNNPE_OK;
}

int gnnpe_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int gnnpe_pinned_alloc(uint64_t bytes, void **host_ptr)
{
    GNNPE_REQUIRE(host_ptr, GNNPE_ERR_ARG, "gnnpe_pinned_alloc: null argument");
    *host_ptr = nullptr;
    GNNPE_HIP_TRY(hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return GNNPE_OK;
}

void gnnpe_pinned_free(void *host_ptr)
{
    if (host_ptr) (void)hipHostFree(host_ptr);
}

int gnnpe_dev_alloc(gnnpe_ctx *c, uint64_t bytes, void **dev_ptr)
{
    GNNPE_REQUIRE(c && dev_ptr, GNNPE_ERR_ARG, "null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    GNNPE_HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 16));
    return GNNPE_OK;
}

int gnnpe_dev_free(gnnpe_ctx *c, void *dev_ptr)
{
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    if (dev_ptr) {
        gnnpe_forget_emit_pref(c, dev_ptr, 1);
        GNNPE_HIP_TRY(hipFree(dev_ptr));
    }
    return GNNPE_OK;
}

int gnnpe_copy_to_host(gnnpe_ctx *c, void *host_dst, const void *dev_src, uint64_t bytes)
{
    GNNPE_REQUIRE(c && (bytes == 0 || (host_dst && dev_src)), GNNPE_ERR_ARG, "null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    if (bytes) GNNPE_HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    return GNNPE_OK;
}

static int alloc_vertex_arrays(gnnpe_ctx *c, uint32_t n)
{
    int rc;
    if ((rc = c->adj_start.reserve((size_t)(n + 1) * 4))) return rc;
    if ((rc = c->adj_deg.reserve((size_t)(n + 1) * 4))) return rc;
    if ((rc = c->present.reserve((size_t)n + 1))) return rc;
    if ((rc = c->labels.reserve((size_t)(n + 1) * 4))) return rc;
    GNNPE_HIP_TRY(hipMemsetAsync(c->adj_start.p, 0, (size_t)(n + 1) * 4, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(c->adj_deg.p, 0, (size_t)(n + 1) * 4, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(c->present.p, 0, (size_t)n + 1, c->stream));
    return GNNPE_OK;
}

static void invalidate_derived(gnnpe_ctx *c)
{
    c->halo_min_rank = 0;  // called by the loaders only: a freshly loaded graph has no halo rows
    c->multigraph = false;
    c->have_vde = false;
    c->have_deg_all = false;
    c->aux_vdl_valid = false;
    c->have_pge = false;
    c->nbr_vde_valid = false;
    c->counted = false;
    c->slab_struct_valid = false;
    c->labels_checked = false;
}

// Validation and the graph-only derived structure of the rows [first held row .., +n_new): reverse
// positions and the hub-row list.  Synchronises the stream.
static int finish_rows(gnnpe_ctx *c, uint64_t n_new, const uint32_t *dev_new_rows)
{
    int rc;
    if ((rc = c->small.reserve(1024))) return rc;
    uint32_t *d_bad = c->small.as<uint32_t>() + 8;
    uint32_t *d_cnt = c->small.as<uint32_t>() + 10;
    if (n_new) {
        GNNPE_HIP_TRY(hipMemsetAsync(d_bad, 0xFF, 4, c->stream));
        hipLaunchKernelGGL(k_validate_rows, dim3(grid_for(n_new * 16)), dim3(kBlock), 0, c->stream, c->n, n_new, dev_new_rows,
                           c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(), d_bad);
        GNNPE_HIP_TRY(hipGetLastError());
        uint64_t bad = 0;
        if ((rc = read_back_u64(c, d_bad, 4, &bad))) return rc;
        GNNPE_REQUIRE((uint32_t)bad == 0xFFFFFFFFu, GNNPE_ERR_ARG,
                      "adjacency row of vertex %u is not a strictly ascending list of ids < n without a self-loop "
                      "(the engine needs a simple graph with sorted rows, as graph.cpp:231-233 leaves them)", (uint32_t)bad);
        hipLaunchKernelGGL(k_revpos, dim3(grid_for(n_new * 16)), dim3(kBlock), 0, c->stream, (uint32_t)n_new, dev_new_rows,
                           c->owned.as<uint8_t>(), c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(),
                           c->nbrs.as<uint32_t>(), c->revpos.as<uint32_t>());
        GNNPE_HIP_TRY(hipGetLastError());
    }
    // hub rows over everything held
    const uint32_t *held = c->rows_identity ? nullptr : c->held.as<uint32_t>();
    c->rblock_valid = false;
    c->n_hub = 0;
    c->hub_entries = 0;
    if (c->n_held) {
        GNNPE_HIP_TRY(hipMemsetAsync(d_cnt, 0, 4, c->stream));
        hipLaunchKernelGGL(k_hub_list, dim3(grid_for(c->n_held)), dim3(kBlock), 0, c->stream, (uint64_t)c->n_held, held,
                           c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), kHubDegree, d_cnt, (uint32_t *)nullptr,
                           (uint32_t *)nullptr, (uint32_t *)nullptr);
        uint64_t nh = 0;
        if ((rc = read_back_u64(c, d_cnt, 4, &nh))) return rc;
        if ((uint32_t)nh) {
            const size_t by = ((size_t)(uint32_t)nh + 1) * 4;
            if ((rc = c->hub_rows.reserve(by)) || (rc = c->hub_beg.reserve(by)) || (rc = c->hub_end.reserve(by))) return rc;
            GNNPE_HIP_TRY(hipMemsetAsync(d_cnt, 0, 4, c->stream));
            hipLaunchKernelGGL(k_hub_list, dim3(grid_for(c->n_held)), dim3(kBlock), 0, c->stream, (uint64_t)c->n_held, held,
                               c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), kHubDegree, d_cnt,
                               c->hub_rows.as<uint32_t>(), c->hub_beg.as<uint32_t>(), c->hub_end.as<uint32_t>());
            GNNPE_HIP_TRY(hipGetLastError());
            c->n_hub = (uint32_t)nh;
            // longest rows first (and a deterministic order: the list was filled through an atomic counter): the kernels that
            // give a hub row to a lane or a wave then work on rows of similar length side by side
            const uint32_t nhub = (uint32_t)nh;
            if ((rc = c->scratch.reserve((size_t)nhub * 16 + 64))) return rc;
            uint32_t *k_in = c->scratch.as<uint32_t>(), *k_out = k_in + nhub, *v_out = k_out + nhub;
            hipLaunchKernelGGL(k_hub_sort_keys, dim3(grid_for(nhub)), dim3(kBlock), 0, c->stream, nhub, c->hub_rows.as<uint32_t>(),
                               c->adj_deg.as<uint32_t>(), k_in);
            // two passes: by row id ascending, then (stable) by degree descending
            size_t tb = 0;
            GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, c->hub_rows.as<uint32_t>(), v_out, (int)nhub, 0, 32, c->stream));
            if ((rc = c->cub_tmp.reserve(tb))) return rc;
            tb = c->cub_tmp.bytes;
            GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(c->cub_tmp.p, tb, c->hub_rows.as<uint32_t>(), v_out, (int)nhub, 0, 32, c->stream));
            hipLaunchKernelGGL(k_hub_sort_keys, dim3(grid_for(nhub)), dim3(kBlock), 0, c->stream, nhub, v_out, c->adj_deg.as<uint32_t>(), k_in);
            tb = 0;
            GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, tb, k_in, k_out, v_out, c->hub_rows.as<uint32_t>(), (int)nhub, 0, 32, c->stream));
            if ((rc = c->cub_tmp.reserve(tb))) return rc;
            tb = c->cub_tmp.bytes;
            GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(c->cub_tmp.p, tb, k_in, k_out, v_out, c->hub_rows.as<uint32_t>(), (int)nhub, 0, 32, c->stream));
            hipLaunchKernelGGL(k_hub_bounds, dim3(grid_for(nhub)), dim3(kBlock), 0, c->stream, nhub, c->hub_rows.as<uint32_t>(),
                               c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->hub_beg.as<uint32_t>(), c->hub_end.as<uint32_t>());
            GNNPE_HIP_TRY(hipGetLastError());
        }
    }
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    return GNNPE_OK;
}

int gnnpe_load_csr(gnnpe_ctx *c, uint32_t n, const uint32_t *offs, const uint32_t *nbrs, const uint32_t *labels)
{
    GNNPE_REQUIRE(c && offs && labels && (nbrs || offs[n] == 0), GNNPE_ERR_ARG, "gnnpe_load_csr: null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint64_t m2 = offs[n];
    for (uint32_t i = 0; i < n; i++)
        GNNPE_REQUIRE(offs[i] <= offs[i + 1], GNNPE_ERR_ARG, "offsets not monotone at %u", i);
    int rc;
    if ((rc = alloc_vertex_arrays(c, n))) return rc;
    if ((rc = c->nbrs.reserve((m2 + 1) * 4))) return rc;
    if ((rc = c->scratch.reserve((size_t)(n + 1) * 4))) return rc;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->scratch.p, offs, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (m2) GNNPE_HIP_TRY(hipMemcpyAsync(c->nbrs.p, nbrs, m2 * 4, hipMemcpyHostToDevice, c->stream));
    if (n) GNNPE_HIP_TRY(hipMemcpyAsync(c->labels.p, labels, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (n)
        hipLaunchKernelGGL(k_offsets_to_start_deg, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, n,
                           c->scratch.as<uint32_t>(), c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(),
                           c->present.as<uint8_t>());
    GNNPE_HIP_TRY(hipGetLastError());
    if ((rc = c->owned.reserve((size_t)n + 1)) || (rc = c->revpos.reserve((m2 + 1) * 4))) return rc;
    if (n) GNNPE_HIP_TRY(hipMemcpyAsync(c->owned.p, c->present.p, n, hipMemcpyDeviceToDevice, c->stream));
    c->have_graph = false;
    c->n = n;
    c->rows_identity = true;
    c->n_rows = n;
    c->n_held = n;
    c->nbr_used = c->nbr_owned = m2;
    c->nbr_cap = c->nbrs.bytes / 4;
    // the neighbours' labels, reverse positions and the hub list are part of the graph structure
    if ((rc = c->nbr_label.reserve((m2 + 1) * 4))) return rc;
    if ((rc = finish_rows(c, n, nullptr))) return rc;  // validates the neighbour ids first
    if (m2)
        hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(m2)), dim3(kBlock), 0, c->stream, m2, c->nbrs.as<uint32_t>(),
                           c->labels.as<uint32_t>(), c->nbr_label.as<uint32_t>());
    GNNPE_HIP_TRY(hipGetLastError());
    c->have_graph = true;
    c->slab_begin = 0;  // a new graph starts with the whole order as its slab (no stale slab)
    c->slab_end = n;
    c->slab_set = false;
    c->have_order = false;
    invalidate_derived(c);
    return GNNPE_OK;
}

int gnnpe_load_rows(gnnpe_ctx *c, uint32_t n, const uint32_t *labels, uint32_t n_rows, const uint32_t *rows,
                    const uint64_t *row_offsets, const uint32_t *row_nbrs, uint64_t nbr_capacity)
{
    GNNPE_REQUIRE(c && labels && (n_rows == 0 || (rows && row_offsets)), GNNPE_ERR_ARG,
                  "gnnpe_load_rows: null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint64_t used = n_rows ? row_offsets[n_rows] : 0;
    GNNPE_REQUIRE(used == 0 || row_nbrs, GNNPE_ERR_ARG, "gnnpe_load_rows: null neighbour buffer");
    for (uint32_t k = 0; k < n_rows; k++)
        GNNPE_REQUIRE(rows[k] < n && row_offsets[k] <= row_offsets[k + 1], GNNPE_ERR_ARG, "bad row %u", k);
    const uint64_t cap = std::max<uint64_t>(nbr_capacity, used) + 1;
    GNNPE_REQUIRE(cap < (1ull << 32), GNNPE_ERR_RANGE, "neighbour buffer of %llu entries exceeds 32-bit addressing",
                  (unsigned long long)cap);
    int rc;
    if ((rc = alloc_vertex_arrays(c, n))) return rc;
    if ((rc = c->nbrs.reserve(cap * 4))) return rc;
    if ((rc = c->rows.reserve((size_t)(n_rows + 1) * 4))) return rc;
    if ((rc = c->held.reserve((size_t)(n + 1) * 4))) return rc;
    if ((rc = c->scratch.reserve((size_t)(n_rows + 1) * 8))) return rc;
    if (n) GNNPE_HIP_TRY(hipMemcpyAsync(c->labels.p, labels, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (n_rows) {
        GNNPE_HIP_TRY(hipMemcpyAsync(c->rows.p, rows, (size_t)n_rows * 4, hipMemcpyHostToDevice, c->stream));
        GNNPE_HIP_TRY(hipMemcpyAsync(c->held.p, rows, (size_t)n_rows * 4, hipMemcpyHostToDevice, c->stream));
        GNNPE_HIP_TRY(hipMemcpyAsync(c->scratch.p, row_offsets, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice,
                                     c->stream));
        if (used) GNNPE_HIP_TRY(hipMemcpyAsync(c->nbrs.p, row_nbrs, used * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_install_rows, dim3(grid_for(n_rows)), dim3(kBlock), 0, c->stream, (uint64_t)n_rows,
                           c->rows.as<uint32_t>(), c->scratch.as<uint64_t>(), (uint64_t)0,
                           c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->present.as<uint8_t>());
        GNNPE_HIP_TRY(hipGetLastError());
    }
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    if ((rc = c->owned.reserve((size_t)n + 1)) || (rc = c->revpos.reserve(c->nbrs.bytes))) return rc;
    // same stream as the kernels that read it: a null-stream copy would not order against c->stream
    if (n) GNNPE_HIP_TRY(hipMemcpyAsync(c->owned.p, c->present.p, n, hipMemcpyDeviceToDevice, c->stream));
    c->have_graph = false;
    c->n = n;
    c->rows_identity = false;
    c->n_rows = n_rows;
    c->n_held = n_rows;
    c->nbr_used = c->nbr_owned = used;
    c->nbr_cap = c->nbrs.bytes / 4;
    if ((rc = c->nbr_label.reserve((used + 1) * 4))) return rc;
    if ((rc = finish_rows(c, n_rows, c->rows.as<uint32_t>()))) return rc;  // validates the neighbour ids first
    if (used)
        hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(used)), dim3(kBlock), 0, c->stream, used, c->nbrs.as<uint32_t>(),
                           c->labels.as<uint32_t>(), c->nbr_label.as<uint32_t>());
    GNNPE_HIP_TRY(hipGetLastError());
    c->have_graph = true;
    c->slab_begin = 0;
    c->slab_end = n;
    c->slab_set = false;
    c->have_order = false;
    invalidate_derived(c);
    return GNNPE_OK;
}

int gnnpe_set_multigraph_rows(gnnpe_ctx *c, uint32_t n_rows, const uint64_t *row_offsets, const uint32_t *row_nbrs)
{
    GNNPE_REQUIRE(c && c->have_graph && row_offsets, GNNPE_ERR_ARG, "gnnpe_set_multigraph_rows: load the simple rows first");
    GNNPE_REQUIRE(n_rows == c->n_rows, GNNPE_ERR_ARG, "gnnpe_set_multigraph_rows: %u rows given, %u loaded", n_rows, c->n_rows);
    GNNPE_REQUIRE(c->n_held == c->n_rows, GNNPE_ERR_ARG, "gnnpe_set_multigraph_rows: call it before halo rows are appended");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    for (uint32_t k = 0; k < n_rows; k++)
        GNNPE_REQUIRE(row_offsets[k] <= row_offsets[k + 1], GNNPE_ERR_ARG, "gnnpe_set_multigraph_rows: offsets not monotone at row %u", k);
    const uint64_t used = row_offsets[n_rows] - row_offsets[0];
    GNNPE_REQUIRE(used == 0 || row_nbrs, GNNPE_ERR_ARG, "gnnpe_set_multigraph_rows: null neighbour buffer");
    GNNPE_REQUIRE(used < (1ull << 32), GNNPE_ERR_RANGE, "gnnpe_set_multigraph_rows: %llu entries exceed 32-bit addressing",
                  (unsigned long long)used);
    int rc;
    if ((rc = c->mg_off.reserve(((size_t)n_rows + 1) * 8)) || (rc = c->mg_label.reserve((used + 1) * 4)) ||
        (rc = c->scratch.reserve((used + 1) * 4)) || (rc = c->small.reserve(1024)))
        return rc;
    std::vector<uint64_t> off0(row_offsets, row_offsets + n_rows + 1);
    for (auto &o : off0) o -= row_offsets[0];
    uint32_t *d_nbr = c->scratch.as<uint32_t>(), *d_bad = c->small.as<uint32_t>() + 8;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->mg_off.p, off0.data(), off0.size() * 8, hipMemcpyHostToDevice, c->stream));
    if (used) GNNPE_HIP_TRY(hipMemcpyAsync(d_nbr, row_nbrs + row_offsets[0], used * 4, hipMemcpyHostToDevice, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(d_bad, 0xFF, 4, c->stream));
    const uint32_t *rows = c->rows_identity ? nullptr : c->rows.as<uint32_t>();
    if (n_rows)
        hipLaunchKernelGGL(k_check_multi_rows, dim3(grid_for(n_rows)), dim3(kBlock), 0, c->stream, c->n, n_rows, rows,
                           c->mg_off.as<uint64_t>(), d_nbr, c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(),
                           c->nbrs.as<uint32_t>(), d_bad);
    GNNPE_HIP_TRY(hipGetLastError());
    uint64_t bad = 0;
    if ((rc = read_back_u64(c, d_bad, 4, &bad))) return rc;  // (synchronises: off0 may go)
    GNNPE_REQUIRE((uint32_t)bad == 0xFFFFFFFFu, GNNPE_ERR_ARG,
                  "gnnpe_set_multigraph_rows: row %u is not an ascending list of ids < n (the vertex itself excluded) whose distinct "
                  "entries are the loaded row (graph.cpp:211-233 / custom.h:68-77)", (uint32_t)bad);
    if (used)
        hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(used)), dim3(kBlock), 0, c->stream, used, d_nbr, c->labels.as<uint32_t>(),
                           c->mg_label.as<uint32_t>());
    if (c->rows_identity && n_rows == c->n) {
        // `degree` is the stored row's length (graph.h:154-156): what the auxiliary index and the online filter compare
        if ((rc = c->deg_all.reserve(((size_t)c->n + 1) * 4))) return rc;
        if (c->n)
            hipLaunchKernelGGL(k_row_lengths_u64, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, c->n, c->mg_off.as<uint64_t>(),
                               c->deg_all.as<uint32_t>());
        c->have_deg_all = true;
        c->aux_vdl_valid = false;
    }
    GNNPE_HIP_TRY(hipGetLastError());
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    c->multigraph = true;
    c->have_vde = false;
    c->nbr_vde_valid = false;
    c->have_pge = false;
    return GNNPE_OK;
}

int gnnpe_set_order(gnnpe_ctx *c, const uint32_t *sorted_nodes, const uint32_t *membership, uint32_t p)
{
    GNNPE_REQUIRE(c && sorted_nodes && membership, GNNPE_ERR_ARG, "gnnpe_set_order: null argument");
    GNNPE_REQUIRE(c->have_graph, GNNPE_ERR_ARG, "gnnpe_set_order: load the graph first");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = c->n;
    // membership.txt must list every vertex exactly once (main.cpp:80-85 never checks; we do)
    {
        std::vector<uint8_t> seen(n, 0);
        for (uint32_t i = 0; i < n; i++) {
            GNNPE_REQUIRE(sorted_nodes[i] < n && !seen[sorted_nodes[i]], GNNPE_ERR_ARG,
                          "processing order is not a permutation (entry %u = %u)", i, sorted_nodes[i]);
            seen[sorted_nodes[i]] = 1;
            GNNPE_REQUIRE(membership[i] < p, GNNPE_ERR_ARG, "membership[%u] = %u >= p = %u", i, membership[i], p);
        }
    }
    int rc;
    if ((rc = c->sorted.reserve((size_t)(n + 1) * 4))) return rc;
    if ((rc = c->rank.reserve((size_t)(n + 1) * 4))) return rc;
    if ((rc = c->member.reserve((size_t)(n + 1) * 4))) return rc;
    if (n) {
        GNNPE_HIP_TRY(hipMemcpyAsync(c->sorted.p, sorted_nodes, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        GNNPE_HIP_TRY(hipMemcpyAsync(c->member.p, membership, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_invert_order, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, n, c->sorted.as<uint32_t>(),
                           c->rank.as<uint32_t>());
        GNNPE_HIP_TRY(hipGetLastError());
    }
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    c->p = p;
    c->have_order = true;
    c->counted = false;
    c->slab_struct_valid = false;
    return GNNPE_OK;
}

int gnnpe_set_slab(gnnpe_ctx *c, uint32_t begin, uint32_t end)
{
    GNNPE_REQUIRE(c && c->have_graph, GNNPE_ERR_ARG, "gnnpe_set_slab: load the graph first");
    GNNPE_REQUIRE(begin <= end && end <= c->n, GNNPE_ERR_ARG, "slab [%u,%u) outside [0,%u]", begin, end, c->n);
    // halo rows installed with min_rank (gnnpe_rows_append) hold only their entries ranked >= min_rank: a slab that starts
    // earlier would count paths over incomplete rows
    GNNPE_REQUIRE(begin >= c->halo_min_rank || begin == end, GNNPE_ERR_ARG,
                  "slab [%u,%u) starts before rank %u, below which the halo rows on this device were truncated "
                  "(gnnpe_rows_drop_halo and fetch them again)", begin, end, c->halo_min_rank);
    c->slab_begin = begin;
    c->slab_end = end;
    c->slab_set = true;
    c->counted = false;
    c->slab_struct_valid = false;
    return GNNPE_OK;
}

int gnnpe_set_label_table(gnnpe_ctx *c, uint32_t n_labels, uint32_t e, const double *x_table)
{
    GNNPE_REQUIRE(c && x_table && n_labels > 0, GNNPE_ERR_ARG, "gnnpe_set_label_table: null/empty table");
    GNNPE_REQUIRE(e >= 1 && e <= 32, GNNPE_ERR_UNSUPPORTED, "embedding dimension e=%u outside [1,32]", e);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->xtab.reserve((size_t)n_labels * e * 8))) return rc;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->xtab.p, x_table, (size_t)n_labels * e * 8, hipMemcpyHostToDevice, c->stream));
    // rank form of the table (see gnnpe_common.h): per dimension the distinct values in ascending order
    std::vector<uint16_t> xr;
    std::vector<double> xs((size_t)n_labels * e, 0.0);
    if (n_labels <= 65536) {
        xr.assign((size_t)n_labels * e, 0);
        std::vector<double> col(n_labels);
        for (uint32_t k = 0; k < e; k++) {
            for (uint32_t l = 0; l < n_labels; l++) col[l] = x_table[(size_t)l * e + k];
            std::sort(col.begin(), col.end());
            const size_t nd = std::unique(col.begin(), col.end()) - col.begin();
            for (size_t r = 0; r < nd; r++) xs[(size_t)k * n_labels + r] = col[r];
            for (uint32_t l = 0; l < n_labels; l++)
                xr[(size_t)l * e + k] = (uint16_t)(std::lower_bound(col.begin(), col.begin() + nd, x_table[(size_t)l * e + k]) - col.begin());
        }
        if ((rc = c->xrank.reserve(xr.size() * 2 + 16)) || (rc = c->xsorted.reserve(xs.size() * 8 + 16))) return rc;
        GNNPE_HIP_TRY(hipMemcpyAsync(c->xrank.p, xr.data(), xr.size() * 2, hipMemcpyHostToDevice, c->stream));
        GNNPE_HIP_TRY(hipMemcpyAsync(c->xsorted.p, xs.data(), xs.size() * 8, hipMemcpyHostToDevice, c->stream));
    }
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_labels = n_labels;
    c->e = e;
    c->have_table = true;
    c->labels_checked = false;
    c->counted = false;  // record layouts of the enumeration depend on e
    c->have_vde = false;
    c->nbr_vde_valid = false;
    return GNNPE_OK;
}

// R3: gen_vde_x (custom.h:492-511) with the same libstdc++ engine/distribution classes the reference
// is compiled against; depends only on the label, so it is tabulated once on the host.
int gnnpe_host_label_table(uint32_t n_labels, uint32_t e, double *out)
{
    GNNPE_REQUIRE(out && e >= 1, GNNPE_ERR_ARG, "gnnpe_host_label_table: null output / e=0");
    for (uint32_t label = 0; label < n_labels; label++) {
        std::mt19937 gen(label);
        std::uniform_real_distribution<double> dis(0.0, 1.0);
        double *row = out + (size_t)label * e;
        for (uint32_t k = 0; k < e; k++) row[k] = dis(gen);
        double sum = 0.0;
        for (uint32_t k = 0; k < e; k++) sum += row[k];
        for (uint32_t k = 0; k < e; k++) row[k] = row[k] / sum;
    }
    return GNNPE_OK;
}

enum { kVarPairWave = 1, kVarRanked = 4, kVarDeep = 5 };
static bool fast_e(uint32_t e) { return e == 1 || e == 2 || e == 3 || e == 4 || e == 8; }

// ---- R4 ---------------------------------------------------------------------------------------
static int run_vde(gnnpe_ctx *c)
{
    GNNPE_REQUIRE(c->have_graph && c->have_table, GNNPE_ERR_ARG, "gnnpe_vde: need graph and label table");
    const uint32_t n = c->n, e = c->e;
    int rc;
    const size_t bytes = (size_t)n * e * 8 + 16;
    if ((rc = c->x.reserve(bytes)) || (rc = c->nx.reserve(bytes)) || (rc = c->vde.reserve(bytes))) return rc;
    if (n == 0) {
        c->have_vde = true;
        return GNNPE_OK;
    }
    // every label must index the table (its size is the loader's labels_count, graph.cpp:223); checked once per
    // label / table upload, not per step
    if (!c->labels_checked) {
        size_t tb = 0;
        if ((rc = c->small.reserve(1024))) return rc;
        uint32_t *d_max = c->small.as<uint32_t>();
        GNNPE_HIP_TRY(hipcub::DeviceReduce::Max(nullptr, tb, c->labels.as<uint32_t>(), d_max, (int)n, c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        tb = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceReduce::Max(c->cub_tmp.p, tb, c->labels.as<uint32_t>(), d_max, (int)n, c->stream));
        uint64_t mx = 0;
        if ((rc = read_back_u64(c, d_max, 4, &mx))) return rc;
        GNNPE_REQUIRE((uint32_t)mx < c->n_labels, GNNPE_ERR_ARG, "label %u has no row in the %u-row label table",
                      (uint32_t)mx, c->n_labels);
        c->labels_checked = true;
    }
    // (with the whole graph loaded every vertex' row is summed below -- k_vde or k_vde_hubs -- and nothing needs clearing)
    if (!(c->rows_identity && c->n_rows == n)) {
        GNNPE_HIP_TRY(hipMemsetAsync(c->nx.p, 0, bytes, c->stream));
        GNNPE_HIP_TRY(hipMemsetAsync(c->vde.p, 0, bytes, c->stream));
    }
    // with every vertex a row of this context k_vde writes x itself (and, when the slab's pair offsets are current, the count
    // kernel's per-vertex gather records: k_pack_vinfo's work) -- two launches less per step at config 3
    const bool whole = c->rows_identity && c->n_rows == n && !c->multigraph;
    if (!whole)
        hipLaunchKernelGGL(k_x_from_labels, dim3(grid_for((uint64_t)n * e)), dim3(kBlock), 0, c->stream, n, e,
                           c->labels.as<uint32_t>(), c->xtab.as<double>(), c->x.as<double>());
    double *x_out = whole ? c->x.as<double>() : nullptr;
    VinfoPack vp = {nullptr, nullptr, nullptr, 0u, 0u};
    c->vinfo_fused = false;
    if (whole && c->n_hub == 0 && c->have_order && c->slab_struct_valid && c->fill_variant == kVarRanked && fast_e(e) &&
        c->vinfo.bytes >= ((size_t)n + 1) * GNNPE_VINFO_STRIDE(e) * 8 && c->poffs.p) {
        vp = {c->vinfo.as<double>(), c->rank.as<uint32_t>(), c->poffs.as<uint32_t>(), c->slab_begin, c->slab_end};
        c->vinfo_fused = true;
        c->vinfo_gen = c->slab_struct_gen;
    }
    const uint32_t nr = c->n_rows;
    if (nr && c->multigraph) {  // (x came from k_x_from_labels above: `whole` is false for a multigraph)
        hipLaunchKernelGGL(k_vde_multi, dim3(grid_for((uint64_t)nr * e)), dim3(kBlock), 0, c->stream, nr, e,
                           c->rows_identity ? (const uint32_t *)nullptr : c->rows.as<uint32_t>(), c->mg_off.as<uint64_t>(),
                           c->mg_label.as<uint32_t>(), c->labels.as<uint32_t>(), c->xtab.as<double>(), c->nx.as<double>(),
                           c->vde.as<double>());
    } else if (nr) {
        const uint32_t *rows = c->rows_identity ? nullptr : c->rows.as<uint32_t>();
        dim3 grid((nr + 255) / 256), block(256);
        const size_t tab_lds = (uint64_t)c->n_labels * e <= (uint64_t)kVdeTabMax ? (size_t)c->n_labels * e * 8 : 0;
        // rows longer than 64 go to k_vde_hubs when the width has an instantiation (the hub list covers every held row;
        // the kernel skips the ones this device does not own)
        const bool hub_split = c->n_hub != 0 && (e == 1 || e == 2 || e == 4 || e == 8);
#define GNNPE_VDE_ARGS                                                                                   \
    nr, rows, c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbr_label.as<uint32_t>(),      \
        c->labels.as<uint32_t>(), c->xtab.as<double>(), c->n_labels, e, c->nx.as<double>(), c->vde.as<double>(), hub_split, x_out, vp
        switch (e) {
        case 1: hipLaunchKernelGGL((k_vde<1>), grid, block, tab_lds, c->stream, GNNPE_VDE_ARGS); break;
        case 2: hipLaunchKernelGGL((k_vde<2>), grid, block, tab_lds, c->stream, GNNPE_VDE_ARGS); break;
        case 3: hipLaunchKernelGGL((k_vde<3>), grid, block, tab_lds, c->stream, GNNPE_VDE_ARGS); break;
        case 4: hipLaunchKernelGGL((k_vde<4>), grid, block, tab_lds, c->stream, GNNPE_VDE_ARGS); break;
        case 8: hipLaunchKernelGGL((k_vde<8>), grid, block, tab_lds, c->stream, GNNPE_VDE_ARGS); break;
        default: hipLaunchKernelGGL((k_vde<0>), grid, block, tab_lds, c->stream, GNNPE_VDE_ARGS); break;
        }
#undef GNNPE_VDE_ARGS
        if (hub_split) {
            const uint8_t *owned = c->rows_identity ? nullptr : c->owned.as<uint8_t>();
            const uint32_t rpw = 64 / e;
            const dim3 hgrid(grid_for(((uint64_t)c->n_hub + rpw - 1) / rpw * 64));
#define GNNPE_VH(EE)                                                                                                  \
    hipLaunchKernelGGL((k_vde_hubs<EE>), hgrid, block, tab_lds, c->stream, c->n_hub, c->hub_rows.as<uint32_t>(), owned, \
                       c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbr_label.as<uint32_t>(),            \
                       c->labels.as<uint32_t>(), c->xtab.as<double>(), c->n_labels, c->nx.as<double>(), c->vde.as<double>())
            switch (e) {
            case 1: GNNPE_VH(1); break;
            case 2: GNNPE_VH(2); break;
            case 4: GNNPE_VH(4); break;
            default: GNNPE_VH(8); break;
            }
#undef GNNPE_VH
        }
    }
    GNNPE_HIP_TRY(hipGetLastError());
    c->have_vde = true;
    c->nbr_vde_valid = false;
    c->ranked_vde_valid = false;
    c->vkey_valid = false;
    return GNNPE_OK;
}

int gnnpe_vde(gnnpe_ctx *c, double *hx, double *hnx, double *hvde)
{
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc = run_vde(c);
    if (rc) return rc;
    const size_t bytes = (size_t)c->n * c->e * 8;
    if (bytes) {
        if (hx) GNNPE_HIP_TRY(hipMemcpyAsync(hx, c->x.p, bytes, hipMemcpyDeviceToHost, c->stream));
        if (hnx) GNNPE_HIP_TRY(hipMemcpyAsync(hnx, c->nx.p, bytes, hipMemcpyDeviceToHost, c->stream));
        if (hvde) GNNPE_HIP_TRY(hipMemcpyAsync(hvde, c->vde.p, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    if (hx || hnx || hvde) GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    return GNNPE_OK;
}

int gnnpe_vde_device_ptr(gnnpe_ctx *c, void **dev_vde, void **dev_x)
{
    GNNPE_REQUIRE(c && c->have_vde, GNNPE_ERR_ARG, "gnnpe_vde_device_ptr: call gnnpe_vde first");
    if (dev_vde) *dev_vde = c->vde.p;
    if (dev_x) *dev_x = c->x.p;
    return GNNPE_OK;
}

int gnnpe_vde_pack_slab(gnnpe_ctx *c, uint32_t begin, uint32_t end, void *dev_buf)
{
    GNNPE_REQUIRE(c && c->have_vde && c->have_order && dev_buf, GNNPE_ERR_ARG, "gnnpe_vde_pack_slab: bad state");
    GNNPE_REQUIRE(begin <= end && end <= c->n, GNNPE_ERR_ARG, "bad slab");
    if (end > begin)
        hipLaunchKernelGGL(k_vde_pack, dim3(grid_for((uint64_t)(end - begin) * c->e)), dim3(kBlock), 0, c->stream, begin,
                           end, c->e, c->sorted.as<uint32_t>(), c->vde.as<double>(), (double *)dev_buf);
    GNNPE_HIP_TRY(hipGetLastError());
    return GNNPE_OK;
}

int gnnpe_vde_unpack_slab(gnnpe_ctx *c, uint32_t begin, uint32_t end, const void *dev_buf)
{
    GNNPE_REQUIRE(c && c->have_vde && c->have_order && dev_buf, GNNPE_ERR_ARG, "gnnpe_vde_unpack_slab: bad state");
    GNNPE_REQUIRE(begin <= end && end <= c->n, GNNPE_ERR_ARG, "bad slab");
    if (end > begin)
        hipLaunchKernelGGL(k_vde_unpack, dim3(grid_for((uint64_t)(end - begin) * c->e)), dim3(kBlock), 0, c->stream,
                           begin, end, c->e, c->sorted.as<uint32_t>(), (const double *)dev_buf, c->vde.as<double>());
    GNNPE_HIP_TRY(hipGetLastError());
    c->nbr_vde_valid = false;
    c->ranked_vde_valid = false;
    c->vkey_valid = false;
    c->vinfo_fused = false;  // (the count kernel's vertex records carried the vde table as it was)
    return GNNPE_OK;
}

int gnnpe_vde_unpack_all(gnnpe_ctx *c, uint32_t n_ranks, const uint32_t *bounds, uint32_t stride, uint32_t skip_rank,
                         const void *dev_buf)
{
    GNNPE_REQUIRE(c && c->have_vde && c->have_order && dev_buf && bounds && n_ranks >= 1, GNNPE_ERR_ARG,
                  "gnnpe_vde_unpack_all: bad state");
    GNNPE_REQUIRE(bounds[0] == 0 && bounds[n_ranks] == c->n, GNNPE_ERR_ARG, "slab bounds must run from 0 to n");
    for (uint32_t r = 0; r < n_ranks; r++)
        GNNPE_REQUIRE(bounds[r] <= bounds[r + 1] && bounds[r + 1] - bounds[r] <= stride, GNNPE_ERR_ARG,
                      "slab %u: [%u, %u) does not fit rows of stride %u", r, bounds[r], bounds[r + 1], stride);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->slab_bounds.reserve(((size_t)n_ranks + 1) * 4))) return rc;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->slab_bounds.p, bounds, ((size_t)n_ranks + 1) * 4, hipMemcpyHostToDevice, c->stream));
    if (c->n)
        hipLaunchKernelGGL(k_vde_unpack_all, dim3(grid_for((uint64_t)c->n * c->e)), dim3(kBlock), 0, c->stream, c->n, c->e, n_ranks,
                           c->slab_bounds.as<uint32_t>(), stride, skip_rank, c->sorted.as<uint32_t>(), (const double *)dev_buf,
                           c->vde.as<double>());
    GNNPE_HIP_TRY(hipGetLastError());
    c->nbr_vde_valid = false;
    c->ranked_vde_valid = false;
    c->vkey_valid = false;
    c->vinfo_fused = false;  // (the count kernel's vertex records carried the vde table as it was)
    return GNNPE_OK;
}

// ---- R2 / R5: enumeration ---------------------------------------------------------------------------
//   4 ranked     one wave per start vertex over rank-sorted neighbour records, hub rows (degree > 64) streamed in id
//                order by the same kernel (gnnpe_fill_ranked.hip.h)                                        default
//   1 pair-wave  one wave per (s, b) pair, direct stores, run-time embedding width (gnnpe_fill_pairwave.hip.h): the
//                generic form for widths without a specialised instantiation, and the A/B baseline
// l = 3 (4-vertex paths, BASELINE config 5) has one implementation: gnnpe_fill_deep.hip.h.


// nbr_vde[q] = vde[nbrs[q]] for every held adjacency entry (the pair-wave and l=3 kernels stream it)
static int ensure_nbr_vde(gnnpe_ctx *c)
{
    if (c->nbr_vde_valid) return GNNPE_OK;
    GNNPE_REQUIRE(c->have_vde, GNNPE_ERR_ARG, "embeddings requested before gnnpe_vde");
    int rc;
    if ((rc = c->nbr_vde.reserve((c->nbr_used + 1) * c->e * 8))) return rc;
    if (c->nbr_used)
        hipLaunchKernelGGL(k_gather_rows_f64, dim3(grid_for(c->nbr_used * c->e)), dim3(kBlock), 0, c->stream,
                           c->nbr_used, c->e, c->nbrs.as<uint32_t>(), c->vde.as<double>(), c->nbr_vde.as<double>());
    GNNPE_HIP_TRY(hipGetLastError());
    c->nbr_vde_valid = true;
    return GNNPE_OK;
}

#define GNNPE_BY_E(e, LAUNCH)     \
    switch (e) {                  \
    case 1: LAUNCH(1); break;     \
    case 2: LAUNCH(2); break;     \
    case 3: LAUNCH(3); break;     \
    case 4: LAUNCH(4); break;     \
    default: LAUNCH(8); break;    \
    }

static bool packed_ids(const gnnpe_ctx *c) { return c->n <= (1u << kPackedIdBits); }
static uint32_t rec_bytes(uint32_t e, bool packed) { return (packed ? 4u : 8u) + 8u * e; }

// lay the row blocks out: header (8e bytes) + records, every block on a 128-byte boundary
static int ensure_row_blocks(gnnpe_ctx *c, uint32_t e)
{
    if (c->rblock_valid && c->rblock_e == e) return GNNPE_OK;
    int rc;
    const uint32_t nh = c->n_held;
    if ((rc = c->rblock.reserve(((size_t)c->n + 1) * 4)) || (rc = c->scratch.reserve(((size_t)nh + 2) * 8))) return rc;
    uint32_t *units = c->scratch.as<uint32_t>(), *off = units + nh + 1;
    const uint32_t *held = c->rows_identity ? nullptr : c->held.as<uint32_t>();
    hipLaunchKernelGGL(k_row_block_units, dim3(grid_for((uint64_t)nh + 1)), dim3(kBlock), 0, c->stream, nh, held,
                       c->adj_deg.as<uint32_t>(), 8u * e, rec_bytes(e, packed_ids(c)), rec_bytes(e, false), units);
    if ((rc = scan_u32(c, units, off, (uint64_t)nh + 1))) return rc;
    uint64_t tot = 0;
    if ((rc = read_back_u64(c, off + nh, 4, &tot))) return rc;
    if (nh)
        hipLaunchKernelGGL(k_row_block_starts, dim3(grid_for(nh)), dim3(kBlock), 0, c->stream, nh, held, off,
                           c->rblock.as<uint32_t>());
    GNNPE_HIP_TRY(hipGetLastError());
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));  // scratch is reused by the caller
    c->rblock_units = (uint32_t)tot;
    c->rblock_valid = true;
    c->rblock_e = e;
    return GNNPE_OK;
}

// row blocks {vde[b] | records by descending rank} + per-pair {block, count, G}; hub rows: id-ordered records, counts
// from a per-row sort of the ranks
static int build_ranked(gnnpe_ctx *c, uint64_t ne)
{
    const uint32_t e = c->have_table ? c->e : 2;
    const bool packed = packed_ids(c);
    int rc;
    if ((rc = ensure_row_blocks(c, e))) return rc;
    if ((rc = c->rpairs.reserve((ne + 1) * sizeof(RankedPair))) || (rc = c->rrecs.reserve((c->rblock_units + 1) * kRowAlign)) ||
        (rc = c->vinfo.reserve(((size_t)c->n + 1) * GNNPE_VINFO_STRIDE(e) * 8)))
        return rc;
    // pairs whose middle row is not on the device stay empty; with the whole graph loaded every pair is written by
    // the row kernels and only the scan's sentinel entry needs clearing
    if (!c->rows_identity) {
        GNNPE_HIP_TRY(hipMemsetAsync(c->rpairs.p, 0, (ne + 1) * sizeof(RankedPair), c->stream));
    } else if (c->rpairs_sentinel_buf != c->rpairs.p || c->rpairs_sentinel_at != ne) {  // (nobody writes slot ne: once per buffer and pair count)
        GNNPE_HIP_TRY(hipMemsetAsync(c->rpairs.as<RankedPair>() + ne, 0, sizeof(RankedPair), c->stream));
        c->rpairs_sentinel_buf = c->rpairs.p;
        c->rpairs_sentinel_at = ne;
    }
    if (!c->rows_identity) c->rpairs_sentinel_buf = nullptr;
    if (c->n_held) {
        const uint32_t *held = c->rows_identity ? nullptr : c->held.as<uint32_t>();
        const double *vde = c->have_vde ? c->vde.as<double>() : nullptr;
        // one row per wave, workgroups in launch order (the kernel's loop is then a single pass): the blocks are written as one
        // dense front; against the resident grid 1.063 -> 1.050 ms for the count phase at config 3 (scripts/count_ab.py)
        [[maybe_unused]] const dim3 grid((unsigned)std::min<uint64_t>(((uint64_t)c->n_held + 3) / 4, 1u << 30));
        const dim3 block(kBlock);
        // rows per wave: 1 (one wave per row) or 4 with the rows' loads batched (GNNPE_ROWS_ILP=1|4 overrides; default below)
        // k_vde wrote the per-vertex records when it ran (run_vde) and the slab's pair offsets have not changed since
        const bool vinfo_current = c->vinfo_fused && c->vinfo_gen == c->slab_struct_gen && c->have_vde;
        c->vinfo_fused = false;  // (good for one count: the next gnnpe_vde decides again)
        if (c->sw.debug) fprintf(stderr, "[count] vertex records: %s\n", vinfo_current ? "written by k_vde" : "k_pack_vinfo");
        // k_start_scan's status words + ticket are zeroed by the row kernel (count_paths passes them through the context)
        uint32_t *clear_words = c->clear_words;
        const uint32_t n_clear = c->n_clear;
        c->clear_done = clear_words != nullptr;
        int rows_ilp = (int)diag_int("GNNPE_ROWS_ILP", 4);  // (diagnostic builds: 1 | 2 | 8 for the A/B of DESIGN 3.2)
        if (rows_ilp != 1 && rows_ilp != 2 && rows_ilp != 8) rows_ilp = 4;
        const dim3 gridk((unsigned)std::min<uint64_t>(((uint64_t)c->n_held + 4 * rows_ilp - 1) / (4 * rows_ilp), 1u << 30));
#define GNNPE_RRM(EE, PK, KK)                                                                                       \
    hipLaunchKernelGGL((k_rows_rank_multi<EE, PK, KK>), gridk, block, 0, c->stream, c->n_held, held,                \
                       c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(),              \
                       c->vinfo.as<double>(), c->revpos.as<uint32_t>(), c->rblock.as<uint32_t>(),                   \
                       c->rrecs.as<char>(), c->rpairs.as<RankedPair>(), clear_words, n_clear,                       \
                       (uint32_t)diag_int("GNNPE_ROWS_PAIR8", 0))
#ifdef GNNPE_DIAG
#define GNNPE_RRK(EE, PK)                                                                                           \
    do {                                                                                                            \
        if (rows_ilp == 8) GNNPE_RRM(EE, PK, 8);                                                                    \
        else if (rows_ilp == 4) GNNPE_RRM(EE, PK, 4);                                                               \
        else if (rows_ilp == 2) GNNPE_RRM(EE, PK, 2);                                                               \
        else {                                                                                                      \
            c->clear_done = false;                                                                                  \
            hipLaunchKernelGGL((k_rows_rank<EE, PK>), grid, block, 0, c->stream, c->n_held, held,                   \
                               c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(),      \
                               c->vinfo.as<double>(), c->revpos.as<uint32_t>(), c->rblock.as<uint32_t>(),           \
                               c->rrecs.as<char>(), c->rpairs.as<RankedPair>());                                    \
        }                                                                                                           \
    } while (0)
#else
#define GNNPE_RRK(EE, PK) GNNPE_RRM(EE, PK, 4)
#endif
#define GNNPE_RR(EE)                                                                                               \
    do {                                                                                                           \
        if (!vinfo_current)                                                                                        \
            hipLaunchKernelGGL((k_pack_vinfo<EE>), dim3(grid_for(c->n)), block, 0, c->stream, c->n, vde,            \
                               c->rank.as<uint32_t>(), c->slab_begin, c->slab_end, c->poffs.as<uint32_t>(),         \
                               c->vinfo.as<double>());                                                              \
        if (packed) GNNPE_RRK(EE, true); else GNNPE_RRK(EE, false);                                                 \
    } while (0)
        GNNPE_BY_E(e, GNNPE_RR)
#undef GNNPE_RR
#undef GNNPE_RRK
#undef GNNPE_RRM
        // diagnostic launches beside the real one (GNNPE_ROWS_PROBE, scripts/count_ab.py): pieces of the kernel on their own,
        // into scratch copies of its outputs; whole graph on one device, e = 2, packed ids only
#ifdef GNNPE_DIAG
        if (const int mode = (int)diag_int("GNNPE_ROWS_PROBE", 0)) {
            if (mode >= 1 && mode <= 3 && c->rows_identity && e == 2 && packed && c->slab_begin == 0 && c->slab_end == c->n) {
                DevBuf probe_recs, probe_pairs;  // freed on return (a diagnostic path may allocate)
                if ((rc = probe_recs.reserve((c->rblock_units + 1) * kRowAlign)) || (rc = probe_pairs.reserve((ne + 1) * sizeof(RankedPair)))) return rc;
#define GNNPE_RP(MM)                                                                                                    \
    hipLaunchKernelGGL((k_rows_rank_probe<2, true, MM>), dim3((c->n_held + 15) / 16), block, 0, c->stream, c->n_held, c->adj_start.as<uint32_t>(), \
                       c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(), c->rank.as<uint32_t>(), vde, c->poffs.as<uint32_t>(), \
                       c->revpos.as<uint32_t>(), c->rblock.as<uint32_t>(), probe_recs.as<char>(), probe_pairs.as<RankedPair>())
                for (int rep = 0; rep < 3; rep++) {
                    if (mode == 1) GNNPE_RP(1); else if (mode == 2) GNNPE_RP(2); else GNNPE_RP(3);
                }
#undef GNNPE_RP
                GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
            }
        }
#endif
        GNNPE_HIP_TRY(hipGetLastError());
    }
    if (c->n_hub) {
        // hub rows (known since the rows were loaded): id-ordered records, the rows' ranks sorted per row, counts by
        // binary search
        if ((rc = c->nbr_rank.reserve((c->nbr_used + 1) * 4)) || (rc = c->rank_sorted.reserve((c->nbr_used + 1) * 4))) return rc;
        const dim3 grid(grid_for((uint64_t)c->n_hub * 64)), block(kBlock);
#define GNNPE_HR(EE)                                                                                               \
    hipLaunchKernelGGL((k_hub_records<EE>), grid, block, 0, c->stream, c->n_hub, c->hub_rows.as<uint32_t>(),        \
                       c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(),              \
                       c->vinfo.as<double>(), c->rblock.as<uint32_t>(), c->rrecs.as<char>(), c->nbr_rank.as<uint32_t>())
        GNNPE_BY_E(e, GNNPE_HR)
#undef GNNPE_HR
        size_t ts = 0;
        GNNPE_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(
            nullptr, ts, c->nbr_rank.as<uint32_t>(), c->rank_sorted.as<uint32_t>(), (int)c->nbr_used, (int)c->n_hub,
            c->hub_beg.as<uint32_t>(), c->hub_end.as<uint32_t>(), 0, 32, c->stream));
        if ((rc = c->cub_tmp.reserve(ts))) return rc;
        ts = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(
            c->cub_tmp.p, ts, c->nbr_rank.as<uint32_t>(), c->rank_sorted.as<uint32_t>(), (int)c->nbr_used, (int)c->n_hub,
            c->hub_beg.as<uint32_t>(), c->hub_end.as<uint32_t>(), 0, 32, c->stream));
#define GNNPE_HP(EE)                                                                                               \
    hipLaunchKernelGGL((k_hub_pairs<EE>), grid, block, 0, c->stream, c->n_hub, c->hub_rows.as<uint32_t>(),          \
                       c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(),              \
                       c->vinfo.as<double>(), c->revpos.as<uint32_t>(), c->rank_sorted.as<uint32_t>(),              \
                       c->rblock.as<uint32_t>(), c->rpairs.as<RankedPair>())
        GNNPE_BY_E(e, GNNPE_HP)
#undef GNNPE_HP
        GNNPE_HIP_TRY(hipGetLastError());
    }
    c->ranked_vde_valid = c->have_vde;
    return GNNPE_OK;
}

// poffs[i] = first pair of the i-th start vertex of the slab (prefix sum of the slab rows' degrees), n_edges = their
// number: the slab's own CSR offsets in processing order.  Depends on the graph, the order and the slab only, so it is
// built when one of them changes, not per step.
static int ensure_slab_struct(gnnpe_ctx *c)
{
    if (c->slab_struct_valid) return GNNPE_OK;
    const uint32_t sb = c->slab_begin, len = c->slab_end - c->slab_begin;
    int rc;
    if ((rc = c->poffs.reserve((size_t)(len + 2) * 4)) || (rc = c->scratch.reserve((size_t)(len + 2) * 8))) return rc;
    uint32_t *pdeg = c->scratch.as<uint32_t>();
    hipLaunchKernelGGL(k_slab_degrees, dim3(grid_for(len + 1)), dim3(kBlock), 0, c->stream, len, sb,
                       c->sorted.as<uint32_t>(), c->adj_deg.as<uint32_t>(), pdeg);
    if ((rc = scan_u32(c, pdeg, c->poffs.as<uint32_t>(), (uint64_t)len + 1))) return rc;
    uint64_t w = 0;
    if ((rc = read_back_u64(c, c->poffs.as<uint32_t>() + len, 4, &w))) return rc;
    c->n_edges = (uint32_t)w;
    c->slab_struct_valid = true;
    c->slab_struct_gen++;
    c->pst_valid = false;
    return GNNPE_OK;
}

// eoff[pair] = output slot of the pair's first path, for the consumers that need it per PAIR (the tile table, the index build's
// pair records, per-start counts, the l = 2 strip jobs): the rank-sorted count leaves only the start records and the total
// (k_start_scan); the scan over the 2.0e7 pair counts runs here, once per count, when somebody asks.
int gnnpe_ensure_eoff(gnnpe_ctx *c)
{
    if (c->eoff_valid) return GNNPE_OK;
    GNNPE_REQUIRE(c->counted && c->counted_variant == kVarRanked, GNNPE_ERR_ARG, "per-pair offsets: no count of the rank-sorted enumeration");
    const uint64_t ne = c->n_edges;
    int rc;
    hipcub::TransformInputIterator<uint64_t, CntOfPair, const RankedPair *> it(c->rpairs.as<RankedPair>(), CntOfPair());
    size_t tb = 0;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, c->eoff.as<uint64_t>(), (int64_t)(ne + 1), c->stream));
    if ((rc = c->cub_tmp.reserve(tb))) return rc;
    tb = c->cub_tmp.bytes;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, it, c->eoff.as<uint64_t>(), (int64_t)(ne + 1), c->stream));
    c->eoff_valid = true;
    return GNNPE_OK;
}

// what the output-tile-driven emit needs beside the count: the pairs' end points (once per graph / order / slab) and the
// tile table of this count for the tiles of rows [0, rows_hi)
static int ensure_tile_table(gnnpe_ctx *c, uint64_t rows_hi, uint32_t ts)
{
    const uint32_t len = c->slab_end - c->slab_begin;
    const uint64_t ne = c->n_edges;
    int rc;
    if (!c->pst_valid) {
        if ((rc = c->pst.reserve((ne + 1) * sizeof(uint2)))) return rc;
        if (len)
            hipLaunchKernelGGL(k_pair_ends, dim3(grid_for((uint64_t)len * 64)), dim3(kBlock), 0, c->stream, len, c->slab_begin,
                               c->sorted.as<uint32_t>(), c->adj_start.as<uint32_t>(), c->poffs.as<uint32_t>(),
                               c->nbrs.as<uint32_t>(), c->pst.as<uint2>());
        GNNPE_HIP_TRY(hipGetLastError());
        c->pst_valid = true;
    }
    const uint64_t need = (rows_hi + ts - 1) / ts + 1;  // tiles + the sentinel entry
    if ((rc = gnnpe_ensure_eoff(c))) return rc;  // (also for the strip jobs of emit shape 3, which read eoff themselves)
    if (c->tile_gen == c->count_gen && c->tile_cap >= need && c->tile_rows == ts) return GNNPE_OK;
    if ((rc = c->tfirst.reserve((need + 1) * 8))) return rc;
    if (ne)
        hipLaunchKernelGGL(k_tile_first, dim3(grid_for(ne)), dim3(kBlock), 0, c->stream, ne, c->eoff.as<uint64_t>(), ts, need,
                           c->tfirst.as<uint64_t>());
    GNNPE_HIP_TRY(hipGetLastError());
    c->tile_gen = c->count_gen;
    c->tile_cap = need;
    c->tile_rows = ts;
    return GNNPE_OK;
}

static int count_paths_impl(gnnpe_ctx *c, uint32_t l, uint64_t *host_per_start, uint64_t *host_total, bool fetch_total)
{
#ifdef GNNPE_DIAG
    if (c) c->sw = read_switches();  // GNNPE_DIAG_REREAD
#endif
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_rank_arrays(c);
    if (rc) return rc;
    // The reference enumerates 3-vertex paths whatever -l says; l=2 is the parity path, l=3 the
    // intended generalisation with the depth fixed (gnnpe_fill_deep.hip.h).
    GNNPE_REQUIRE(l == 2 || l == 3, GNNPE_ERR_UNSUPPORTED, "path length l=%u: only l=2 and l=3 are implemented", l);
    const uint32_t sb = c->slab_begin, se = c->slab_end, len = se - sb;
    const uint32_t e = c->have_table ? c->e : 2;
    c->counted = false;

    // which implementation runs: embedding widths without a specialised kernel use the generic pair-wave kernel
    int var = c->fill_variant;
    if (!fast_e(e)) var = kVarPairWave;
    if (l == 3) var = kVarDeep;

    // 1. slab rows -> directed (s, b) pairs in emission order: pair index = poffs[i] + position of b in N(s)
    if ((rc = ensure_slab_struct(c))) return rc;
    const uint64_t ne = c->n_edges;
    if ((rc = c->eoff.reserve((ne + 2) * 8))) return rc;
    uint64_t w = 0;

    // 2. per-pair path counts
    if (var == kVarRanked) {
        // k_start_scan's status words and ticket counter (step 4) are zeroed by the row kernel in front of it
        const uint32_t n_tiles_c = (len + kStartTile - 1) / kStartTile;
        if ((rc = c->scan_status.reserve((size_t)n_tiles_c * 8 + 64))) return rc;
        c->clear_words = c->scan_status.as<uint32_t>();
        c->n_clear = n_tiles_c * 2 + 16;
        c->clear_done = false;
        rc = build_ranked(c, ne);
        c->clear_words = nullptr;
        c->n_clear = 0;
        if (rc) return rc;
    } else {
        if ((rc = c->ecnt.reserve((ne + 2) * 4)) || (rc = c->nbr_rank.reserve((c->nbr_used + 1) * 4))) return rc;
        // rank of every held neighbour entry: the rank test becomes a contiguous stream
        if (c->nbr_used)
            hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(c->nbr_used)), dim3(kBlock), 0, c->stream, c->nbr_used,
                               c->nbrs.as<uint32_t>(), c->rank.as<uint32_t>(), c->nbr_rank.as<uint32_t>());
        GNNPE_HIP_TRY(hipMemsetAsync(c->ecnt.p, 0, (ne + 1) * 4, c->stream));
        if ((rc = c->erow.reserve((ne + 1) * 4)) || (rc = c->pnbr.reserve((ne + 1) * 4))) return rc;
        if (len)
            hipLaunchKernelGGL(k_perm_edges, dim3(grid_for((uint64_t)len * 16)), dim3(kBlock), 0, c->stream, len, sb,
                               c->sorted.as<uint32_t>(), c->adj_start.as<uint32_t>(), c->poffs.as<uint32_t>(),
                               c->nbrs.as<uint32_t>(), c->erow.as<uint32_t>(), c->pnbr.as<uint32_t>());
        if (var == kVarDeep) {
            // the emit walk with the stores compiled out; rows two hops from the slab must be on this device
            uint32_t *d_missing = c->small.as<uint32_t>() + 16;
            GNNPE_HIP_TRY(hipMemsetAsync(d_missing, 0xFF, 4, c->stream));
            FillParams P = {};
            P.erow = c->erow.as<uint32_t>();
            P.pnbr = c->pnbr.as<uint32_t>();
            P.adj_start = c->adj_start.as<uint32_t>();
            P.adj_deg = c->adj_deg.as<uint32_t>();
            P.nbrs = c->nbrs.as<uint32_t>();
            P.nbr_rank = c->nbr_rank.as<uint32_t>();
            P.sorted = c->sorted.as<uint32_t>();
            P.n_edges = ne;
            P.slab_begin = sb;
            P.e = e;
            // work units: (pair, batch of 64 third vertices); ufirst = exclusive scan of the units per pair
            if ((rc = c->ufirst.reserve((ne + 2) * 8))) return rc;
            uint64_t *ufirst = c->ufirst.as<uint64_t>();
            hipLaunchKernelGGL(k_deep_unit_counts, dim3(grid_for(ne + 1)), dim3(kBlock), 0, c->stream, ne,
                               c->pnbr.as<uint32_t>(), c->adj_deg.as<uint32_t>(), ufirst);
            size_t tb = 0;
            GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, ufirst, ufirst, (int64_t)(ne + 1), c->stream));
            if ((rc = c->cub_tmp.reserve(tb))) return rc;
            tb = c->cub_tmp.bytes;
            GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, ufirst, ufirst, (int64_t)(ne + 1), c->stream));
            uint64_t nu = 0;
            if ((rc = read_back_u64(c, ufirst + ne, 8, &nu))) return rc;
            c->n_units = nu;
            if ((rc = c->upair.reserve((nu + 1) * 4)) || (rc = c->uoff.reserve((nu + 2) * 8))) return rc;
            hipLaunchKernelGGL(k_deep_unit_pairs, dim3(grid_for(ne + 1)), dim3(kBlock), 0, c->stream, ne, ufirst,
                               c->upair.as<uint32_t>());
            // every row's neighbour ranks in ascending order, with the entry each came from: the count is then one forward merge
            // per (row, third vertex) over all the row's slab neighbours (k_deep3_count_rows)
            if ((rc = c->rank_sorted.reserve((c->nbr_used + 1) * 4)) || (rc = c->adj_end.reserve(((size_t)c->n + 1) * 4)) ||
                (rc = c->rank_arg.reserve((c->nbr_used + 1) * 4)) || (rc = c->scratch.reserve((c->nbr_used + 1) * 4)) ||
                (rc = c->rb_first.reserve(((size_t)c->n + 2) * 4)) || (rc = c->rb_cnt.reserve(((size_t)c->n + 2) * 4)))
                return rc;
            GNNPE_HIP_TRY(hipMemsetAsync(c->uoff.p, 0, (nu + 1) * 8, c->stream));
            if (c->nbr_used) {
                hipLaunchKernelGGL(k_row_ends, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, c->n, c->adj_start.as<uint32_t>(),
                                   c->adj_deg.as<uint32_t>(), c->adj_end.as<uint32_t>());
                hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(c->nbr_used)), dim3(kBlock), 0, c->stream, c->nbr_used, c->scratch.as<uint32_t>());
                size_t ts = 0;
                GNNPE_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(
                    nullptr, ts, c->nbr_rank.as<uint32_t>(), c->rank_sorted.as<uint32_t>(), c->scratch.as<uint32_t>(), c->rank_arg.as<uint32_t>(),
                    (int)c->nbr_used, (int)c->n, c->adj_start.as<uint32_t>(), c->adj_end.as<uint32_t>(), 0, 32, c->stream));
                if ((rc = c->cub_tmp.reserve(ts))) return rc;
                ts = c->cub_tmp.bytes;
                GNNPE_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortPairs(
                    c->cub_tmp.p, ts, c->nbr_rank.as<uint32_t>(), c->rank_sorted.as<uint32_t>(), c->scratch.as<uint32_t>(), c->rank_arg.as<uint32_t>(),
                    (int)c->nbr_used, (int)c->n, c->adj_start.as<uint32_t>(), c->adj_end.as<uint32_t>(), 0, 32, c->stream));
            }
            hipLaunchKernelGGL(k_deep_row_batches, dim3(grid_for((uint64_t)c->n + 1)), dim3(kBlock), 0, c->stream, c->n, c->adj_deg.as<uint32_t>(),
                               c->rb_cnt.as<uint32_t>());
            if ((rc = scan_u32(c, c->rb_cnt.as<uint32_t>(), c->rb_first.as<uint32_t>(), (uint64_t)c->n + 1))) return rc;
            // the entry-major count (k_deep3_count_hist) unless GNNPE_DEEP_COUNT=merge asks for the pointer walk (tests run both)
            if (c->sw.deep_merge) {
                hipLaunchKernelGGL(k_deep3_count_rows, dim3(kMaxGrid), dim3(kBlock), 0, c->stream, P, c->n, len,
                                   c->rows_identity ? (const uint8_t *)nullptr : c->present.as<uint8_t>(), c->rank.as<uint32_t>(),
                                   c->rank_sorted.as<uint32_t>(), c->rank_arg.as<uint32_t>(), c->revpos.as<uint32_t>(), c->poffs.as<uint32_t>(),
                                   c->rb_first.as<uint32_t>(), ufirst, c->uoff.as<uint64_t>(), nu, d_missing);
            } else {
                uint32_t *lowcnt = c->rb_cnt.as<uint32_t>();  // (the row-batch counts are dead behind their scan)
                hipLaunchKernelGGL(k_deep_lowcnt, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, c->n,
                                   c->rows_identity ? (const uint8_t *)nullptr : c->present.as<uint8_t>(), c->adj_start.as<uint32_t>(),
                                   c->adj_deg.as<uint32_t>(), c->rank.as<uint32_t>(), c->rank_sorted.as<uint32_t>(), lowcnt);
                // the row-batches of the long rows: counted, then listed
                uint32_t *d_nhb = c->small.as<uint32_t>() + 20;
                uint64_t n_hub_batches = 0;
                for (int pass = 0; pass < 2; pass++) {
                    GNNPE_HIP_TRY(hipMemsetAsync(d_nhb, 0, 4, c->stream));
                    hipLaunchKernelGGL(k_deep_hub_batches, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, c->n, c->adj_deg.as<uint32_t>(), d_nhb,
                                       pass ? c->deep_hub_batches.as<uint2>() : (uint2 *)nullptr);
                    if (pass == 0) {
                        if ((rc = read_back_u64(c, d_nhb, 4, &n_hub_batches))) return rc;
                        n_hub_batches &= 0xFFFFFFFFull;
                        if (!n_hub_batches) break;
                        if ((rc = c->deep_hub_batches.reserve((n_hub_batches + 1) * 8))) return rc;
                    }
                }
                if ((rc = c->deep_ubase.reserve((c->nbr_used + 1) * 8))) return rc;
                hipLaunchKernelGGL(k_deep_unit_bases, dim3(grid_for((uint64_t)c->n * 16)), dim3(kBlock), 0, c->stream, c->n, sb, se,
                                   c->rows_identity ? (const uint8_t *)nullptr : c->present.as<uint8_t>(), c->adj_start.as<uint32_t>(),
                                   c->adj_deg.as<uint32_t>(), c->rank_sorted.as<uint32_t>(), c->rank_arg.as<uint32_t>(), c->revpos.as<uint32_t>(),
                                   c->poffs.as<uint32_t>(), ufirst, c->deep_ubase.as<uint64_t>());
                hipLaunchKernelGGL(k_deep3_count_hist, dim3(kMaxGrid), dim3(kBlock), 0, c->stream, P, c->n, len,
                                   c->rows_identity ? (const uint8_t *)nullptr : c->present.as<uint8_t>(), c->rank.as<uint32_t>(),
                                   c->rank_sorted.as<uint32_t>(), c->rank_arg.as<uint32_t>(), c->deep_ubase.as<uint64_t>(),
                                   c->rb_first.as<uint32_t>(), lowcnt, c->uoff.as<uint64_t>(), nu, d_missing);
                if (n_hub_batches)
                    hipLaunchKernelGGL(k_deep3_count_hist_coop, dim3((unsigned)std::min<uint64_t>(n_hub_batches, (uint64_t)c->num_cus * 4)),
                                       dim3(64 * kCoopWaves), 0, c->stream, P, len,
                                       c->rows_identity ? (const uint8_t *)nullptr : c->present.as<uint8_t>(), c->rank.as<uint32_t>(),
                                       c->rank_sorted.as<uint32_t>(), c->rank_arg.as<uint32_t>(), c->deep_ubase.as<uint64_t>(),
                                       c->deep_hub_batches.as<uint2>(), (uint32_t)n_hub_batches, lowcnt, c->uoff.as<uint64_t>(), nu,
                                       d_missing);
            }
            uint64_t miss = 0;
            if ((rc = read_back_u64(c, d_missing, 4, &miss))) return rc;
            GNNPE_REQUIRE((uint32_t)miss == 0xFFFFFFFFu, GNNPE_ERR_ARG,
                          "l=3: the adjacency row of vertex %u (two hops from the slab) is not on this device",
                          (uint32_t)miss);
        } else {  // kVarPairWave
            hipLaunchKernelGGL(k_count_edges, dim3(grid_for(ne * 16 + 1)), dim3(kBlock), 0, c->stream, ne, sb,
                               c->erow.as<uint32_t>(), c->pnbr.as<uint32_t>(), c->adj_start.as<uint32_t>(),
                               c->adj_deg.as<uint32_t>(), c->nbr_rank.as<uint32_t>(), c->ecnt.as<uint32_t>());
        }
        GNNPE_HIP_TRY(hipGetLastError());
    }

    // 3. exclusive scan: eoff[pair] = output slot of the pair's first path; eoff[ne] = total.  The rank-sorted enumeration's
    // default emit needs the STARTS' offsets only: k_start_scan below; eoff itself is built on demand (ensure_eoff)
    c->eoff_valid = true;
    if (var == kVarRanked) {
        c->eoff_valid = false;
    } else if (var == kVarDeep) {
        // unit counts -> unit offsets (in place), then the pair offsets the per-start counts read
        uint64_t *uoff = c->uoff.as<uint64_t>();
        size_t tb = 0;
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, uoff, uoff, (int64_t)(c->n_units + 1), c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        tb = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, uoff, uoff, (int64_t)(c->n_units + 1), c->stream));
        hipLaunchKernelGGL(k_deep_pair_offsets, dim3(grid_for(ne + 1)), dim3(kBlock), 0, c->stream, ne,
                           c->ufirst.as<uint64_t>(), uoff, c->eoff.as<uint64_t>());
    } else if ((rc = scan_u32_to_u64(c, c->ecnt.as<uint32_t>(), c->eoff.as<uint64_t>(), ne + 1))) {
        return rc;
    }

    // 4. per-start records that shorten the emit kernel's dependent-load chain; enqueued before the one read-back
    if (var == kVarRanked) {
        const uint32_t n_tiles = (len + kStartTile - 1) / kStartTile;
        if ((rc = c->srec.reserve((size_t)(len + 1) * sizeof(StartRec))) || (rc = c->scan_status.reserve((size_t)n_tiles * 8 + 64))) return rc;
        // status words, then the ticket counter (a line of its own behind them): zeroed by the row kernel of step 2 where one ran
        if (!c->clear_done) GNNPE_HIP_TRY(hipMemsetAsync(c->scan_status.p, 0, (size_t)n_tiles * 8 + 64, c->stream));
        c->clear_done = false;
        if (!len) GNNPE_HIP_TRY(hipMemsetAsync(c->eoff.as<uint64_t>() + ne, 0, 8, c->stream));  // (no start vertices: no paths; else the last tile writes the total)
        if ((rc = c->tk_ctl.reserve(kStartHeadsBytes + 64))) return rc;
        c->heads_clean = false;
        if (len) {
            unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)blocks_per_cu(k_start_scan) * c->num_cus));
            grid = (unsigned)std::max<long>(1, std::min<long>((long)n_tiles, diag_int("GNNPE_START_SCAN_GRID", (long)grid)));
            if (c->sw.debug) fprintf(stderr, "[count] k_start_scan: %u tiles, grid %u (%d per CU)\n", n_tiles, grid, blocks_per_cu(k_start_scan));
            hipLaunchKernelGGL(k_start_scan, dim3(grid), dim3(kStartTile), 0, c->stream, len, sb, c->sorted.as<uint32_t>(),
                               c->member.as<uint32_t>(), c->adj_start.as<uint32_t>(), c->poffs.as<uint32_t>(), c->rpairs.as<RankedPair>(),
                               c->scan_status.as<unsigned long long>(), reinterpret_cast<uint32_t *>(c->scan_status.as<char>() + (size_t)n_tiles * 8 + 32),
                               c->srec.as<StartRec>(), c->eoff.as<uint64_t>() + ne, c->tk_ctl.as<uint32_t>(), kStartHeadsBytes / 4u);
            c->heads_clean = true;  // (the emit kernel's ticket heads: no memset in front of the first fill of this count)
        }
        GNNPE_HIP_TRY(hipGetLastError());
    }
    c->l = l;
    c->counted = true;
    c->count_gen++;
    c->counted_variant = var;
    c->total_known = false;
    c->total_paths = 0;
    if (!fetch_total) return GNNPE_OK;  // enqueue-only: the total stays in eoff[ne] until somebody asks (resolve_total)
    if ((rc = resolve_total(c))) {
        c->counted = false;
        return rc;
    }
    w = c->total_paths;
    // embeddings of the adjacency entries, when the vde table is already there (keeps it out of the fill)
    if (c->have_vde && (var == kVarPairWave || var == kVarDeep) && (rc = ensure_nbr_vde(c))) return rc;
    if (host_total) *host_total = w;
    if (host_per_start && len) {
        if ((rc = gnnpe_ensure_eoff(c))) return rc;
        if ((rc = c->scratch.reserve((size_t)len * 8))) return rc;
        hipLaunchKernelGGL(k_per_start_counts, dim3(grid_for(len)), dim3(kBlock), 0, c->stream, len,
                           c->poffs.as<uint32_t>(), c->eoff.as<uint64_t>(), c->scratch.as<uint64_t>());
        GNNPE_HIP_TRY(hipMemcpyAsync(host_per_start, c->scratch.p, (size_t)len * 8, hipMemcpyDeviceToHost, c->stream));
        GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return GNNPE_OK;
}

int gnnpe_count_paths(gnnpe_ctx *c, uint32_t l, uint64_t *host_per_start, uint64_t *host_total)
{
    return count_paths_impl(c, l, host_per_start, host_total, true);
}

int gnnpe_count_paths_enqueue(gnnpe_ctx *c, uint32_t l)
{
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_REQUIRE(l == 2 && c->fill_variant == kVarRanked && fast_e(c->have_table ? c->e : 2), GNNPE_ERR_UNSUPPORTED,
                  "gnnpe_count_paths_enqueue: l=2 with a specialised embedding width only (use gnnpe_count_paths)");
    return count_paths_impl(c, l, nullptr, nullptr, false);
}

int gnnpe_count_total(gnnpe_ctx *c, uint64_t *host_total)
{
    GNNPE_REQUIRE(c && c->counted && host_total, GNNPE_ERR_ARG, "gnnpe_count_total: no count on this context");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc = resolve_total(c);
    if (rc) return rc;
    *host_total = c->total_paths;
    return GNNPE_OK;
}

int gnnpe_count_total_device(gnnpe_ctx *c, void *dev_u64)
{
    GNNPE_REQUIRE(c && c->counted && dev_u64, GNNPE_ERR_ARG, "gnnpe_count_total_device: no count on this context");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    GNNPE_HIP_TRY(hipMemcpyAsync(dev_u64, c->eoff.as<uint64_t>() + c->n_edges, 8, hipMemcpyDeviceToDevice, c->stream));
    return GNNPE_OK;
}

// capped: rows [0, min(total, end)) with the total read by nobody on the host -- the kernel clips against the start
// records, which carry every start vertex' output range
static int fill_device(gnnpe_ctx *c, uint64_t begin, uint64_t end, void *d_vids, void *d_pde, void *d_pdl,
                       void *d_part, bool capped = false)
{
#ifdef GNNPE_DIAG
    if (c) c->sw = read_switches();  // GNNPE_DIAG_REREAD
#endif
    GNNPE_REQUIRE(c && c->counted, GNNPE_ERR_ARG, "gnnpe_fill_paths: call gnnpe_count_paths first");
    if (capped) {
        GNNPE_REQUIRE(begin == 0 && !d_pdl && !d_part && c->counted_variant == kVarRanked, GNNPE_ERR_UNSUPPORTED,
                      "capped fill: ids and pde of the rank-sorted l=2 enumeration only");
    } else {
        int rc0 = resolve_total(c);
        if (rc0) return rc0;
        GNNPE_REQUIRE(begin <= end && end <= c->total_paths, GNNPE_ERR_ARG, "path range [%llu,%llu) outside [0,%llu]",
                      (unsigned long long)begin, (unsigned long long)end, (unsigned long long)c->total_paths);
    }
    GNNPE_REQUIRE((!d_pde && !d_pdl) || c->have_vde, GNNPE_ERR_ARG, "gnnpe_fill_paths: embeddings requested before gnnpe_vde");
    if (begin == end) return GNNPE_OK;
    const int var = c->counted_variant;
    const uint32_t e = c->have_table ? c->e : 2;
    const uint32_t len = c->slab_end - c->slab_begin;
    int rc;
    if (d_pde && (var == kVarPairWave || var == kVarDeep) && (rc = ensure_nbr_vde(c))) return rc;
    if (d_pde && var == kVarRanked && !c->ranked_vde_valid && (rc = build_ranked(c, c->n_edges))) return rc;

    FillParams P;
    P.erow = c->erow.as<uint32_t>();
    P.pnbr = c->pnbr.as<uint32_t>();
    P.adj_start = c->adj_start.as<uint32_t>();
    P.adj_deg = c->adj_deg.as<uint32_t>();
    P.nbrs = c->nbrs.as<uint32_t>();
    P.nbr_rank = c->nbr_rank.as<uint32_t>();
    P.sorted = c->sorted.as<uint32_t>();
    P.member = c->member.as<uint32_t>();
    P.eoff = c->eoff.as<uint64_t>();
    P.vde = c->vde.as<double>();
    P.x = c->x.as<double>();
    P.nbr_vde = c->nbr_vde.as<double>();
    P.n_edges = c->n_edges;
    P.begin = begin;
    P.end = end;
    P.slab_begin = c->slab_begin;
    P.e = e;
    P.out_ids = (uint32_t *)d_vids;
    P.out_pde = (double *)d_pde;
    P.out_pdl = (double *)d_pdl;
    P.out_part = (uint32_t *)d_part;

    if (var == kVarDeep) {
        // only the units whose output overlaps [begin, end)
        uint64_t *d_range = reinterpret_cast<uint64_t *>(c->small.as<char>() + 192);
        hipLaunchKernelGGL(k_deep_unit_range, dim3(1), dim3(64), 0, c->stream, c->n_units, c->uoff.as<uint64_t>(), begin, end,
                           d_range);
        GNNPE_HIP_TRY(hipMemcpyAsync(c->h_pinned, d_range, 16, hipMemcpyDeviceToHost, c->stream));
        GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
        const uint64_t u_lo = c->h_pinned[0], u_hi = c->h_pinned[1];
        // slices where hub rows put 10^5 candidates behind one unit; one workgroup per unit (a single kernel) where every
        // unit is a few hundred candidates.  GNNPE_DEEP_EMIT=slices|units overrides (tests run both on the same graphs).
        const bool slices = c->sw.deep_emit == 1 ? true : c->sw.deep_emit == 2 ? false : c->n_hub != 0;
        const uint32_t sparse_num = (uint32_t)diag_int("GNNPE_DEEP_SPARSE_NUM", kSparseNum);  // (gnnpe_fill_deep.hip.h: kSparseNum)
        if (slices && u_hi > u_lo) {
            // slices (gnnpe_fill_deep.hip.h): units -> slices per unit -> kept rows per slice -> emit, one wave per slice
            GNNPE_REQUIRE(u_hi - u_lo < (1ull << 31), GNNPE_ERR_ARG, "l=3 range covers %llu units; emit in smaller chunks",
                          (unsigned long long)(u_hi - u_lo));
            const uint32_t n_u = (uint32_t)(u_hi - u_lo);
            // first slice of every unit, then (16-byte aligned, behind them) the units' records for the emitting waves
            const size_t sf_bytes = (((size_t)n_u + 2) * 4 + 15) & ~(size_t)15;
            if ((rc = c->dsl_first.reserve(sf_bytes + ((size_t)n_u + 1) * 16))) return rc;
            uint32_t *sfirst = c->dsl_first.as<uint32_t>();
            uint4 *uinfo = reinterpret_cast<uint4 *>(c->dsl_first.as<char>() + sf_bytes);
            hipLaunchKernelGGL(k_deep_slice_counts, dim3(grid_for(((uint64_t)n_u + 1) * 64)), dim3(kBlock), 0, c->stream, P,
                               c->upair.as<uint32_t>(), c->ufirst.as<uint64_t>(), c->uoff.as<uint64_t>(), u_lo, n_u, sfirst,
                               c->rank.as<uint32_t>(), uinfo);
            size_t tb = 0;
            GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, sfirst, sfirst, (int64_t)n_u + 1, c->stream));
            if ((rc = c->cub_tmp.reserve(tb))) return rc;
            tb = c->cub_tmp.bytes;
            GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, sfirst, sfirst, (int64_t)n_u + 1, c->stream));
            GNNPE_HIP_TRY(hipMemcpyAsync(c->h_pinned, sfirst + n_u, 4, hipMemcpyDeviceToHost, c->stream));
            GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
            const uint32_t n_sl = *reinterpret_cast<const uint32_t *>(c->h_pinned);
            if (n_sl == 0) return GNNPE_OK;
            if ((rc = c->dsl_kept.reserve(((size_t)n_sl + 1) * 16 + 64))) return rc;
            // one launch (round 6, k_deep3_slices_fused): the slices' kept rows, their first slots (look-back inside the unit) and
            // the rows; GNNPE_DEEP_TWO_PASS=1 in a diagnostic build keeps the count kernel + scan + emit kernel of rounds 3-5
            if (!diag_int("GNNPE_DEEP_TWO_PASS", 0)) {
                unsigned long long *sstat = c->dsl_kept.as<unsigned long long>();  // status word per slice; behind them the count of
                uint32_t *sfall = reinterpret_cast<uint32_t *>(sstat + n_sl + 1);   // waves that gave up the look-back (never seen)
                GNNPE_HIP_TRY(hipMemsetAsync(sstat, 0, ((size_t)n_sl + 2) * 8, c->stream));
#define GNNPE_L(EE)                                                                                                      \
    hipLaunchKernelGGL((k_deep3_slices_fused<EE>), dim3((n_sl + 3u) / 4u), dim3(256), 0, c->stream, P, c->uoff.as<uint64_t>(), uinfo, \
                       u_lo, n_u, sfirst, n_sl, sstat, sfall, sparse_num)
                if (fast_e(e)) {
                    GNNPE_BY_E(e, GNNPE_L)
                } else {
                    GNNPE_L(0);
                }
#undef GNNPE_L
                GNNPE_HIP_TRY(hipGetLastError());
                if (c->sw.debug) {  // GNNPE_DEBUG=1: how many waves gave up the look-back and counted their unit's earlier slices themselves
                    uint32_t nf = 0;
                    GNNPE_HIP_TRY(hipMemcpyAsync(&nf, sfall, 4, hipMemcpyDeviceToHost, c->stream));
                    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
                    fprintf(stderr, "[deep] k_deep3_slices_fused: %u slices over %u units, %u look-back fallbacks\n", n_sl, n_u, nf);
                }
                return GNNPE_OK;
            }
#ifdef GNNPE_DIAG
            uint64_t *skept = c->dsl_kept.as<uint64_t>(), *sscan = skept + n_sl + 1;
            const dim3 sgrid(grid_for((uint64_t)n_sl * 64)), sblock(256);
#define GNNPE_L(EE)                                                                                                      \
    do {                                                                                                                 \
        hipLaunchKernelGGL((k_deep3_slices<EE, false>), sgrid, sblock, 0, c->stream, P, c->upair.as<uint32_t>(),         \
                           c->ufirst.as<uint64_t>(), c->uoff.as<uint64_t>(), u_lo, n_u, sfirst, n_sl, skept, sscan);     \
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, skept, sscan, (int64_t)n_sl, c->stream));            \
        if ((rc = c->cub_tmp.reserve(tb))) return rc;                                                                    \
        tb = c->cub_tmp.bytes;                                                                                           \
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, skept, sscan, (int64_t)n_sl, c->stream));       \
        hipLaunchKernelGGL((k_deep3_slices<EE, true>), sgrid, sblock, 0, c->stream, P, c->upair.as<uint32_t>(),          \
                           c->ufirst.as<uint64_t>(), c->uoff.as<uint64_t>(), u_lo, n_u, sfirst, n_sl, skept, sscan);     \
    } while (0)
            if (fast_e(e)) {
                GNNPE_BY_E(e, GNNPE_L)
            } else {
                GNNPE_L(0);
            }
#undef GNNPE_L
#endif
            GNNPE_HIP_TRY(hipGetLastError());
            return GNNPE_OK;
        }
        // waves per unit: 16 on graphs with hub rows (a unit behind a hub holds 10^5 candidates), 4 otherwise
        const unsigned n_blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(u_hi - u_lo, 1u << 20));
#define GNNPE_LW(EE, WW)                                                                                           \
    hipLaunchKernelGGL((k_deep3<EE, WW>), dim3(n_blocks), dim3(64 * WW), 0, c->stream, P, c->upair.as<uint32_t>(),   \
                       c->ufirst.as<uint64_t>(), c->uoff.as<uint64_t>(), u_lo, u_hi, sparse_num)
#define GNNPE_L(EE)                                     \
    do {                                                \
        if (c->n_hub) GNNPE_LW(EE, 16); else GNNPE_LW(EE, 4); \
    } while (0)
        if (fast_e(e)) {
            GNNPE_BY_E(e, GNNPE_L)
        } else {
            GNNPE_L(0);
        }
#undef GNNPE_LW
#undef GNNPE_L
    } else if (var == kVarRanked) {
        // a resident grid: every wave walks its share of the start vertices (w, w + waves, ...): the waves in flight
        // write one moving window of the output
        const StartRec *sr = c->srec.as<StartRec>();
        const bool packed = packed_ids(c);
        // Occupancy cap of the start-vertex kernel at e <= 2 (dynamic LDS the kernel never touches; occupancy_plan): with a
        // start's loads batched a wave keeps many bytes in flight, and FEWER resident waves write faster.  Same process, same
        // buffers: in an allocation of the fast class 8 / 7 / 6 / 5 / 4 / 3 workgroups per CU 3.37 / 3.05 / 2.95 / 2.93 / 3.00 /
        // 3.27 ms (round 4, scripts/emit_ab_libs.py; profiles/r05_emit_ab.txt), in one of the slow class 5 / 4 / 3 / 2:
        // 3.65 / 3.59 / 3.51 / 4.40 ms.  So: five (shape 1), or three (shape 4) where gnnpe_emit_calibrate_device measured that
        // faster into the buffer.
        int want_per_cu = e <= 2 ? 5 : 0;
        // Round 6, shape 1 is ONE-SHOT: one wave per start vertex, workgroups in launch order, exit -- the launch order keeps the rows
        // the chip writes at one moment together as the tickets do, without the atomic per start vertex, and a wave that is done does
        // not wait for the acknowledgement of its stores before it loads again (one counter for loads and stores, completed in issue
        // order: what made the resident ticket form of the l = 3 emit slower).  Same process, same buffers, eight
        // allocations (profiles/r06_emit_oneshot.txt): fast class 2.81-2.84 -> 2.73-2.76 ms (0.834-0.842), in between 3.04 -> 2.95,
        // slow class of the kind that prefers many waves 3.36 -> 3.26; the kind that prefers few keeps the resident grid at three
        // workgroups per CU (shape 4: 3.17 against 3.29-3.31 either way), so the calibration still times both.
        // (Shape 4 and diagnostic builds:) start vertices from ticket counters, in order (the kernel's comment): 16 heads; GNNPE_RANKED_TICKETS=0 restores the
        // static assignment w, w + waves, ... for A/B runs.  Same process, same buffers, fast / slow class, five workgroups
        // per CU: 2.77 / 3.46 ms against 2.88 / 3.68; three per CU into the slow class 3.32 against 3.51.
        uint32_t *rk_heads = nullptr;
        const uint32_t rk_nh = (uint32_t)std::max<long>(0, std::min<long>(64, diag_int("GNNPE_RANKED_TICKETS", 16)));
#define GNNPE_LK(KERN)                                                                                                  \
    do {                                                                                                                \
        auto kern = KERN;                                                                                               \
        OccPlan plan = occupancy_plan(kern, want_per_cu, c->device);                                                    \
        if (const long pad_ab = diag_int("GNNPE_FILL_LDS_PAD", -1); pad_ab >= 0) { /* A/B aid: this much dynamic LDS */  \
            plan.pad = (size_t)std::min<long>(48 * 1024, pad_ab);                                                       \
            plan.per_cu = blocks_per_cu_at(reinterpret_cast<const void *>(kern), plan.pad);                             \
        }                                                                                                               \
        /* shape 1 since round 6: ONE-SHOT -- a wave per start vertex, workgroups in launch order, exit; no ticket, no   \
           occupancy cap (the comment above GNNPE_LK's use).  Shape 4 stays the resident grid at three workgroups per CU.   \
           GNNPE_FILL_ONESHOT=0 in a diagnostic build restores the resident five-per-CU grid of rounds 4-5 for shape 1 */   \
        const bool oneshot = want_per_cu != 3 && diag_int("GNNPE_FILL_ONESHOT", 1) != 0;                                \
        if (oneshot && diag_int("GNNPE_FILL_LDS_PAD", -1) < 0) plan.pad = 0;                                              \
        const uint64_t want = ((uint64_t)len + 3) / 4, fit = oneshot ? want : (uint64_t)plan.per_cu * c->num_cus;       \
        const dim3 grid((unsigned)std::max<uint64_t>(1, std::min(want, fit))), block(kBlock);                           \
        const uint32_t nh_l = oneshot ? 0u : std::min<uint32_t>(rk_nh, grid.x * 4u); /* every head needs a wave that serves it */ \
        if (c->sw.debug)                                                                                                \
            fprintf(stderr, "[emit] k_fill_ranked: %s, %d workgroups per CU wanted, %d planned, %zu B of dynamic LDS, grid %u, %u ticket heads\n", \
                    oneshot ? "one wave per start vertex in launch order" : "resident grid", want_per_cu, plan.per_cu, plan.pad, grid.x, nh_l); \
        if (nh_l) {                                                                                                     \
            if ((rc = c->tk_ctl.reserve(kStartHeadsBytes + 64))) return rc;                                             \
            if (!c->heads_clean) GNNPE_HIP_TRY(hipMemsetAsync(c->tk_ctl.p, 0, (size_t)nh_l * kStartHeadWords * 4, c->stream)); \
            c->heads_clean = false;                                                                                     \
            rk_heads = c->tk_ctl.as<uint32_t>();                                                                        \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, block, plan.pad, c->stream, P, sr, c->rpairs.as<RankedPair>(), c->rrecs.as<char>(), len, rk_heads, nh_l); \
    } while (0)
#ifdef GNNPE_DIAG
#define GNNPE_L(EE)                                                                                                     \
    do {                                                                                                                \
        const long rr = diag_int("GNNPE_FILL_ROWS", 0); /* A/B aid: rows staged per wave between flushes */              \
        if (packed && EE <= 2 && rr) {                                                                                  \
            if (rr == 256) GNNPE_LK((k_fill_ranked<EE <= 2 ? EE : 2, true, 256>));                                        \
            else if (rr == 64) GNNPE_LK((k_fill_ranked<EE <= 2 ? EE : 2, true, 64>));                                     \
            else GNNPE_LK((k_fill_ranked<EE <= 2 ? EE : 2, true, 128>));                                                  \
        }                                                                                                               \
        else if (packed) GNNPE_LK((k_fill_ranked<EE, true, fill_rows(EE)>));                                            \
        else GNNPE_LK((k_fill_ranked<EE, false, fill_rows(EE)>));                                                       \
    } while (0)
#else
#define GNNPE_L(EE)                                                                                                     \
    do {                                                                                                                \
        if (packed) GNNPE_LK((k_fill_ranked<EE, true, fill_rows(EE)>));                                                 \
        else GNNPE_LK((k_fill_ranked<EE, false, fill_rows(EE)>));                                                       \
    } while (0)
#endif
        // pde_label is gathered from the emitted ids; without an id output of the caller's they go to scratch
        if (d_pdl && !d_vids) {
            if ((rc = c->scratch.reserve((end - begin) * 12 + 16))) return rc;
            P.out_ids = c->scratch.as<uint32_t>();
        }
        // which shape emits: one wave per start vertex (default: it is the faster one, profiles/r04_emit_ab.txt), or one
        // wave per output tile when asked for (graphs without hub rows)
        int shape = c->emit_shape;
        if (shape == 0) {  // by what was measured into this buffer, if anything was
            const void *key = d_pde ? d_pde : d_vids;
            for (const auto &pr : c->emit_prefs)
                if (pr.key == key) shape = pr.shape;
        }
        if (c->sw.emit) shape = c->sw.emit;
        if (shape == 4 && e <= 2) want_per_cu = 3;
        // emit shape 3 (persistent ticket waves, gnnpe_fill_tickets.hip.h) lost in every allocation and lives in
        // diagnostic builds only; the shipped library answers a request for it with the one-shot tile kernel.  It exists for
        // e <= 2 and addresses the record blocks and the vde table with 32-bit offsets.
#ifdef GNNPE_DIAG
        const bool tickets = c->n_hub == 0 && shape == 3 && c->n_edges != 0 && end > begin && e <= 2 &&
                             c->rrecs.bytes < (1ull << 32) && (uint64_t)c->n * e * 8 < (1ull << 32);
#else
        const bool tickets = false;
#endif
        const bool tiles = c->n_hub == 0 && (shape == 2 || (shape == 3 && !tickets)) && c->n_edges != 0;
#ifdef GNNPE_DIAG
        if (tickets && (P.out_ids || d_pde)) {
            // persistent waves, tiles in ticket order, three tiles in flight per wave (gnnpe_fill_tickets.hip.h); the tiles its
            // pipeline does not take go through a job list to k_fill_tile_jobs in a second launch
            const uint32_t ts = 64u;
            if ((rc = ensure_tile_table(c, end, ts))) return rc;
            const uint64_t t_lo = begin / ts, t_hi = (end + ts - 1) / ts;
            GNNPE_REQUIRE(t_hi < (1ull << 32), GNNPE_ERR_ARG, "emit shape 3: more than 2^32 output tiles");
            const uint64_t total_arg = c->total_known ? c->total_paths : ~0ull;
            const uint32_t tpt = (uint32_t)std::max<long>(1, std::min<long>(64, diag_int("GNNPE_TICKET_TILES", 4)));
            const uint32_t nh = (uint32_t)std::max<long>(4, std::min<long>(kTicketHeads, diag_int("GNNPE_TICKET_HEADS", kTicketHeads) & ~3L));
            const uint64_t job_cap = (t_hi - t_lo) + c->n_edges / kJobStrip + 8;
            if ((rc = c->tk_ctl.reserve(kTicketCtlWords * 4 + 64)) || (rc = c->tk_jobs.reserve(job_cap * sizeof(uint2)))) return rc;
            GNNPE_HIP_TRY(hipMemsetAsync(c->tk_ctl.p, 0, kTicketCtlWords * 4 + 64, c->stream));
            c->heads_clean = false;
            uint32_t *ctl = c->tk_ctl.as<uint32_t>();
            const int occ_env = (int)diag_int("GNNPE_TICKET_OCC", 0);
            const uint32_t tk_exp = (uint32_t)diag_int("GNNPE_TICKET_EXP", 0);  // 1 no stores, 2 no record loads, 16 stamps
#define GNNPE_LQ(EE, PK, SP, OCC)                                                                                          \
    do {                                                                                                                   \
        auto kern = k_fill_tickets<EE, PK, SP, 4, OCC>;                                                                    \
        int per_cu = blocks_per_cu(kern);                                                                                  \
        if (occ_env > 0) per_cu = std::min(per_cu, occ_env);                                                               \
        const uint64_t want = (t_hi - t_lo + 3) / 4;                                                                       \
        const dim3 grid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)per_cu * c->num_cus))), block(kBlock); \
        const uint32_t nh_l = std::min<uint32_t>(nh, grid.x * 4u); /* every head needs a workgroup that serves it */       \
        hipLaunchKernelGGL(kern, grid, block, 0, c->stream, P, c->tfirst.as<uint64_t>(), c->rpairs.as<RankedPair>(),       \
                           c->pst.as<uint2>(), c->rrecs.as<char>(), (uint32_t)c->rrecs.bytes, (uint32_t)((uint64_t)c->n * e * 8), \
                           t_lo, t_hi, total_arg, tpt, nh_l, tk_exp, ctl, c->tk_jobs.as<uint2>());                         \
        auto jkern = k_fill_tile_jobs<EE, PK, (int)kJobStrip, 4>;                                                          \
        const dim3 jgrid((unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)blocks_per_cu(jkern) * c->num_cus))); \
        hipLaunchKernelGGL(jkern, jgrid, block, 0, c->stream, P, c->tfirst.as<uint64_t>(), c->rpairs.as<RankedPair>(),     \
                           c->pst.as<uint2>(), c->rrecs.as<char>(), total_arg, ctl + kTicketHeads * kTicketHeadWords,      \
                           c->tk_jobs.as<uint2>());                                                                        \
    } while (0)
#define GNNPE_LQS(EE, SP, OCC)                                                    \
    do {                                                                          \
        if (packed) GNNPE_LQ(EE, true, SP, OCC); else GNNPE_LQ(EE, false, SP, OCC); \
    } while (0)
            if (e == 1) GNNPE_LQS(1, 32, 5);
            else GNNPE_LQS(2, 32, 5);
#undef GNNPE_LQS
#undef GNNPE_LQ
            if (tk_exp & 16u) {
                unsigned long long h[8];
                GNNPE_HIP_TRY(hipMemcpyAsync(h, ctl + kTicketCtlWords, 64, hipMemcpyDeviceToHost, c->stream));
                GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
                const double tiles_w = (double)(t_hi - t_lo) / 16.0;  // tiles of the stamped waves (one wave in 16), about
                fprintf(stderr, "[ticket stamps] %llu waves stamped; cycles per tile: wait %.0f  park %.0f  strip+records %.0f  pairs %.0f  ticket %.0f  stores %.0f\n",
                        h[7], h[0] / tiles_w, h[1] / tiles_w, h[2] / tiles_w, h[3] / tiles_w, 0.0, h[4] / tiles_w);
            }
            c->last_emit_kernel = "k_fill_tickets";
        } else
#endif
        if (tiles && (P.out_ids || d_pde)) {
            // rows per tile in units of 64 (GNNPE_TILE_SHAPE=<units> for A/B runs)
            int kt = diag_int("GNNPE_TILE_SHAPE", 1) == 2 ? 2 : 1;
            if (e > 2) kt = 1;
            const uint32_t ts = 64u * (uint32_t)kt;
            if ((rc = ensure_tile_table(c, end, ts))) return rc;
            const uint64_t t_lo = begin / ts, t_hi = (end + ts - 1) / ts;
            const uint64_t total_arg = c->total_known ? c->total_paths : ~0ull;
            // GNNPE_TILE_EXP (diagnostic instantiation, e = 2 with packed ids only): bit 0 no stores, bit 1 no record loads,
            // bit 4 in-kernel stamps (cycles per phase of one wave in 64, printed to stderr)
            uint32_t xf = (uint32_t)diag_int("GNNPE_TILE_EXP", 0);  // diagnostic builds only: the knock-outs produce wrong rows by design
            if (!(e == 2 && packed)) xf = 0;
            const dim3 grid((unsigned)((t_hi - t_lo + 3) / 4)), block(kBlock);
            unsigned long long *d_stamps = nullptr;
            if (xf & 16u) {
                d_stamps = reinterpret_cast<unsigned long long *>(c->small.as<char>() + 2048);
                GNNPE_HIP_TRY(hipMemsetAsync(d_stamps, 0, 64, c->stream));
            }
#define GNNPE_LTK(EE, PK, KT, SP, DG)                                                                                    \
    hipLaunchKernelGGL((k_fill_tiles<EE, PK, KT, SP, 4, DG>), grid, block, 0, c->stream, P, c->tfirst.as<uint64_t>(),    \
                       c->rpairs.as<RankedPair>(), c->pst.as<uint2>(), c->rrecs.as<char>(), t_lo, t_hi, total_arg, xf, d_stamps)
#define GNNPE_LTS(EE, KT, SP)                                                         \
    do {                                                                              \
        if (packed) GNNPE_LTK(EE, true, KT, SP, false); else GNNPE_LTK(EE, false, KT, SP, false); \
    } while (0)
#ifdef GNNPE_DIAG
            if (xf) {
                if (kt == 2) GNNPE_LTK(2, true, 2, 64, true); else GNNPE_LTK(2, true, 1, 64, true);
            } else
#endif
#ifdef GNNPE_DIAG
            if (kt == 2 && e == 1) GNNPE_LTS(1, 2, 64);
            else if (kt == 2 && e == 2) GNNPE_LTS(2, 2, 64);
            else
#endif
            if (e == 1) {
                GNNPE_LTS(1, 1, 64);
            } else if (e == 2) {
                GNNPE_LTS(2, 1, 64);
            } else if (e == 3) {
                GNNPE_LTS(3, 1, 32);
            } else if (e == 4) {
                GNNPE_LTS(4, 1, 32);
            } else {
                GNNPE_LTS(8, 1, 16);
            }
#undef GNNPE_LTS
#undef GNNPE_LTK
            if (d_stamps) {
                unsigned long long h[8];
                GNNPE_HIP_TRY(hipMemcpyAsync(h, d_stamps, 64, hipMemcpyDeviceToHost, c->stream));
                GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
                const double nwv = (double)((t_hi - t_lo + 63) / 64);
                fprintf(stderr, "[tile stamps] cycles per wave: table %.0f  pairs %.0f  records %.0f  park+issue %.0f  store drain %.0f\n",
                        h[0] / nwv, h[1] / nwv, h[2] / nwv, h[3] / nwv, h[4] / nwv);
            }
            c->last_emit_kernel = "k_fill_tiles";
        } else if (P.out_ids || d_pde) {
            GNNPE_BY_E(e, GNNPE_L)
            c->last_emit_kernel = "k_fill_ranked";
            c->last_emit_per_cu = want_per_cu;
        }
#undef GNNPE_L
#undef GNNPE_LK
        if (d_part)
            hipLaunchKernelGGL(k_start_parts, dim3(grid_for((uint64_t)len * 64)), dim3(kBlock), 0, c->stream, len, sr, begin, end,
                               (uint32_t *)d_part);
        if (d_pdl)
            hipLaunchKernelGGL(k_pdl_from_ids, dim3(grid_for((end - begin) * 3 * e)), dim3(kBlock), 0, c->stream, end - begin,
                               3u, e, P.out_ids, c->x.as<double>(), (double *)d_pdl);
    } else {  // kVarPairWave: runtime embedding width
        hipLaunchKernelGGL(k_fill_edge_wave, dim3(grid_for(c->n_edges * 64)), dim3(kBlock), 0, c->stream, P);
    }
    GNNPE_HIP_TRY(hipGetLastError());
    return GNNPE_OK;
}

int gnnpe_fill_paths_device(gnnpe_ctx *c, uint64_t begin, uint64_t end, void *dev_vids, void *dev_pde,
                            void *dev_pde_label)
{
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    return fill_device(c, begin, end, dev_vids, dev_pde, dev_pde_label, nullptr);
}

int gnnpe_fill_paths_capped_device(gnnpe_ctx *c, uint64_t cap_rows, void *dev_vids, void *dev_pde)
{
    GNNPE_REQUIRE(c, GNNPE_ERR_ARG, "null context");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    return fill_device(c, 0, cap_rows, dev_vids, dev_pde, nullptr, nullptr, true);
}

int gnnpe_path_partitions_device(gnnpe_ctx *c, uint64_t begin, uint64_t end, void *dev_part)
{
    GNNPE_REQUIRE(c && dev_part, GNNPE_ERR_ARG, "null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    return fill_device(c, begin, end, nullptr, nullptr, nullptr, dev_part);
}

int gnnpe_fill_paths(gnnpe_ctx *c, uint64_t begin, uint64_t end, uint32_t *hv, double *hpde, double *hpdl)
{
    GNNPE_REQUIRE(c && c->counted, GNNPE_ERR_ARG, "gnnpe_fill_paths: call gnnpe_count_paths first");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc_t = resolve_total(c);
    if (rc_t) return rc_t;
    GNNPE_REQUIRE(begin <= end && end <= c->total_paths, GNNPE_ERR_ARG, "bad path range");
    const uint64_t cnt = end - begin;
    if (!cnt) return GNNPE_OK;
    const uint32_t e = c->have_table ? c->e : 2, L = c->l + 1, D = L * e;
    DevBuf bv, bp, bl;
    int rc = GNNPE_OK;
    if (hv) rc = bv.reserve(cnt * L * 4);
    if (!rc && hpde) rc = bp.reserve(cnt * D * 8);
    if (!rc && hpdl) rc = bl.reserve(cnt * D * 8);
    if (!rc) rc = fill_device(c, begin, end, hv ? bv.p : nullptr, hpde ? bp.p : nullptr, hpdl ? bl.p : nullptr, nullptr);
    hipError_t he = hipSuccess;
    if (!rc && hv) he = hipMemcpyAsync(hv, bv.p, cnt * L * 4, hipMemcpyDeviceToHost, c->stream);
    if (!rc && he == hipSuccess && hpde) he = hipMemcpyAsync(hpde, bp.p, cnt * D * 8, hipMemcpyDeviceToHost, c->stream);
    if (!rc && he == hipSuccess && hpdl) he = hipMemcpyAsync(hpdl, bl.p, cnt * D * 8, hipMemcpyDeviceToHost, c->stream);
    if (!rc && he == hipSuccess) he = hipStreamSynchronize(c->stream);
    if (!rc && he != hipSuccess) {
        set_error("gnnpe_fill_paths copy-back: %s", hipGetErrorString(he));
        rc = GNNPE_ERR_HIP;
    }
    (void)hipStreamSynchronize(c->stream);
    bv.release();
    bp.release();
    bl.release();
    return rc;
}

int gnnpe_rows_checksum_device(gnnpe_ctx *c, uint64_t n_rows, uint32_t L, const void *dev_ids, uint64_t first_id,
                               uint64_t *host_sum)
{
    GNNPE_REQUIRE(c && host_sum && L >= 1 && (n_rows == 0 || dev_ids), GNNPE_ERR_ARG, "gnnpe_rows_checksum_device: bad argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->small.reserve(256))) return rc;
    unsigned long long *d_sum = reinterpret_cast<unsigned long long *>(c->small.as<char>() + 128);
    GNNPE_HIP_TRY(hipMemsetAsync(d_sum, 0, 8, c->stream));
    if (n_rows)
        hipLaunchKernelGGL(k_rows_checksum, dim3(grid_for(n_rows)), dim3(kBlock), 0, c->stream, n_rows, L,
                           (const uint32_t *)dev_ids, first_id, d_sum);
    GNNPE_HIP_TRY(hipGetLastError());
    return read_back_u64(c, d_sum, 8, host_sum);
}

int gnnpe_set_emit_shape(gnnpe_ctx *c, int shape)
{
    GNNPE_REQUIRE(c && shape >= 0 && shape <= 4, GNNPE_ERR_ARG,
                  "emit shape must be 0 (as calibrated), 1 (start waves), 2 (output tiles), 3 (ticket waves) or 4 (start waves, three workgroups per CU)");
    c->emit_shape = shape;
    return GNNPE_OK;
}

const char *gnnpe_emit_kernel_name(gnnpe_ctx *c) { return c ? c->last_emit_kernel : ""; }

int gnnpe_emit_calibrate_device(gnnpe_ctx *c, uint64_t rows_cap, void *dev_vids, void *dev_pde, float *ms_by_shape, int *shape_kept)
{
    GNNPE_REQUIRE(c && (dev_vids || dev_pde), GNNPE_ERR_ARG, "gnnpe_emit_calibrate_device: no output buffer");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc = resolve_total(c);
    if (rc) return rc;
    GNNPE_REQUIRE(c->counted && c->counted_variant == kVarRanked && c->l == 2, GNNPE_ERR_ARG,
                  "gnnpe_emit_calibrate_device: needs an l=2 count of the rank-sorted enumeration on this context");
    GNNPE_REQUIRE(c->total_paths <= rows_cap, GNNPE_ERR_ARG, "gnnpe_emit_calibrate_device: the buffers hold %llu rows, the count is %llu paths",
                  (unsigned long long)rows_cap, (unsigned long long)c->total_paths);
    const void *key = dev_pde ? dev_pde : dev_vids;
    const uint32_t e_now = c->have_table ? c->e : 2;
    for (const auto &pr : c->emit_prefs)  // measured before, for this buffer and this count: the same answer without the launches
        if (pr.key == key && pr.total == c->total_paths && pr.n_edges == c->n_edges && pr.e == e_now) {
            for (int k = 0; ms_by_shape && k < 5; k++) ms_by_shape[k] = pr.ms[k];
            if (shape_kept) *shape_kept = pr.shape;
            return GNNPE_OK;
        }
    forget_emit_pref(c, key);
    float ms[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int kept = 1;
    // only one shape exists for graphs with hub rows or without paths; below 2^24 paths a launch is too short to tell
    if (c->n_hub == 0 && c->total_paths >= (1ull << 24)) {
        const int saved = c->emit_shape;
        const uint32_t e = c->have_table ? c->e : 2;
        GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
        const int shapes[3] = {1, 4, 2};  // start-vertex waves at five and at three workgroups per CU, output tiles
        for (int k = 0; k < 3 && rc == GNNPE_OK; k++) {
            const int shape = shapes[k];
            if (shape == 4 && e > 2) continue;  // (no occupancy cap at these widths: the same launch as shape 1)
            c->emit_shape = shape;
            float best = 1e30f;
            for (int rep = 0; rep < 3 && rc == GNNPE_OK; rep++) {  // the first launch touches the pages and builds the tile table
                const auto t0 = std::chrono::steady_clock::now();
                rc = fill_device(c, 0, c->total_paths, dev_vids, dev_pde, nullptr, nullptr);
                if (rc == GNNPE_OK && hipStreamSynchronize(c->stream) != hipSuccess) {
                    set_error("gnnpe_emit_calibrate_device: launch failed: %s", hipGetErrorString(hipGetLastError()));
                    rc = GNNPE_ERR_HIP;
                }
                const float t = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
                if (rep > 0) best = std::min(best, t);
            }
            ms[shape] = best;
        }
        c->emit_shape = saved;
        if (rc) return rc;
        for (int k = 0; k < 3; k++)
            if (ms[shapes[k]] > 0.f && ms[shapes[k]] < ms[kept]) kept = shapes[k];
    }
    if (c->emit_prefs.size() >= 16) c->emit_prefs.erase(c->emit_prefs.begin());
    c->emit_prefs.push_back(gnnpe_ctx::EmitPref{key, kept, c->total_paths, c->n_edges, e_now, {ms[0], ms[1], ms[2], ms[3], ms[4]}});
    for (int k = 0; ms_by_shape && k < 5; k++) ms_by_shape[k] = ms[k];
    if (shape_kept) *shape_kept = kept;
    return GNNPE_OK;
}

int gnnpe_set_fill_variant(gnnpe_ctx *c, int variant)
{
    GNNPE_REQUIRE(c && (variant == kVarPairWave || variant == kVarRanked), GNNPE_ERR_ARG, "fill variant must be 1 or 4");
    if (variant != c->fill_variant) c->counted = false;
    c->fill_variant = variant;
    return GNNPE_OK;
}

// ---- halo helpers -------------------------------------------------------------------------------
int gnnpe_halo_need(gnnpe_ctx *c, uint32_t n_ranks, const uint32_t *bounds, void *dev_ids, uint64_t cap,
                    uint64_t *host_counts)
{
    GNNPE_REQUIRE(c && bounds && host_counts && n_ranks >= 1 && n_ranks <= 64, GNNPE_ERR_ARG,
                  "gnnpe_halo_need: null argument / more than 64 ranks");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_rank_arrays(c);
    if (rc) return rc;
    const uint32_t n = c->n;
    for (uint32_t r = 0; r < n_ranks; r++) host_counts[r] = 0;
    if (n == 0) return GNNPE_OK;
    // layout of `mark`: mark[n] | key_in[n] | key_out[n] | (aligned) ids_in[n] | ids_out[n]
    const size_t a4 = ((size_t)3 * n + 15) & ~(size_t)15;
    if ((rc = c->mark.reserve(a4 + (size_t)8 * n + 64)) || (rc = c->small.reserve(256 + 65 * 4 + 64 * 8))) return rc;
    uint8_t *mark = c->mark.as<uint8_t>(), *key_in = mark + n, *key_out = mark + 2 * (size_t)n;
    uint32_t *ids_in = reinterpret_cast<uint32_t *>(mark + a4), *ids_out = ids_in + n;
    uint32_t *d_bounds = reinterpret_cast<uint32_t *>(c->small.as<char>() + 256);
    unsigned long long *d_hist = reinterpret_cast<unsigned long long *>(c->small.as<char>() + 256 + 65 * 4 + 4);
    GNNPE_HIP_TRY(hipMemsetAsync(mark, 0, n, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(d_hist, 0, 64 * 8, c->stream));
    GNNPE_HIP_TRY(hipMemcpyAsync(d_bounds, bounds, (size_t)(n_ranks + 1) * 4, hipMemcpyHostToDevice, c->stream));
    // every vertex referenced by a row on the device (owned rows on the first hop; owned + 1-hop rows when the
    // caller asks again without dropping the halo: the 2-hop rows l=3 needs)
    if (c->nbr_used)
        hipLaunchKernelGGL(k_mark_needed, dim3(grid_for(c->nbr_used)), dim3(kBlock), 0, c->stream, c->nbr_used,
                           c->nbrs.as<uint32_t>(), mark);
    hipLaunchKernelGGL(k_need_owner, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, n, n_ranks, d_bounds, mark,
                       c->present.as<uint8_t>(), c->rank.as<uint32_t>(), key_in, ids_in);
    GNNPE_HIP_TRY(hipGetLastError());
    size_t tb = 0;
    GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, key_in, key_out, ids_in, ids_out, (int)n, 0, 8, c->stream));
    if ((rc = c->cub_tmp.reserve(tb))) return rc;
    tb = c->cub_tmp.bytes;
    GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, key_in, key_out, ids_in, ids_out, (int)n, 0, 8, c->stream));
    hipLaunchKernelGGL(k_key_counts, dim3(1), dim3(64), 0, c->stream, n, n_ranks, key_out, d_hist);
    GNNPE_HIP_TRY(hipGetLastError());
    GNNPE_HIP_TRY(hipMemcpyAsync(c->h_pinned, d_hist, (size_t)n_ranks * 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));  // bounds[] (caller memory) and the histogram are done
    uint64_t used = 0;
    for (uint32_t r = 0; r < n_ranks; r++) {
        host_counts[r] = c->h_pinned[r];
        used += host_counts[r];
    }
    if (used) {
        GNNPE_REQUIRE(dev_ids && used <= cap, GNNPE_ERR_ARG, "gnnpe_halo_need: id buffer too small (%llu needed)",
                      (unsigned long long)used);
        GNNPE_HIP_TRY(hipMemcpyAsync(dev_ids, ids_out, used * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    return GNNPE_OK;
}

int gnnpe_rows_drop_halo(gnnpe_ctx *c)
{
    GNNPE_REQUIRE(c && c->have_graph, GNNPE_ERR_ARG, "gnnpe_rows_drop_halo: no graph");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    if (c->rows_identity || c->n_held == c->n_rows) return GNNPE_OK;
    hipLaunchKernelGGL(k_drop_halo, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, c->n, c->owned.as<uint8_t>(),
                       c->present.as<uint8_t>(), c->adj_deg.as<uint32_t>());
    GNNPE_HIP_TRY(hipGetLastError());
    c->nbr_used = c->nbr_owned;
    c->n_held = c->n_rows;
    c->halo_min_rank = 0;
    c->nbr_vde_valid = false;
    c->counted = false;
    return finish_rows(c, 0, nullptr);  // the hub list covers the held rows
}

int gnnpe_rows_degree(gnnpe_ctx *c, uint64_t n_req, const void *dev_ids, void *dev_deg)
{
    GNNPE_REQUIRE(c && c->have_graph, GNNPE_ERR_ARG, "gnnpe_rows_degree: no graph");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    if (!n_req) return GNNPE_OK;
    GNNPE_REQUIRE(dev_ids && dev_deg, GNNPE_ERR_ARG, "null argument");
    hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(n_req)), dim3(kBlock), 0, c->stream, n_req, (const uint32_t *)dev_ids,
                       c->adj_deg.as<uint32_t>(), (uint32_t *)dev_deg);
    GNNPE_HIP_TRY(hipGetLastError());
    return GNNPE_OK;
}

int gnnpe_rows_pack(gnnpe_ctx *c, uint64_t n_req, const void *dev_ids, void *dev_out, uint64_t cap)
{
    GNNPE_REQUIRE(c && c->have_graph, GNNPE_ERR_ARG, "gnnpe_rows_pack: no graph");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    if (!n_req) return GNNPE_OK;
    GNNPE_REQUIRE(dev_ids && dev_out, GNNPE_ERR_ARG, "null argument");
    int rc;
    if ((rc = c->scratch.reserve((n_req + 1) * 12 + 64))) return rc;
    uint64_t *roff = c->scratch.as<uint64_t>();
    uint32_t *deg = reinterpret_cast<uint32_t *>(roff + n_req + 1);
    hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(n_req)), dim3(kBlock), 0, c->stream, n_req, (const uint32_t *)dev_ids,
                       c->adj_deg.as<uint32_t>(), deg);
    GNNPE_HIP_TRY(hipMemsetAsync(deg + n_req, 0, 4, c->stream));
    if ((rc = scan_u32_to_u64(c, deg, roff, n_req + 1))) return rc;
    uint64_t tot = 0;
    if ((rc = read_back_u64(c, roff + n_req, 8, &tot))) return rc;
    GNNPE_REQUIRE(tot <= cap, GNNPE_ERR_ARG, "gnnpe_rows_pack: output holds %llu entries, need %llu",
                  (unsigned long long)cap, (unsigned long long)tot);
    hipLaunchKernelGGL(k_rows_pack, dim3(grid_for(n_req * 16)), dim3(kBlock), 0, c->stream, n_req,
                       (const uint32_t *)dev_ids, roff, c->adj_start.as<uint32_t>(), c->nbrs.as<uint32_t>(),
                       (uint32_t *)dev_out);
    GNNPE_HIP_TRY(hipGetLastError());
    return GNNPE_OK;
}

int gnnpe_rows_append(gnnpe_ctx *c, uint64_t n_rows, const void *dev_ids, const void *dev_deg, const void *dev_nbrs,
                      uint64_t n_nbrs, uint32_t min_rank)
{
    GNNPE_REQUIRE(c && c->have_graph, GNNPE_ERR_ARG, "gnnpe_rows_append: no graph");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    if (!n_rows) return GNNPE_OK;
    GNNPE_REQUIRE(dev_ids && dev_deg && (dev_nbrs || !n_nbrs), GNNPE_ERR_ARG, "null argument");
    GNNPE_REQUIRE(min_rank == 0 || c->have_order, GNNPE_ERR_ARG, "gnnpe_rows_append: min_rank needs gnnpe_set_order first");
    GNNPE_REQUIRE(!c->rows_identity && (uint64_t)c->n_held + n_rows <= c->n, GNNPE_ERR_ARG,
                  "gnnpe_rows_append: more rows than vertices");
    int rc;
    // scratch: src_off u64[n_rows+1] | dst_off u64[n_rows+1] | deg u32[n_rows+1] | kept u32[n_rows+1]
    if ((rc = c->scratch.reserve((n_rows + 1) * 24 + 64))) return rc;
    uint64_t *src_off = c->scratch.as<uint64_t>(), *dst_off = src_off + n_rows + 1;
    uint32_t *deg = reinterpret_cast<uint32_t *>(dst_off + n_rows + 1), *kept = deg + n_rows + 1;
    GNNPE_HIP_TRY(hipMemcpyAsync(deg, dev_deg, n_rows * 4, hipMemcpyDeviceToDevice, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(deg + n_rows, 0, 4, c->stream));
    if ((rc = scan_u32_to_u64(c, deg, src_off, n_rows + 1))) return rc;
    uint64_t tot = 0;
    if ((rc = read_back_u64(c, src_off + n_rows, 8, &tot))) return rc;
    GNNPE_REQUIRE(tot == n_nbrs, GNNPE_ERR_ARG, "gnnpe_rows_append: degrees sum to %llu but %llu entries given",
                  (unsigned long long)tot, (unsigned long long)n_nbrs);
    uint64_t n_keep = n_nbrs;
    const uint64_t *roff = src_off;
    if (min_rank > 0 && n_nbrs) {
        // drop the entries ranked before the slab: they can never close a path of this rank
        hipLaunchKernelGGL(k_rows_kept_counts, dim3(grid_for(n_rows * 16)), dim3(kBlock), 0, c->stream, n_rows, src_off,
                           (const uint32_t *)dev_nbrs, c->rank.as<uint32_t>(), c->n, min_rank, kept);
        if ((rc = scan_u32_to_u64(c, kept, dst_off, n_rows + 1))) return rc;
        if ((rc = read_back_u64(c, dst_off + n_rows, 8, &n_keep))) return rc;
        roff = dst_off;
    }
    GNNPE_REQUIRE(c->nbr_used + n_keep <= c->nbr_cap, GNNPE_ERR_ARG,
                  "gnnpe_rows_append: neighbour buffer full (%llu + %llu > %llu); reserve more in gnnpe_load_rows",
                  (unsigned long long)c->nbr_used, (unsigned long long)n_keep, (unsigned long long)c->nbr_cap);
    if (roff == dst_off)
        hipLaunchKernelGGL(k_rows_compact, dim3(grid_for(n_rows * 64)), dim3(kBlock), 0, c->stream, n_rows, src_off,
                           (const uint32_t *)dev_nbrs, c->rank.as<uint32_t>(), c->n, min_rank, dst_off,
                           c->nbrs.as<uint32_t>() + c->nbr_used);
    else if (n_nbrs)
        GNNPE_HIP_TRY(hipMemcpyAsync(c->nbrs.as<uint32_t>() + c->nbr_used, dev_nbrs, n_nbrs * 4, hipMemcpyDeviceToDevice,
                                     c->stream));
    GNNPE_HIP_TRY(hipMemcpyAsync(c->held.as<uint32_t>() + c->n_held, dev_ids, n_rows * 4, hipMemcpyDeviceToDevice,
                                 c->stream));
    hipLaunchKernelGGL(k_install_rows, dim3(grid_for(n_rows)), dim3(kBlock), 0, c->stream, n_rows,
                       (const uint32_t *)dev_ids, roff, c->nbr_used, c->adj_start.as<uint32_t>(),
                       c->adj_deg.as<uint32_t>(), c->present.as<uint8_t>());
    GNNPE_HIP_TRY(hipGetLastError());
    const uint32_t first_new = c->n_held;
    c->halo_min_rank = std::max(c->halo_min_rank, min_rank);  // these rows lack their entries ranked below min_rank
    c->nbr_used += n_keep;
    c->n_held += (uint32_t)n_rows;
    c->nbr_vde_valid = false;
    c->counted = false;
    // validation, reverse positions and the hub list: graph structure, built when the rows arrive (synchronises)
    return finish_rows(c, n_rows, c->held.as<uint32_t>() + first_new);
}

}  // extern "C"
