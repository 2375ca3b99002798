// gnnpe_filter.hip -- SURVEY 8(f) row 4: the online FILTER, data side.
//
// The reference answers a query by walking each partition's R*-tree best-first (Partition::query,
// GNN-PE/include/custom.h:366-489); the traversal only prunes -- what it reports is defined by its leaf test
// (custom.h:404-431): data path p matches query path j iff, position by position, labels are equal and the query
// degree does not exceed the data degree, and in no embedding dimension the query's pde exceeds the data pde by
// more than epsilon; each match inserts p's vertices into the candidate sets of j's vertices (custom.h:429-432),
// and the sets of all partitions are united (main.cpp:165-171).  On the GPU the index is unnecessary: the test
// is applied to every enumerated path, 2e8 paths against a plan of a few query paths in a few milliseconds,
// with no files, no R-tree and no 100-second text re-parse (custom.h:546-572) in between.  With the ranked records of
// variant 4 on the device the test is fused into the enumeration (gnnpe_filter_ranked.hip.h) and nothing is emitted;
// k_filter_paths below is the general form over emitted ids (any degree, any embedding width).
// Candidate sets are bitmaps: row u (query vertex) x ceil(n/32) words, bit v = data vertex v is a candidate.
#include "gnnpe_common.h"

namespace gnnpe {

constexpr int kMaxPlan = 512;  // query paths held in LDS (same limit as kMaxPlanPaths of the fused kernel)

__global__ __launch_bounds__(256) void k_filter_paths(uint64_t cnt, const uint32_t *__restrict__ ids,
                                                      const uint32_t *__restrict__ labels,
                                                      const uint32_t *__restrict__ adj_deg,
                                                      const double *__restrict__ vde, uint32_t e, uint32_t n_qp,
                                                      const uint32_t *__restrict__ q_vids,
                                                      const uint32_t *__restrict__ q_labels,
                                                      const uint32_t *__restrict__ q_deg,
                                                      const double *__restrict__ q_pde, double eps, uint64_t words,
                                                      uint32_t *__restrict__ bitmap)
{
    __shared__ uint32_t s_lab[kMaxPlan * 3], s_deg[kMaxPlan * 3], s_vid[kMaxPlan * 3];
    for (uint32_t i = threadIdx.x; i < n_qp * 3; i += blockDim.x) {
        s_lab[i] = q_labels[i];
        s_deg[i] = q_deg[i];
        s_vid[i] = q_vids[i];
    }
    __syncthreads();
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < cnt; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v0 = ids[p * 3], v1 = ids[p * 3 + 1], v2 = ids[p * 3 + 2];
        const uint32_t l0 = labels[v0], l1 = labels[v1], l2 = labels[v2];
        for (uint32_t j = 0; j < n_qp; j++) {
            if (s_lab[j * 3] != l0 || s_lab[j * 3 + 1] != l1 || s_lab[j * 3 + 2] != l2) continue;          // custom.h:410
            if (s_deg[j * 3] > adj_deg[v0] || s_deg[j * 3 + 1] > adj_deg[v1] || s_deg[j * 3 + 2] > adj_deg[v2]) continue;
            bool ok = true;
            for (uint32_t t = 0; t < 3 * e && ok; t++) {                                                      // custom.h:420-426
                const uint32_t v = t / e == 0 ? v0 : (t / e == 1 ? v1 : v2);
                const double q = q_pde[(uint64_t)j * 3 * e + t], d = vde[(uint64_t)v * e + t % e];
                if (q > d && fabs(q - d) > eps) ok = false;
            }
            if (!ok) continue;
            atomicOr(&bitmap[s_vid[j * 3] * words + (v0 >> 5)], 1u << (v0 & 31u));                          // custom.h:429-432
            atomicOr(&bitmap[s_vid[j * 3 + 1] * words + (v1 >> 5)], 1u << (v1 & 31u));
            atomicOr(&bitmap[s_vid[j * 3 + 2] * words + (v2 >> 5)], 1u << (v2 & 31u));
        }
    }
}

}  // namespace gnnpe

using namespace gnnpe;

extern "C" {

int gnnpe_set_degrees(gnnpe_ctx *c, const uint32_t *host_degrees)
{
    GNNPE_REQUIRE(c && host_degrees && c->have_graph, GNNPE_ERR_ARG, "gnnpe_set_degrees: load the rows first");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->deg_all.reserve(((size_t)c->n + 1) * 4))) return rc;
    GNNPE_HIP_TRY(hipMemcpyAsync(c->deg_all.p, host_degrees, (size_t)c->n * 4, hipMemcpyHostToDevice, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_deg_all = true;
    return GNNPE_OK;
}

int gnnpe_filter_candidates(gnnpe_ctx *c, uint32_t n_paths, const uint32_t *q_vids, const uint32_t *q_labels,
                            const uint32_t *q_degrees, const double *q_pde, uint32_t n_query_vertices, double epsilon,
                            uint32_t *host_bitmap, double *device_ms)
{
    GNNPE_REQUIRE(c && host_bitmap && n_query_vertices, GNNPE_ERR_ARG, "gnnpe_filter_candidates: null argument");
    GNNPE_REQUIRE(n_paths == 0 || (q_vids && q_labels && q_degrees && q_pde), GNNPE_ERR_ARG, "null query plan");
    GNNPE_REQUIRE(n_paths <= (uint32_t)kMaxPlan, GNNPE_ERR_UNSUPPORTED, "query plan of %u paths (limit %d)", n_paths, kMaxPlan);
    GNNPE_REQUIRE(c->counted && c->have_vde && c->l == 2, GNNPE_ERR_ARG,
                  "gnnpe_filter_candidates: call gnnpe_vde and gnnpe_count_paths(l = 2) first");
    GNNPE_REQUIRE(c->rows_identity || c->have_deg_all, GNNPE_ERR_UNSUPPORTED,
                  "the filter needs every vertex' degree: load the whole graph (gnnpe_load_csr) or call gnnpe_set_degrees");
    for (uint32_t i = 0; i < n_paths * 3; i++)
        GNNPE_REQUIRE(q_vids[i] < n_query_vertices, GNNPE_ERR_ARG, "query path vertex %u >= %u", q_vids[i], n_query_vertices);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint32_t e = c->e;
    const uint64_t words = ((uint64_t)c->n + 31) / 32, bm_bytes = (uint64_t)n_query_vertices * words * 4;
    const uint64_t total = c->total_paths, chunk = std::min<uint64_t>(std::max<uint64_t>(total, 1), 64ull << 20);
    DevBuf &ids = c->q_ids, &plan = c->q_plan, &bm = c->q_bitmap;  // context-owned: a query allocates nothing new
    int rc;
    const size_t np3 = (size_t)n_paths * 3;
    if ((rc = plan.reserve(np3 * 12 + np3 * e * 8 + 64)) || (rc = bm.reserve(std::max<uint64_t>(bm_bytes, 4)))) return rc;
    double *d_pde = plan.as<double>();  // doubles first (alignment), then the three uint32 arrays
    uint32_t *d_vids = reinterpret_cast<uint32_t *>(d_pde + np3 * e), *d_lab = d_vids + np3, *d_deg = d_lab + np3;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipError_t he = hipSuccess;
    if (np3) {
        he = hipMemcpyAsync(d_pde, q_pde, np3 * e * 8, hipMemcpyHostToDevice, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(d_vids, q_vids, np3 * 4, hipMemcpyHostToDevice, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(d_lab, q_labels, np3 * 4, hipMemcpyHostToDevice, c->stream);
        if (he == hipSuccess) he = hipMemcpyAsync(d_deg, q_degrees, np3 * 4, hipMemcpyHostToDevice, c->stream);
    }
    if (he == hipSuccess) he = hipMemsetAsync(bm.p, 0, std::max<uint64_t>(bm_bytes, 4), c->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(c->stream);  // the plan arrays are caller memory
    if (he == hipSuccess && device_ms) he = hipEventCreate(&ev0);
    if (he == hipSuccess && device_ms) he = hipEventCreate(&ev1);
    if (he == hipSuccess && device_ms) he = hipEventRecord(ev0, c->stream);
    rc = GNNPE_OK;
    bool fused = false;  // ranked records on the device: filter while enumerating, nothing is emitted
    if (he == hipSuccess && n_paths)
        rc = filter_fused(c, n_paths, d_vids, d_lab, d_deg, d_pde, epsilon, words, bm.as<uint32_t>(), &fused);
    if (he == hipSuccess && !rc && !fused && n_paths) rc = ids.reserve(chunk * 12);
    for (uint64_t b = 0; he == hipSuccess && !rc && !fused && b < total && n_paths; b += chunk) {
        const uint64_t cnt = std::min(total, b + chunk) - b;
        if ((rc = gnnpe_fill_paths_device(c, b, b + cnt, ids.p, nullptr, nullptr))) break;
        hipLaunchKernelGGL(k_filter_paths, dim3(grid_for(cnt)), dim3(256), 0, c->stream, cnt, ids.as<uint32_t>(),
                           c->labels.as<uint32_t>(), c->have_deg_all ? c->deg_all.as<uint32_t>() : c->adj_deg.as<uint32_t>(),
                           c->vde.as<double>(), e, n_paths, d_vids,
                           d_lab, d_deg, d_pde, epsilon, words, bm.as<uint32_t>());
        he = hipGetLastError();
    }
    if (he == hipSuccess && !rc && device_ms) he = hipEventRecord(ev1, c->stream);
    if (he == hipSuccess && !rc) he = hipMemcpyAsync(host_bitmap, bm.p, bm_bytes, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess && !rc) he = hipStreamSynchronize(c->stream);
    if (he == hipSuccess && !rc && device_ms) {
        float ms = 0.f;
        he = hipEventElapsedTime(&ms, ev0, ev1);
        *device_ms = ms;
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    (void)hipStreamSynchronize(c->stream);
    if (!rc && he != hipSuccess) {
        set_error("gnnpe_filter_candidates: %s", hipGetErrorString(he));
        rc = GNNPE_ERR_HIP;
    }
    return rc;
}

}  // extern "C"
