// gnnpe_filter.hip -- SURVEY 8(f) row 4: the online FILTER, data side.
//
// The reference answers a query by walking each partition's R*-tree best-first (Partition::query,
// GNN-PE/include/custom.h:366-489); the traversal only prunes -- what it reports is defined by its leaf test
// (custom.h:404-431): data path p matches query path j iff, position by position, labels are equal and the query
// degree does not exceed the data degree, and in no embedding dimension the query's pde exceeds the data pde by
// more than epsilon; each match inserts p's vertices into the candidate sets of j's vertices (custom.h:429-432),
// and the sets of all partitions are united (main.cpp:165-171).  On the GPU the index is unnecessary: the test
// is applied to every enumerated path, 2e8 paths against a plan of a few query paths in a few milliseconds,
// with no files, no R-tree and no 100-second text re-parse (custom.h:546-572) in between -- and without emitting a
// single path: the test is fused into the enumeration itself (k_filter_starts).
// Candidate sets are bitmaps: row u (query vertex) x ceil(n/32) words, bit v = data vertex v is a candidate.
#include "gnnpe_common.h"

namespace gnnpe {

constexpr int kMaxPlan = 512;  // query paths held in LDS

// One wave per start vertex s of the slab.  A pair (s, b) is dropped as soon as no plan path begins with
// (label s, label b) within the two degrees -- with |labels|^2 label pairs and a handful of plan paths that is almost
// every pair, before the row of b is touched.  Only the surviving pairs scan N(b): c is kept iff rank[c] > rank[s]
// (the enumeration's rule, custom.h:66-92 in closed form), then the full leaf test runs.  Needs the CSR rows of the
// slab and its 1-hop halo, labels, ranks, vde and every vertex' degree; no counts, no records, any degree, any e.
__global__ __launch_bounds__(256) void k_filter_starts(uint32_t slab_begin, uint32_t slab_len,
                                                       const uint32_t *__restrict__ sorted,
                                                       const uint32_t *__restrict__ adj_start,
                                                       const uint32_t *__restrict__ adj_deg,
                                                       const uint32_t *__restrict__ nbrs,
                                                       const uint32_t *__restrict__ labels,
                                                       const uint32_t *__restrict__ rank, const uint32_t *__restrict__ deg,
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
    const unsigned lane = threadIdx.x & 63u;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < slab_len; w += nw) {
        const uint32_t thr = slab_begin + (uint32_t)w, s = sorted[thr];
        const uint32_t a_s = adj_start[s], ds = adj_deg[s], ls = labels[s];
        bool any = false;  // wave-uniform: does any plan path start like s?
        for (uint32_t j = 0; j < n_qp && !any; j++) any = s_lab[j * 3] == ls && s_deg[j * 3] <= ds;
        if (!any) continue;
        for (uint32_t k0 = 0; k0 < ds; k0 += 64) {
            const uint32_t k = k0 + lane;
            uint32_t b = 0, lb = 0, db = 0;
            bool hit = false;
            if (k < ds) {
                b = nbrs[a_s + k];
                lb = labels[b];
                db = deg[b];
                for (uint32_t j = 0; j < n_qp && !hit; j++)
                    hit = s_lab[j * 3] == ls && s_lab[j * 3 + 1] == lb && s_deg[j * 3] <= ds && s_deg[j * 3 + 1] <= db;
            }
            uint64_t live = __ballot(hit);
            while (live) {  // surviving pairs, one at a time, the row of b over the lanes
                const int kk = __ffsll((long long)live) - 1;
                live &= live - 1;
                const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)b, kk);
                const uint32_t lbb = (uint32_t)__builtin_amdgcn_readlane((int)lb, kk);
                const uint32_t dbb = (uint32_t)__builtin_amdgcn_readlane((int)db, kk);
                const uint32_t b_st = adj_start[bb], b_d = adj_deg[bb];
                for (uint32_t j0 = 0; j0 < b_d; j0 += 64) {
                    if (j0 + lane >= b_d) continue;
                    const uint32_t c = nbrs[b_st + j0 + lane];
                    if (rank[c] <= thr) continue;  // not emitted from s: c == s, or the path belongs to start c
                    const uint32_t lc = labels[c];
                    for (uint32_t q = 0; q < n_qp; q++) {
                        if (s_lab[q * 3] != ls || s_lab[q * 3 + 1] != lbb || s_lab[q * 3 + 2] != lc) continue;  // custom.h:410
                        if (s_deg[q * 3] > ds || s_deg[q * 3 + 1] > dbb || s_deg[q * 3 + 2] > deg[c]) continue;
                        bool ok = true;
                        const double *qp = q_pde + (uint64_t)q * 3 * e;
                        for (uint32_t t = 0; t < e && ok; t++) {                                              // custom.h:420-426
                            const double a0 = vde[(uint64_t)s * e + t], a1 = vde[(uint64_t)bb * e + t], a2 = vde[(uint64_t)c * e + t];
                            if (qp[t] > a0 && fabs(qp[t] - a0) > eps) ok = false;
                            if (qp[e + t] > a1 && fabs(qp[e + t] - a1) > eps) ok = false;
                            if (qp[2 * e + t] > a2 && fabs(qp[2 * e + t] - a2) > eps) ok = false;
                        }
                        if (!ok) continue;
                        atomicOr(&bitmap[s_vid[q * 3] * words + (s >> 5)], 1u << (s & 31u));                // custom.h:429-432
                        atomicOr(&bitmap[s_vid[q * 3 + 1] * words + (bb >> 5)], 1u << (bb & 31u));
                        atomicOr(&bitmap[s_vid[q * 3 + 2] * words + (c >> 5)], 1u << (c & 31u));
                    }
                }
            }
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
    c->aux_vdl_valid = false;
    return GNNPE_OK;
}

int gnnpe_filter_candidates(gnnpe_ctx *c, uint32_t n_paths, const uint32_t *q_vids, const uint32_t *q_labels,
                            const uint32_t *q_degrees, const double *q_pde, uint32_t n_query_vertices, double epsilon,
                            uint32_t *host_bitmap, double *device_ms)
{
    GNNPE_REQUIRE(c && host_bitmap && n_query_vertices, GNNPE_ERR_ARG, "gnnpe_filter_candidates: null argument");
    GNNPE_REQUIRE(n_paths == 0 || (q_vids && q_labels && q_degrees && q_pde), GNNPE_ERR_ARG, "null query plan");
    GNNPE_REQUIRE(n_paths <= (uint32_t)kMaxPlan, GNNPE_ERR_UNSUPPORTED, "query plan of %u paths (limit %d)", n_paths, kMaxPlan);
    GNNPE_REQUIRE(c->have_graph && c->have_order && c->have_vde, GNNPE_ERR_ARG,
                  "gnnpe_filter_candidates: needs the graph, the order (gnnpe_set_order) and gnnpe_vde");
    GNNPE_REQUIRE(c->rows_identity || c->have_deg_all, GNNPE_ERR_UNSUPPORTED,
                  "the filter needs every vertex' degree: load the whole graph (gnnpe_load_csr) or call gnnpe_set_degrees");
    for (uint32_t i = 0; i < n_paths * 3; i++)
        GNNPE_REQUIRE(q_vids[i] < n_query_vertices, GNNPE_ERR_ARG, "query path vertex %u >= %u", q_vids[i], n_query_vertices);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint32_t e = c->e;
    const uint64_t words = ((uint64_t)c->n + 31) / 32, bm_bytes = (uint64_t)n_query_vertices * words * 4;
    DevBuf &plan = c->q_plan, &bm = c->q_bitmap;  // context-owned: a query allocates nothing new
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
    const uint32_t len = c->slab_end - c->slab_begin;
    if (he == hipSuccess && n_paths && len) {
        hipLaunchKernelGGL(k_filter_starts, dim3(grid_for((uint64_t)len * 64)), dim3(256), 0, c->stream, c->slab_begin, len,
                           c->sorted.as<uint32_t>(), c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(),
                           c->nbrs.as<uint32_t>(), c->labels.as<uint32_t>(), c->rank.as<uint32_t>(),
                           c->have_deg_all ? c->deg_all.as<uint32_t>() : c->adj_deg.as<uint32_t>(), c->vde.as<double>(), e,
                           n_paths, d_vids, d_lab, d_deg, d_pde, epsilon, words, bm.as<uint32_t>());
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
