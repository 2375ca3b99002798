// gnnpe_refine.hip -- the REFINEMENT half of the reference's online step on the device.
//
// Definition (GNN-PE/include/custom.h:634-932, see host/refine.h): the number of embeddings of the query graph in
// the data graph -- injective, labels equal, query degree <= data degree, query edges on data edges -- whose START
// vertex maps into its candidate set, counted up to the answer limit.  host/refine.cpp is the host form; this is the
// same backtracking search with one thread per (start candidate, neighbour slot of its image): the first two levels
// of the search tree are spread over the grid, everything below is a per-thread depth-first walk over the data
// graph's adjacency with explicit iterators.  Counts are exact; the limit only stops the walk early.
#include <hipcub/hipcub.hpp>

#include <string>
#include <vector>

#include "../host/graph_loader.h"
#include "../host/refine.h"
#include "gnnpe_common.h"

namespace gnnpe {

constexpr int kMaxQueryVertices = 32;

struct RefinePlan {  // indexed by POSITION in the matching order
    uint32_t nq;
    uint32_t label[kMaxQueryVertices], degree[kMaxQueryVertices], pivot[kMaxQueryVertices];
    uint32_t back_off[kMaxQueryVertices + 1], back[kMaxQueryVertices * kMaxQueryVertices];
};

__device__ __forceinline__ bool has_edge(const uint32_t *__restrict__ nbrs, uint32_t st, uint32_t d, uint32_t target)
{
    uint32_t lo = 0, hi = d;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t x = nbrs[st + mid];
        if (x == target) return true;
        if (x < target) lo = mid + 1; else hi = mid;
    }
    return false;
}

__global__ void k_cand_degrees(uint32_t n_cand, const uint32_t *__restrict__ cand, const uint32_t *__restrict__ adj_deg,
                               uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_cand) out[i] = i < n_cand ? adj_deg[cand[i]] : 0u;
}

__global__ __launch_bounds__(256) void k_refine(RefinePlan P, uint32_t n_cand, const uint32_t *__restrict__ cand,
                                                const uint64_t *__restrict__ item_off, uint64_t n_items,
                                                const uint32_t *__restrict__ adj_start,
                                                const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ nbrs,
                                                const uint32_t *__restrict__ labels, unsigned long long limit,
                                                unsigned long long *__restrict__ total)
{
    const uint32_t nq = P.nq;
    uint32_t image[kMaxQueryVertices], it[kMaxQueryVertices], end[kMaxQueryVertices];
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < n_items; q += (uint64_t)gridDim.x * blockDim.x) {
        if (*(volatile unsigned long long *)total >= limit) return;
        // item -> (start candidate, neighbour slot): largest i with item_off[i] <= q
        uint32_t lo = 0, hi = n_cand;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (item_off[mid] <= q) lo = mid; else hi = mid;
        }
        const uint32_t v0 = cand[lo];
        image[0] = v0;
        unsigned long long found = 0;
        if (nq == 1) {
            found = 1;  // n_items = n_cand in this case
        } else {
            // level 1: its pivot is position 0
            const uint32_t v1 = nbrs[adj_start[v0] + (uint32_t)(q - item_off[lo])];
            if (labels[v1] == P.label[1] && adj_deg[v1] >= P.degree[1] && v1 != v0) {
                if (nq == 2) {
                    found = 1;
                } else {
                    image[1] = v1;
                    uint32_t depth = 2;
                    uint32_t p = image[P.pivot[2]];
                    it[2] = adj_start[p];
                    end[2] = it[2] + adj_deg[p];
                    while (depth >= 2) {
                        if (it[depth] == end[depth]) {
                            depth--;
                            continue;
                        }
                        const uint32_t v = nbrs[it[depth]++];
                        if (labels[v] != P.label[depth]) continue;
                        const uint32_t dv = adj_deg[v];
                        if (dv < P.degree[depth]) continue;
                        bool ok = true;
                        for (uint32_t i = 0; i < depth && ok; i++) ok = image[i] != v;
                        const uint32_t vs = adj_start[v];
                        for (uint32_t j = P.back_off[depth]; j < P.back_off[depth + 1] && ok; j++)
                            ok = has_edge(nbrs, vs, dv, image[P.back[j]]);
                        if (!ok) continue;
                        if (depth == nq - 1) {
                            found++;
                            if ((found & 1023ull) == 0 && *(volatile unsigned long long *)total + found >= limit) break;
                        } else {
                            image[depth] = v;
                            depth++;
                            p = image[P.pivot[depth]];
                            it[depth] = adj_start[p];
                            end[depth] = it[depth] + adj_deg[p];
                        }
                    }
                }
            }
        }
        if (found) atomicAdd(total, found);
    }
}

}  // namespace gnnpe

using namespace gnnpe;

extern "C" {

int gnnpe_refine(gnnpe_ctx *c, const char *query_graph_path, const uint32_t *candidate_bitmap, uint64_t limit,
                 uint64_t *answers, double *device_ms)
{
    GNNPE_REQUIRE(c && query_graph_path && candidate_bitmap && answers, GNNPE_ERR_ARG, "gnnpe_refine: null argument");
    GNNPE_REQUIRE(c->have_graph && c->rows_identity, GNNPE_ERR_UNSUPPORTED, "gnnpe_refine: the whole graph must be on the device (gnnpe_load_csr)");
    GNNPE_REQUIRE(!c->multigraph, GNNPE_ERR_UNSUPPORTED, "gnnpe_refine: simple graphs only (gnnpe_set_multigraph_rows was called)");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    *answers = 0;
    gnnpe_host::StaticGraph q;
    std::string err;
    int rc = q.load(query_graph_path, &err);
    if (rc != 0) {
        set_error("%s", err.c_str());
        return rc;
    }
    const uint32_t nq = q.n;
    GNNPE_REQUIRE(nq >= 1 && nq <= (uint32_t)kMaxQueryVertices, GNNPE_ERR_UNSUPPORTED, "query graphs of 1..%d vertices (got %u)",
                  kMaxQueryVertices, nq);
    if (limit == 0) return GNNPE_OK;
    const uint64_t words = ((uint64_t)c->n + 31) / 32;
    std::vector<uint64_t> cnt(nq, 0);
    for (uint32_t u = 0; u < nq; u++)
        for (uint64_t w = 0; w < words; w++) cnt[u] += (uint64_t)__builtin_popcount(candidate_bitmap[(size_t)u * words + w]);
    gnnpe_host::MatchOrder mo;
    if (gnnpe_host::build_match_order(q, cnt, &mo, &err) != 0) {
        set_error("%s", err.c_str());
        return GNNPE_ERR_ARG;
    }
    // plan by position in the order
    RefinePlan P = {};
    P.nq = nq;
    std::vector<uint32_t> pos_of(nq, 0);
    for (uint32_t i = 0; i < nq; i++) pos_of[mo.order[i]] = i;
    for (uint32_t i = 0; i < nq; i++) {
        P.label[i] = q.labels[mo.order[i]];
        P.degree[i] = q.degree(mo.order[i]);
        P.pivot[i] = pos_of[mo.pivot[i]];
        P.back_off[i] = mo.back_off[i];
    }
    P.back_off[nq] = mo.back_off[nq];
    GNNPE_REQUIRE(mo.back.size() <= (size_t)kMaxQueryVertices * kMaxQueryVertices, GNNPE_ERR_UNSUPPORTED, "query graph too dense");
    for (size_t j = 0; j < mo.back.size(); j++) P.back[j] = pos_of[mo.back[j]];
    // start candidates
    const uint32_t start = mo.order[0];
    std::vector<uint32_t> cand;
    cand.reserve(cnt[start]);
    for (uint64_t w = 0; w < words; w++)
        for (uint32_t bits = candidate_bitmap[(size_t)start * words + w]; bits; bits &= bits - 1)
            cand.push_back((uint32_t)(w * 32 + __builtin_ctz(bits)));
    const uint32_t n_cand = (uint32_t)cand.size();
    if (n_cand == 0) return GNNPE_OK;

    // context-owned scratch, carved from one grow-only buffer: [total u64 | item_off u64 x (n+1) | cand u32 x n | deg u32 x (n+1)]
    DevBuf &d_tmp = c->q_tmp;
    if ((rc = c->q_work.reserve(8 + ((size_t)n_cand + 1) * 8 + (size_t)n_cand * 4 + ((size_t)n_cand + 1) * 4 + 64))) return rc;
    unsigned long long *d_total = c->q_work.as<unsigned long long>();
    uint64_t *item_off = reinterpret_cast<uint64_t *>(d_total + 1);
    uint32_t *d_cand = reinterpret_cast<uint32_t *>(item_off + n_cand + 1), *d_deg = d_cand + n_cand;
    hipError_t he = hipMemcpyAsync(d_cand, cand.data(), (size_t)n_cand * 4, hipMemcpyHostToDevice, c->stream);
    if (he == hipSuccess) he = hipMemsetAsync(d_total, 0, 8, c->stream);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (he == hipSuccess && device_ms) he = hipEventCreate(&ev0);
    if (he == hipSuccess && device_ms) he = hipEventCreate(&ev1);
    if (he == hipSuccess && device_ms) he = hipEventRecord(ev0, c->stream);
    uint64_t n_items = n_cand;
    if (he == hipSuccess) {
        // items = (start candidate, neighbour slot); a single-vertex query has one item per candidate
        hipLaunchKernelGGL(k_cand_degrees, dim3((n_cand + 256) / 256), dim3(256), 0, c->stream, n_cand, d_cand,
                           c->adj_deg.as<uint32_t>(), d_deg);
        size_t tb = 0;
        hipcub::TransformInputIterator<uint64_t, hipcub::CastOp<uint64_t>, const uint32_t *> in(d_deg,
                                                                                                 hipcub::CastOp<uint64_t>());
        he = hipcub::DeviceScan::ExclusiveSum(nullptr, tb, in, item_off, (int)(n_cand + 1), c->stream);
        if (he == hipSuccess && (rc = d_tmp.reserve(tb)) == 0)
            he = hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, in, item_off, (int)(n_cand + 1), c->stream);
        if (he == hipSuccess && !rc && nq > 1) {
            he = hipMemcpyAsync(c->h_pinned, item_off + n_cand, 8, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            n_items = *c->h_pinned;
        }
    }
    if (he == hipSuccess && !rc && nq == 1) {
        // one item per candidate: item_off = 0, 1, 2, ... (degree table replaced by ones)
        std::vector<uint64_t> iota(n_cand + 1);
        for (uint32_t i = 0; i <= n_cand; i++) iota[i] = i;
        he = hipMemcpyAsync(item_off, iota.data(), ((size_t)n_cand + 1) * 8, hipMemcpyHostToDevice, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
    }
    if (he == hipSuccess && !rc && n_items)
        hipLaunchKernelGGL(k_refine, dim3(grid_for(n_items)), dim3(256), 0, c->stream, P, n_cand, d_cand, item_off,
                           n_items, c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(),
                           c->labels.as<uint32_t>(), (unsigned long long)limit, d_total);
    if (he == hipSuccess && !rc) he = hipGetLastError();
    if (he == hipSuccess && !rc && device_ms) he = hipEventRecord(ev1, c->stream);
    if (he == hipSuccess && !rc) he = hipMemcpyAsync(c->h_pinned, d_total, 8, hipMemcpyDeviceToHost, c->stream);
    if (he == hipSuccess && !rc) he = hipStreamSynchronize(c->stream);
    if (he == hipSuccess && !rc) *answers = std::min<uint64_t>(*c->h_pinned, limit);
    if (he == hipSuccess && !rc && device_ms) {
        float ms = 0.f;
        he = hipEventElapsedTime(&ms, ev0, ev1);
        *device_ms = ms;
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    (void)hipStreamSynchronize(c->stream);
    if (!rc && he != hipSuccess) {
        set_error("gnnpe_refine: %s", hipGetErrorString(he));
        rc = GNNPE_ERR_HIP;
    }
    return rc;
}

}  // extern "C"
