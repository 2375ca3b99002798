// gnnpe_filter_ranked.hip.h -- the online filter fused with the enumeration (ranked records, variant 4).
//
// gnnpe_filter.hip applies the leaf test of Partition::query (custom.h:404-431) to emitted path ids.  Here no path is
// emitted at all: one wave per start vertex s walks its pairs (s, b) and drops a pair as soon as no plan path starts
// with (label s, label b) and fits the two degrees -- with |labels|^2 label pairs and a handful of plan paths almost
// every pair goes, before the row of b is touched.  Only the surviving pairs read their kept records (the rank-sorted
// suffix, which already carries c and vde[c]) and run the full test.  Same candidate bitmaps, a tenth of the time.
#pragma once

#include "gnnpe_fill_ranked.hip.h"

namespace gnnpe {

constexpr int kMaxPlanPaths = 512;  // query paths held in LDS (gnnpe_filter.hip enforces the limit)

template <int E>
__global__ __launch_bounds__(256) void k_filter_ranked(const StartRec *__restrict__ srec, uint32_t slab_len,
                                                       const RankedPair *__restrict__ pairs,
                                                       const RankedNbr<E> *__restrict__ recs,
                                                       const uint32_t *__restrict__ nbrs,
                                                       const uint32_t *__restrict__ labels,
                                                       const uint32_t *__restrict__ deg, const double *__restrict__ vde,
                                                       uint32_t n_qp, const uint32_t *__restrict__ q_vids,
                                                       const uint32_t *__restrict__ q_labels,
                                                       const uint32_t *__restrict__ q_deg,
                                                       const double *__restrict__ q_pde, double eps, uint64_t words,
                                                       uint32_t *__restrict__ bitmap)
{
    __shared__ uint32_t s_lab[kMaxPlanPaths * 3], s_deg[kMaxPlanPaths * 3], s_vid[kMaxPlanPaths * 3];
    for (uint32_t i = threadIdx.x; i < n_qp * 3; i += blockDim.x) {
        s_lab[i] = q_labels[i];
        s_deg[i] = q_deg[i];
        s_vid[i] = q_vids[i];
    }
    __syncthreads();
    const unsigned lane = lane_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < slab_len; w += nw) {
        const StartRec sr = srec[w];
        if (sr.end == sr.base) continue;
        const uint32_t s = sr.s, ds = sr.ds, ls = labels[s];
        bool any = false;  // wave-uniform: does any plan path start like s?
        for (uint32_t j = 0; j < n_qp && !any; j++) any = s_lab[j * 3] == ls && s_deg[j * 3] <= ds;
        if (!any) continue;
        for (uint32_t k0 = 0; k0 < ds; k0 += 64) {
            const uint32_t k = k0 + lane;
            uint32_t b = 0, lb = 0, db = 0, sst = 0, cnt = 0;
            bool hit = false;
            if (k < ds) {
                const RankedPair pr = pairs[sr.e0 + k];
                cnt = (uint32_t)__popcll(pr.G);
                sst = pr.sstart;
                if (cnt) {
                    b = nbrs[sr.a_s + k];
                    lb = labels[b];
                    db = deg[b];
                    for (uint32_t j = 0; j < n_qp && !hit; j++)
                        hit = s_lab[j * 3] == ls && s_lab[j * 3 + 1] == lb && s_deg[j * 3] <= ds && s_deg[j * 3 + 1] <= db;
                }
            }
            uint64_t live = __ballot(hit);
            while (live) {  // surviving pairs, one at a time, their kept records over the lanes
                const int kk = __ffsll((long long)live) - 1;
                live &= live - 1;
                const uint32_t bb = rl32(b, kk), lbb = rl32(lb, kk), dbb = rl32(db, kk), s0 = rl32(sst, kk), n_c = rl32(cnt, kk);
                for (uint32_t j0 = 0; j0 < n_c; j0 += 64) {
                    if (j0 + lane >= n_c) continue;
                    const RankedNbr<E> rec = recs[s0 + j0 + lane];
                    const uint32_t c = rec.id, lc = labels[c];
                    for (uint32_t q = 0; q < n_qp; q++) {
                        if (s_lab[q * 3] != ls || s_lab[q * 3 + 1] != lbb || s_lab[q * 3 + 2] != lc) continue;  // custom.h:410
                        if (s_deg[q * 3] > ds || s_deg[q * 3 + 1] > dbb || s_deg[q * 3 + 2] > deg[c]) continue;
                        bool ok = true;
                        const double *qp = q_pde + (uint64_t)q * 3 * E;
#pragma unroll
                        for (int t = 0; t < E; t++) {                                                         // custom.h:420-426
                            const double a0 = vde[(uint64_t)s * E + t], a1 = vde[(uint64_t)bb * E + t], a2 = rec.vde[t];
                            if (qp[t] > a0 && fabs(qp[t] - a0) > eps) ok = false;
                            if (qp[E + t] > a1 && fabs(qp[E + t] - a1) > eps) ok = false;
                            if (qp[2 * E + t] > a2 && fabs(qp[2 * E + t] - a2) > eps) ok = false;
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
