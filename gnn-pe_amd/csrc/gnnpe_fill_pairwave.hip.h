// gnnpe_fill_pairwave.hip.h -- enumeration variant 1: one wave per (start, middle) pair.  The first correct
// version, kept as the simplest restatement of the closed form and as the A/B baseline.
#pragma once

#include "gnnpe_kernels.hip.h"

namespace gnnpe {

// Directed (start, middle) pairs of the slab in emission order: 16 lanes per start vertex.
__global__ void k_perm_edges(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted,
                             const uint32_t *__restrict__ adj_start, const uint32_t *__restrict__ poffs,
                             const uint32_t *__restrict__ nbrs, uint32_t *__restrict__ erow, uint32_t *__restrict__ pnbr)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; g < len; g += ng) {
        uint32_t s = sorted[slab_begin + g];
        uint32_t a = adj_start[s];
        uint32_t o = poffs[g], d = poffs[g + 1] - o;
        for (uint32_t j = sub; j < d; j += 16) {
            erow[o + j] = (uint32_t)g;
            pnbr[o + j] = nbrs[a + j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// R2 count (closed form of dfs + VectorHash, custom.h:52-92): for the directed pair e = (s, b),
// cnt[e] = |{ c in N(b) : rank[c] > rank[s] }|  (c != s is implied).  16 lanes per pair; the
// neighbour RANKS of b are a contiguous 4-byte stream.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_edges(uint64_t n_edges, uint32_t slab_begin,
                                                     const uint32_t *__restrict__ erow,
                                                     const uint32_t *__restrict__ pnbr,
                                                     const uint32_t *__restrict__ adj_start,
                                                     const uint32_t *__restrict__ adj_deg,
                                                     const uint32_t *__restrict__ nbr_rank,
                                                     uint32_t *__restrict__ ecnt)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    // all 16 lanes of a group iterate together; groups past the end still take part in shuffles
    const uint64_t g_end = (n_edges + 3) & ~(uint64_t)3;  // whole waves (4 groups per wave)
    for (; g < g_end; g += ng) {
        uint32_t cnt = 0;
        if (g < n_edges) {
            const uint32_t thr = slab_begin + erow[g];
            const uint32_t b = pnbr[g];
            const uint32_t st = adj_start[b], d = adj_deg[b];
            for (uint32_t j = sub; j < d; j += 16) cnt += nbr_rank[st + j] > thr ? 1u : 0u;
        }
        cnt += __shfl_xor(cnt, 8);
        cnt += __shfl_xor(cnt, 4);
        cnt += __shfl_xor(cnt, 2);
        cnt += __shfl_xor(cnt, 1);
        if (sub == 0 && g < n_edges) ecnt[g] = cnt;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ecnt[n_edges] = 0;
}

// nbr_vde[q][k] = vde[nbrs[q]][k]: one random gather per ADJACENCY ENTRY (2m of them) instead of one
// per emitted path (sum deg^2 of them): the fill kernels then read the endpoint's embedding from
// the same contiguous neighbour segment they scan for the rank test.
__global__ void k_gather_rows_f64(uint64_t cnt, uint32_t e, const uint32_t *__restrict__ idx,
                                  const double *__restrict__ table, double *__restrict__ out)
{
    const uint64_t tot = cnt * e;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = i / e;
        const uint32_t k = (uint32_t)(i % e);
        out[i] = table[(uint64_t)idx[q] * e + k];
    }
}

// ------------------------------------------------------------------------------------------------
// R2 + R5 fill, variant 1 (first correct version, kept for A/B): one wave per (s, b) pair, kept
// candidates compacted with ballot/popcount and stored straight to global memory.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fill_edge_wave(FillParams P)
{
    const unsigned lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t e = P.e, D = 3 * P.e;
    for (; w < P.n_edges; w += nw) {
        uint64_t base = P.eoff[w];
        const uint64_t nxt = P.eoff[w + 1];
        if (nxt == base || base >= P.end || nxt <= P.begin) continue;
        const uint32_t i = P.erow[w], b = P.pnbr[w];
        const uint32_t s = P.sorted[P.slab_begin + i], thr = P.slab_begin + i;
        const uint32_t st = P.adj_start[b], d = P.adj_deg[b];
        for (uint32_t j0 = 0; j0 < d; j0 += 64) {
            const uint32_t j = j0 + lane;
            uint32_t c = 0, r = 0;
            if (j < d) {
                c = P.nbrs[st + j];
                r = P.nbr_rank[st + j];
            }
            const bool keep = j < d && r > thr;
            const uint64_t mask = __ballot(keep);
            const uint64_t pos = base + __popcll(mask & lt);
            base += __popcll(mask);
            if (keep && pos >= P.begin && pos < P.end) {
                const uint64_t o = pos - P.begin;
                if (P.out_ids) {
                    P.out_ids[o * 3 + 0] = s;
                    P.out_ids[o * 3 + 1] = b;
                    P.out_ids[o * 3 + 2] = c;
                }
                if (P.out_pde)
                    for (uint32_t k = 0; k < e; k++) {
                        P.out_pde[o * D + k] = P.vde[(uint64_t)s * e + k];
                        P.out_pde[o * D + e + k] = P.vde[(uint64_t)b * e + k];
                        P.out_pde[o * D + 2 * e + k] = P.nbr_vde[(uint64_t)(st + j) * e + k];
                    }
                if (P.out_pdl)
                    for (uint32_t k = 0; k < e; k++) {
                        P.out_pdl[o * D + k] = P.x[(uint64_t)s * e + k];
                        P.out_pdl[o * D + e + k] = P.x[(uint64_t)b * e + k];
                        P.out_pdl[o * D + 2 * e + k] = P.x[(uint64_t)c * e + k];
                    }
                if (P.out_part) P.out_part[o] = P.member[s];
            }
        }
    }
}


}  // namespace gnnpe
