// gnnpe_fill_start.hip.h -- enumeration variant 3: one wave per start vertex over the id-sorted rows (any degree).
// Fallback of the ranked variant for graphs with rows longer than 64.
#pragma once

#include "gnnpe_kernels.hip.h"

namespace gnnpe {

// cnt(s = u, b) for the adjacency entry q = (b -> u): one THREAD per entry, looping over the row's
// rank stream (lanes of the same row read the same addresses, so the loads are broadcasts).  Work per
// row is deg^2 / 64 wave-iterations whatever the degree, so hubs spread over many waves.
__global__ __launch_bounds__(256) void k_count_flat(uint64_t n_entries, uint32_t slab_begin, uint32_t slab_end,
                                                    const uint32_t *__restrict__ nbr_row,
                                                    const uint32_t *__restrict__ adj_start,
                                                    const uint32_t *__restrict__ adj_deg,
                                                    const uint32_t *__restrict__ nbr_rank,
                                                    const uint32_t *__restrict__ revpos,
                                                    const uint32_t *__restrict__ poffs, uint32_t *__restrict__ rev,
                                                    uint32_t *__restrict__ ecnt)
{
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < n_entries; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t ri = nbr_rank[q];
        const uint32_t rv = pair_index(revpos[q], ri, slab_begin, slab_end, poffs);
        if (rev) rev[q] = rv;
        if (rv == kNoEdge) continue;
        const uint32_t b = nbr_row[q];
        const uint32_t st = adj_start[b], d = adj_deg[b];
        uint32_t cnt = 0;
        for (uint32_t j0 = 0; j0 < d; j0 += 8) {  // 8 independent loads in flight per lane
            uint32_t r[8];
#pragma unroll
            for (int t = 0; t < 8; t++) r[t] = (j0 + t < d) ? nbr_rank[st + j0 + t] : 0u;
#pragma unroll
            for (int t = 0; t < 8; t++) cnt += r[t] > ri ? 1u : 0u;
        }
        ecnt[rv] = cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// R2 + R5 fill, variant 3: one wave per START vertex s over the id-sorted rows.
//
// All paths of s occupy ONE contiguous run of the output, so a wave that emits them in order writes
// its region front to back (sequential, every byte once).  Per s: its StartRec, the PairRecs of its
// middle vertices (one lane each), a wave scan of their degrees into a per-wave LDS strip; the
// candidates (neighbour lists of the middle vertices) are flattened over the 64 lanes -- a 6-step
// binary search in the strip maps candidate -> pair -- and processed in batches of R rounds:
// phase A issues every id / rank / embedding load of the batch (contiguous segments of nbrs,
// nbr_rank, nbr_vde), phase B compacts (ballot + popcount against rank[s]), parks kept rows in a
// 64-row LDS staging strip and flushes it with consecutive lanes on consecutive 16-byte (pde) /
// 4-byte (ids) pieces, non-temporal.  Works for any degree; reads every candidate (kept or not).
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) PairRec {
    uint32_t b, st, dg, pad;
};
__global__ void k_pair_recs(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted,
                            const uint32_t *__restrict__ adj_start, const uint32_t *__restrict__ adj_deg,
                            const uint32_t *__restrict__ poffs, const uint32_t *__restrict__ nbrs,
                            PairRec *__restrict__ recs)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; g < len; g += ng) {
        const uint32_t s = sorted[slab_begin + g];
        const uint32_t a = adj_start[s];
        const uint32_t o = poffs[g], d = poffs[g + 1] - o;
        for (uint32_t j = sub; j < d; j += 16) {
            const uint32_t b = nbrs[a + j];
            PairRec r = {b, adj_start[b], adj_deg[b], 0u};
            recs[o + j] = r;
        }
    }
}

template <int E, int R>
__global__ __launch_bounds__(256) void k_fill_s_rec(FillParams P, const StartRec *__restrict__ srec,
                                                    const PairRec *__restrict__ prec, uint32_t slab_len)
{
    constexpr int D = 3 * E;
    __shared__ uint32_t s_cs[4][65], s_st[4][64], s_b[4][64];
    __shared__ __attribute__((aligned(16))) double s_vb[4][64 * E];
    __shared__ __attribute__((aligned(16))) uint32_t s_ids[4][64 * 3];
    __shared__ __attribute__((aligned(16))) double s_pde[4][64 * D];
    const unsigned lane = lane_id(), wv = wave_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const bool want_pde = P.out_pde != nullptr;
    uint32_t *const my_ids = s_ids[wv];
    double *const my_pde = s_pde[wv];

    for (; w < slab_len; w += nw) {
        const StartRec sr = srec[w];
        if (sr.end == sr.base || sr.base >= P.end || sr.end <= P.begin) continue;
        const uint32_t thr = P.slab_begin + (uint32_t)w;
        const uint32_t s = sr.s, e0 = sr.e0, ds = sr.ds;
        double vs[E];
#pragma unroll
        for (int k = 0; k < E; k++) vs[k] = want_pde ? P.vde[(uint64_t)s * E + k] : 0.0;
        uint64_t fbase = sr.base;
        uint32_t fill = 0;

        auto flush = [&]() {
            __builtin_amdgcn_wave_barrier();
            const uint64_t lo = max(fbase, P.begin), hi = min(fbase + fill, P.end);
            if (hi > lo) {
                const uint32_t r0 = (uint32_t)(lo - fbase), nr = (uint32_t)(hi - lo);
                const uint64_t o = lo - P.begin;
                if (P.out_ids) {
                    for (uint32_t g = lane; g < nr * 3; g += 64) {
                        __builtin_nontemporal_store(my_ids[r0 * 3 + g], &P.out_ids[o * 3 + g]);
                    }
                }
                if (want_pde) {
                    if ((D & 1) == 0) {
                        typedef double dbl2 __attribute__((ext_vector_type(2)));
                        const dbl2 *src = reinterpret_cast<const dbl2 *>(my_pde + (size_t)r0 * D);
                        dbl2 *dst = reinterpret_cast<dbl2 *>(P.out_pde + o * D);
                        for (uint32_t g = lane; g < nr * (D / 2); g += 64) {
                            __builtin_nontemporal_store(src[g], &dst[g]);
                        }
                    } else {
                        for (uint32_t g = lane; g < nr * D; g += 64) P.out_pde[o * D + g] = my_pde[(size_t)r0 * D + g];
                    }
                }
                if (P.out_part)
                    for (uint32_t g = lane; g < nr; g += 64) P.out_part[o + g] = sr.part;
                if (P.out_pdl) {
                    for (uint32_t g = lane; g < nr; g += 64) {
                        const uint32_t bb = my_ids[(r0 + g) * 3 + 1], cv = my_ids[(r0 + g) * 3 + 2];
#pragma unroll
                        for (int k = 0; k < E; k++) {
                            P.out_pdl[(o + g) * D + k] = P.x[(uint64_t)s * E + k];
                            P.out_pdl[(o + g) * D + E + k] = P.x[(uint64_t)bb * E + k];
                            P.out_pdl[(o + g) * D + 2 * E + k] = P.x[(uint64_t)cv * E + k];
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            fbase += fill;
            fill = 0;
        };

        for (uint32_t k0 = 0; k0 < ds; k0 += 64) {
            const uint32_t k = k0 + lane;
            PairRec pr = {0u, 0u, 0u, 0u};
            if (k < ds) pr = prec[e0 + k];
            uint32_t incl = pr.dg;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= (unsigned)off) incl += t;
            }
            const uint32_t C = rl32(incl, 63);
            s_cs[wv][lane] = incl - pr.dg;
            s_st[wv][lane] = pr.st;
            s_b[wv][lane] = pr.b;
            if (lane == 0) s_cs[wv][64] = C;
            if (want_pde && k < ds) {
#pragma unroll
                for (int kk = 0; kk < E; kk++) s_vb[wv][lane * E + kk] = P.vde[(uint64_t)pr.b * E + kk];
            }
            __builtin_amdgcn_wave_barrier();

            for (uint32_t q0 = 0; q0 < C; q0 += 64 * R) {
                uint32_t cc[R], rc[R], kk[R];
                double vc[R][E];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t q = q0 + r * 64 + lane;
                    cc[r] = 0;
                    rc[r] = 0;
                    kk[r] = 0;
#pragma unroll
                    for (int k2 = 0; k2 < E; k2++) vc[r][k2] = 0.0;
                    if (q < C) {
                        uint32_t a = 0, bnd = 64;
#pragma unroll
                        for (int it = 0; it < 6; it++) {
                            const uint32_t mid = (a + bnd) >> 1;
                            if (s_cs[wv][mid] <= q) a = mid; else bnd = mid;
                        }
                        const uint32_t idx = s_st[wv][a] + (q - s_cs[wv][a]);
                        kk[r] = a;
                        cc[r] = P.nbrs[idx];
                        rc[r] = P.nbr_rank[idx];
                        if (want_pde) {
#pragma unroll
                            for (int k2 = 0; k2 < E; k2++) vc[r][k2] = P.nbr_vde[(uint64_t)idx * E + k2];
                        }
                    }
                }
                bool kp[R];
                uint32_t slot[R], cnt[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t q = q0 + r * 64 + lane;
                    kp[r] = q < C && rc[r] > thr;
                    const uint64_t mask = __ballot(kp[r]);
                    slot[r] = (uint32_t)__popcll(mask & lt);
                    cnt[r] = (uint32_t)__popcll(mask);
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (cnt[r] == 0) continue;
                    if (fill + cnt[r] > 64) flush();
                    if (kp[r]) {
                        const uint32_t row = fill + slot[r];
                        my_ids[row * 3 + 0] = s;
                        my_ids[row * 3 + 1] = s_b[wv][kk[r]];
                        my_ids[row * 3 + 2] = cc[r];
                        if (want_pde) {
#pragma unroll
                            for (int k2 = 0; k2 < E; k2++) {
                                my_pde[row * D + k2] = vs[k2];
                                my_pde[row * D + E + k2] = s_vb[wv][kk[r] * E + k2];
                                my_pde[row * D + 2 * E + k2] = vc[r][k2];
                            }
                        }
                    }
                    fill += cnt[r];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (fill) flush();
    }
}


}  // namespace gnnpe
