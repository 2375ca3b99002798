// gnnpe_fill_middle.hip.h -- enumeration variant 2: middle-vertex-centric count + fill.
#pragma once

#include "gnnpe_kernels.hip.h"

namespace gnnpe {

// ------------------------------------------------------------------------------------------------
// MIDDLE-VERTEX-CENTRIC enumeration (variant 2).
//
// Every path (s, b, c) is a pair of neighbours of its middle vertex b, oriented from the lower to
// the higher rank.  So one pass over the rows is enough: a wave takes row b, keeps N(b) -- ids,
// ranks, embeddings, and for every neighbour u_i the output offset of the pair (s = u_i, b) -- in
// registers, and for each i emits the neighbours c with rank[c] > rank[u_i] in ascending-id order
// at that offset.  The adjacency is read ONCE (2m entries) instead of once per (s, b) pair
// (sum deg^2 entries), and there is no dependent load inside the emit loop.  The price is that a
// pair's rows (cnt x 60 B) are written as one short contiguous run per wave iteration.
// ------------------------------------------------------------------------------------------------
// ---- packed wave layout for a row of degree d <= 64 -----------------------------------------------
// A wave holds PER = 64 / d complete copies of the row: lane L works on the ordered pair
// (i = ib + L / d, j = L % d).  The j side (candidate c = u_j) never changes while the row is
// processed, the i side (start s = u_i) advances by PER per iteration, so a row of degree 20 takes
// 7 iterations with 60 of 64 lanes busy instead of 20 iterations with 20 lanes.
struct RowLanes {
    uint32_t j, iq, per;
    bool lane_ok;  // lane belongs to a complete copy
};
__device__ __forceinline__ RowLanes row_lanes(uint32_t d, unsigned lane)
{
    RowLanes r;
    r.per = 64u / d;
    r.iq = lane / d;
    r.j = lane - r.iq * d;
    r.lane_ok = r.iq < r.per;
    return r;
}
// kept-lane bits of this lane's copy, and the number of kept lanes before it inside the copy
__device__ __forceinline__ uint64_t copy_bits(uint64_t mask, uint32_t iq, uint32_t d)
{
    const uint64_t seg = mask >> (iq * d);  // iq * d <= 63 whenever the lane is valid
    return d >= 64 ? seg : (seg & ((1ull << d) - 1ull));
}

// cnt(s = u_i, b) = |{ j : rank[u_j] > rank[u_i] }| for every neighbour u_i of b that starts a path
// here, stored at the pair's emission index; rev[q] keeps that index for the fill.
__global__ __launch_bounds__(256) void k_count_b(uint32_t n_held, const uint32_t *__restrict__ held,
                                                 uint32_t slab_begin, uint32_t slab_end,
                                                 const uint32_t *__restrict__ adj_start,
                                                 const uint32_t *__restrict__ adj_deg,
                                                 const uint32_t *__restrict__ nbr_rank,
                                                 const uint32_t *__restrict__ revpos,
                                                 const uint32_t *__restrict__ poffs, uint32_t *__restrict__ rev,
                                                 uint32_t *__restrict__ ecnt)
{
    __shared__ uint32_t s_rank[4][64], s_rev[4][64];
    const unsigned lane = lane_id(), wv = wave_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < n_held; w += nw) {
        const uint32_t b = held ? held[w] : (uint32_t)w;
        const uint32_t st = adj_start[b], d = adj_deg[b];
        if (d == 0) continue;
        if (d <= 64) {
            uint32_t rt = 0xFFFFFFFFu, rv = kNoEdge;
            if (lane < d) {
                rt = nbr_rank[st + lane];
                rv = pair_index(revpos[st + lane], rt, slab_begin, slab_end, poffs);
                rev[st + lane] = rv;
            }
            if (__ballot(rv != kNoEdge) == 0) continue;
            s_rank[wv][lane] = rt;
            s_rev[wv][lane] = rv;
            __builtin_amdgcn_wave_barrier();
            const RowLanes R = row_lanes(d, lane);
            const uint32_t rc = s_rank[wv][R.j];
            for (uint32_t ib = 0; ib < d; ib += R.per) {
                const uint32_t i = ib + R.iq;
                const bool act = R.lane_ok && i < d;
                const uint32_t rs = act ? s_rank[wv][i] : 0xFFFFFFFFu;
                const uint32_t ri = act ? s_rev[wv][i] : kNoEdge;
                const uint64_t mask = __ballot(act && ri != kNoEdge && rc > rs);
                if (act && R.j == 0 && ri != kNoEdge) ecnt[ri] = (uint32_t)__popcll(copy_bits(mask, R.iq, d));
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            for (uint32_t i0 = 0; i0 < d; i0 += 64) {
                const uint32_t i = i0 + lane;
                const uint32_t ri = i < d ? nbr_rank[st + i] : 0xFFFFFFFFu;
                const uint32_t rv = i < d ? pair_index(revpos[st + i], ri, slab_begin, slab_end, poffs) : kNoEdge;
                if (i < d) rev[st + i] = rv;
                if (__ballot(rv != kNoEdge) == 0) continue;
                uint32_t cnt = 0;
                for (uint32_t j = 0; j < d; j++) cnt += nbr_rank[st + j] > ri ? 1u : 0u;
                if (rv != kNoEdge) ecnt[rv] = cnt;
            }
        }
    }
}

struct FillBParams {
    const uint32_t *held, *adj_start, *adj_deg, *nbrs, *nbr_rank, *rev, *member;
    const uint64_t *eoff;
    const double *vde, *x;
    uint32_t n_held, e;
    uint64_t begin, end;
    uint32_t *out_ids;
    double *out_pde, *out_pdl;
    uint32_t *out_part;
};

// One wave per middle vertex b.  Rows of degree <= 64 use the packed layout above: the row's ids,
// ranks, first-slot offsets and embeddings are fetched once (one lane per neighbour) into a
// per-wave LDS strip; every iteration compares PER starts against all candidates, compacts the kept
// ones per copy (ballot + popcount), and each kept lane stores its own 12-byte id triple and
// 24e-byte embedding row -- consecutive kept lanes hit consecutive rows of the pair's output run.
template <int E>
__global__ __launch_bounds__(256) void k_fill_b(FillBParams P)
{
    __shared__ uint32_t s_u[4][64], s_r[4][64];
    __shared__ uint64_t s_off[4][64];
    __shared__ __attribute__((aligned(16))) double s_v[4][64 * E];

    const unsigned lane = lane_id(), wv = wave_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const bool want_pde = P.out_pde != nullptr;

    for (; w < P.n_held; w += nw) {
        const uint32_t b = P.held ? P.held[w] : (uint32_t)w;
        const uint32_t st = P.adj_start[b], d = P.adj_deg[b];
        if (d < 2) continue;
        double vb[E];
#pragma unroll
        for (int k = 0; k < E; k++) vb[k] = want_pde ? P.vde[(uint64_t)b * E + k] : 0.0;

        if (d <= 64) {
            // ---- one lane per neighbour: fetch, park in the wave's strip ----
            uint32_t ut = 0, rt = 0;
            uint64_t ot = kNoOff;
            if (lane < d) {
                ut = P.nbrs[st + lane];
                rt = P.nbr_rank[st + lane];
                const uint32_t rv = P.rev[st + lane];
                if (rv != kNoEdge) ot = P.eoff[rv];
            }
            if (__ballot(ot != kNoOff && ot < P.end) == 0) continue;
            s_u[wv][lane] = ut;
            s_r[wv][lane] = rt;
            s_off[wv][lane] = ot;
            if (want_pde && lane < d) {
#pragma unroll
                for (int k = 0; k < E; k++) s_v[wv][lane * E + k] = P.vde[(uint64_t)ut * E + k];
            }
            __builtin_amdgcn_wave_barrier();
            const RowLanes R = row_lanes(d, lane);
            const uint32_t c = s_u[wv][R.j], rc = s_r[wv][R.j];
            double vc[E];
#pragma unroll
            for (int k = 0; k < E; k++) vc[k] = want_pde ? s_v[wv][R.j * E + k] : 0.0;
            const uint64_t jbits = (1ull << R.j) - 1ull;
            for (uint32_t ib = 0; ib < d; ib += R.per) {
                const uint32_t i = ib + R.iq;
                const bool act = R.lane_ok && i < d;
                const uint64_t off = act ? s_off[wv][i] : kNoOff;
                const uint32_t rs = act ? s_r[wv][i] : 0xFFFFFFFFu;
                const bool keep = act && off != kNoOff && rc > rs;
                const uint64_t mask = __ballot(keep);
                if (mask == 0) continue;
                if (keep) {
                    const uint64_t pos = off + (uint64_t)__popcll(copy_bits(mask, R.iq, d) & jbits);
                    if (pos >= P.begin && pos < P.end) {
                        double vs[E];
#pragma unroll
                        for (int k = 0; k < E; k++) vs[k] = want_pde ? s_v[wv][i * E + k] : 0.0;
                        emit_path<E, FillBParams>(P, pos, s_u[wv][i], b, c, vs, vb, vc);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // the strip is reused by the wave's next row
        } else {
            // ---- long rows: one start per iteration, candidates in 64-lane chunks ----
            for (uint32_t i = 0; i < d; i++) {
                const uint32_t rv = P.rev[st + i];
                if (rv == kNoEdge) continue;
                uint64_t run = P.eoff[rv];
                if (run >= P.end) continue;
                const uint32_t s = P.nbrs[st + i], rs = P.nbr_rank[st + i];
                double vs[E];
#pragma unroll
                for (int k = 0; k < E; k++) vs[k] = want_pde ? P.vde[(uint64_t)s * E + k] : 0.0;
                for (uint32_t j0 = 0; j0 < d; j0 += 64) {
                    const uint32_t jj = j0 + lane;
                    const bool jv = jj < d;
                    const uint32_t c = jv ? P.nbrs[st + jj] : 0u;
                    const uint32_t rc = jv ? P.nbr_rank[st + jj] : 0u;
                    const bool keep = jv && rc > rs;
                    const uint64_t mask = __ballot(keep);
                    const uint64_t pos = run + (uint64_t)__popcll(mask & lt);
                    run += (uint64_t)__popcll(mask);
                    if (keep && pos >= P.begin && pos < P.end) {
                        double vc[E];
#pragma unroll
                        for (int k = 0; k < E; k++) vc[k] = want_pde ? P.vde[(uint64_t)c * E + k] : 0.0;
                        emit_path<E, FillBParams>(P, pos, s, b, c, vs, vb, vc);
                    }
                }
            }
        }
    }
}


}  // namespace gnnpe
