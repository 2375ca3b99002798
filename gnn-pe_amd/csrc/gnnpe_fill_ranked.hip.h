// gnnpe_fill_ranked.hip.h -- enumeration variant 4 (default): rank-sorted neighbour records.
//
// Why: the wave-per-start fill over id-sorted rows (variant 3) is bound by the traffic it moves -- every (s, b) pair
// re-reads the WHOLE neighbour list of b (ids, ranks, embeddings: 24 B per candidate, ~half of them
// discarded by the rank test), 10-20 GB per launch next to 12 GB of output (profiles/r01_pmc_fill.json).
// Here each row's neighbours are stored a second time as 8+8e-byte records {id, id-position, vde}
// SORTED BY RANK.  The neighbours c of b with rank[c] > rank[s] are then exactly the records after
// s's own position: one contiguous suffix, no rank stream, nothing read and thrown away.  The output
// order inside a pair is ascending id, so every pair also carries G = the 64-bit set of id-positions
// with greater rank: a kept record with id-position ip lands at slot popcount(G & ((1 << ip) - 1)).
// Per-pair counts are popcount(G) -- the count pass needs no scan of candidates either.
//
// Restriction: rows of degree <= 64 (one bit per id-position).  gnnpe_count_paths falls back to
// variant 3 when the held graph has a longer row; on power-law inputs, where hub rows carry most paths and
// stream well, variant 3 reaches the same fraction of the HBM peak (0.63) as this one does on G(n,m).
#pragma once

#include "gnnpe_kernels.hip.h"

namespace gnnpe {

// per (s, b) pair, indexed by the pair's emission index (poffs[rank[s]] + position of b in N(s)).
// 16 bytes so the scattered store of the row kernel is ONE dwordx4; the middle vertex itself is
// re-read from N(s) (contiguous) by the fill, the count is popcount(G).
struct __attribute__((aligned(16))) RankedPair {
    uint32_t sstart;  // first kept record: adj_start[b] + (rank-position of s in N(b)) + 1
    uint32_t pad;
    uint64_t G;       // id-positions of N(b) with rank > rank[s]
};

// one gather per adjacency entry instead of three: {vde[v], rank[v], first pair slot of v as a start vertex} side
// by side, VINFO_STRIDE(E) doubles per vertex (E = 2 -> 32 bytes: one aligned half cache line).  The kernel issues
// about as many random requests per second as the chip sustains (6-8e10/s), so requests are what to save.
#define GNNPE_VINFO_STRIDE(E) ((E) + 2)
template <int E>
__global__ void k_pack_vinfo(uint32_t n, const double *__restrict__ vde, const uint32_t *__restrict__ rank,
                             uint32_t slab_begin, uint32_t slab_end, const uint32_t *__restrict__ poffs,
                             double *__restrict__ vinfo)
{
    constexpr int S = GNNPE_VINFO_STRIDE(E);
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int k = 0; k < E; k++) vinfo[v * S + k] = vde ? vde[v * E + k] : 0.0;
        const uint32_t r = rank[v];
        const uint32_t po = (r >= slab_begin && r < slab_end) ? poffs[r - slab_begin] : kNoEdge;  // kNoEdge: not a start here
        reinterpret_cast<uint64_t *>(vinfo)[v * S + E] = ((uint64_t)po << 32) | r;
        vinfo[v * S + E + 1] = 0.0;
    }
}

template <int E> struct __attribute__((aligned(8))) RankedNbr {
    uint32_t id, idpos;
    double vde[E];
};

struct CntOfPair {
    __host__ __device__ uint64_t operator()(const RankedPair &p) const { return (uint64_t)__popcll(p.G); }
};

// One wave per held row b (degree <= 64), one lane per neighbour u_j:
//   G_j  = { i : rank[u_i] > rank[u_j] }          (d wave-uniform readlanes)
//   p_j  = d - 1 - |G_j|                            rank-position of u_j inside the row
//   recs[adj_start + p_j] = { u_j, j, vde[u_j] }    the row, sorted by rank
//   pairs[index of (s = u_j, b)] = { adj_start + p_j + 1, G_j }              when u_j starts paths here
template <int E>
__global__ __launch_bounds__(256) void k_rows_rank(uint32_t n_held, const uint32_t *__restrict__ held,
                                                   const uint32_t *__restrict__ adj_start,
                                                   const uint32_t *__restrict__ adj_deg,
                                                   const uint32_t *__restrict__ nbrs, const double *__restrict__ vinfo,
                                                   const uint32_t *__restrict__ revpos,
                                                   RankedNbr<E> *__restrict__ recs, RankedPair *__restrict__ pairs)
{
    const unsigned lane = lane_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < n_held; w += nw) {
        const uint32_t b = held ? held[w] : (uint32_t)w;
        const uint32_t st = adj_start[b], d = adj_deg[b];
        if (d == 0 || d > 64) continue;  // longer rows: the caller does not select this variant
        constexpr int S = GNNPE_VINFO_STRIDE(E);
        uint32_t u = 0, r = 0, rp = kNoEdge, po = kNoEdge;
        double vu[E];
#pragma unroll
        for (int k = 0; k < E; k++) vu[k] = 0.0;
        if (lane < d) {
            u = nbrs[st + lane];
            const double *vi = vinfo + (uint64_t)u * S;
#pragma unroll
            for (int k = 0; k < E; k++) vu[k] = vi[k];
            const uint64_t rw = reinterpret_cast<const uint64_t *>(vi)[E];
            r = (uint32_t)rw;
            po = (uint32_t)(rw >> 32);
            rp = revpos[st + lane];
        }
        uint64_t G = 0;
        for (uint32_t i = 0; i < d; i++) G |= (uint64_t)(rl32(r, (int)i) > r ? 1u : 0u) << i;
        if (lane < d) {
            const uint32_t cnt = (uint32_t)__popcll(G);
            const uint32_t p = d - 1 - cnt;
            RankedNbr<E> rec;
            rec.id = u;
            rec.idpos = lane;
#pragma unroll
            for (int k = 0; k < E; k++) rec.vde[k] = vu[k];
            recs[st + p] = rec;
            if (rp != kNoEdge && po != kNoEdge) {  // (u, b) is a pair of this slab: u starts here and its row is held
                const uint32_t pi = po + rp;
                RankedPair pr = {st + p + 1, 0u, G};
                pairs[pi] = pr;
            }
        }
    }
}

// One wave per start vertex s.  The pairs of s are taken in batches of whole pairs with at most kBatch
// kept records; the batch's records are flattened over the lanes (binary search in the wave's LDS strip
// maps record -> pair), each lane reads ONE contiguous record, computes its slot from G, parks the
// row (12 B ids + 24e B embeddings) in the wave's staging strip, and the strip is flushed with
// consecutive lanes on consecutive 16-byte / 4-byte pieces (non-temporal).
// rows staged per wave: 128 at e <= 2; wider rows take 64 (the minimum: a pair holds up to 63 records) so that the
// staging strips leave room for more than one workgroup per CU
template <int E> struct FillBatch { static constexpr int rows = E <= 2 ? 128 : 64; };

template <int E>
__global__ __launch_bounds__(256) void k_fill_ranked(FillParams P, const StartRec *__restrict__ srec,
                                                     const RankedPair *__restrict__ pairs,
                                                     const RankedNbr<E> *__restrict__ recs, uint32_t slab_len)
{
    constexpr int D = 3 * E;
    constexpr int kBatch = FillBatch<E>::rows;
    __shared__ uint32_t s_cs[4][65], s_ss[4][64], s_b[4][64];
    __shared__ uint64_t s_G[4][64];
    __shared__ __attribute__((aligned(16))) double s_vb[4][64 * E];
    __shared__ __attribute__((aligned(16))) uint32_t s_ids[4][kBatch * 3];
    __shared__ __attribute__((aligned(16))) double s_pde[4][kBatch * D];
    const unsigned lane = lane_id(), wv = wave_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const bool want_pde = P.out_pde != nullptr;
    uint32_t *const my_ids = s_ids[wv];
    double *const my_pde = s_pde[wv];

    for (; w < slab_len; w += nw) {
        const StartRec sr = srec[w];
        if (sr.end == sr.base || sr.base >= P.end || sr.end <= P.begin) continue;
        const uint32_t s = sr.s, e0 = sr.e0, ds = sr.ds;
        double vs[E];
#pragma unroll
        for (int k = 0; k < E; k++) vs[k] = want_pde ? P.vde[(uint64_t)s * E + k] : 0.0;
        uint64_t chunk_base = sr.base;  // output slot of the chunk's first row

        for (uint32_t k0 = 0; k0 < ds; k0 += 64) {
            const uint32_t k = k0 + lane;
            RankedPair pr = {0u, 0u, 0ull};
            uint32_t bk = 0;
            if (k < ds) {
                pr = pairs[e0 + k];
                bk = P.nbrs[sr.a_s + k];  // the k-th neighbour of s is the pair's middle vertex
            }
            const uint32_t pcnt = (uint32_t)__popcll(pr.G);
            uint32_t incl = pcnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= (unsigned)off) incl += t;
            }
            const uint32_t C = rl32(incl, 63);
            s_cs[wv][lane] = incl - pcnt;
            s_ss[wv][lane] = pr.sstart;
            s_b[wv][lane] = bk;
            s_G[wv][lane] = pr.G;
            if (lane == 0) s_cs[wv][64] = C;
            if (want_pde && k < ds) {
#pragma unroll
                for (int kk = 0; kk < E; kk++) s_vb[wv][lane * E + kk] = P.vde[(uint64_t)bk * E + kk];
            }
            __builtin_amdgcn_wave_barrier();

            // batches of whole pairs: [kb, ke) with cs[ke] - cs[kb] <= kBatch (a pair holds <= 63 records)
            uint32_t kb = 0;
            while (kb < 64 && s_cs[wv][kb] < C) {
                const uint32_t lo = s_cs[wv][kb];
                // largest ke in (kb, 64] with cs[ke] - lo <= kBatch: lanes test ke = lane + 1
                const bool fits = (lane + 1 > kb) && (s_cs[wv][lane + 1] - lo <= (uint32_t)kBatch);
                const uint64_t m = __ballot(fits);
                const uint32_t ke = 64u - (uint32_t)__clzll(m);  // m != 0: ke = kb + 1 always fits
                const uint32_t hi = s_cs[wv][ke];
                for (uint32_t f = lo + lane; f < hi; f += 64) {
                    uint32_t a = kb, bnd = ke;  // largest a in [kb, ke) with cs[a] <= f
                    while (bnd - a > 1) {
                        const uint32_t mid = (a + bnd) >> 1;
                        if (s_cs[wv][mid] <= f) a = mid; else bnd = mid;
                    }
                    const RankedNbr<E> rec = recs[s_ss[wv][a] + (f - s_cs[wv][a])];
                    const uint64_t below = s_G[wv][a] & ((1ull << rec.idpos) - 1ull);
                    const uint32_t row = (s_cs[wv][a] - lo) + (uint32_t)__popcll(below);
                    my_ids[row * 3 + 0] = s;
                    my_ids[row * 3 + 1] = s_b[wv][a];
                    my_ids[row * 3 + 2] = rec.id;
                    if (want_pde) {
#pragma unroll
                        for (int k2 = 0; k2 < E; k2++) {
                            my_pde[row * D + k2] = vs[k2];
                            my_pde[row * D + E + k2] = s_vb[wv][a * E + k2];
                            my_pde[row * D + 2 * E + k2] = rec.vde[k2];
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // flush rows [0, hi - lo) -> output slots chunk_base + lo ...
                {
                    const uint64_t fb = chunk_base + lo;
                    const uint64_t glo = max(fb, P.begin), ghi = min(fb + (hi - lo), P.end);
                    if (ghi > glo) {
                        const uint32_t r0 = (uint32_t)(glo - fb), nr = (uint32_t)(ghi - glo);
                        const uint64_t o = glo - P.begin;
                        if (P.out_ids)
                            for (uint32_t g = lane; g < nr * 3; g += 64)
                                __builtin_nontemporal_store(my_ids[r0 * 3 + g], &P.out_ids[o * 3 + g]);
                        if (want_pde) {
                            if ((D & 1) == 0) {
                                typedef double dbl2 __attribute__((ext_vector_type(2)));
                                const dbl2 *src = reinterpret_cast<const dbl2 *>(my_pde + (size_t)r0 * D);
                                dbl2 *dst = reinterpret_cast<dbl2 *>(P.out_pde + o * D);
                                for (uint32_t g = lane; g < nr * (D / 2); g += 64) __builtin_nontemporal_store(src[g], &dst[g]);
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
                                for (int k2 = 0; k2 < E; k2++) {
                                    P.out_pdl[(o + g) * D + k2] = P.x[(uint64_t)s * E + k2];
                                    P.out_pdl[(o + g) * D + E + k2] = P.x[(uint64_t)bb * E + k2];
                                    P.out_pdl[(o + g) * D + 2 * E + k2] = P.x[(uint64_t)cv * E + k2];
                                }
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                kb = ke;
            }
            chunk_base += C;
        }
    }
}

}  // namespace gnnpe
