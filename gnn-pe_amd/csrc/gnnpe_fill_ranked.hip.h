// gnnpe_fill_ranked.hip.h -- the enumeration (R2 count + emit, R5 embeddings; custom.h:52-92, 546-572) at l = 2.
//
// Layout idea: every adjacency row of degree <= 64 is stored a second time as a ROW BLOCK on a 128-byte boundary:
// header vde[b], then one record per neighbour {id | id-position, vde} (4+8e bytes; 8+8e for graphs beyond 2^26
// vertices) in DESCENDING RANK order (k_rows_rank).  The neighbours c of b with rank[c] > rank[s] are then exactly the
// block's first cnt records: every read of the emit kernel starts on a line boundary, no rank stream, nothing read and
// thrown away but the tail of the last line.  The output order inside a pair is ascending id, so every (s, b) pair also
// carries G = the 64-bit set of id-positions with greater rank: a kept record with id-position ip lands at slot
// popcount(G & ((1 << ip) - 1)); the pair's count is popcount(G), so the count pass scans no candidates either.  The
// emit kernel has no dependent gather: start record -> pair records -> row block (header + records) -> stores.
//
// Rows longer than 64 ("hub" rows; Test/data_graph.graph has one of degree 168) keep id order: their records are
// {id, rank, vde}, their pair counts come from a per-row sort of the ranks (hipCUB segmented sort, hubs only) and the
// emit kernel streams them with a ballot compaction -- in the same launch, pair by pair, so one graph may mix both.
#pragma once

#include "gnnpe_dpp.hip.h"
#include "gnnpe_kernels.hip.h"
#include "gnnpe_records.h"

namespace gnnpe {

// one gather per adjacency entry instead of three: {vde[v], rank[v], first pair slot of v as a start vertex} side
// by side, VINFO_STRIDE(E) doubles per vertex (E = 2 -> 32 bytes: one aligned half cache line).  The row kernel issues
// about as many random requests per second as the chip sustains (5-8e10/s), so requests are what to save.
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

struct CntOfPair {
    __host__ __device__ uint64_t operator()(const RankedPair &p) const { return (uint64_t)(p.cnt & ~kHubFlag); }
};

// 128-byte units of a held row's block: header (vde of the row's vertex) + its records
__global__ void k_row_block_units(uint32_t n_held, const uint32_t *__restrict__ held, const uint32_t *__restrict__ adj_deg,
                                  uint32_t hdr_bytes, uint32_t rec_bytes, uint32_t hub_rec_bytes,
                                  uint32_t *__restrict__ units)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k <= n_held; k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t u = 0;
        if (k < n_held) {
            const uint32_t d = adj_deg[held ? held[k] : (uint32_t)k];
            const uint64_t bytes = hdr_bytes + (uint64_t)d * (d > kHubDegree ? hub_rec_bytes : rec_bytes);
            u = d ? (uint32_t)((bytes + kRowAlign - 1) / kRowAlign) : 0u;
        }
        units[k] = u;
    }
}
__global__ void k_row_block_starts(uint32_t n_held, const uint32_t *__restrict__ held, const uint32_t *__restrict__ off,
                                   uint32_t *__restrict__ rblock)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_held; k += (uint64_t)gridDim.x * blockDim.x)
        rblock[held ? held[k] : (uint32_t)k] = off[k];
}

// One wave per held row b (degree <= 64), one lane per neighbour u_j:
//   G_j  = { i : rank[u_i] > rank[u_j] }          (d wave-uniform readlanes)
//   block[b] = [ vde[b] | records in DESCENDING rank order ]: u_j's record {u_j, j, vde[u_j]} sits at position |G_j|,
//              so the neighbours ranked after any s are the block's FIRST |G_s| records -- every read of the emit
//              kernel starts on the block's (line-aligned) first byte
//   pairs[index of (s = u_j, b)] = { block, |G_j|, G_j }              when u_j starts paths here
template <int E, bool PACKED>
__global__ __launch_bounds__(256) void k_rows_rank(uint32_t n_held, const uint32_t *__restrict__ held,
                                                   const uint32_t *__restrict__ adj_start,
                                                   const uint32_t *__restrict__ adj_deg,
                                                   const uint32_t *__restrict__ nbrs, const double *__restrict__ vinfo,
                                                   const uint32_t *__restrict__ revpos,
                                                   const uint32_t *__restrict__ rblock, char *__restrict__ recs,
                                                   RankedPair *__restrict__ pairs)
{
    typedef typename RecOf<E, PACKED>::type Rec;
    const unsigned lane = lane_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int S = GNNPE_VINFO_STRIDE(E);
    for (; w < n_held; w += nw) {
        const uint32_t b = held ? held[w] : (uint32_t)w;
        const uint32_t st = adj_start[b], d = adj_deg[b];
        if (d == 0 || d > kHubDegree) continue;  // hub rows: k_hub_records / k_hub_pairs
        const uint32_t blk = rblock[b];
        char *const base = recs + (uint64_t)blk * kRowAlign;
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
        if (lane < (unsigned)E) reinterpret_cast<double *>(base)[lane] = vinfo[(uint64_t)b * S + lane];  // header
        // G of lane i = the lanes ranked after u_i: one wave-wide compare against u_i's rank, handed to lane i (idle
        // lanes hold rank 0 and never rank after anything).  The kernel is bound by its random traffic, not by this loop
        // (scripts/count_ab.py: the gathers and the pair scatter are 0.3 of its 0.75-0.85 ms; 2 or 4 rows in flight per
        // wave change nothing).
        const uint32_t du = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
        uint32_t Glo = 0, Ghi = 0;
        for (uint32_t i = 0; i < du; i++) {
            const uint64_t m = __ballot(r > rl32(r, (int)i));
            Glo = writelane32((uint32_t)m, i, Glo);
            if (du > 32) Ghi = writelane32((uint32_t)(m >> 32), i, Ghi);
        }
        const uint64_t G = ((uint64_t)Ghi << 32) | Glo;
        if (lane < d) {
            const uint32_t cnt = (uint32_t)__popcll(G);
            Rec rec;
            if constexpr (PACKED) {
                rec.idp = u | (lane << kPackedIdBits);
            } else {
                rec.id = u;
                rec.aux = lane;
            }
#pragma unroll
            for (int k = 0; k < E; k++) rec.vde[k] = vu[k];
            reinterpret_cast<Rec *>(base + 8 * E)[cnt] = rec;
            if (rp != kNoEdge && po != kNoEdge) {  // (u, b) is a pair of this slab: u starts here and its row is held
                RankedPair pr = {blk, cnt, G};
                // the one random store per adjacency entry.  Round-3 knock-outs at config 3 (count phase, same process: 1.07 ms):
                // without this store 0.74; stored at the entry's own, coalesced position instead 0.80; 8 bytes instead of 16
                // 1.07; a 32-byte slot written in two halves 1.19 -- it is the REQUEST that costs, and moving it to the reading
                // side (the emit kernel gathering from b-major pairs) costs the dominant kernel more than it saves here
                pairs[po + rp] = pr;
            }
        }
    }
}

// The same computation with K rows per wave, one after the other, but with the K rows' loads batched: the descriptors of all K
// rows, then their neighbour / reverse-position loads, then their vinfo gathers, then K times {G, record, pair}.  A wave per
// row spends its life in three dependent round trips (row -> neighbours -> vinfo) for ~20 useful lanes; the trace of the
// per-rank steps (scripts/emulate_rank.py) fits 0.40 ms per million ROWS + 0.005 per million entries + 0.02 per million
// scattered pairs -- per rank the kernel is bound by the number of waves, not by the entries.  No load sits under a lane
// mask (idle lanes re-read the row's last entry): hipcc would wait for it at the join.
template <int E, bool PACKED, int K>
__global__ __launch_bounds__(256) void k_rows_rank_multi(uint32_t n_held, const uint32_t *__restrict__ held,
                                                         const uint32_t *__restrict__ adj_start,
                                                         const uint32_t *__restrict__ adj_deg,
                                                         const uint32_t *__restrict__ nbrs, const double *__restrict__ vinfo,
                                                         const uint32_t *__restrict__ revpos,
                                                         const uint32_t *__restrict__ rblock, char *__restrict__ recs,
                                                         RankedPair *__restrict__ pairs, uint32_t *__restrict__ clear_words,
                                                         uint32_t n_clear, uint32_t pair8)
{
    typedef typename RecOf<E, PACKED>::type Rec;
    constexpr int S = GNNPE_VINFO_STRIDE(E);
    const unsigned lane = lane_id();
    // (k_start_scan's status words and ticket, zeroed here instead of by a memset between the two kernels: a launch less per step)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_clear; i += gridDim.x * blockDim.x) clear_words[i] = 0u;
    const uint64_t w = (uint64_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    uint32_t b[K], st[K], d[K], blk[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint64_t idx = w * K + k;
        const bool ok = idx < n_held;
        b[k] = ok ? (held ? held[idx] : (uint32_t)idx) : 0u;
        st[k] = adj_start[b[k]];
        d[k] = ok ? adj_deg[b[k]] : 0u;
        if (d[k] > kHubDegree) d[k] = 0;  // hub rows: k_hub_records / k_hub_pairs
        blk[k] = rblock[b[k]];
    }
    uint32_t u[K], rp[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t q = d[k] ? st[k] + min(lane, d[k] - 1u) : 0u;
        const uint32_t uq = nbrs[q];
        u[k] = d[k] ? uq : 0u;  // (a row without entries gathers vertex 0's record: entry 0 may not exist)
        rp[k] = revpos[q];
    }
    double vu[K][E], hb[K];
    uint64_t rw[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const double *vi = vinfo + (uint64_t)u[k] * S;
#pragma unroll
        for (int j = 0; j < E; j++) vu[k][j] = vi[j];
        rw[k] = reinterpret_cast<const uint64_t *>(vi)[E];
        hb[k] = vinfo[(uint64_t)b[k] * S + min(lane, (unsigned)(E - 1))];  // header: vde of the row's vertex
    }
    // every gathered value is waited for HERE, once: hipcc otherwise waits in front of each row's first use, between the
    // stores of the rows before it -- and with one counter for loads and stores that means waiting for those stores
#pragma unroll
    for (int k = 0; k < K; k++) {
        asm volatile("" : "+v"(rw[k]), "+v"(hb[k]), "+v"(rp[k]));
#pragma unroll
        for (int j = 0; j < E; j++) asm volatile("" : "+v"(vu[k][j]));
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t du = (uint32_t)__builtin_amdgcn_readfirstlane((int)d[k]);
        if (du == 0) continue;
        char *const base = recs + (uint64_t)blk[k] * kRowAlign;
        const uint32_t r = lane < du ? (uint32_t)rw[k] : 0u, po = (uint32_t)(rw[k] >> 32);
        if (lane < (unsigned)E) reinterpret_cast<double *>(base)[lane] = hb[k];
        uint32_t Glo = 0, Ghi = 0;
        for (uint32_t i = 0; i < du; i++) {
            const uint64_t m = __ballot(r > rl32(r, (int)i));
            Glo = writelane32((uint32_t)m, i, Glo);
            if (du > 32) Ghi = writelane32((uint32_t)(m >> 32), i, Ghi);
        }
        const uint64_t G = ((uint64_t)Ghi << 32) | Glo;
        if (lane < du) {
            const uint32_t cnt = (uint32_t)__popcll(G);
            Rec rec;
            if constexpr (PACKED) {
                rec.idp = u[k] | (lane << kPackedIdBits);
            } else {
                rec.id = u[k];
                rec.aux = lane;
            }
#pragma unroll
            for (int j = 0; j < E; j++) rec.vde[j] = vu[k][j];
            reinterpret_cast<Rec *>(base + 8 * E)[cnt] = rec;
            if (rp[k] != kNoEdge && po != kNoEdge) {
                if (pair8) {  // diagnostic builds, GNNPE_ROWS_PAIR8=1: what an 8-byte pair record {block, count} would cost the scatter
                    reinterpret_cast<uint2 *>(pairs)[po + rp[k]] = make_uint2(blk[k], cnt);  // (no G: the emit kernel cannot use these)
                } else {
                    RankedPair pr = {blk[k], cnt, G};
                    pairs[po + rp[k]] = pr;
                }
            }
        }
    }
}

// DIAGNOSTIC (GNNPE_ROWS_PROBE=1|2|3, scripts/count_ab.py; never launched by the product, its outputs are not the
// enumeration's): what the pieces of k_rows_rank cost on their own, for the question whether a split into a rank-only pass
// over an L2-sized table and a payload pass would beat the fused kernel.
//   MODE 1  the G pass alone: one 4-byte gather per entry from the 4 MB rank table, G, the pair scatter; no payload, no records
//   MODE 2  MODE 1 without the pair scatter (the pairs go to the entries' own, coalesced positions)
//   MODE 3  the payload pass alone: one 16-byte vde gather per entry, the record written at the entry's id position
template <int E, bool PACKED, int MODE>
__global__ __launch_bounds__(256) void k_rows_rank_probe(uint32_t n_held, const uint32_t *__restrict__ adj_start,
                                                         const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ nbrs,
                                                         const uint32_t *__restrict__ rank, const double *__restrict__ vde,
                                                         const uint32_t *__restrict__ poffs, const uint32_t *__restrict__ revpos,
                                                         const uint32_t *__restrict__ rblock, char *__restrict__ recs,
                                                         RankedPair *__restrict__ pairs)
{
    // four rows per wave with their loads batched, like k_rows_rank_multi: the pieces are compared at the product's shape
    typedef typename RecOf<E, PACKED>::type Rec;
    constexpr int K = 4;
    const unsigned lane = lane_id();
    const uint64_t w = (uint64_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    uint32_t st[K], d[K], blk[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint64_t idx = w * K + k;
        const uint32_t b = idx < n_held ? (uint32_t)idx : 0u;
        st[k] = adj_start[b];
        d[k] = idx < n_held ? adj_deg[b] : 0u;
        if (d[k] > kHubDegree) d[k] = 0;
        blk[k] = rblock[b];
    }
    uint32_t u[K], rp[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t q = d[k] ? st[k] + min(lane, d[k] - 1u) : 0u;
        u[k] = nbrs[q];
        rp[k] = MODE == 3 ? 0u : revpos[q];
    }
    if constexpr (MODE == 3) {
        double v[K][E];
#pragma unroll
        for (int k = 0; k < K; k++)
#pragma unroll
            for (int j = 0; j < E; j++) v[k][j] = vde[(uint64_t)u[k] * E + j];
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (lane < d[k]) {
                Rec rec;
                if constexpr (PACKED) rec.idp = u[k] | (lane << kPackedIdBits); else { rec.id = u[k]; rec.aux = lane; }
#pragma unroll
                for (int j = 0; j < E; j++) rec.vde[j] = v[k][j];
                reinterpret_cast<Rec *>(recs + (uint64_t)blk[k] * kRowAlign + 8 * E)[lane] = rec;
            }
        }
        return;
    }
    uint32_t rk[K], po[K];
#pragma unroll
    for (int k = 0; k < K; k++) rk[k] = rank[u[k]];
#pragma unroll
    for (int k = 0; k < K; k++) po[k] = MODE == 1 ? poffs[rk[k]] : 0u;  // (whole graph = one slab: a start's first pair slot is poffs[rank])
#pragma unroll
    for (int k = 0; k < K; k++) {
        const uint32_t du = (uint32_t)__builtin_amdgcn_readfirstlane((int)d[k]);
        if (du == 0) continue;
        const uint32_t r = lane < du ? rk[k] : 0u;
        uint32_t Glo = 0, Ghi = 0;
        for (uint32_t i = 0; i < du; i++) {
            const uint64_t m = __ballot(r > rl32(r, (int)i));
            Glo = writelane32((uint32_t)m, i, Glo);
            if (du > 32) Ghi = writelane32((uint32_t)(m >> 32), i, Ghi);
        }
        const uint64_t G = ((uint64_t)Ghi << 32) | Glo;
        if (lane < du && rp[k] != kNoEdge) {
            RankedPair pr = {blk[k], (uint32_t)__popcll(G), G};
            if constexpr (MODE == 1) pairs[po[k] + rp[k]] = pr;
            else pairs[st[k] + lane] = pr;
        }
    }
}

// Hub rows, pass 1 (one wave per hub row): header + records {id, rank, vde} in id order, and the row's neighbour ranks
// as a stream for the per-row sort.
template <int E>
__global__ __launch_bounds__(256) void k_hub_records(uint32_t n_hub, const uint32_t *__restrict__ hub_rows,
                                                     const uint32_t *__restrict__ adj_start,
                                                     const uint32_t *__restrict__ adj_deg,
                                                     const uint32_t *__restrict__ nbrs, const double *__restrict__ vinfo,
                                                     const uint32_t *__restrict__ rblock, char *__restrict__ recs,
                                                     uint32_t *__restrict__ nbr_rank)
{
    const unsigned lane = lane_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int S = GNNPE_VINFO_STRIDE(E);
    for (; w < n_hub; w += nw) {
        const uint32_t b = hub_rows[w];
        const uint32_t st = adj_start[b], d = adj_deg[b];
        char *const base = recs + (uint64_t)rblock[b] * kRowAlign;
        if (lane < (unsigned)E) reinterpret_cast<double *>(base)[lane] = vinfo[(uint64_t)b * S + lane];
        for (uint32_t j = lane; j < d; j += 64) {
            const uint32_t u = nbrs[st + j];
            const double *vi = vinfo + (uint64_t)u * S;
            RecWide<E> rec;
            rec.id = u;
            rec.aux = (uint32_t) reinterpret_cast<const uint64_t *>(vi)[E];
#pragma unroll
            for (int k = 0; k < E; k++) rec.vde[k] = vi[k];
            reinterpret_cast<RecWide<E> *>(base + 8 * E)[j] = rec;
            nbr_rank[st + j] = rec.aux;
        }
    }
}

// Hub rows, pass 2 (after the per-row sort of the ranks): pair (s = u_j, b) keeps the neighbours ranked after s --
// deg(b) minus the number of neighbours ranked up to and including s.
template <int E>
__global__ __launch_bounds__(256) void k_hub_pairs(uint32_t n_hub, const uint32_t *__restrict__ hub_rows,
                                                   const uint32_t *__restrict__ adj_start,
                                                   const uint32_t *__restrict__ adj_deg,
                                                   const uint32_t *__restrict__ nbrs, const double *__restrict__ vinfo,
                                                   const uint32_t *__restrict__ revpos,
                                                   const uint32_t *__restrict__ rank_sorted,
                                                   const uint32_t *__restrict__ rblock, RankedPair *__restrict__ pairs)
{
    const unsigned lane = lane_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int S = GNNPE_VINFO_STRIDE(E);
    for (; w < n_hub; w += nw) {
        const uint32_t b = hub_rows[w];
        const uint32_t st = adj_start[b], d = adj_deg[b], blk = rblock[b];
        for (uint32_t j = lane; j < d; j += 64) {
            const uint32_t rp = revpos[st + j];
            if (rp == kNoEdge) continue;
            const uint32_t u = nbrs[st + j];
            const uint64_t rw = reinterpret_cast<const uint64_t *>(vinfo + (uint64_t)u * S)[E];
            const uint32_t r = (uint32_t)rw, po = (uint32_t)(rw >> 32);
            if (po == kNoEdge) continue;
            uint32_t lo = 0, hi = d;  // first position with a rank > r
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (rank_sorted[st + mid] <= r) lo = mid + 1; else hi = mid;
            }
            RankedPair pr = {blk, (d - lo) | kHubFlag, (uint64_t)d};
            pairs[po + rp] = pr;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The emit kernel: one wave per start vertex s, all of whose paths are one contiguous run of the output.
//   strip   : the pairs of s, 64 at a time, one lane each -- {row block, G, b} and the scan of their counts sit in a
//             per-wave LDS strip.
//   batch   : whole pairs with at most kBatch kept records.  The batch's records are flattened over the lanes (binary
//             search in the strip maps record -> pair); a lane reads ONE record {c, id-position, vde[c]} from the
//             front of the pair's row block and parks {c, pair slot, vde[c]} at its output row (from G) in the
//             wave's staging rows (8+8e B per row; s, b, vde[s], vde[b] are not copied per row: the flush picks them
//             out of the strip).  The lane on a pair's first record also reads the block's header, vde[b] -- same line.
//   hub pair: streamed instead -- 64 id-ordered records per step, kept iff rank > rank[s], ballot compaction.
//   flush   : rows -> global memory, consecutive lanes on consecutive 16-byte pieces of the pde rows and on
//             consecutive 12-byte id rows, non-temporal.
// Measured (config 3, profiles/r02*): the launch moves what the fabric can move for this read/write mix (20.4 GB at
// 5.0 TB/s before the aligned blocks); waves per SIMD beyond 5 change nothing, spilling to reach 8 costs 6-25 %.
// ------------------------------------------------------------------------------------------------------------------
// rows staged per wave between flushes.  Round 2, static start vertices, same-process A/B at config 3 (scripts/fill_ab.py): 64 rows
// 3.440 ms, 128 rows 3.466, 192 rows 3.462; round 4 (five workgroups per CU): 128 rows 3.18 against 2.98.  Round 5, start vertices in
// order from ticket counters (profiles/r05_emit_ab.txt section 6, six allocations, same buffers): 128 rows are as fast or faster in EVERY
// allocation -- fast class 2.73 ms against 2.77 (0.843 of the spec), in between the same, slow class 3.29 against 3.44 / 3.38 against
// 3.49 -- 192 / 256 rows gain further in one kind of slow allocation (3.24 / 3.22) and lose in the others (256: 2.82 in the fast
// class).  So 128 at the widths whose rows are small (e <= 2: 21 bytes of LDS per row); the wide embeddings keep 64.
constexpr int kFillRows = 64;
__host__ __device__ constexpr int fill_rows(int e) { return e <= 2 ? 128 : kFillRows; }

struct __attribute__((packed, aligned(4))) IdRow {
    uint32_t s, b, c;
};

// ticket counters of k_fill_ranked's start vertices: up to 64 heads, each on a 128-byte line of its own
constexpr uint32_t kStartHeadWords = 32;
constexpr uint32_t kStartHeadsBytes = 64 * kStartHeadWords * 4;

template <int E, bool PACKED, int kBatch>
__global__ __launch_bounds__(256, (E <= 2 ? 5 : 1)) void k_fill_ranked(FillParams P, const StartRec *__restrict__ srec,
                                                     const RankedPair *__restrict__ pairs,
                                                     const char *__restrict__ recs, uint32_t slab_len,
                                                     uint32_t *__restrict__ heads, uint32_t nh)
{
    typedef typename RecOf<E, PACKED>::type Rec;
    constexpr int D = 3 * E;
    constexpr int EP = E + (E & 1);  // even number of doubles per LDS embedding slot (16-byte slots)
    static_assert(kBatch >= 64 && kBatch % 64 == 0, "a pair holds up to 63 records; the hub path stages 64 per step");
    __shared__ uint32_t s_cs[4][66], s_blk[4][64], s_b[4][64];
    __shared__ uint64_t s_G[4][64];
    __shared__ __attribute__((aligned(16))) double s_vb[4][64 * EP];
    __shared__ __attribute__((aligned(16))) double s_vs[4][EP];
    __shared__ uint32_t s_id[4][kBatch];
    __shared__ uint8_t s_a[4][kBatch];
    __shared__ __attribute__((aligned(16))) double s_v[4][kBatch * EP];
    const unsigned lane = lane_id(), wv = wave_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t nw = (uint32_t)((gridDim.x * (uint64_t)blockDim.x) >> 6);
    const bool want_pde = P.out_pde != nullptr;
    uint32_t *const cs = s_cs[wv], *const sblk = s_blk[wv], *const sb = s_b[wv], *const sid = s_id[wv];
    uint64_t *const sG = s_G[wv];
    uint8_t *const sa = s_a[wv];
    double *const svb = s_vb[wv], *const svs = s_vs[wv], *const sv = s_v[wv];

    // rows [0, nr) of the staging area -> output slots [slot0, slot0 + nr) clipped to [P.begin, P.end)
    auto flush = [&](uint32_t s, uint64_t slot0, uint32_t nr_all) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint64_t glo = max(slot0, P.begin), ghi = min(slot0 + nr_all, P.end);
        if (ghi > glo) {
            const uint32_t r0 = (uint32_t)(glo - slot0), nr = (uint32_t)(ghi - glo);
            const uint64_t o = glo - P.begin;
            if (P.out_ids) {
                IdRow *dst = reinterpret_cast<IdRow *>(P.out_ids) + o;
#pragma unroll 1
                for (uint32_t g = lane; g < nr; g += 64) {
                    // non-temporal like the pde rows: same-process A/B at config 3 (scripts/fill_exp.py, same records and
                    // buffers) 3.40 -> 3.30 ms; sc1 / sc0 sc1 instead 3.45; any policy but nt for the pde rows 3.6-3.9
                    uint32_t *q = reinterpret_cast<uint32_t *>(&dst[g]);
                    __builtin_nontemporal_store(s, q);
                    __builtin_nontemporal_store(sb[sa[r0 + g]], q + 1);
                    __builtin_nontemporal_store(sid[r0 + g], q + 2);
                }
            }
            if (want_pde) {
                if constexpr ((E & 1) == 0) {
                    typedef double dbl2 __attribute__((ext_vector_type(2)));
                    constexpr uint32_t H = E / 2, PR = 3 * H;  // 16-byte pieces per vertex / per row
                    dbl2 *dst = reinterpret_cast<dbl2 *>(P.out_pde + o * D);
#pragma unroll 1
                    for (uint32_t g = lane; g < nr * PR; g += 64) {
                        const uint32_t row = r0 + g / PR, within = g % PR, which = within / H, sub = within % H;
                        const double *src = which == 0 ? svs + 2 * sub
                                            : which == 1 ? svb + (uint32_t)sa[row] * EP + 2 * sub
                                                         : sv + row * EP + 2 * sub;
                        __builtin_nontemporal_store(*reinterpret_cast<const dbl2 *>(src), &dst[g]);
                    }
                } else {
                    double *dst = P.out_pde + o * D;
#pragma unroll 1
                    for (uint32_t g = lane; g < nr * D; g += 64) {
                        const uint32_t row = r0 + g / D, within = g % D, which = within / E, sub = within % E;
                        const double *src = which == 0 ? svs + sub : which == 1 ? svb + (uint32_t)sa[row] * EP + sub : sv + row * EP + sub;
                        __builtin_nontemporal_store(*src, &dst[g]);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // the strip's pairs, one per lane.  Every lane loads (lanes past the start's last pair re-read it and are zeroed after):
    // a load under a lane mask makes hipcc wait for every outstanding access at the join
    auto load_pairs = [&](const StartRec &sr, uint32_t k0, RankedPair &pr, uint32_t &bk) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const uint32_t k = k0 + lane, kc = min(k, sr.ds - 1u);  // (sr.ds >= 1 for a start vertex with paths)
        // read once, never again: non-temporal (same-process A/B at config 3: 3.085 -> 3.036 ms; the RECORD loads must
        // stay cached -- a pair's header and first records share lines across load instructions: 4.02 -> 4.37 ms)
        const u32x4 pw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(&pairs[sr.e0 + kc]));  // one 16-byte load
        const uint32_t bq = __builtin_nontemporal_load(&P.nbrs[sr.a_s + kc]);  // the k-th neighbour of s is the pair's middle vertex
        const bool in = k < sr.ds;
        pr.block = in ? pw.x : 0u;
        pr.cnt = in ? pw.y : 0u;
        pr.G = in ? (((uint64_t)pw.w << 32) | pw.z) : 0ull;
        bk = in ? bq : 0u;
    };

    // a wave's start vertices: w, w + nw, ... (wave-uniform, so the start record arrives through the scalar cache); the NEXT
    // start's record is requested one start ahead
    // heads != nullptr: start vertices IN ORDER from ticket counters instead (ticket k of head h = start k * nh + h; one returning
    // increment per start vertex, requested one start ahead): waves that get ahead cannot run away from the others, so the rows
    // the chip writes at one moment stay one window of the output whatever the waves' speeds
    uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const uint32_t head = heads ? w % nh : 0u;
    auto draw = [&]() -> uint32_t {
        uint32_t tk = 0;
        if (lane == 0) tk = __builtin_amdgcn_atomic_inc32(heads + head * kStartHeadWords, 0xFFFFFFFFu, __ATOMIC_RELAXED, "agent");
        return tk;
    };
    uint32_t tk = 0;
    if (heads) {
        tk = draw();
        w = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk) * nh + head;
        tk = draw();
    }
    if (w >= slab_len) return;
    StartRec sr = srec[w];
    while (w < slab_len) {
        uint32_t w_next = w + nw;
        if (heads) {
            w_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk) * nh + head;
            tk = draw();
        }
        const StartRec sr_next = srec[min(w_next, slab_len - 1u)];
        RankedPair pr;
        uint32_t bk;
        if (sr.end != sr.base && sr.base < P.end && sr.end > P.begin) {
            const uint32_t s = sr.s, ds = sr.ds, thr = P.slab_begin + w;
            // vde[s] and the first strip's pairs are in flight together (the LDS copy of vde[s] used to wait for its load
            // BEFORE the pair loads were issued: one more round trip per start vertex)
            double vs_val = 0.0;
            if (want_pde) vs_val = P.vde[(uint64_t)s * E + min(lane, (unsigned)(E - 1))];
            load_pairs(sr, 0, pr, bk);
            if (want_pde && lane < (unsigned)E) svs[lane] = vs_val;
            uint64_t chunk_base = sr.base;  // output slot of the strip's first row
            for (uint32_t k0 = 0; k0 < ds; k0 += 64) {
                if (k0) load_pairs(sr, k0, pr, bk);
                const uint32_t pcnt = pr.cnt & ~kHubFlag;
                const uint32_t incl = wave_scan_add(pcnt);  // DPP: no LDS round trips (six ds_bpermute before)
                const uint32_t C = rl32(incl, 63);
                const uint64_t hubs = __ballot((pr.cnt & kHubFlag) != 0);
                cs[lane] = incl - pcnt;
                sblk[lane] = pr.block;
                sb[lane] = bk;
                sG[lane] = pr.G;
                if (lane == 0) cs[64] = C;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                // batches: whole pairs with at most kBatch kept records, or up to kBatch kept records of one hub pair
                uint32_t kb = 0, hub_j = 0, hub_done = 0;
                while (kb < 64 && cs[kb] < C) {
                    const uint32_t lo = cs[kb];
                    uint32_t nr, first_row;
                    if ((hubs >> kb) & 1ull) {
                        // hub pair: stream the id-ordered row from entry hub_j on, keep what ranks after s
                        const char *const base = recs + (uint64_t)sblk[kb] * kRowAlign;
                        const uint32_t d = (uint32_t)sG[kb];
                        if (want_pde && hub_j == 0 && lane < (unsigned)E) svb[kb * EP + lane] = reinterpret_cast<const double *>(base)[lane];
                        uint32_t fill = 0;
#pragma unroll 1
                        while (hub_j < d && fill + 64 <= (uint32_t)kBatch) {
                            const uint32_t j = hub_j + lane;
                            RecWide<E> rec;
                            rec.id = 0;
                            rec.aux = 0;
                            if (j < d) rec = reinterpret_cast<const RecWide<E> *>(base + 8 * E)[j];
                            const bool keep = j < d && rec.aux > thr;
                            const uint64_t mask = __ballot(keep);
                            if (keep) {
                                const uint32_t row = fill + (uint32_t)__popcll(mask & lt);
                                sid[row] = rec.id;
                                sa[row] = (uint8_t)kb;
                                if (want_pde) {
#pragma unroll
                                    for (int k2 = 0; k2 < E; k2++) sv[row * EP + k2] = rec.vde[k2];
                                }
                            }
                            fill += (uint32_t)__popcll(mask);
                            hub_j += 64;
                        }
                        nr = fill;
                        first_row = lo + hub_done;
                        hub_done += fill;
                        if (hub_j >= d) {
                            kb++;
                            hub_j = 0;
                            hub_done = 0;
                        }
                    } else {
                        // whole pairs [kb, ke): cs[ke] - cs[kb] <= kBatch, no hub pair inside
                        const uint32_t first_hub = (hubs >> kb) ? kb + (uint32_t)__builtin_ctzll(hubs >> kb) : 64u;
                        const bool fits = (lane + 1 > kb) && (lane + 1 <= first_hub) && (cs[lane + 1] - lo <= (uint32_t)kBatch);
                        const uint64_t m = __ballot(fits);
                        const uint32_t ke = 64u - (uint32_t)__clzll(m);  // m != 0: ke = kb + 1 always fits (<= 63 records)
                        const uint32_t hi = cs[ke];
                        if (hi > lo) {  // (wave-uniform; empty when the pairs in front of a hub pair hold no path)
                            // a batch holds at most kBatch records: kBatch / 64 per lane.  The records and their pairs' headers
                            // (vde[b], same line) are loaded as raw dwords by EVERY lane (idle lanes re-read the batch's last
                            // record: no load under a lane mask) and pinned by an empty asm statement, so that all loads are
                            // issued back to back -- hipcc otherwise sinks the payload load below the first use of the id
                            // word and the header load into its branch: three dependent round trips per batch instead of one
                            constexpr int NB = kBatch / 64;
                            constexpr int RW = (int)(sizeof(Rec) / 4), VW = RW - 2 * E;
                            uint32_t rw[NB][RW], hw[NB][2 * E], pa[NB], pca[NB];
#pragma unroll
                            for (int p = 0; p < NB; p++) {
                                const uint32_t fc = min(lo + lane + 64u * p, hi - 1u);
                                uint32_t a = kb, bnd = ke;  // largest a in [kb, ke) with cs[a] <= fc
                                while (bnd - a > 1) {
                                    const uint32_t mid = (a + bnd) >> 1;
                                    if (cs[mid] <= fc) a = mid; else bnd = mid;
                                }
                                const uint32_t ca = cs[a];
                                pa[p] = a;
                                pca[p] = ca;
                                const char *const base = recs + (uint64_t)sblk[a] * kRowAlign;
                                const uint32_t *rq = reinterpret_cast<const uint32_t *>(base + 8 * E) + (uint64_t)(fc - ca) * RW;
#pragma unroll
                                for (int z = 0; z < RW; z++) rw[p][z] = rq[z];
                                if (want_pde) {
#pragma unroll
                                    for (int z = 0; z < 2 * E; z++) hw[p][z] = reinterpret_cast<const uint32_t *>(base)[z];
                                }
                            }
#pragma unroll
                            for (int p = 0; p < NB; p++) {
#pragma unroll
                                for (int z = 0; z < RW; z++) asm volatile("" : "+v"(rw[p][z]));
                                if (want_pde) {
#pragma unroll
                                    for (int z = 0; z < 2 * E; z++) asm volatile("" : "+v"(hw[p][z]));
                                }
                            }
#pragma unroll
                            for (int p = 0; p < NB; p++) {
                                const uint32_t f = lo + lane + 64u * p;
                                if (f < hi) {
                                    const uint32_t a = pa[p], ca = pca[p];
                                    uint32_t id, ip;
                                    if constexpr (PACKED) {
                                        id = rw[p][0] & ((1u << kPackedIdBits) - 1u);
                                        ip = rw[p][0] >> kPackedIdBits;
                                    } else {
                                        id = rw[p][0];
                                        ip = rw[p][1];
                                    }
                                    const uint64_t below = sG[a] & ((1ull << ip) - 1ull);
                                    const uint32_t row = (ca - lo) + (uint32_t)__popcll(below);
                                    sid[row] = id;
                                    sa[row] = (uint8_t)a;
                                    if (want_pde) {
                                        uint32_t *q = reinterpret_cast<uint32_t *>(sv + row * EP);
#pragma unroll
                                        for (int z = 0; z < 2 * E; z++) q[z] = rw[p][VW + z];
                                        if (f == ca) {  // first record of the pair: its lane parks the header
                                            uint32_t *qh = reinterpret_cast<uint32_t *>(svb + a * EP);
#pragma unroll
                                            for (int z = 0; z < 2 * E; z++) qh[z] = hw[p][z];
                                        }
                                    }
                                }
                            }
                        }
                        nr = hi - lo;
                        first_row = lo;
                        kb = ke;
                    }
                    flush(s, chunk_base + first_row, nr);
                }
                chunk_base += C;
            }
        }
        sr = sr_next;
        w = w_next;
    }
}

// Secondary outputs, from what the emit kernel wrote.  Partition of every path's start vertex (what main.cpp:98-108
// groups partition_paths.txt by): one wave per start vertex fills its run.
__global__ void k_start_parts(uint32_t slab_len, const StartRec *__restrict__ srec, uint64_t begin, uint64_t end,
                              uint32_t *__restrict__ out_part)
{
    const unsigned lane = lane_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < slab_len; w += nw) {
        const uint64_t lo = max(srec[w].base, begin), hi = min(srec[w].end, end);
        const uint32_t part = srec[w].part;
        for (uint64_t o = lo + lane; o < hi; o += 64) out_part[o - begin] = part;
    }
}
// pde_label (gen_pde, custom.h:561-567): the label features x of the path's vertices, gathered from the emitted ids
__global__ void k_pdl_from_ids(uint64_t n_rows, uint32_t L, uint32_t e, const uint32_t *__restrict__ ids,
                               const double *__restrict__ x, double *__restrict__ out)
{
    const uint64_t D = (uint64_t)L * e, tot = n_rows * D;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / D;
        const uint32_t k = (uint32_t)(i % D);
        out[i] = x[(uint64_t)ids[r * L + k / e] * e + k % e];
    }
}

}  // namespace gnnpe
