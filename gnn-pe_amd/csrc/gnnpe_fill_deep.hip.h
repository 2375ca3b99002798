// gnnpe_fill_deep.hip.h -- 4-vertex paths (l = 3), BASELINE config 5's "deep path" case.
//
// The reference cannot run l != 2 (dfs is always started at depth path_length-2, main.cpp:95, and
// the writer then reads a 4th column out of bounds), so this is the intended generalisation of
// custom.h:66-92 with the depth fixed: simple paths (s, b, c, d) from every start in processing order,
// neighbours ascending at every level, kept iff neither the path nor its reverse was kept before --
// i.e. iff rank[d] > rank[s] (the reverse starts at d).  The CPU checker under tests/ restates both the
// hash-set DFS and this closed form for any L; the GPU tests compare against them.
//
// Work unit = one (s, b) pair x one batch of 64 third vertices c (a pair with deg(b) <= 64 is one unit; a hub middle
// vertex gives ceil(deg/64) units), so that the millions of candidates behind a hub are spread over many waves instead
// of one: `upair[u]` names the unit's pair, `ufirst[pair]` its first unit, and the count pass leaves one 64-bit count
// per unit whose scan (`uoff`) is the unit's first output slot.
// One wave per unit, walking like variant 1 walks a pair.  The pair's candidates are the entries of N(c) for
// every c in N(b) \ {s}, in (c ascending, position ascending) order = emission order.  c's are taken 64
// at a time (one lane each: row start and degree), their degrees are scanned into a per-wave LDS prefix,
// and the flattened candidate space is walked 64 candidates per step with every lane busy: a 6-step binary
// search in the prefix maps candidate -> (c, j); keep = rank[d] > rank[s] and d != b (d != s is implied,
// d != c and c != b hold in a simple graph); ballot/popcount gives the output slot.  The kept rows of a step
// are one contiguous piece of the output: (c, d, entry) are compacted into LDS and the wave then writes the
// ids (16 bytes per lane) and the 4e doubles per row with consecutive lanes on consecutive 16-byte pieces (e = 8:
// 256-byte rows; one lane per row reached 0.64 TB/s, this 4.9 TB/s).  The count pass (k_deep3_count) does not walk
// candidates at all; its per-unit totals are 64-bit (hub pairs of a power-law graph pass 2^32).
#pragma once

#include "gnnpe_dpp.hip.h"
#include "gnnpe_kernels.hip.h"

namespace gnnpe {

constexpr int kDeepWaves = 4;  // waves per workgroup

// units of a pair: one per 64 neighbours of the middle vertex, at least one
__global__ void k_deep_unit_counts(uint64_t n_pairs, const uint32_t *__restrict__ pnbr, const uint32_t *__restrict__ adj_deg,
                                   uint64_t *__restrict__ out)
{
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w <= n_pairs; w += (uint64_t)gridDim.x * blockDim.x)
        out[w] = w < n_pairs ? (uint64_t)max(1u, (adj_deg[pnbr[w]] + 63u) / 64u) : 0ull;
}
__global__ void k_deep_unit_pairs(uint64_t n_pairs, const uint64_t *__restrict__ ufirst, uint32_t *__restrict__ upair)
{
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w < n_pairs; w += (uint64_t)gridDim.x * blockDim.x)
        for (uint64_t u = ufirst[w]; u < ufirst[w + 1]; u++) upair[u] = (uint32_t)w;
}
// eoff[pair] = first output slot of the pair = offset of its first unit (eoff[n_pairs] = total)
__global__ void k_deep_pair_offsets(uint64_t n_pairs, const uint64_t *__restrict__ ufirst, const uint64_t *__restrict__ uoff,
                                    uint64_t *__restrict__ eoff)
{
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w <= n_pairs; w += (uint64_t)gridDim.x * blockDim.x)
        eoff[w] = uoff[ufirst[w]];
}
// units overlapping the output range [begin, end): out[0] = last unit starting at or before begin, out[1] = first unit
// starting at or after end
__global__ void k_deep_unit_range(uint64_t n_units, const uint64_t *__restrict__ uoff, uint64_t begin, uint64_t end,
                                  uint64_t *__restrict__ out)
{
    if (threadIdx.x > 1 || blockIdx.x) return;
    const uint64_t key = threadIdx.x ? end : begin;
    uint64_t lo = 0, hi = n_units;  // uoff has n_units + 1 entries, non-decreasing
    if (threadIdx.x == 0) {
        while (hi - lo > 1) {  // largest u with uoff[u] <= begin
            const uint64_t mid = (lo + hi) >> 1;
            if (uoff[mid] <= key) lo = mid; else hi = mid;
        }
        out[0] = lo;
    } else {
        while (lo < hi) {  // smallest u with uoff[u] >= end
            const uint64_t mid = (lo + hi) >> 1;
            if (uoff[mid] < key) lo = mid + 1; else hi = mid;
        }
        out[1] = lo;
    }
}

__global__ void k_row_ends(uint32_t n, const uint32_t *__restrict__ adj_start, const uint32_t *__restrict__ adj_deg,
                           uint32_t *__restrict__ adj_end)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x)
        adj_end[v] = adj_start[v] + adj_deg[v];
}

// ---- the count, row-major (round 3) -----------------------------------------------------------------------------------
// With every row's neighbour RANKS sorted (one segmented sort per order), the kept fourth vertices of (s, b, c) number
// |{d in N(c): rank d > rank s}| - [rank b > rank s].  Rounds 1-2 answered every (s, b, c) with its own binary search in row
// c (one wave per unit, one lane per third vertex): 1.4e11 searches of ~12 dependent loads at config 5 -- 11.7 s, all of
// the count (profiles/r03_deep_kernel_stats.csv, k_deep3_count).  For a fixed (b, c) the answer is monotone in
// rank[s], and the start vertices that have b as a neighbour are N(b) itself: so one wave takes a ROW-BATCH (b, 64 third
// vertices c = one lane each) and answers ALL of b's slab neighbours s in ascending rank order with one merge per lane --
// a pointer into row c's sorted ranks that only moves forward (after one binary search for the first threshold).  Per
// threshold: the lanes advance, subtract, a DPP wave sum, and the unit (s, b, batch) gets its count.  Thresholds are taken
// 64 at a time (their rank, start vertex and unit slot loaded by the lanes, broadcast by readlane), so the loop body has no
// dependent global load but the pointer advance, which walks consecutive words of a line.
// rb_first[v] = first row-batch of row v (prefix of ceil(deg / 64) over the rows); rb_first[n] = their number.
__global__ void k_deep_row_batches(uint32_t n, const uint32_t *__restrict__ adj_deg, uint32_t *__restrict__ out)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v <= n; v += (uint64_t)gridDim.x * blockDim.x)
        out[v] = v < n ? (adj_deg[v] + 63u) / 64u : 0u;
}
__global__ void k_iota_u32(uint64_t n, uint32_t *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_deep3_count_rows(FillParams P, uint32_t n, uint32_t slab_len, const uint8_t *__restrict__ present,
                                                          const uint32_t *__restrict__ rank, const uint32_t *__restrict__ sorted_ranks,
                                                          const uint32_t *__restrict__ rank_arg, const uint32_t *__restrict__ revpos,
                                                          const uint32_t *__restrict__ poffs, const uint32_t *__restrict__ rb_first,
                                                          const uint64_t *__restrict__ ufirst, uint64_t *__restrict__ uoff,
                                                          uint64_t n_units, uint32_t *__restrict__ missing_row)
{
    const unsigned lane = lane_id();
    const uint32_t n_rb = rb_first[n];
    uint64_t rb = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t sb = P.slab_begin, se = P.slab_begin + slab_len;
    for (; rb < n_rb; rb += nw) {
        uint32_t lo = 0, hi = n;  // the row of this batch: largest b with rb_first[b] <= rb (rows without entries share a value)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rb_first[mid] <= rb) lo = mid; else hi = mid;
        }
        const uint32_t b = lo, j = (uint32_t)rb - rb_first[b];
        const uint32_t bst = P.adj_start[b], bd = P.adj_deg[b];
        const uint32_t *sr_b = sorted_ranks + bst;
        // b's neighbours that start paths here = its sorted ranks inside [sb, se)
        uint32_t t_lo = 0, t_hi = bd;
        {
            uint32_t a = 0, z = bd;
            while (a < z) {
                const uint32_t mid = (a + z) >> 1;
                if (sr_b[mid] < sb) a = mid + 1; else z = mid;
            }
            t_lo = a;
            z = bd;
            while (a < z) {
                const uint32_t mid = (a + z) >> 1;
                if (sr_b[mid] < se) a = mid + 1; else z = mid;
            }
            t_hi = a;
        }
        if (t_lo == t_hi) continue;  // (a row two hops out: a third vertex only)
        const uint32_t rank_b = rank[b];
        const uint32_t k = j * 64u + lane;
        uint32_t c = 0xFFFFFFFFu, cd = 0, cst = 0;
        if (k < bd) {
            c = P.nbrs[bst + k];
            if (present && !present[c]) {
                atomicMin(missing_row, c);  // 2-hop row not on this device
            } else {
                cd = P.adj_deg[c];
                cst = P.adj_start[c];
            }
        }
        const uint32_t *sr_c = sorted_ranks + cst;
        // first position of row c ranked after the first threshold: one binary search per (b, c); from there on the pointer
        // only moves forward.  nxt = the rank under the pointer (all ones past the row's end: no rank reaches it)
        uint32_t ptr = 0;
        {
            const uint32_t r0 = sr_b[t_lo];
            uint32_t a = 0, z = cd;
            while (a < z) {
                const uint32_t mid = (a + z) >> 1;
                if (sr_c[mid] <= r0) a = mid + 1; else z = mid;
            }
            ptr = a;
        }
        uint32_t nxt = ptr < cd ? sr_c[ptr] : 0xFFFFFFFFu;
        for (uint32_t t0 = t_lo; t0 < t_hi; t0 += 64) {
            const uint32_t tt = t0 + lane;
            uint32_t r_l = 0, s_l = 0xFFFFFFFFu;
            uint64_t u_l = 0;
            if (tt < t_hi) {
                r_l = sr_b[tt];
                const uint32_t q = rank_arg[bst + tt];  // entry of row b that holds this neighbour
                s_l = P.nbrs[q];
                const uint32_t rp = revpos[q];  // position of b inside N(s): s starts paths here, so its row is an owned one
                u_l = rp != 0xFFFFFFFFu ? ufirst[poffs[r_l - sb] + rp] + j : ~0ull;  // the pair (s, b), this batch
            }
            const uint32_t nt = min(64u, t_hi - t0);
            uint32_t my_sum = 0;
            for (uint32_t i = 0; i < nt; i++) {
                const uint32_t r_t = rl32(r_l, (int)i), s = rl32(s_l, (int)i);
                if (nxt <= r_t) {
                    // advance to the first entry ranked after r_t: gallop (1, 2, 4, ... entries), then bisect the last stride --
                    // a hub third vertex behind a low-degree middle vertex jumps hundreds of entries per threshold (stepping
                    // one entry at a time: count phase 1.08 s at config 5; galloping: 0.66 s)
                    uint32_t lo_p = ptr + 1, step = 1;  // every entry before lo_p is ranked <= r_t
                    while (lo_p + step <= cd && sr_c[lo_p + step - 1] <= r_t) {
                        lo_p += step;
                        step <<= 1;
                    }
                    uint32_t hi_p = min(cd, lo_p + step - 1);
                    while (lo_p < hi_p) {
                        const uint32_t mid = (lo_p + hi_p) >> 1;
                        if (sr_c[mid] <= r_t) lo_p = mid + 1; else hi_p = mid;
                    }
                    ptr = lo_p;
                    nxt = ptr < cd ? sr_c[ptr] : 0xFFFFFFFFu;
                }
                const uint32_t b_kept = rank_b > r_t ? 1u : 0u;  // b is a neighbour of every c and must not close the path
                const uint32_t cnt = (cd && c != s) ? cd - ptr - b_kept : 0u;
                const uint32_t sum = wave_sum_u32(cnt);
                if (lane == i) my_sum = sum;
            }
            if (tt < t_hi && u_l < n_units) uoff[u_l] = my_sum;  // counts; scanned in place by the caller
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) uoff[n_units] = 0;
}

// ---- the count, entry-major (round 5) -----------------------------------------------------------------------------------
// k_deep3_count_rows above walks, per lane, a pointer through row c's sorted ranks: one DEPENDENT global load per threshold and
// lane, 2.3e9 wave iterations of a round trip each at config 5 -- the 0.63 s are latency.  The same numbers without a dependent
// load: for a row-batch (b, 64 third vertices) and b's start vertices s_0 < s_1 < ... (ascending rank, thresholds T),
//   count(s_t, b, batch) = sum over the batch's c != s_t of ( deg c - [rank b > T_t] - |{d in N(c): rank d <= T_t}| )
// and the last term summed over the batch is P(t) = |{(c, d): rank d <= T_t}|: a HISTOGRAM of the batch's rows over the
// thresholds.  So the wave keeps T and the histogram in LDS and STREAMS the batch's rows (64 consecutive sorted ranks per step,
// coalesced, every load independent of every other), finds each entry's bin by a binary search in LDS, and -- the entries of a
// step being ascending, equal bins are runs -- adds each run's length with one LDS atomic per run (no two lanes of an
// instruction on one address).  A prefix over the bins gives P(t); the start vertex' own term, when s_t is one of the batch's
// third vertices, is deg s - [rank b > T_t] - lowcnt[s] with lowcnt[v] = |{d in N(v): rank d < rank v}| (k_deep_lowcnt, once
// per count).  Thresholds beyond kHistChunk (hub rows b) take further passes over the batch's rows.
// Two launches.  Rows b of up to kHistChunk neighbours: one WAVE per row-batch, T and the bins in the wave's own 4.25 KB of
// LDS.  Longer rows (k_deep_hub_batches lists their batches): one WORKGROUP of eight waves per row-batch with up to kCoopChunk
// thresholds in 37 KB shared by the waves, which take the batch's rows in pieces of 256 entries in turn -- chunking the
// thresholds instead means streaming the batch's rows once per chunk, and at config 5 the long rows b are where the long rows
// c are: 3.96e9 steps with chunks of 512 (2.71e9 with 1 024) against 1.72e9 without.
constexpr int kHistChunk = 511, kCoopWaves = 8, kCoopChunk = 4095;  // (tables of 512 / 4 096 words: a power of two ABOVE the thresholds)
__global__ void k_deep_lowcnt(uint32_t n, const uint8_t *__restrict__ present, const uint32_t *__restrict__ adj_start,
                              const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ rank,
                              const uint32_t *__restrict__ sorted_ranks, uint32_t *__restrict__ lowcnt)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        if (present && !present[v]) {  // (a row that is not on this device starts no path here)
            lowcnt[v] = 0u;
            continue;
        }
        const uint32_t *row = sorted_ranks + adj_start[v];
        const uint32_t r = rank[v];
        uint32_t a = 0, z = adj_deg[v];
        while (a < z) {
            const uint32_t mid = (a + z) >> 1;
            if (row[mid] < r) a = mid + 1; else z = mid;
        }
        lowcnt[v] = a;
    }
}
// the row-batches {row, batch} of the rows longer than kHistChunk, in no particular order; counter[0] = their number
__global__ void k_deep_hub_batches(uint32_t n, const uint32_t *__restrict__ adj_deg, uint32_t *__restrict__ counter, uint2 *__restrict__ list)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = adj_deg[v];
        if (d <= (uint32_t)kHistChunk) continue;
        const uint32_t nb = (d + 63u) / 64u, at = atomicAdd(counter, nb);
        if (list)
            for (uint32_t j = 0; j < nb; j++) list[at + j] = make_uint2((uint32_t)v, j);
    }
}

// ubase[position of s in row b's sorted ranks] = first unit of the pair (s, b) (all ones: s starts no path here).  A row-batch
// writes one count per start vertex of its row, and every batch of the row needs the same five dependent gathers to find where
// (entry of b -> s -> position of b in N(s) -> first pair of s -> first unit of the pair): 1.72e9 units x 5 at config 5.  Once
// per count instead: 16 lanes per row.
__global__ void k_deep_unit_bases(uint32_t n, uint32_t sb, uint32_t se, const uint8_t *__restrict__ present, const uint32_t *__restrict__ adj_start,
                                  const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ sorted_ranks,
                                  const uint32_t *__restrict__ rank_arg, const uint32_t *__restrict__ revpos, const uint32_t *__restrict__ poffs,
                                  const uint64_t *__restrict__ ufirst, uint64_t *__restrict__ ubase)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t v = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t nv = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; v < n; v += nv) {
        if (present && !present[v]) continue;
        const uint32_t st = adj_start[v], d = adj_deg[v];
        for (uint32_t t = sub; t < d; t += 16) {
            const uint32_t r = sorted_ranks[st + t];
            uint64_t u = ~0ull;
            if (r >= sb && r < se) {
                const uint32_t rp = revpos[rank_arg[st + t]];  // position of v inside N(s): s starts paths here, so its row is an owned one
                if (rp != 0xFFFFFFFFu) u = ufirst[poffs[r - sb] + rp];
            }
            ubase[st + t] = u;
        }
    }
}

// NS steps side by side: 64 consecutive sorted ranks of a row each -> bins.  bin(x) = thresholds below x (x == T_t is the start
// vertex itself: rank d <= T_t, bin t); bin ntc = ranked after every threshold of the pass: counted nowhere.  The search is a
// chain of dependent LDS reads; a wave that walked one chain at a time spent its life waiting for the LDS (the whole count
// 774 ms, slower than the pointer walk's 648; four chains 595).  And the kernel is bound by VALU issue (a wave64 instruction is
// four cycles; ~100 of them per step made 420 cycles per step and SIMD): so the table is padded with all-ones to a power of
// two above ntc (1 << L words), which leaves an iteration of the search three instructions -- read at a constant offset,
// compare, add -- and the runs of equal bins (the entries of a step ascend) are credited by their boundaries alone: the
// lane where the bin changes adds its position + 1 to its own bin and takes it off the next lane's (bin ntc collects what is
// left and is never read).
template <int NS, int L>
__device__ __forceinline__ void deep_hist_steps(const uint32_t *T, uint32_t *H, unsigned lane, const uint32_t *x)
{
    uint32_t g4[NS];  // bin as a byte offset
#pragma unroll
    for (int u = 0; u < NS; u++) g4[u] = 0;
#pragma unroll
    for (int k = L - 1; k >= 0; k--) {
        uint32_t tv[NS];
#pragma unroll
        for (int u = 0; u < NS; u++) tv[u] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(T) + g4[u] + ((1u << k) - 1u) * 4u);
#pragma unroll
        for (int u = 0; u < NS; u++) g4[u] += tv[u] < x[u] ? (4u << k) : 0u;
    }
#pragma unroll
    for (int u = 0; u < NS; u++) {
        // lane + 1's bin (wave_shl:1; lane 63 reads the all-ones `old`: a boundary)
        const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)g4[u], 0x130, 0xF, 0xF, false);
        if (nx != g4[u]) {
            atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(H) + g4[u]), lane + 1u);
            if (lane != 63u) atomicSub(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(H) + nx), lane + 1u);
        }
    }
}
// entries [off, off + 256) of a row: up to four steps
template <int L>
__device__ __forceinline__ void deep_hist_piece(const uint32_t *T, uint32_t *H, unsigned lane, const uint32_t *__restrict__ row, uint32_t off,
                                                uint32_t rd)
{
    if (off + 256u <= rd) {
        const uint32_t x[4] = {row[off + lane], row[off + 64u + lane], row[off + 128u + lane], row[off + 192u + lane]};
        deep_hist_steps<4, L>(T, H, lane, x);
    } else if (off + 64u < rd) {  // two to four steps: the same four chains, idle entries ranked after everything
        uint32_t x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = off + 64u * u + lane < rd ? row[off + 64u * u + lane] : 0xFFFFFFFFu;
        deep_hist_steps<4, L>(T, H, lane, x);
    } else if (off < rd) {
        const uint32_t x[1] = {off + lane < rd ? row[off + lane] : 0xFFFFFFFFu};
        deep_hist_steps<1, L>(T, H, lane, x);
    }
}
// the batch's rows (the share of wave `wv` of `n_waves`: pieces of 256 entries dealt in turn) into the bins
template <int L, uint32_t n_waves>
__device__ __forceinline__ void deep_hist_rows(const uint32_t *T, uint32_t *H, unsigned lane, const uint32_t *__restrict__ sorted_ranks,
                                               uint64_t have, uint32_t cd, uint32_t cst, uint32_t wv)
{
    static_assert((n_waves & (n_waves - 1u)) == 0u, "the deal is a mask, not a division");
    uint32_t piece0 = 0;  // pieces of the rows before this one
    for (uint64_t m = have; m; m &= m - 1) {
        const int ci = __builtin_ctzll(m);
        const uint32_t rd = rl32(cd, ci);
        const uint32_t *row = sorted_ranks + rl32(cst, ci);
        const uint32_t np = (rd + 255u) / 256u;
        for (uint32_t pc = (wv + n_waves - piece0 % n_waves) % n_waves; pc < np; pc += n_waves) deep_hist_piece<L>(T, H, lane, row, pc * 256u, rd);
        piece0 += np;
    }
}
// what a batch's waves share: the row, its start vertices' range of sorted ranks, the third vertices in the lanes
struct DeepBatch {
    uint32_t b, j, bst, bd, t_lo, t_hi, rank_b, cd, cst, deg_sum, n_third;
    uint64_t have;
};
__device__ __forceinline__ DeepBatch deep_batch(const FillParams &P, uint32_t b, uint32_t j, uint32_t sb, uint32_t se,
                                                const uint8_t *__restrict__ present, const uint32_t *__restrict__ rank,
                                                const uint32_t *__restrict__ sorted_ranks, uint32_t *__restrict__ missing_row, unsigned lane)
{
    DeepBatch B;
    B.b = b;
    B.j = j;
    B.bst = P.adj_start[b];
    B.bd = P.adj_deg[b];
    const uint32_t *sr_b = sorted_ranks + B.bst;
    {  // b's neighbours that start paths here = its sorted ranks inside [sb, se)
        uint32_t a = 0, z = B.bd;
        while (a < z) {
            const uint32_t mid = (a + z) >> 1;
            if (sr_b[mid] < sb) a = mid + 1; else z = mid;
        }
        B.t_lo = a;
        z = B.bd;
        while (a < z) {
            const uint32_t mid = (a + z) >> 1;
            if (sr_b[mid] < se) a = mid + 1; else z = mid;
        }
        B.t_hi = a;
    }
    B.rank_b = rank[b];
    B.cd = B.cst = 0;
    const uint32_t k = j * 64u + lane;
    if (B.t_lo != B.t_hi && k < B.bd) {
        const uint32_t c = P.nbrs[B.bst + k];
        if (present && !present[c]) {
            atomicMin(missing_row, c);  // 2-hop row not on this device
        } else {
            B.cd = P.adj_deg[c];
            B.cst = P.adj_start[c];
        }
    }
    B.have = __ballot(B.cd != 0u);
    B.deg_sum = wave_sum_u32(B.cd);
    B.n_third = (uint32_t)__popcll(B.have);
    return B;
}
// threshold a0 + tt of the batch with incl = P(t): its unit's count
__device__ __forceinline__ void deep_hist_unit(const FillParams &P, const DeepBatch &B, uint32_t a0, uint32_t tt, uint32_t r_t, uint32_t incl,
                                               const uint32_t *__restrict__ rank_arg, const uint64_t *__restrict__ ubase,
                                               const uint32_t *__restrict__ lowcnt, uint64_t *__restrict__ uoff, uint64_t n_units)
{
    const uint64_t u0 = ubase[B.bst + a0 + tt];      // the pair (s, b)'s first unit
    const uint32_t ks = rank_arg[B.bst + a0 + tt] - B.bst;  // s among b's neighbours: one of this batch's third vertices?
    const uint32_t b_kept = B.rank_b > r_t ? 1u : 0u;  // b is a neighbour of every c and must not close the path
    uint32_t cnt = B.deg_sum - B.n_third * b_kept - incl;
    if (ks / 64u == B.j && ((B.have >> (ks & 63u)) & 1ull)) {  // c == s
        const uint32_t s = P.nbrs[B.bst + ks];
        cnt -= P.adj_deg[s] - b_kept - lowcnt[s];
    }
    if (u0 != ~0ull && u0 + B.j < n_units) uoff[u0 + B.j] = cnt;  // counts; scanned in place by the caller
}

__global__ __launch_bounds__(256) void k_deep3_count_hist(FillParams P, uint32_t n, uint32_t slab_len, const uint8_t *__restrict__ present,
                                                          const uint32_t *__restrict__ rank, const uint32_t *__restrict__ sorted_ranks,
                                                          const uint32_t *__restrict__ rank_arg, const uint64_t *__restrict__ ubase,
                                                          const uint32_t *__restrict__ rb_first, const uint32_t *__restrict__ lowcnt,
                                                          uint64_t *__restrict__ uoff, uint64_t n_units, uint32_t *__restrict__ missing_row)
{
    __shared__ uint32_t s_T[4][kHistChunk + 1];
    __shared__ uint32_t s_H[4][kHistChunk + 1 + 64];  // (the prefix stage reads whole steps of 64 bins)
    const unsigned lane = lane_id();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *const T = s_T[wv], *const H = s_H[wv];
    const uint32_t n_rb = rb_first[n];
    uint64_t rb = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t sb = P.slab_begin, se = P.slab_begin + slab_len;
    for (; rb < n_rb; rb += nw) {
        uint32_t lo = 0, hi = n;  // the row of this batch: largest b with rb_first[b] <= rb (rows without entries share a value)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rb_first[mid] <= rb) lo = mid; else hi = mid;
        }
        if (P.adj_deg[lo] > (uint32_t)kHistChunk) continue;  // k_deep3_count_hist_coop
        const DeepBatch B = deep_batch(P, lo, (uint32_t)rb - rb_first[lo], sb, se, present, rank, sorted_ranks, missing_row, lane);
        if (B.t_lo == B.t_hi) continue;  // (a row two hops out: a third vertex only)
        const uint32_t ntc = B.t_hi - B.t_lo, a0 = B.t_lo;
        // table: the thresholds, then all-ones up to the power of two the search walks (3, 5, 7 or 9 levels)
        const uint32_t levels = max(3u, (32u - (uint32_t)__builtin_clz(ntc)) | 1u), padded = 1u << levels;
        for (uint32_t i = lane; i < padded; i += 64) T[i] = i < ntc ? sorted_ranks[B.bst + a0 + i] : 0xFFFFFFFFu;
        for (uint32_t i = lane; i < ((ntc + 64u) & ~63u); i += 64) H[i] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        switch (levels) {
        case 3: deep_hist_rows<3, 1u>(T, H, lane, sorted_ranks, B.have, B.cd, B.cst, 0u); break;
        case 5: deep_hist_rows<5, 1u>(T, H, lane, sorted_ranks, B.have, B.cd, B.cst, 0u); break;
        case 7: deep_hist_rows<7, 1u>(T, H, lane, sorted_ranks, B.have, B.cd, B.cst, 0u); break;
        default: deep_hist_rows<9, 1u>(T, H, lane, sorted_ranks, B.have, B.cd, B.cst, 0u); break;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // bins -> P(t), 64 thresholds at a time; every threshold's unit gets its count
        uint32_t carry = 0;
        for (uint32_t t0 = 0; t0 < ntc; t0 += 64) {
            const uint32_t tt = t0 + lane;
            const uint32_t incl = wave_scan_add(H[tt]) + carry;
            carry = rl32(incl, 63);
            if (tt < ntc) deep_hist_unit(P, B, a0, tt, T[tt], incl, rank_arg, ubase, lowcnt, uoff, n_units);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) uoff[n_units] = 0;
}

__global__ __launch_bounds__(64 * kCoopWaves) void k_deep3_count_hist_coop(FillParams P, uint32_t slab_len, const uint8_t *__restrict__ present,
                                                                           const uint32_t *__restrict__ rank,
                                                                           const uint32_t *__restrict__ sorted_ranks,
                                                                           const uint32_t *__restrict__ rank_arg,
                                                                           const uint64_t *__restrict__ ubase,
                                                                           const uint2 *__restrict__ batches, uint32_t n_batches,
                                                                           const uint32_t *__restrict__ lowcnt,
                                                                           uint64_t *__restrict__ uoff, uint64_t n_units,
                                                                           uint32_t *__restrict__ missing_row)
{
    constexpr int NT = 64 * kCoopWaves, kPer = (kCoopChunk + NT - 1) / NT;
    typedef hipcub::BlockScan<uint32_t, NT> Scan;
    __shared__ typename Scan::TempStorage s_scan;
    __shared__ uint32_t T[kCoopChunk + 1];
    __shared__ uint32_t H[kPer * NT];  // (bins, then their inclusive prefix in place)
    static_assert(kPer * NT > kCoopChunk, "bin ntc has a word");
    const unsigned lane = lane_id(), tid = threadIdx.x;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t sb = P.slab_begin, se = P.slab_begin + slab_len;
    for (uint32_t it = blockIdx.x; it < n_batches; it += gridDim.x) {
        const uint2 bj = batches[it];
        const DeepBatch B = deep_batch(P, bj.x, bj.y, sb, se, present, rank, sorted_ranks, missing_row, lane);  // (every wave the same)
        for (uint32_t a0 = B.t_lo; a0 < B.t_hi; a0 += (uint32_t)kCoopChunk) {
            const uint32_t ntc = min((uint32_t)kCoopChunk, B.t_hi - a0);
            const uint32_t lv = 32u - (uint32_t)__builtin_clz(ntc), levels = lv <= 6u ? 6u : lv <= 9u ? 9u : 12u, padded = 1u << levels;
            for (uint32_t i = tid; i < padded; i += NT) T[i] = i < ntc ? sorted_ranks[B.bst + a0 + i] : 0xFFFFFFFFu;
            for (uint32_t i = tid; i < (uint32_t)(kPer * NT); i += NT) H[i] = 0u;
            __syncthreads();
            // the batch's rows in pieces of 256 entries, dealt to the waves in turn
            switch (levels) {
            case 6: deep_hist_rows<6, (uint32_t)kCoopWaves>(T, H, lane, sorted_ranks, B.have, B.cd, B.cst, wv); break;
            case 9: deep_hist_rows<9, (uint32_t)kCoopWaves>(T, H, lane, sorted_ranks, B.have, B.cd, B.cst, wv); break;
            default: deep_hist_rows<12, (uint32_t)kCoopWaves>(T, H, lane, sorted_ranks, B.have, B.cd, B.cst, wv); break;
            }
            __syncthreads();
            // bins -> inclusive prefix, in place: kPer consecutive bins per thread
            uint32_t h[kPer], mine = 0;
#pragma unroll
            for (int z = 0; z < kPer; z++) {
                h[z] = H[tid * kPer + z];
                mine += h[z];
            }
            uint32_t excl = 0;
            Scan(s_scan).ExclusiveSum(mine, excl);
#pragma unroll
            for (int z = 0; z < kPer; z++) {
                excl += h[z];
                H[tid * kPer + z] = excl;
            }
            __syncthreads();
            for (uint32_t tt = tid; tt < ntc; tt += NT)
                deep_hist_unit(P, B, a0, tt, T[tt], H[tt], rank_arg, ubase, lowcnt, uoff, n_units);
            __syncthreads();  // T and H are rewritten by the next pass
        }
    }
}

// The kept rows of one 64-candidate step -- (c, d, entry) compacted in LDS by the caller -- written as one contiguous piece of
// every output: rows [r_lo, r_lo + rows) of the step to output rows o0 ....  pde: a row is 4e doubles = 2e pieces of 16 bytes.
// For e = 2, 4, 8 the pieces of a row (or of half a row) divide the wave, so a lane keeps ONE column for the whole step: the
// columns of s and b hold the same 16 bytes in every row (loaded once per unit: `fixed`), the columns of c and d take one
// 16-byte load per row (vde[c], nbr_vde[entry]), and all loads of a step are issued before its first store.  (Rounds 1-2
// computed (row, column) per element with two 8-byte loads each, s and b included, a pass of the wave at a time: a load ->
// store -> next load chain per KB written, each link waiting for the stores before it.)  Other widths keep that form.
typedef double dbl2 __attribute__((ext_vector_type(2)));

template <int E>
__device__ __forceinline__ dbl2 deep_fixed_piece(const FillParams &P, uint32_t s, uint32_t b, unsigned lane)
{
    dbl2 v = {0.0, 0.0};
    if constexpr (E == 4 || E == 8) {  // lane -> one of the 2H pieces of the row's first half (s | b)
        constexpr uint32_t H = E / 2;
        const uint32_t piece = lane % (2 * H);
        if (P.out_pde) v = reinterpret_cast<const dbl2 *>(P.vde)[(uint64_t)(piece / H ? b : s) * H + piece % H];
    } else if constexpr (E == 2) {  // lane -> one of the row's four pieces (s, b, c, d)
        const uint32_t col = lane % 4;
        if (P.out_pde && col < 2) v = reinterpret_cast<const dbl2 *>(P.vde)[col ? b : s];
    }
    return v;
}

template <int E>
__device__ __forceinline__ void deep_emit_rows(const FillParams &P, uint32_t s, uint32_t b, dbl2 fixed, const uint32_t *kc,
                                               const uint32_t *kd, const uint32_t *kp, uint32_t r_lo, uint32_t rows, uint64_t o0,
                                               unsigned lane, const double *d_table = nullptr)
{
    // where the fourth vertex' embedding comes from: the per-entry copy (kp = the entry: consecutive entries of a row are
    // consecutive lines) or, d_table given, a table indexed by what the caller put into kp (the vertex table by vertex id)
    const double *const dsrc = d_table ? d_table : P.nbr_vde;
    const uint32_t e = E > 0 ? (uint32_t)E : P.e, D = 4 * e;
    if (rows == 0) return;  // (wave-uniform; the clamped loads below index row rows - 1)
    // Order matters on this hardware (one vmcnt counter for loads AND stores, completed in issue order): a load issued
    // after a store cannot be waited for without waiting for that store's acknowledgement, and hipcc places a wait in front
    // of each use of a loaded value -- with the uses between the stores that made every store pair wait for the stores
    // before it (`global_store; s_waitcnt vmcnt(1); global_store`, round 3's binary).  So: ALL loads of the step first, by
    // every lane (rows past the step's last re-read it: no load under a lane mask), pinned by an empty asm statement = one
    // wait; then the id rows and the embedding rows, stores only.
    auto store_ids = [&]() {
        if (P.out_ids && lane < rows) {  // one 16-byte row per lane, consecutive lanes on consecutive rows
            const uint32_t r = r_lo + lane;
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 v;
            v.x = s;
            v.y = b;
            v.z = kc[r];
            v.w = kd[r];
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(P.out_ids) + (o0 + lane));
        }
    };
    if (!P.out_pde || !(E == 2 || E == 4 || E == 8)) store_ids();
    if (P.out_pde) {
        // rows are D = 4e doubles: always an even count, and o0 * D * 8 is a multiple of 16 bytes
        dbl2 *dst = reinterpret_cast<dbl2 *>(P.out_pde + o0 * D);
        if constexpr (E == 4 || E == 8) {
            // half a row (s | b, then c | d) is 2H pieces: the wave takes 64 / 2H rows per pass, every lane one piece of the
            // c | d half: the lane's fixed piece goes into the first half of its rows, the loaded pieces into the second
            // (e = 8: 128-byte halves, whole lines per store)
            constexpr uint32_t H = E / 2, HP = 2 * H, RP = 64 / HP;
            const uint32_t piece = lane % HP, rsub = lane / HP, cp = piece % H;
            const bool is_d = piece >= H;
            const dbl2 *vde2 = reinterpret_cast<const dbl2 *>(P.vde), *nv2 = reinterpret_cast<const dbl2 *>(dsrc);
            dbl2 v[HP];
#pragma unroll
            for (uint32_t j = 0; j < HP; j++) {
                const uint32_t rc = r_lo + min(j * RP + rsub, rows - 1u);
                const dbl2 *src = is_d ? nv2 + (uint64_t)kp[rc] * H + cp : vde2 + (uint64_t)kc[rc] * H + cp;
                v[j] = *src;
            }
#pragma unroll
            for (uint32_t j = 0; j < HP; j++) asm volatile("" : "+v"(v[j]));
            store_ids();
#pragma unroll
            for (uint32_t j = 0; j < HP; j++) {
                const uint32_t r = j * RP + rsub;
                if (r < rows) {
                    __builtin_nontemporal_store(fixed, dst + (uint64_t)r * (2 * HP) + piece);
                    __builtin_nontemporal_store(v[j], dst + (uint64_t)r * (2 * HP) + HP + piece);
                }
            }
        } else if constexpr (E == 2) {
            // a row is four pieces (s, b, c, d): a lane keeps one column, sixteen rows per pass
            const uint32_t col = lane % 4, rsub = lane / 4;
            const dbl2 *vde2 = reinterpret_cast<const dbl2 *>(P.vde), *nv2 = reinterpret_cast<const dbl2 *>(dsrc);
            dbl2 v[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t rc = r_lo + min(j * 16 + rsub, rows - 1u);
                const dbl2 *src = col == 3 ? nv2 + kp[rc] : vde2 + kc[rc];  // (columns 0 and 1 load too and keep `fixed`)
                const dbl2 got = *src;
                v[j] = col >= 2 ? got : fixed;
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) asm volatile("" : "+v"(v[j]));
            store_ids();
#pragma unroll
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t r = j * 16 + rsub;
                if (r < rows) __builtin_nontemporal_store(v[j], dst + (uint64_t)j * 64 + lane);
            }
        } else {
            auto elem = [&](uint32_t t) -> double {
                const uint32_t r = r_lo + t / D, k2 = t % D, which = k2 / e, comp = k2 % e;
                return which == 0   ? P.vde[(uint64_t)s * e + comp]
                       : which == 1 ? P.vde[(uint64_t)b * e + comp]
                       : which == 2 ? P.vde[(uint64_t)kc[r] * e + comp]
                                    : dsrc[(uint64_t)kp[r] * e + comp];
            };
            for (uint32_t t2 = lane; t2 < rows * (D / 2); t2 += 64) {
                dbl2 v;
                v.x = elem(2 * t2);
                v.y = elem(2 * t2 + 1);
                __builtin_nontemporal_store(v, dst + t2);
            }
        }
    }
    if (P.out_pdl)
        for (uint32_t t = lane; t < rows * D; t += 64) {
            const uint32_t r = r_lo + t / D, k2 = t % D, which = k2 / e, comp = k2 % e;
            const uint32_t v = which == 0 ? s : which == 1 ? b : which == 2 ? kc[r] : kd[r];
            P.out_pdl[o0 * D + t] = P.x[(uint64_t)v * e + comp];
        }
    if (P.out_part && lane < rows) P.out_part[o0 + lane] = P.member[s];
}

// Where a piece of work rejects candidates (it keeps fewer than kSparseNum / kSparseDen of them) the fourth vertices' embeddings are
// gathered from the vertex table by id instead of streamed from the per-entry copy (round 6, DESIGN 3.5: 0.62 -> 0.74 of spec on the
// sparse sampled range of config 5; 0.6 and "always" measured too)
constexpr uint32_t kSparseNum = 19, kSparseDen = 20;  // (the launches pass the numerator: diagnostic builds sweep it, 0 = never, 21 = always)

// E > 0: compile-time embedding width (divisions by constants); E = 0: runtime P.e
//
// One WORKGROUP (WAVES waves) per unit since round 3; round 1 gave a unit to one wave.  On a power-law graph a unit
// behind a hub middle vertex holds 64 third vertices of degree up to thousands -- 10^5 candidates, 80 MB of output -- and
// a sampled range of config 5 (2^24 paths) is covered by a few hundred such units: a few hundred waves on a chip that
// holds eight thousand, the longest of them running for milliseconds (bench.py config5 leg, round 3: 0.06-0.10 of the HBM
// spec).  The flattened candidate space of the unit is now cut into WAVES equal pieces at CANDIDATE granularity (a
// single hub third vertex is split too).  Output slots need the kept rows before a piece, so the waves walk their
// (WAVES = 4 where no row is longer than 64 entries -- units of a few hundred candidates -- and 16 on graphs with hub rows.)
// pieces twice: pass A reads the candidates' ranks only and counts (4 bytes per candidate against ~140 written per
// candidate in pass B), an LDS prefix over the waves gives every piece its first slot, pass B emits.  A piece whose slots
// lie outside the requested range [P.begin, P.end) skips pass B.
template <int E, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_deep3(FillParams P, const uint32_t *__restrict__ upair,
                                                           const uint64_t *__restrict__ ufirst,
                                                           const uint64_t *__restrict__ uoff, uint64_t u_begin, uint64_t u_end,
                                                           uint32_t sparse_num)
{
    // the unit's segment table (first candidate, row start and id of each of its 64 third vertices): built by wave 0, read by
    // every wave; two copies, used alternately, so that wave 0 may build the next unit's table while the others still emit
    __shared__ uint32_t s_off[2][65], s_st[2][64], s_c[2][64];
    __shared__ uint32_t s_kc[WAVES][64], s_kd[WAVES][64], s_kp[WAVES][64];  // kept rows of one step
    __shared__ uint32_t s_piece[WAVES];  // kept rows of every wave's piece (pass A)
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t *kc = s_kc[wv], *kd = s_kd[wv], *kp = s_kp[wv];
    unsigned flip = 0;
    for (uint64_t u = u_begin + blockIdx.x; u < u_end; u += gridDim.x) {  // block-uniform: every wave takes part in the barriers
        const uint64_t base = uoff[u], nxt = uoff[u + 1];
        if (nxt == base || base >= P.end || nxt <= P.begin) continue;
        const uint32_t w = upair[u];
        const uint32_t i = P.erow[w], b = P.pnbr[w];
        const uint32_t s = P.sorted[P.slab_begin + i], thr = P.slab_begin + i;
        const dbl2 fixed = deep_fixed_piece<E>(P, s, b, lane);
        flip ^= 1u;
        uint32_t *off = s_off[flip], *rst = s_st[flip], *rc = s_c[flip];
        if (wv == 0) {
            const uint32_t bst = P.adj_start[b], bd = P.adj_deg[b];
            const uint32_t k = (uint32_t)(u - ufirst[w]) * 64u + lane;  // this unit's 64 third vertices, one lane each
            uint32_t c = 0, cd = 0, cst = 0;
            if (k < bd) {
                c = P.nbrs[bst + k];
                if (c != s) {  // (a missing 2-hop row was reported by the count pass)
                    cd = P.adj_deg[c];
                    cst = P.adj_start[c];
                }
            }
            uint32_t incl = cd;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o, 64);
                if ((int)lane >= o) incl += t;
            }
            off[lane] = incl - cd;
            rst[lane] = cst;
            rc[lane] = c;
            if (lane == 63) off[64] = incl;
        }
        __syncthreads();
        const uint32_t n_cand = off[64];
        // this wave's piece of the flattened candidates: whole steps of 64, the pieces as equal as that allows
        const uint32_t n_steps = (n_cand + 63u) / 64u;
        const uint32_t q_lo = min(n_cand, (n_steps * wv / WAVES) * 64u), q_hi = min(n_cand, (n_steps * (wv + 1) / WAVES) * 64u);
        auto locate = [&](uint32_t q, uint32_t &seg, uint32_t &pos) {
            uint32_t lo = 0;  // last segment whose first candidate is <= q (skips empty segments)
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (off[lo + step] <= q) lo += step;
            seg = lo;
            pos = rst[lo] + (q - off[lo]);
        };
        // pass A: kept rows of the piece (ranks only)
        uint32_t mine = 0;
        for (uint32_t q0 = q_lo; q0 < q_hi; q0 += 64) {
            const uint32_t q = q0 + lane;
            bool keep = false;
            if (q < q_hi) {
                uint32_t seg, pos;
                locate(q, seg, pos);
                keep = P.nbr_rank[pos] > thr && P.nbrs[pos] != b;
            }
            mine += (uint32_t)__popcll(__ballot(keep));
        }
        if (lane == 0) s_piece[wv] = mine;
        __syncthreads();
        uint64_t running = 0;
        for (unsigned v = 0; v < wv; v++) running += s_piece[v];
        __syncthreads();  // s_piece is rewritten by the next unit
        const uint64_t piece0 = base + running;
        if (mine == 0 || piece0 >= P.end || piece0 + mine <= P.begin) continue;  // (after both barriers: block-uniform control flow above)
        const bool by_vertex = (uint64_t)mine * kSparseDen < (uint64_t)(q_hi - q_lo) * sparse_num;  // (see kSparseNum)
        running = 0;
        // pass B: flattened candidates of the piece, 64 per step
        for (uint32_t q0 = q_lo; q0 < q_hi; q0 += 64) {
            const uint32_t q = q0 + lane;
            const bool act = q < q_hi;
            uint32_t lo = 0, pos = 0;
            if (act) locate(q, lo, pos);
            uint32_t d = 0, rd = 0;
            if (act) {
                d = P.nbrs[pos];
                rd = P.nbr_rank[pos];
            }
            const bool keep = act && rd > thr && d != b;
            const uint64_t mask = __ballot(keep);
            {
                // rows of this step occupy slots [slot0, slot0 + cnt): compact (c, d, entry) into LDS, then the
                // wave writes the rows as one contiguous region, consecutive lanes on consecutive elements
                const uint32_t cnt = (uint32_t)__popcll(mask);
                const uint64_t slot0 = piece0 + running;
                if (cnt && slot0 < P.end && slot0 + cnt > P.begin) {
                    if (keep) {
                        const uint32_t r = (uint32_t)__popcll(mask & lt);
                        kc[r] = rc[lo];
                        kd[r] = d;
                        kp[r] = by_vertex ? d : pos;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const uint32_t r_lo = slot0 < P.begin ? (uint32_t)(P.begin - slot0) : 0u;
                    const uint32_t r_hi = slot0 + cnt > P.end ? (uint32_t)(P.end - slot0) : cnt;
                    const uint64_t o0 = slot0 + r_lo - P.begin;  // first output row of this step
                    const uint32_t rows = r_hi - r_lo;
                    deep_emit_rows<E>(P, s, b, fixed, kc, kd, kp, r_lo, rows, o0, lane, by_vertex ? P.vde : nullptr);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            running += __popcll(mask);
        }
    }
}

// ---- emission by slices (round 3) ---------------------------------------------------------------------------------------
// k_deep3 above gives a unit to one workgroup.  Counters of a 2^26-path range of config 5 under it (profiles/
// r03_deep_pmc.txt): 1 376 units, 22 016 waves over the kernel's life on a chip that holds 8 192 at once, the waves waiting
// 87 % of their cycles.  Behind a hub the unit of WORK is therefore a SLICE: kSliceSteps x 64 consecutive candidates of a
// unit's flattened candidate space, one wave each, however many a unit needs:
//   k_deep_slice_counts       one wave per unit of the requested range: its candidates -> its number of slices   (+ scan)
//   k_deep3_slices<E, false>  one wave per slice: the kept rows among its candidates (ranks only)                 (+ scan)
//   k_deep3_slices<E, true>   one wave per slice: first slot = the unit's + the kept rows of its earlier slices; emit
// Every wave rebuilds its unit's segment table (64 lanes x three loads, served by the L2 for the slices of one unit) in its
// own corner of the LDS: no workgroup barrier anywhere; the candidates of the next step are fetched before the rows of this
// one are written.  Same process, same output buffers (scripts/deep_ab.py, e = 8, fraction of the 8 TB/s spec at 352
// algorithmic bytes per path): config-5 graph, ranges of 2^24 paths -- units 0.47-0.49, slices 0.50-0.58; ranges of 2^26 --
// 0.53-0.61 both ways; G(100K, 1M), every unit a few hundred candidates -- units 0.72, slices 0.58-0.60 (three kernels and
// three table builds per unit).  The library takes slices on graphs with hub rows, units elsewhere (fill_device).
// (Tried and dropped: collecting kept rows across steps in an LDS ring and writing them 64 at a time -- within 1 % on every
// range, the sparse ones behind high-ranked starts included: the steps' round trips are not what the slices wait for.)
constexpr uint32_t kSliceSteps = 16;  // 8 ... 32 time alike on the config-5 graph; 2 and 4 lose to the table rebuilds (round 6, one-launch
                                      // kernel: 32 gains 1 % on the sparsest sampled range and loses 1-4 % on the others)

// segment table of unit u for one wave: first candidate, row start and id of each of its 64 third vertices; returns the
// unit's candidates
__device__ __forceinline__ uint32_t deep_unit_table(const FillParams &P, uint32_t w, uint32_t part, uint32_t s, uint32_t b,
                                                    unsigned lane, uint32_t *off, uint32_t *rst, uint32_t *rc)
{
    const uint32_t bst = P.adj_start[b], bd = P.adj_deg[b];
    const uint32_t k = part * 64u + lane;  // this unit's 64 third vertices, one lane each
    uint32_t c = 0, cd = 0, cst = 0;
    if (k < bd) {
        c = P.nbrs[bst + k];
        if (c != s) {  // (a missing 2-hop row was reported by the count pass)
            cd = P.adj_deg[c];
            cst = P.adj_start[c];
        }
    }
    uint32_t incl = cd;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if ((int)lane >= o) incl += t;
    }
    if (off) {
        off[lane] = incl - cd;
        rst[lane] = cst;
        rc[lane] = c;
        if (lane == 63) off[64] = incl;
    }
    return rl32(incl, 63);
}

// (uinfo, when given: {rank of the start vertex, middle vertex, part of the pair's row, rank of the middle vertex} of every unit, so that
// the emitting waves find them with one 16-byte load instead of a chain of four dependent ones)
__global__ __launch_bounds__(256) void k_deep_slice_counts(FillParams P, const uint32_t *__restrict__ upair,
                                                           const uint64_t *__restrict__ ufirst, const uint64_t *__restrict__ uoff,
                                                           uint64_t u_lo, uint32_t n_u, uint32_t *__restrict__ nsl,
                                                           const uint32_t *__restrict__ rank, uint4 *__restrict__ uinfo)
{
    const unsigned lane = lane_id();
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t ui = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6; ui <= n_u; ui += nw) {
        uint32_t out = 0;
        if (ui < n_u) {
            const uint64_t u = u_lo + ui, base = uoff[u], nxt = uoff[u + 1];
            if (nxt != base && base < P.end && nxt > P.begin) {
                const uint32_t w = upair[u];
                const uint32_t i = P.erow[w], b = P.pnbr[w], s = P.sorted[P.slab_begin + i], part = (uint32_t)(u - ufirst[w]);
                const uint32_t n_cand = deep_unit_table(P, w, part, s, b, lane, nullptr, nullptr, nullptr);
                out = ((n_cand + 63u) / 64u + kSliceSteps - 1u) / kSliceSteps;
                if (uinfo && lane == 0) uinfo[ui] = make_uint4(P.slab_begin + i, b, part, rank[b]);
            }
        }
        if (lane == 0) nsl[ui] = out;
    }
}

#ifdef GNNPE_DIAG  // (the two-pass form of rounds 3-5: kept for the A/B against k_deep3_slices_fused below)
template <int E, bool EMIT>
__global__ __launch_bounds__(256) void k_deep3_slices(FillParams P, const uint32_t *__restrict__ upair,
                                                      const uint64_t *__restrict__ ufirst, const uint64_t *__restrict__ uoff,
                                                      uint64_t u_lo, uint32_t n_u, const uint32_t *__restrict__ sfirst,
                                                      uint32_t n_slices, uint64_t *__restrict__ skept,
                                                      const uint64_t *__restrict__ sscan)
{
    __shared__ uint32_t s_off[4][65], s_st[4][64], s_c[4][64];
    __shared__ uint32_t s_kc[4][64], s_kd[4][64], s_kp[4][64];  // kept rows of one step
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t *off = s_off[wv], *rst = s_st[wv], *rc = s_c[wv];
    uint32_t *kc = s_kc[wv], *kd = s_kd[wv], *kp = s_kp[wv];
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t sl = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6; sl < n_slices; sl += nw) {
        uint32_t lo_u = 0, hi_u = n_u;  // the slice's unit: the last one whose first slice is <= sl (units without slices share a value)
        while (hi_u - lo_u > 1) {
            const uint32_t mid = (lo_u + hi_u) >> 1;
            if (sfirst[mid] <= sl) lo_u = mid; else hi_u = mid;
        }
        const uint32_t first_sl = sfirst[lo_u], part = (uint32_t)sl - first_sl;
        const uint64_t u = u_lo + lo_u;
        uint64_t piece0 = 0;
        if (EMIT) {
            const uint64_t mine = skept[sl];
            piece0 = uoff[u] + (sscan[sl] - sscan[first_sl]);
            if (mine == 0 || piece0 >= P.end || piece0 + mine <= P.begin) continue;
        }
        const uint32_t w = upair[u];
        const uint32_t i = P.erow[w], b = P.pnbr[w];
        const uint32_t s = P.sorted[P.slab_begin + i], thr = P.slab_begin + i;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the previous slice's table reads are done
        __builtin_amdgcn_wave_barrier();
        const uint32_t n_cand = deep_unit_table(P, w, (uint32_t)(u - ufirst[w]), s, b, lane, off, rst, rc);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t q_lo = part * (kSliceSteps * 64u), q_hi = min(n_cand, q_lo + kSliceSteps * 64u);
        auto locate = [&](uint32_t q, uint32_t &seg, uint32_t &pos) {
            uint32_t lo = 0;  // last segment whose first candidate is <= q (skips empty segments)
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (off[lo + step] <= q) lo += step;
            seg = lo;
            pos = rst[lo] + (q - off[lo]);
        };
        if (!EMIT) {  // kept rows of the slice (ranks only)
            uint32_t mine = 0;
            for (uint32_t q0 = q_lo; q0 < q_hi; q0 += 64) {
                const uint32_t q = q0 + lane;
                bool keep = false;
                if (q < q_hi) {
                    uint32_t seg, pos;
                    locate(q, seg, pos);
                    keep = P.nbr_rank[pos] > thr && P.nbrs[pos] != b;
                }
                mine += (uint32_t)__popcll(__ballot(keep));
            }
            if (lane == 0) skept[sl] = mine;
            continue;
        }
        const dbl2 fixed = deep_fixed_piece<E>(P, s, b, lane);
        uint64_t running = 0;
        // the candidates of a step: segment, entry, fourth vertex and its rank -- fetched one step ahead of the rows being written
        auto fetch = [&](uint32_t q0, uint32_t &seg, uint32_t &pos, uint32_t &d, uint32_t &rd) {
            const uint32_t q = q0 + lane;
            seg = pos = d = rd = 0;
            if (q < q_hi) {
                locate(q, seg, pos);
                d = P.nbrs[pos];
                rd = P.nbr_rank[pos];
            }
            return q < q_hi;
        };
        uint32_t lo, pos, d, rd;
        bool act = fetch(q_lo, lo, pos, d, rd);
        for (uint32_t q0 = q_lo; q0 < q_hi; q0 += 64) {
            uint32_t n_lo = 0, n_pos = 0, n_d = 0, n_rd = 0;
            bool n_act = false;
            if (q0 + 64 < q_hi) n_act = fetch(q0 + 64, n_lo, n_pos, n_d, n_rd);
            const bool keep = act && rd > thr && d != b;
            const uint64_t mask = __ballot(keep);
            // rows of this step occupy slots [slot0, slot0 + cnt): compact (c, d, entry) into LDS, then the wave writes the
            // rows as one contiguous region, consecutive lanes on consecutive elements
            const uint32_t cnt = (uint32_t)__popcll(mask);
            const uint64_t slot0 = piece0 + running;
            running += cnt;
            if (cnt && slot0 < P.end && slot0 + cnt > P.begin) {
                if (keep) {
                    const uint32_t r = (uint32_t)__popcll(mask & lt);
                    kc[r] = rc[lo];
                    kd[r] = d;
                    kp[r] = pos;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t r_lo = slot0 < P.begin ? (uint32_t)(P.begin - slot0) : 0u;
                const uint32_t r_hi = slot0 + cnt > P.end ? (uint32_t)(P.end - slot0) : cnt;
                const uint64_t o0 = slot0 + r_lo - P.begin;  // first output row of this step
                deep_emit_rows<E>(P, s, b, fixed, kc, kd, kp, r_lo, r_hi - r_lo, o0, lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            act = n_act;
            lo = n_lo;
            pos = n_pos;
            d = n_d;
            rd = n_rd;
        }
    }
}

#endif

// ---- emission by slices in ONE launch (round 6) ---------------------------------------------------------------------------
// The two-pass form above needs the kept rows of every slice before any row can be written: a kernel that reads every candidate's
// rank, a scan, then the kernel that reads them again and writes (config 5, a range of 2^26 paths: 0.25-0.52 ms + two scan launches
// in front of a 4.2-4.6 ms writer; profiles/r06_deep_emit_calls.txt).  Here a wave does both for its slice:
//   pass A   the slice's 16 steps of 64 candidates: rank only (4 bytes per candidate; "is it b" is a rank comparison -- ranks are a
//            permutation), the kept lanes of every step as a 64-bit mask in LDS, their number;
//   offset   first output slot = the unit's (uoff[u], from the count) + the kept rows of the unit's EARLIER slices: decoupled
//            look-back over the status words of the slices before it (value << 2 | state, as in k_start_scan), which never leaves the
//            unit -- the unit's first slice publishes an inclusive prefix at once;
//   pass B   the steps again from the masks: only KEPT lanes fetch their fourth vertex; rows as before (deep_emit_rows).
// One wave per slice, workgroups in launch order, exit -- NOT a resident grid taking tickets: a wave's first load of its next slice
// would wait for the acknowledgement of every store of the slice before (one counter for loads and stores, completed in issue
// order), and that form measured 4-12 % slower than the two passes it replaces (profiles/r06_deep_emit_calls.txt).  The look-back
// therefore leans on the dispatcher starting workgroups in index order (every slice waited for has a smaller index); it does not
// DEPEND on it: after kLookbackPolls polls without an answer the wave counts the unit's earlier slices itself (their candidates
// are behind the table it already holds), so no wave can wait for ever.
constexpr uint32_t kLookbackPolls = 2048;
template <int E>
__global__ __launch_bounds__(256) void k_deep3_slices_fused(FillParams P, const uint64_t *__restrict__ uoff,
                                                            const uint4 *__restrict__ uinfo, uint64_t u_lo, uint32_t n_u,
                                                            const uint32_t *__restrict__ sfirst, uint32_t n_slices,
                                                            unsigned long long *__restrict__ status, uint32_t *__restrict__ fallbacks,
                                                            uint32_t sparse_num)
{
    __shared__ uint32_t s_off[4][65], s_st[4][64], s_c[4][64];
    __shared__ uint32_t s_kc[4][64], s_kd[4][64], s_kp[4][64];  // kept rows of one step
    __shared__ uint64_t s_mask[4][kSliceSteps];                 // kept lanes of every step of the slice
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    uint32_t *off = s_off[wv], *rst = s_st[wv], *rc = s_c[wv];
    uint32_t *kc = s_kc[wv], *kd = s_kd[wv], *kp = s_kp[wv];
    uint64_t *msk = s_mask[wv];
    const uint64_t sl64 = (uint64_t)blockIdx.x * 4u + wv;
    if (sl64 >= n_slices) return;  // (no workgroup barrier anywhere in this kernel)
    const uint32_t sl = (uint32_t)sl64;
    // the slice's unit: the last one whose first slice is <= sl (units without slices share a value) -- a 64-ary search, one probe
    // per lane and round: two round trips for up to 4 096 units (a binary search was twelve dependent loads per slice)
    uint32_t lo_u = 0, len_u = n_u;
    while (len_u > 1) {
        const uint32_t stride = (len_u + 63u) / 64u, idx = lo_u + lane * stride;
        const uint64_t le = __ballot(lane * stride < len_u && sfirst[min(idx, lo_u + len_u - 1u)] <= sl);  // (lane 0 always: the invariant)
        const uint32_t k = (uint32_t)__popcll(le) - 1u;
        len_u = min(stride, len_u - k * stride);
        lo_u += k * stride;
    }
    const uint32_t first_sl = sfirst[lo_u], part = sl - first_sl;
    const uint64_t u = u_lo + lo_u;
    const uint4 ui4 = uinfo[lo_u];  // {rank[s], b, part of b's row, rank[b]}: k_deep_slice_counts
    const uint32_t thr = ui4.x, b = ui4.y, rb = ui4.w;
    const uint32_t s = P.sorted[thr];
    const uint32_t n_cand = deep_unit_table(P, 0u, ui4.z, s, b, lane, off, rst, rc);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    auto locate = [&](uint32_t q, uint32_t &seg, uint32_t &pos) {
        uint32_t lo = 0;  // last segment whose first candidate is <= q (skips empty segments)
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (off[lo + step] <= q) lo += step;
        seg = lo;
        pos = rst[lo] + (q - off[lo]);
    };
    // kept rows of slice `pt` of this unit: ranks only, all the steps' loads before the first ballot; the masks go to LDS for
    // the wave's own slice
    auto slice_kept = [&](uint32_t pt, bool keep_masks) -> uint32_t {
        const uint32_t a_lo = pt * (kSliceSteps * 64u), a_hi = min(n_cand, a_lo + kSliceSteps * 64u);
        constexpr uint32_t kChunk = 16;  // steps whose loads are in flight together
        uint32_t kept = 0;
        for (uint32_t s0 = 0; s0 < kSliceSteps; s0 += kChunk) {
            if (a_lo + s0 * 64u >= a_hi) {  // (wave-uniform) nothing behind: empty masks
                if (keep_masks && lane == 0)
                    for (uint32_t st = s0; st < kSliceSteps; st++) msk[st] = 0ull;
                break;
            }
            uint32_t rk[kChunk];
#pragma unroll
            for (uint32_t st = 0; st < kChunk; st++) {
                const uint32_t q = a_lo + (s0 + st) * 64u + lane;
                rk[st] = 0u;  // (rank 0 is never kept: thr >= 0)
                if (q < a_hi) {
                    uint32_t seg, pos;
                    locate(q, seg, pos);
                    rk[st] = P.nbr_rank[pos];
                }
            }
#pragma unroll
            for (uint32_t st = 0; st < kChunk; st++) {
                const uint64_t m = __ballot(rk[st] > thr && rk[st] != rb);
                if (keep_masks && lane == 0) msk[s0 + st] = m;
                kept += (uint32_t)__popcll(m);
            }
        }
        return kept;
    };
    const uint32_t q_lo = part * (kSliceSteps * 64u), q_hi = min(n_cand, q_lo + kSliceSteps * 64u);
    const uint32_t mine = slice_kept(part, true);
    // the slice's first slot: look back over the unit's earlier slices
    uint64_t prefix = 0;
    if (lane == 0)
        __hip_atomic_store(&status[sl], (unsigned long long)(((uint64_t)mine << 2) | (part ? 1ull : 2ull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    {
        int64_t top = (int64_t)sl - 1;  // nearest predecessor; never below first_sl
        uint32_t polls = 0;
        bool gave_up = false;
        while (top >= (int64_t)first_sl) {
            const int64_t idx = top - (int64_t)lane;
            // (in front of the unit's first slice: an inclusive prefix of zero)
            const unsigned long long wd = idx >= (int64_t)first_sl ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 2ull;
            const uint64_t inc = __ballot((wd & 3ull) == 2ull), none = __ballot((wd & 3ull) == 0ull);
            const uint32_t first_inc = inc ? (uint32_t)__builtin_ctzll(inc) : 64u;
            const uint64_t need = first_inc >= 63u ? ~0ull : ((1ull << (first_inc + 1)) - 1ull);  // lanes up to the first inclusive prefix
            if (none & need) {  // one of them has not published yet
                if (++polls > kLookbackPolls) {
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            uint64_t v = lane <= first_inc ? (uint64_t)(wd >> 2) : 0ull;
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
            prefix += v;
            if (first_inc < 64u) break;
            top -= 64;
        }
        if (gave_up) {  // (never seen; the dispatcher starts workgroups in index order) -- count the earlier slices here
            prefix = 0;
            for (uint32_t pt = 0; pt < part; pt++) prefix += slice_kept(pt, false);
            if (lane == 0) atomicAdd(fallbacks, 1u);
        }
    }
    if (lane == 0 && part)
        __hip_atomic_store(&status[sl], (unsigned long long)(((prefix + mine) << 2) | 2ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t piece0 = uoff[u] + prefix;
    if (mine == 0 || piece0 >= P.end || piece0 + mine <= P.begin) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // lane 0's masks are visible to the wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pass B: the kept lanes' fourth vertices, one step ahead of the rows being written
    const dbl2 fixed = deep_fixed_piece<E>(P, s, b, lane);
    // a slice that rejects candidates reads the fourth vertices' embeddings from the vertex table (by id: 256 MB at config 5, mostly
    // answered by L2 / Infinity Cache) instead of the per-entry copy, of whose lines it would use a fraction (kSparseNum above)
    const bool by_vertex = (uint64_t)mine * kSparseDen < (uint64_t)(q_hi - q_lo) * sparse_num;
    uint64_t running = 0;
    auto fetch = [&](uint32_t st, uint64_t &mask, uint32_t &seg, uint32_t &pos, uint32_t &d) {
        mask = msk[st];
        seg = pos = d = 0;
        if ((mask >> lane) & 1ull) {
            locate(q_lo + st * 64u + lane, seg, pos);
            d = P.nbrs[pos];
        }
    };
    const uint32_t n_steps = (q_hi - q_lo + 63u) / 64u;
    uint64_t mask;
    uint32_t lo, pos, d;
    fetch(0, mask, lo, pos, d);
    for (uint32_t st = 0; st < n_steps; st++) {
        uint64_t n_mask = 0;
        uint32_t n_lo = 0, n_pos = 0, n_d = 0;
        if (st + 1 < n_steps) fetch(st + 1, n_mask, n_lo, n_pos, n_d);
        const bool keep = (mask >> lane) & 1ull;
        const uint32_t cnt = (uint32_t)__popcll(mask);
        const uint64_t slot0 = piece0 + running;
        running += cnt;
        if (cnt && slot0 < P.end && slot0 + cnt > P.begin) {
            if (keep) {
                const uint32_t r = (uint32_t)__popcll(mask & lt);
                kc[r] = rc[lo];
                kd[r] = d;
                kp[r] = by_vertex ? d : pos;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t r_lo = slot0 < P.begin ? (uint32_t)(P.begin - slot0) : 0u;
            const uint32_t r_hi = slot0 + cnt > P.end ? (uint32_t)(P.end - slot0) : cnt;
            const uint64_t o0 = slot0 + r_lo - P.begin;  // first output row of this step
            deep_emit_rows<E>(P, s, b, fixed, kc, kd, kp, r_lo, r_hi - r_lo, o0, lane, by_vertex ? P.vde : nullptr);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        mask = n_mask;
        lo = n_lo;
        pos = n_pos;
        d = n_d;
    }
}

// 64-bit checksum of a chunk of emitted rows (sum of per-row hashes; each hash includes the row's global path id, so order matters), so that outputs
// too large to keep (config 5) can still be compared between runs, rank counts and the CPU checker.
__global__ void k_rows_checksum(uint64_t n_rows, uint32_t L, const uint32_t *__restrict__ ids, uint64_t first_id,
                                unsigned long long *__restrict__ sum)
{
    uint64_t acc = 0;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t h = (first_id + r) * 0x9E3779B97F4A7C15ull;  // the row's global path id takes part: order matters
        for (uint32_t k = 0; k < L; k++) {
            h ^= ids[r * L + k] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            h *= 0xBF58476D1CE4E5B9ull;
        }
        acc += h ^ (h >> 31);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, (unsigned long long)acc);
}

}  // namespace gnnpe
