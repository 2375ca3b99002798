// gnnpe_fill_tiles.hip.h -- the OUTPUT-TILE-DRIVEN emit kernel of the l = 2 enumeration (R2 emit + R5; custom.h:66-92,
// 546-572).  Same inputs as k_fill_ranked (gnnpe_fill_ranked.hip.h: row blocks in descending rank order, one
// RankedPair per directed (s, b) pair, eoff = scan of the pairs' counts), other decomposition:
//
//   k_fill_ranked : one wave per START VERTEX, resident grid; a wave issues store after store for the life of the launch
//   k_fill_tiles  : one wave per 64-ROW OUTPUT TILE (3 KiB of pde rows + 768 B of id rows at e = 2), workgroups in launch
//                   order, every wave stores its tile in one burst and ends -- the shape that writes at the same rate into
//                   every allocation (profiles/r03_buffer_classes.txt: one-shot 4 KiB tiles 6.9-7.0 TB/s in all buffers,
//                   resident store loops 5.0-6.3 depending on the allocation)
//
// A tile starts in the middle of a pair in general.  The tile table (k_tile_first, one streaming pass over eoff) names
// the pair that holds the tile's first row and how many of the pair's rows lie before it; the pairs of a tile are the
// consecutive pair records up to the next tile's first pair.  The records of a pair are in RANK order while its rows are
// in ID order (row = popcount(G below the record's id-position)), so a pair cut by a tile boundary is read whole by both
// tiles and each keeps the rows that fall inside it.
#pragma once

#include "gnnpe_fill_ranked.hip.h"

namespace gnnpe {

// {start vertex, middle vertex} of every pair of the slab in emission order: structure of (graph, order, slab), built
// once beside poffs.  One wave per start vertex.
__global__ __launch_bounds__(256) void k_pair_ends(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted,
                                                   const uint32_t *__restrict__ adj_start,
                                                   const uint32_t *__restrict__ poffs, const uint32_t *__restrict__ nbrs,
                                                   uint2 *__restrict__ pst)
{
    const unsigned lane = lane_id();
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < len; w += nw) {
        const uint32_t s = sorted[slab_begin + w], e0 = poffs[w], d = poffs[w + 1] - e0, a = adj_start[s];
        for (uint32_t k = lane; k < d; k += 64) pst[e0 + k] = make_uint2(s, nbrs[a + k]);
    }
}

// tile table for tiles of `ts` rows: tfirst[t] = {pair holding output slot ts * t | rows of that pair before the slot << 32};
// entry ceil(total / ts) names the last non-empty pair (the upper end of the last tile's pair range).  Pair-driven: a
// pair names the tiles whose first slot falls inside it (at most one unless it is a hub pair).
__global__ void k_tile_first(uint64_t n_pairs, const uint64_t *__restrict__ eoff, uint32_t ts, uint64_t n_tiles_cap,
                             uint64_t *__restrict__ tfirst)
{
    const uint64_t total = eoff[n_pairs];
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_pairs; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = eoff[e], z = eoff[e + 1];
        if (z == a) continue;
        for (uint64_t t = (a + ts - 1) / ts; t * ts < z && t < n_tiles_cap; t++) tfirst[t] = e | ((t * ts - a) << 32);
        if (z == total) {
            const uint64_t T = (total + ts - 1) / ts;
            if (T < n_tiles_cap) tfirst[T] = e;
        }
    }
}

// One wave per output tile of TS = 64 KT rows.  SP = pairs per strip (one lane each); a tile whose pairs do not fit one
// strip (runs of empty pairs behind high-ranked start vertices) takes several, flushing the rows of each.  A wave lives
// for three dependent memory round trips and nothing else, so everything between them is kept off LDS round trips:
//   hop 1  tile table (scalar loads: the tile index is wave-uniform)
//   hop 2  the strip's pair records {block, count, G} (one 16-byte load) and end points {s, b}, one lane per pair; DPP scan
//   hop 3  the pairs that own rows of the tile fetch vde[s] and their block's header vde[b]; ALL records of those pairs, KT + 2
//          wave-wide loads issued back to back (a pair cut by the tile's edge is read whole).  record -> pair: every pair
//          marks its first record's slot in LDS, a DPP prefix maximum over the marks names the pair (no binary search)
//   burst  records parked at their rows, rows -> global memory: consecutive lanes on consecutive 16-byte pieces of the pde
//          rows, 12-byte id rows
// DIAG: the diagnostic instantiation (in-kernel stamps, knock-outs); the product launches DIAG = false.
// The body is a device function of the tile index and a range of the tile's pairs [k_lo, k_hi): the one-wave-per-tile kernel
// passes the whole tile; the job kernel behind k_fill_tickets (gnnpe_fill_tickets.hip.h) passes one strip of a tile whose
// pairs do not fit that kernel's pipeline -- strips are independent of each other, the tile row of a strip's first record
// being eoff[its first pair] - the tile's first slot.
template <int E, bool PACKED, int KT, int SP, int WPB, bool DIAG>
__device__ __forceinline__ void fill_tile_strips(const FillParams &P, const uint64_t *__restrict__ tfirst,
                                                 const RankedPair *__restrict__ pairs, const uint2 *__restrict__ pst,
                                                 const char *__restrict__ recs, const uint64_t t, const uint64_t total,
                                                 const uint32_t k_lo, const uint32_t k_hi, uint32_t exp_flags,
                                                 unsigned long long *__restrict__ stamps)
{
    typedef typename RecOf<E, PACKED>::type Rec;
    constexpr int D = 3 * E;
    constexpr int EP = E + (E & 1);
    constexpr int TS = 64 * KT;
    constexpr int NP = KT + 1;            // record passes every tile makes; a tile's pairs hold at most TS + 2 * 62 records, so
    constexpr int NMARK = 64 * (KT + 2);  // one more pass exists for the few that hold more than 64 NP (record slots: NMARK)
    static_assert(SP <= 64 && (SP & (SP - 1)) == 0, "one lane per pair of the strip");
    __shared__ uint32_t s_cs[WPB][SP], s_blk[WPB][SP], s_b[WPB][SP], s_s[WPB][SP];  // (s_cs without a sentinel: 20 480 B per workgroup at e = 2 = eight per CU)
    __shared__ uint64_t s_G[WPB][SP];
    __shared__ __attribute__((aligned(16))) double s_vb[WPB][SP * EP];
    __shared__ __attribute__((aligned(16))) double s_vs[WPB][SP * EP];
    __shared__ uint32_t s_id[WPB][TS];
    __shared__ uint8_t s_a[WPB][TS];
    __shared__ __attribute__((aligned(4))) uint8_t s_mark[WPB][NMARK];
    __shared__ __attribute__((aligned(16))) double s_v[WPB][TS * EP];
    const unsigned lane = lane_id(), wv = wave_id();
    unsigned long long st_prev = 0;
    const bool st_on = DIAG && stamps && ((blockIdx.x * WPB + wv) & 63u) == 0;  // one wave in 64 is timed (its waits are real)
    auto stamp = [&](int phase, bool drain) {
        if constexpr (DIAG) {
            if (st_on) {
                if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned long long now;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                if (phase >= 0 && lane == 0) atomicAdd(&stamps[phase], now - st_prev);
                st_prev = now;
            }
        }
    };
    stamp(-1, false);
    const uint64_t slot0 = t * TS;
    if (slot0 >= total) return;  // capped launches cover the buffer's capacity, not the count
    const bool want_pde = P.out_pde != nullptr;
    uint32_t *const cs = s_cs[wv], *const sblk = s_blk[wv], *const sb = s_b[wv], *const ss = s_s[wv], *const sid = s_id[wv];
    uint64_t *const sG = s_G[wv];
    uint8_t *const sa = s_a[wv], *const smark = s_mark[wv];
    double *const svb = s_vb[wv], *const svs = s_vs[wv], *const sv = s_v[wv];

    const uint64_t tf0 = tfirst[t], tf1 = tfirst[t + 1];
    if (lane < (unsigned)(NMARK / 4)) reinterpret_cast<uint32_t *>(smark)[lane] = 0u;  // under the table's latency
    const uint32_t e0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tf0);
    const uint32_t e1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tf1);
    int32_t carry = -(int32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tf0 >> 32));  // tile row of the strip's first record
    const uint32_t np = min(e1 - e0 + 1, k_hi);
    if (k_lo) carry = (int32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(P.eoff[(uint64_t)e0 + k_lo] - slot0));
    stamp(0, false);

    // rows [r_lo, r_hi) of the tile -> output slots slot0 + row, clipped to [P.begin, P.end)
    auto flush = [&](uint32_t r_lo, uint32_t r_hi) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint64_t glo = max(slot0 + r_lo, P.begin), ghi = min(slot0 + r_hi, P.end);
        if (ghi > glo && !(DIAG && (exp_flags & 1u))) {
            const uint32_t r0 = (uint32_t)(glo - slot0), nr = (uint32_t)(ghi - glo);
            const uint64_t o = glo - P.begin;
            if (P.out_ids) {
                IdRow *dst = reinterpret_cast<IdRow *>(P.out_ids) + o;
#pragma unroll 1
                for (uint32_t i = 0; i < (uint32_t)KT; i++) {
                    const uint32_t g = lane + 64 * i;
                    if (g < nr) {
                        const uint32_t a = sa[r0 + g];
                        uint32_t *q = reinterpret_cast<uint32_t *>(&dst[g]);
                        __builtin_nontemporal_store(ss[a], q);
                        __builtin_nontemporal_store(sb[a], q + 1);
                        __builtin_nontemporal_store(sid[r0 + g], q + 2);
                    }
                }
            }
            if (want_pde) {
                if constexpr ((E & 1) == 0) {
                    typedef double dbl2 __attribute__((ext_vector_type(2)));
                    constexpr uint32_t H = E / 2, PR = 3 * H;  // 16-byte pieces per vertex / per row
                    dbl2 *dst = reinterpret_cast<dbl2 *>(P.out_pde + o * D);
#pragma unroll 1
                    for (uint32_t i = 0; i < PR * KT; i++) {
                        const uint32_t g = lane + 64 * i;
                        if (g < nr * PR) {
                            const uint32_t row = r0 + g / PR, within = g % PR, which = within / H, sub = within % H;
                            const uint32_t a = sa[row];
                            const double *src = which == 0 ? svs + a * EP + 2 * sub
                                                : which == 1 ? svb + a * EP + 2 * sub
                                                             : sv + row * EP + 2 * sub;
                            __builtin_nontemporal_store(*reinterpret_cast<const dbl2 *>(src), &dst[g]);
                        }
                    }
                } else {
                    double *dst = P.out_pde + o * D;
#pragma unroll 1
                    for (uint32_t i = 0; i < (uint32_t)(D * KT); i++) {
                        const uint32_t g = lane + 64 * i;
                        if (g < nr * D) {
                            const uint32_t row = r0 + g / D, within = g % D, which = within / E, sub = within % E;
                            const uint32_t a = sa[row];
                            const double *src = which == 0 ? svs + a * EP + sub : which == 1 ? svb + a * EP + sub : sv + row * EP + sub;
                            __builtin_nontemporal_store(*src, &dst[g]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    for (uint32_t k0 = k_lo; k0 < np && carry < TS; k0 += SP) {
        // strip: SP pairs, one lane each
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 pw = {0u, 0u, 0u, 0u};
        uint2 sbv = make_uint2(0u, 0u);
        const bool valid = lane < (unsigned)SP && k0 + lane < np;
        if (valid) {
            pw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(&pairs[(uint64_t)e0 + k0 + lane]));
            sbv = pst[(uint64_t)e0 + k0 + lane];
        }
        stamp(1, true);
        const uint32_t blk = pw.x;
        const uint64_t G = ((uint64_t)pw.w << 32) | pw.z;
        const uint32_t pcnt = (pw.y & kHubFlag) ? 0u : pw.y;  // hub pairs never reach this kernel (dispatcher)
        const uint32_t incl = wave_scan_add(pcnt);
        const uint32_t excl = incl - pcnt;
        const uint32_t C = rl32(incl, SP - 1);
        // pairs that own rows of this tile: non-empty, first row before the tile's end (the strip's first pair reaches
        // the tile by construction)
        const uint64_t rel = __ballot(valid && pcnt != 0 && carry + (int32_t)excl < TS);
        const bool mine = (rel >> lane) & 1ull;
        if (lane < (unsigned)SP) {
            cs[lane] = excl;
            sblk[lane] = blk;
            sb[lane] = sbv.y;
            ss[lane] = sbv.x;
            sG[lane] = G;
        }
        if (mine) smark[excl] = (uint8_t)(lane + 1);  // excl < TS + 62 <= NMARK for a pair that reaches the tile
        // the pairs that own rows fetch their start vertex' embedding and their row block's header (vde[b]: the line the
        // pair's first records sit in) now; both land in the strip after the record loop, so neither load is a hop of its
        // own.  Nothing loaded before is used after this point: the compiler's wait for an older load would wait for these
        // (every lane loads -- an idle lane reads block 0's header and vde[0] -- so that no value is defined on one path only:
        // hipcc carries such values around the strip loop and waits for every outstanding access at its end)
        double vsr[E], vbr[E];
        if (want_pde) {
            const double *hb = reinterpret_cast<const double *>(recs + (uint64_t)(mine ? blk : 0u) * kRowAlign);
            const double *hs = P.vde + (uint64_t)(mine ? sbv.x : 0u) * E;
#pragma unroll
            for (int k2 = 0; k2 < E; k2++) {
                vsr[k2] = hs[k2];
                vbr[k2] = hb[k2];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (rel != 0) {
            const uint32_t a_hi = 63u - (uint32_t)__clzll(rel);
            const uint32_t f_hi = a_hi + 1 < (uint32_t)SP ? cs[a_hi + 1] : C;  // one past the last record of the last pair that reaches the tile
            // every record of those pairs: all loads of the tile first, then the rows
            constexpr int RW = (int)(sizeof(Rec) / 4), VW = RW - 2 * E;  // dwords per record / in front of its vde
            uint32_t rw[NP][RW];
            uint32_t pa[NP];
            uint32_t run = 0;  // pair of the previous pass' last record
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const uint32_t f = lane + 64u * p;
                uint32_t m = smark[f];
                if (lane == 0) m = max(m, run);
                m = wave_scan_max(m);
                run = rl32(m, 63);
                // lanes past the last record re-read it (same reason as above: no load under a lane mask)
                const uint32_t fc = min(f, f_hi - 1u);
                const uint32_t ac = f < f_hi ? m - 1u : a_hi;
                pa[p] = ac;
                const uint32_t *rp = reinterpret_cast<const uint32_t *>(recs + (uint64_t)sblk[ac] * kRowAlign + 8 * E) + (fc - cs[ac]) * RW;
#pragma unroll
                for (int k2 = 0; k2 < RW; k2++) rw[p][k2] = (DIAG && (exp_flags & 2u)) ? 0u : rp[k2];
            }
            // the loads above are all in flight before the first record is looked at: the empty statement reads every loaded
            // register, so hipcc can neither sink a load into the branch that uses it nor start parking between two loads
#pragma unroll
            for (int p = 0; p < NP; p++) {
#pragma unroll
                for (int k2 = 0; k2 < RW; k2++) asm volatile("" : "+v"(rw[p][k2]));
            }
            stamp(2, true);
#pragma unroll
            for (int p = 0; p < NP; p++) {
                const uint32_t f = lane + 64u * p;
                if (f < f_hi) {
                    const uint32_t a = pa[p];
                    uint32_t id, ip;
                    if constexpr (PACKED) {
                        id = rw[p][0] & ((1u << kPackedIdBits) - 1u);
                        ip = rw[p][0] >> kPackedIdBits;
                    } else {
                        id = rw[p][0];
                        ip = rw[p][1];
                    }
                    const uint64_t below = sG[a] & ((1ull << ip) - 1ull);
                    const int32_t row = carry + (int32_t)cs[a] + (int32_t)__popcll(below);
                    if (row >= 0 && row < TS) {
                        sid[row] = id;
                        sa[row] = (uint8_t)a;
                        if (want_pde) {
                            uint32_t *q = reinterpret_cast<uint32_t *>(sv + row * EP);
#pragma unroll
                            for (int k2 = 0; k2 < 2 * E; k2++) q[k2] = rw[p][VW + k2];
                        }
                    }
                }
            }
            if (f_hi > 64u * NP) {  // wave-uniform and rare: the tile's pairs hold more records than the passes above cover
                const uint32_t f = lane + 64u * NP;
                uint32_t m = smark[f];
                if (lane == 0) m = max(m, run);
                m = wave_scan_max(m);
                if (f < f_hi) {
                    const uint32_t a = m - 1u;
                    const Rec rec = reinterpret_cast<const Rec *>(recs + (uint64_t)sblk[a] * kRowAlign + 8 * E)[f - cs[a]];
                    uint32_t id, ip;
                    if constexpr (PACKED) {
                        id = rec.idp & ((1u << kPackedIdBits) - 1u);
                        ip = rec.idp >> kPackedIdBits;
                    } else {
                        id = rec.id;
                        ip = rec.aux;
                    }
                    const uint64_t below = sG[a] & ((1ull << ip) - 1ull);
                    const int32_t row = carry + (int32_t)cs[a] + (int32_t)__popcll(below);
                    if (row >= 0 && row < TS) {
                        sid[row] = id;
                        sa[row] = (uint8_t)a;
                        if (want_pde) {
#pragma unroll
                            for (int k2 = 0; k2 < E; k2++) sv[row * EP + k2] = rec.vde[k2];
                        }
                    }
                }
            }
            if (want_pde && mine) {
#pragma unroll
                for (int k2 = 0; k2 < E; k2++) {
                    svs[lane * EP + k2] = vsr[k2];
                    svb[lane * EP + k2] = vbr[k2];
                }
            }
            const int32_t r_lo = carry < 0 ? 0 : carry, r_hi = carry + (int32_t)C > TS ? TS : carry + (int32_t)C;
            if (r_hi > r_lo) {
                flush((uint32_t)r_lo, (uint32_t)r_hi);
                stamp(3, false);
                stamp(4, true);
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        carry += (int32_t)C;
        if (k0 + SP < np && carry < TS) {  // another strip: clear the marks of this one
            if (lane < (unsigned)(NMARK / 4)) reinterpret_cast<uint32_t *>(smark)[lane] = 0u;
        }
    }
}

template <int E, bool PACKED, int KT, int SP, int WPB, bool DIAG>
__global__ __launch_bounds__(64 * WPB, (KT == 1 && E <= 2) ? 8 : 1) void k_fill_tiles(FillParams P, const uint64_t *__restrict__ tfirst,
                                                         const RankedPair *__restrict__ pairs, const uint2 *__restrict__ pst,
                                                         const char *__restrict__ recs, uint64_t tile_lo, uint64_t tile_hi,
                                                         uint64_t total_arg, uint32_t exp_flags,
                                                         unsigned long long *__restrict__ stamps)
{
    // the tile index is wave-uniform, and the compiler must know it: the table entries then come through the scalar cache, past
    // the vector memory pipeline in which this CU's loads and stores queue (stamps: 2 500 cycles for the table hop through it)
    const uint64_t t = tile_lo + (uint64_t)blockIdx.x * WPB + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_id());
    if (t >= tile_hi) return;
    const uint64_t total = total_arg != ~0ull ? total_arg : P.eoff[P.n_edges];
    fill_tile_strips<E, PACKED, KT, SP, WPB, DIAG>(P, tfirst, pairs, pst, recs, t, total, 0u, ~0u, exp_flags, stamps);
}

// Strip jobs {tile, first pair of the strip within the tile}.  A resident grid walks the list (its length is known on the
// device only).  Written by k_fill_tickets for the tiles its pipeline does not take.
template <int E, bool PACKED, int SP, int WPB>
__global__ __launch_bounds__(64 * WPB, (E <= 2) ? 8 : 1) void k_fill_tile_jobs(FillParams P, const uint64_t *__restrict__ tfirst,
                                                         const RankedPair *__restrict__ pairs, const uint2 *__restrict__ pst,
                                                         const char *__restrict__ recs, uint64_t total_arg,
                                                         const uint32_t *__restrict__ job_count, const uint2 *__restrict__ jobs)
{
    const uint32_t n_jobs = *job_count;
    const uint64_t total = total_arg != ~0ull ? total_arg : P.eoff[P.n_edges];
    const uint32_t nw = gridDim.x * WPB;
    for (uint32_t j = blockIdx.x * WPB + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_id()); j < n_jobs; j += nw) {
        const uint2 jb = jobs[j];
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)jb.x), k_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)jb.y);
        fill_tile_strips<E, PACKED, 1, SP, WPB, false>(P, tfirst, pairs, pst, recs, (uint64_t)t, total, k_lo, k_lo + SP, 0u, nullptr);
    }
}

}  // namespace gnnpe
