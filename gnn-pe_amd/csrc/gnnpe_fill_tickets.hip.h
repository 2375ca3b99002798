// gnnpe_fill_tickets.hip.h -- the third emit shape of the l = 2 enumeration (R2 emit + R5; custom.h:66-92, 546-572):
// PERSISTENT waves that take 64-row output tiles IN ORDER from ticket counters and keep three tiles in flight each.
//
//   k_fill_ranked  : one wave per start vertex, resident grid        -- stores all the time, but which rows the chip writes at
//                    one moment drifts apart (every wave walks its own start vertices), and the rate depends on the buffer
//   k_fill_tiles   : one wave per output tile, launch order, exit    -- the chip writes one moving window of the output (nearly
//                    the same rate into every buffer), but a wave lives for three DEPENDENT round trips (tile table -> pair
//                    records -> neighbour records) plus the issue and the drain of its stores
//   k_fill_tickets : a resident grid; a wave draws TICKETS (one returning atomic each, on one of `nh` heads: a single word
//                    saturates at ~88 dequeues per microsecond) = runs of `tpt` consecutive tiles, so the tiles in flight are
//                    still one moving window; and it runs the three round trips of three DIFFERENT tiles side by side:
//                        iteration i:   (the wait) tile i's records have arrived
//                                       next ticket requested when the run ends -- the oldest operation of the iteration
//                                       tile i: records -> the tile's OUTPUT IMAGE in LDS (rows assembled whole)
//                                       tile i+1: pairs have arrived -> scan, its RECORD loads issued
//                                       tile i+2: table entry has arrived -> its PAIR loads issued
//                                       tile i+3: its TABLE load issued (scalar)
//                                       tile i: image -> global memory, a linear copy (buffer stores: lanes past the tile's
//                                       rows are dropped by the range check, so the instruction stream is the same in every
//                                       iteration and the compiler's waits count the same operations)
//                    All loads of an iteration are issued BEFORE its stores and the counter of outstanding operations
//                    retires them in order, so the wait at the top of the next iteration leaves the stores in flight.
//
// Everything a record lane needs later rides in its registers from the moment its loads are issued (its pair's scan offset, G,
// start and middle vertex; vde[s] and the row block's header vde[b] are loaded by the record lane itself -- the same line as
// its record, or one line per pair), so parking a tile makes no LDS round trip and storing it makes one.
//
// The pipeline takes the tiles that fit it: at most SP pairs and at most 128 neighbour records behind them (99 % at BASELINE
// config 3 with SP = 32: a tile holds 7.3 pairs on average).  The others -- tiles behind the highest-ranked start vertices,
// whose pairs are almost all empty: up to 6 965 pairs for 64 rows -- are cut into STRIP JOBS of 64 pairs each (strips are
// independent: eoff gives the tile row of any pair) and appended to a job list that k_fill_tile_jobs (gnnpe_fill_tiles.hip.h)
// works off in a second launch: ~1e5 jobs at config 3 (0.05 ms), each one wave life of the one-shot kernel, none of them long.
//
// Record blocks and the vde table are addressed through buffer descriptors with 32-bit offsets (the dispatcher takes another
// shape when one of them reaches 4 GiB); an offset out of range reads zero, so a stage without a tile needs no safe address.
#pragma once

#include "gnnpe_fill_tiles.hip.h"

namespace gnnpe {

constexpr uint32_t kTicketHeads = 64;      // most ticket heads a launch uses (a multiple of the waves per workgroup)
constexpr uint32_t kTicketHeadWords = 32;  // heads sit on lines of their own
constexpr uint32_t kTicketCtlWords = (kTicketHeads + 1) * kTicketHeadWords;  // + the job counter (diagnostic builds: 8 stamp words behind)
constexpr uint32_t kJobStrip = 64;         // pairs per strip job = k_fill_tile_jobs' strip
constexpr int kBufWord3 = 0x00020000;      // buffer descriptor word 3: raw dwords

// N dwords at byte offset `off` of a buffer; N in {1, 2, 4}
template <int N> __device__ __forceinline__ void tk_load(uint32_t *dst, __amdgpu_buffer_rsrc_t rs, uint32_t off)
{
    if constexpr (N == 1) {
        dst[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
    } else if constexpr (N == 2) {
        typedef uint32_t v2 __attribute__((ext_vector_type(2)));
        const v2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
        dst[0] = v.x;
        dst[1] = v.y;
    } else {
        static_assert(N == 4, "tk_load: 1, 2 or 4 dwords");
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        dst[0] = v.x;
        dst[1] = v.y;
        dst[2] = v.z;
        dst[3] = v.w;
    }
}
// 2 E dwords (E doubles) in 16-byte pieces (8-byte for odd E)
template <int E> __device__ __forceinline__ void tk_load_vde(uint32_t *dst, __amdgpu_buffer_rsrc_t rs, uint32_t off)
{
    if constexpr (E % 2 == 0) {
#pragma unroll
        for (int k = 0; k < E / 2; k++) tk_load<4>(dst + 4 * k, rs, off + 16u * k);
    } else {
#pragma unroll
        for (int k = 0; k < E; k++) tk_load<2>(dst + 2 * k, rs, off + 8u * k);
    }
}

template <int E, bool PACKED, int SP, int WPB, int OCC>
__global__ __launch_bounds__(64 * WPB, OCC) void k_fill_tickets(FillParams P, const uint64_t *__restrict__ tfirst,
                                                                const RankedPair *__restrict__ pairs, const uint2 *__restrict__ pst,
                                                                const char *__restrict__ recs, uint32_t recs_bytes, uint32_t vde_bytes,
                                                                uint64_t tile_lo, uint64_t tile_hi, uint64_t total_arg, uint32_t tpt,
                                                                uint32_t nh, uint32_t exp_flags, uint32_t *__restrict__ ctl,
                                                                uint2 *__restrict__ jobs)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    constexpr int D = 3 * E;
    constexpr int TS = 64;
    constexpr int NP = 2;                 // record passes of a tile the pipeline takes (at most 128 records)
    constexpr int NMARK = 64 * NP;
    constexpr int VW = PACKED ? 1 : 2;    // dwords of a record in front of its vde
    constexpr int RB = (VW + 2 * E) * 4;  // bytes of a record
    constexpr int ROWB = D * 8;           // bytes of an output pde row
    constexpr int GR = (E % 2 == 0) ? 16 : 8;  // bytes a lane copies per store instruction
    constexpr int NST = ROWB / GR;        // store instructions for a tile's pde rows
    constexpr int IMG = TS * ROWB;        // the tile's pde rows, then its id rows (TS x 12 bytes)
    static_assert(SP <= 64 && (SP & (SP - 1)) == 0, "one lane per pair of the strip");
    static_assert(sizeof(typename RecOf<E, PACKED>::type) == RB, "record layout");
    __shared__ __attribute__((aligned(16))) u32x4 s_p4[WPB][SP];  // per pair of the strip: {scan offset, row block, s, b}
    __shared__ uint64_t s_G[WPB][SP];
    __shared__ __attribute__((aligned(4))) uint8_t s_mark[WPB][NMARK];
    __shared__ __attribute__((aligned(16))) char s_img[WPB][IMG + TS * 12];
    // (the wave's index is wave-uniform, and the compiler must know it: everything derived from the ticket -- tile, table entry,
    // output offsets, the stores' buffer descriptors -- stays in scalar registers)
    const unsigned lane = lane_id(), wv = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_id());
    const bool want_pde = P.out_pde != nullptr;
    uint8_t *const smark = s_mark[wv];
    char *const img = s_img[wv];
    auto lds_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto sfirst = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };

    const uint64_t total = total_arg != ~0ull ? total_arg : P.eoff[P.n_edges];
    const uint64_t tile_end = min(tile_hi, (total + TS - 1) / TS);  // capped launches cover the buffer's capacity, not the count
    const uint64_t n_tiles = tile_end > tile_lo ? tile_end - tile_lo : 0;
    const uint32_t n_groups = (uint32_t)((n_tiles + tpt - 1) / tpt);
    uint32_t *const job_count = ctl + kTicketHeads * kTicketHeadWords;
    const __amdgpu_buffer_rsrc_t rs_rec = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(recs), 0, (int)recs_bytes, kBufWord3);
    const __amdgpu_buffer_rsrc_t rs_vde = __builtin_amdgcn_make_buffer_rsrc(
        want_pde ? reinterpret_cast<char *>(const_cast<double *>(P.vde)) : const_cast<char *>(recs), 0, (int)(want_pde ? vde_bytes : 0u), kBufWord3);

    // ---- tickets: ticket k of head h is group (k * nh + h) = tiles tile_lo + group * tpt + [0, tpt).  A workgroup's four waves
    // own four ADJACENT heads and workgroup j owns heads 4 (j mod nh/4) ..: with blocks dealt round-robin over the XCDs that
    // is the launch-order map of k_fill_tiles (tiles 4i .. 4i+3 on XCD i mod 8), so neighbouring tiles -- which share pair
    // and record lines -- meet in one L2.  Every head has waves whatever the placement, so no head is left undrained.
    const uint32_t head = (blockIdx.x % (nh / WPB)) * WPB + wv;
    auto draw = [&]() -> uint32_t {  // returning atomic, one lane
        // (an atomic INCREMENT with wrap-around at 2^32 - 1, not an add: the compiler rewrites an add to a wave-uniform address
        // into its one-lane-adds-the-popcount form and reads the result back at once -- a wait for every outstanding operation
        // in the middle of the pipeline; it leaves the increment alone)
        uint32_t tk = 0;
        if (lane == 0) tk = __builtin_amdgcn_atomic_inc32(ctl + head * kTicketHeadWords, 0xFFFFFFFFu, __ATOMIC_RELAXED, "agent");
        return tk;
    };
    uint32_t tk_next = draw();
    asm volatile("" : "+v"(tk_next));  // the first ticket is waited for here, so that no wait inside the loop has to assume it is still in flight
    uint32_t grp = 0, pos = tpt;  // pos == tpt: the run is used up
    bool more = true;
    auto push_jobs = [&](uint32_t t, uint32_t np) {  // strips of kJobStrip pairs of tile t
        const uint32_t nj = (np + kJobStrip - 1) / kJobStrip;
        uint32_t base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(job_count, nj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        base = sfirst(base);
        for (uint32_t j = lane; j < nj; j += 64) jobs[base + j] = make_uint2(t, j * kJobStrip);
    };

    // ---- pipeline state (wave-uniform unless noted) ---------------------------------------------------------------------------
    // stage A: records in flight, then parked and stored
    bool okA = false;
    uint64_t slotA = 0;  // first output slot of the tile
    int32_t carryA = 0;  // tile row of the strip's first record (<= 0)
    uint32_t fhiA = 0, CA = 0;
    // per record lane and pass: the record, its pair's header vde[b] and vde[s], and the pair's {scan offset, G, s, b}
    uint32_t rw[NP][VW + 2 * E], hw[NP][2 * E], sw[NP][2 * E], pex[NP], pGl[NP], pGh[NP], ps[NP], pb[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        pex[p] = pGl[p] = pGh[p] = ps[p] = pb[p] = 0;
#pragma unroll
        for (int k = 0; k < VW + 2 * E; k++) rw[p][k] = 0;
#pragma unroll
        for (int k = 0; k < 2 * E; k++) hw[p][k] = sw[p][k] = 0;
    }
    // stage B: pair records in flight
    bool okB = false;
    uint64_t tB = 0;
    uint32_t npB = 1;
    int32_t carryB = 0;
    u32x4 pw = {0u, 0u, 0u, 0u};  // per lane
    uint2 sbv = make_uint2(0u, 0u);
    // stage C: table entry in flight
    bool okC = false;
    uint64_t tC = 0, tf0 = 0, tf1 = 0;

#ifdef GNNPE_DIAG
    // in-kernel stamps (exp_flags bit 4): shader clock per phase, summed over the iterations of one wave in 16
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0;
    unsigned long long *const stamps = reinterpret_cast<unsigned long long *>(ctl + kTicketCtlWords);
    const bool st_on = (exp_flags & 16u) && ((blockIdx.x * WPB + wv) & 15u) == 0;
#define GNNPE_STAMP(PH)                                                                     \
    do {                                                                                    \
        if (st_on) {                                                                        \
            unsigned long long now;                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");     \
            if ((PH) >= 0) st_acc[(PH) & 7] += now - st_prev;                               \
            st_prev = now;                                                                  \
        }                                                                                   \
    } while (0)
#else
#define GNNPE_STAMP(PH) do {} while (0)
#endif
    GNNPE_STAMP(-1);
    while (okA || okB || okC || more) {
        // ---- the wait: everything loaded for tile A (the empty statement reads every loaded register) ------------------------
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int k = 0; k < VW + 2 * E; k++) asm volatile("" : "+v"(rw[p][k]));
#pragma unroll
            for (int k = 0; k < 2 * E; k++) asm volatile("" : "+v"(hw[p][k]), "+v"(sw[p][k]));
        }
        asm volatile("" : "+v"(tk_next));  // (older than all of them: read here, the ticket costs no wait of its own below)
        GNNPE_STAMP(0);
        // ---- ticket: the one requested a run of tiles ago is older than everything just waited for; the next request is
        // this iteration's first memory operation, so the same holds for it
        if (more && pos == tpt) {
            const uint32_t g = sfirst(tk_next) * nh + head;
            if (g >= n_groups) {
                more = false;  // this head has run dry
            } else {
                grp = g;
                pos = 0;
                tk_next = draw();
            }
        }

        // ---- 1. tile A: records -> the tile's output image in LDS ---------------------------------------------------------------
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const uint32_t f = lane + 64u * p;
            if (okA && f < fhiA) {
                uint32_t id, ip;
                if constexpr (PACKED) {
                    id = rw[p][0] & ((1u << kPackedIdBits) - 1u);
                    ip = rw[p][0] >> kPackedIdBits;
                } else {
                    id = rw[p][0];
                    ip = rw[p][1];
                }
                const uint64_t G = ((uint64_t)pGh[p] << 32) | pGl[p];
                const uint64_t below = G & ((1ull << ip) - 1ull);
                const int32_t row = carryA + (int32_t)pex[p] + (int32_t)__popcll(below);
                if (row >= 0 && row < TS) {
                    uint32_t *q = reinterpret_cast<uint32_t *>(img + row * ROWB);
                    if constexpr (E % 2 == 0) {
#pragma unroll
                        for (int k = 0; k < E / 2; k++) {
                            reinterpret_cast<u32x4 *>(q)[k] = u32x4{sw[p][4 * k], sw[p][4 * k + 1], sw[p][4 * k + 2], sw[p][4 * k + 3]};
                            reinterpret_cast<u32x4 *>(q + 2 * E)[k] = u32x4{hw[p][4 * k], hw[p][4 * k + 1], hw[p][4 * k + 2], hw[p][4 * k + 3]};
                            reinterpret_cast<u32x4 *>(q + 4 * E)[k] =
                                u32x4{rw[p][VW + 4 * k], rw[p][VW + 4 * k + 1], rw[p][VW + 4 * k + 2], rw[p][VW + 4 * k + 3]};
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < E; k++) {
                            reinterpret_cast<u32x2 *>(q)[k] = u32x2{sw[p][2 * k], sw[p][2 * k + 1]};
                            reinterpret_cast<u32x2 *>(q + 2 * E)[k] = u32x2{hw[p][2 * k], hw[p][2 * k + 1]};
                            reinterpret_cast<u32x2 *>(q + 4 * E)[k] = u32x2{rw[p][VW + 2 * k], rw[p][VW + 2 * k + 1]};
                        }
                    }
                    uint32_t *qi = reinterpret_cast<uint32_t *>(img + IMG + row * 12);
                    qi[0] = ps[p];
                    qi[1] = pb[p];
                    qi[2] = id;
                }
            }
        }
        GNNPE_STAMP(1);

        // ---- 2. tile B: its pairs have arrived -> strip state, record loads issued (B becomes the next A) --------------------
        bool okN;
        uint32_t fhiN, CN;
        {
            const bool valid = okB && lane < (unsigned)SP && lane < npB;
            const uint32_t blk = pw.x;
            const uint32_t pcnt = (!valid || (pw.y & kHubFlag)) ? 0u : pw.y;  // hub pairs never reach this kernel (dispatcher)
            const uint32_t incl = wave_scan_add(pcnt);
            const uint32_t excl = incl - pcnt;
            CN = rl32(incl, 63);
            const uint64_t relN = __ballot(valid && pcnt != 0 && carryB + (int32_t)excl < TS);  // pairs that own rows of the tile
            const bool mine = (relN >> lane) & 1ull;
            if (lane < (unsigned)(NMARK / 4)) reinterpret_cast<uint32_t *>(smark)[lane] = 0u;
            if (lane < (unsigned)SP) {
                s_p4[wv][lane] = u32x4{excl, blk, sbv.x, sbv.y};
                s_G[wv][lane] = ((uint64_t)pw.w << 32) | pw.z;
            }
            // f_hi = one past the last record of the last pair that reaches the tile
            const uint32_t a_hi = relN ? 63u - (uint32_t)__clzll(relN) : 0u;
            fhiN = relN ? rl32(incl, (int)a_hi) : 0u;
            okN = okB && relN != 0 && fhiN <= (uint32_t)NMARK;
            if (okB && relN != 0 && fhiN > (uint32_t)NMARK) push_jobs((uint32_t)tB, npB);  // wave-uniform, rare: more records than two passes
            if (mine && excl < (uint32_t)NMARK) smark[excl] = (uint8_t)(lane + 1);
            lds_sync();
            uint32_t run = 0;
#pragma unroll
            for (int p = 0; p < NP; p++) {
                // record f -> its pair: every pair marked its first record's slot, a prefix maximum over the marks names the pair
                const uint32_t f = lane + 64u * p;
                uint32_t m = smark[f];
                if (lane == 0) m = max(m, run);
                m = wave_scan_max(m);
                run = rl32(m, 63);
                // lanes past the last record re-read it (no load under a lane mask); a stage without a tile reads whatever the
                // strip holds -- an offset out of range returns zero
                const uint32_t fc = min(f, fhiN - 1u);
                const uint32_t ac = (f < fhiN ? m - 1u : a_hi) & (uint32_t)(SP - 1);
                const u32x4 p4 = s_p4[wv][ac];
                const uint64_t G = s_G[wv][ac];
                pex[p] = p4.x;
                ps[p] = p4.z;
                pb[p] = p4.w;
                pGl[p] = (uint32_t)G;
                pGh[p] = (uint32_t)(G >> 32);
                uint32_t off_b = p4.y * kRowAlign;                  // the row block: header vde[b] ...
                uint32_t off_r = off_b + 8 * E + (fc - p4.x) * RB;  // ... then the records
                uint32_t off_s = p4.z * (uint32_t)(8 * E);
#ifdef GNNPE_DIAG
                if (exp_flags & 2u) off_b = off_r = 0xFFFFFF00u;  // knock-out: no record loads
#endif
                if (!okN) off_b = off_r = off_s = 0xFFFFFF00u;
                tk_load<VW>(rw[p], rs_rec, off_r);
                tk_load_vde<E>(rw[p] + VW, rs_rec, off_r + 4 * VW);
                tk_load_vde<E>(hw[p], rs_rec, off_b);
                tk_load_vde<E>(sw[p], rs_vde, off_s);
            }
        }
        GNNPE_STAMP(2);

        // ---- 3. tile C: its table entry has arrived -> pair loads issued (C becomes the next B) -------------------------------
        uint32_t npN = 1;
        int32_t carryN = 0;
        bool okBN = false;
        {
            uint32_t e0N = 0;
            if (okC) {
                e0N = sfirst((uint32_t)tf0);
                const uint32_t e1 = sfirst((uint32_t)tf1);
                carryN = -(int32_t)sfirst((uint32_t)(tf0 >> 32));
                npN = e1 - e0N + 1;
                okBN = npN <= (uint32_t)SP;
                if (!okBN) push_jobs((uint32_t)tC, npN);  // wave-uniform: a tile of many (mostly empty) pairs
            }
            const uint64_t q = (uint64_t)e0N + min(lane, npN - 1u);  // lanes past the tile's last pair re-read it
            pw = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(&pairs[q]));
            sbv = pst[q];
        }
        GNNPE_STAMP(3);

        // ---- 4. the tile after C ----------------------------------------------------------------------------------------------
        uint64_t tD = ~0ull;
        if (more && pos < tpt) {
            const uint64_t t = tile_lo + (uint64_t)grp * tpt + pos;
            pos++;
            if (t < tile_end) tD = t;  // (a run may reach past the last tile)
        }
        const bool okD = tD != ~0ull;

        // ---- 5. tile A: image -> global memory ----------------------------------------------------------------------------------
        {
            lds_sync();
            const int32_t r_lo_i = carryA < 0 ? 0 : carryA, r_hi_i = carryA + (int32_t)CA > TS ? TS : carryA + (int32_t)CA;
            uint64_t glo = slotA + (uint32_t)r_lo_i, ghi = slotA + (uint32_t)r_hi_i;
            glo = max(glo, P.begin);
            ghi = min(ghi, P.end);
            const bool any = okA && r_hi_i > r_lo_i && ghi > glo;
            const uint32_t r0 = any ? (uint32_t)(glo - slotA) : 0u;
            uint32_t nr = any ? (uint32_t)(ghi - glo) : 0u;
#ifdef GNNPE_DIAG
            if (exp_flags & 1u) nr = 0;  // knock-out: no stores (every lane out of range)
#endif
            const uint64_t o = any ? glo - P.begin : 0ull;
            // every store below is issued in every iteration: an output that is not wanted (or a tile that is not stored) is a
            // buffer of zero bytes, whose range check drops every lane -- with a branch around a store the compiler would have to
            // assume that it was not issued, and its wait for the loads issued BEFORE the stores would wait for the stores too
            {
                char *const base = want_pde ? reinterpret_cast<char *>(P.out_pde) + o * (uint64_t)ROWB : const_cast<char *>(recs);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(want_pde ? nr * (uint32_t)ROWB : 0u), kBufWord3);
                // (scalar arithmetic spelled out: the compiler otherwise folds r0 * ROWB + lane * GR into a 64-bit multiply-add whose
                // unused upper half reads whatever register sits beside the addend -- the ticket in flight -- and waits for it)
                const char *src = img + sfirst(r0 * (uint32_t)ROWB) + lane * GR;
#pragma unroll
                for (int i = 0; i < NST; i++) {
                    if constexpr (GR == 16) {
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + i * 64 * GR);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)(lane * GR + i * 64 * GR), 0, 2);  // nt; past nr rows: dropped
                    } else {
                        const u32x2 v = *reinterpret_cast<const u32x2 *>(src + i * 64 * GR);
                        __builtin_amdgcn_raw_buffer_store_b64(v, rs, (int)(lane * GR + i * 64 * GR), 0, 2);
                    }
                }
            }
            {
                char *const base = P.out_ids ? reinterpret_cast<char *>(P.out_ids) + o * 12 : const_cast<char *>(recs);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(P.out_ids ? nr * 12u : 0u), kBufWord3);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(img + IMG + sfirst(r0 * 12u)) + lane;
#pragma unroll
                for (int i = 0; i < 3; i++) __builtin_amdgcn_raw_buffer_store_b32(src[i * 64], rs, (int)(lane * 4 + i * 256), 0, 2);
            }
            lds_sync();
        }
        GNNPE_STAMP(4);

        // ---- table load of tile D, behind the stores' LDS reads (scalar loads and LDS share a counter) --------------------------
        {
            const uint64_t tq = okD ? tD : tile_lo;
            tf0 = tfirst[tq];
            tf1 = tfirst[tq + 1];
        }

        // ---- rotate ---------------------------------------------------------------------------------------------------------------
        okA = okN;
        slotA = tB * TS;
        carryA = carryB;
        fhiA = fhiN;
        CA = CN;
        okB = okBN;
        tB = tC;
        npB = npN;
        carryB = carryN;
        okC = okD;
        tC = okD ? tD : tile_lo;
    }
    (void)exp_flags;
#ifdef GNNPE_DIAG
    if (st_on && lane == 0) {
        for (int k = 0; k < 5; k++) atomicAdd(&stamps[k], st_acc[k]);
        atomicAdd(&stamps[7], 1ull);
    }
#endif
#undef GNNPE_STAMP
}

}  // namespace gnnpe
