// gnnpe_index_deep.hip.h -- R6 at l = 3: the partition's index.dat straight from the enumeration's structures (round 6).
//
// Included by gnnpe_index.hip behind the pair-major build (it uses that file's kBlockLen, index_fanout, kLeafWaves).
//
// Until round 6 an l = 3 context built its index from a tuple array: the whole enumeration emitted twice in chunks to pick the
// partition's tuples, one key and one sort entry per PATH, and the generic leaf kernel (one random 16-byte read and four vde row
// gathers per point, a byte at a time into LDS).  The l = 2 build sorts PAIRS instead of paths (gnnpe_index.hip, "pair-major");
// this is the same idea one level down.  Reference: Partition's constructor inserts path after path (custom.h:235-257); the tree
// shape is free, the file format and the consumer's constraints are not.
//
// Unit = a triple (s, b, c) x one 64-entry piece of c's adjacency row (rows of at most 64 entries: one piece).  Its points are
// the paths (s, b, c, d) with d in that piece, rank[d] > rank[s] and d != b (gnnpe_fill_deep.hip.h): a 64-bit mask, in id order
// = emission order, so the r-th set bit is the path with index son0 + r inside the partition.  Units are sorted by
//   [partition(s) | label(s) | label(b) | label(c) | z-order of the quantised vde of s, b, c]     (the 3-vertex key of k_path_keys)
// -- a tenth as many keys as paths at G(100K, 1M) -- and a leaf takes F consecutive points of the sorted units.
#pragma once

namespace gnnpe {

struct __attribute__((aligned(16))) TripX {
    uint32_t s, b, c;  // the unit's fixed vertices
    uint32_t chunk;    // first entry (index into nbrs) of the unit's piece of c's row
    uint64_t mask;     // bit p: entry chunk + p is a kept fourth vertex
    uint32_t son0;     // index inside the partition of the unit's first path (custom.h:243)
    uint32_t pad;
};
static_assert(sizeof(TripX) == 32, "two 16-byte pieces");

// per adjacency entry q = (b -> c): pieces of c's row in front of it inside b's row (cpre[q]); per row: pieces of all its
// neighbours' rows (row_units[b]).  One wave per row.  Depends on the graph only.
__global__ __launch_bounds__(256) void k_tx_row_pieces(uint32_t n, const uint8_t *__restrict__ present, const uint32_t *__restrict__ adj_start,
                                                       const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ nbrs,
                                                       uint32_t *__restrict__ cpre, uint32_t *__restrict__ row_units)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t b = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6; b < n; b += nw) {
        if (present && !present[b]) {  // (present == nullptr: every row is held)
            if (lane == 0) row_units[b] = 0u;
            continue;
        }
        const uint32_t st = adj_start[b], d = adj_deg[b];
        uint32_t running = 0;
        for (uint32_t p0 = 0; p0 < d; p0 += 64) {
            const uint32_t p = p0 + lane;
            uint32_t ch = 0;
            if (p < d) {
                const uint32_t c = nbrs[st + p];
                ch = (!present || present[c]) ? max(1u, (adj_deg[c] + 63u) / 64u) : 1u;
            }
            const uint32_t incl = wave_scan_add(ch);
            if (p < d) cpre[st + p] = running + incl - ch;
            running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) row_units[b] = running;
    }
}

// units of every (s, b) pair = pieces behind b's row; scanned into toff
__global__ void k_tx_pair_units(uint64_t n_pairs, const uint32_t *__restrict__ pnbr, const uint32_t *__restrict__ row_units,
                                uint64_t *__restrict__ out)
{
    for (uint64_t w = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; w <= n_pairs; w += (uint64_t)gridDim.x * blockDim.x)
        out[w] = w < n_pairs ? (uint64_t)row_units[pnbr[w]] : 0ull;
}

// every unit starts out empty: key = partition field n_parts (sorts behind every partition), value = its own index
template <typename KeyT>
__global__ void k_tx_fill_empty(uint64_t nu, KeyT empty_key, KeyT *__restrict__ keys, uint32_t *__restrict__ vals)
{
    for (uint64_t u = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; u < nu; u += (uint64_t)gridDim.x * blockDim.x) {
        keys[u] = empty_key;
        vals[u] = (uint32_t)u;
    }
}

// partition-local index of every start vertex' first path minus its first output slot (son0 = slot + padj[start])
__global__ void k_tx_start_parts(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted, const uint32_t *__restrict__ member,
                                 uint32_t *__restrict__ part, uint32_t *__restrict__ idx)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        part[i] = member[sorted[slab_begin + i]];
        idx[i] = (uint32_t)i;
    }
}
__global__ void k_tx_sorted_counts(uint32_t len, const uint32_t *__restrict__ poffs, const uint64_t *__restrict__ eoff,
                                   const uint32_t *__restrict__ sorted_idx, uint64_t *__restrict__ cnt)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k <= len; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        if (k < len) {
            const uint32_t i = sorted_idx[k];
            v = eoff[poffs[i + 1]] - eoff[poffs[i]];
        }
        cnt[k] = v;
    }
}
// pbase (paths of the same partition in front of the start vertex; nullptr: one partition) -> padj
__global__ void k_tx_padj(uint32_t len, const uint32_t *__restrict__ poffs, const uint64_t *__restrict__ eoff,
                          const uint64_t *__restrict__ pbase, int64_t *__restrict__ padj)
{
    const uint64_t first = eoff[0];
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        padj[i] = pbase ? (int64_t)pbase[i] - (int64_t)eoff[poffs[i]] : -(int64_t)first;
}

// One wave per work unit of the l = 3 enumeration (a pair (s, b) x 64 consecutive third vertices, gnnpe_fill_deep.hip.h): lane =
// third vertex c.  The rows of the 64 c's are walked flattened, 64 entries per step with every lane busy (k_deep3's walk): a
// step's kept mask is one ballot, lane t keeps step t's, and a lane then cuts its own row's bits out of the (at most two) steps
// they fell into.  A c whose row is longer than 64 entries is left out of the walk and taken piece by piece by the whole wave.
// Non-empty units get {record, key}; empty ones keep the fill's key.  The first path of a unit: the work unit's first output slot
// (uoff, from the count) + the kept rows of the lanes in front.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_tx_units(uint64_t n_wu, const uint32_t *__restrict__ upair, const uint64_t *__restrict__ ufirst,
                                                  const uint64_t *__restrict__ uoff, const uint32_t *__restrict__ erow,
                                                  const uint32_t *__restrict__ pnbr, const uint32_t *__restrict__ sorted, uint32_t slab_begin,
                                                  const uint32_t *__restrict__ member, const uint32_t *__restrict__ adj_start,
                                                  const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ nbrs,
                                                  const uint32_t *__restrict__ nbr_rank, const uint32_t *__restrict__ rank,
                                                  const uint32_t *__restrict__ cpre, const uint64_t *__restrict__ toff,
                                                  const int64_t *__restrict__ padj, const uint64_t *__restrict__ vkey, uint32_t e, uint32_t lb,
                                                  uint32_t zbits, TripX *__restrict__ recs, KeyT *__restrict__ keys)
{
    __shared__ uint32_t s_off[4][65], s_st[4][64];
    __shared__ uint64_t s_step[4][64];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t u = blockIdx.x * 4ull + wv;
    if (u >= n_wu) return;  // (no workgroup barrier below: the waves are independent)
    const uint32_t w = upair[u];
    const uint32_t i = erow[w], b = pnbr[w];
    const uint32_t s = sorted[slab_begin + i], thr = slab_begin + i, rb = rank[b];
    const uint32_t bst = adj_start[b], bd = adj_deg[b];
    const uint32_t k = (uint32_t)(u - ufirst[w]) * 64u + lane;
    const bool valid = k < bd;
    const uint32_t q = bst + (valid ? k : 0u);
    uint32_t c = 0, cd = 0, cst = 0;
    if (valid) {
        c = nbrs[q];
        if (c != s) {
            cd = adj_deg[c];
            cst = adj_start[c];
        }
    }
    // what the records and keys need of s, b and c: asked for here, beside the rows' descriptors, not behind the walk's fences
    const uint64_t ws = vkey[s], wb = vkey[b], wc = vkey[c];
    const uint32_t part = member[s];
    const int64_t padj_i = padj[i];
    const uint64_t uoff_u = uoff[u], toff_w = toff[w];
    const uint32_t cpre_q = valid ? cpre[q] : 0u;
    const bool hub = cd > 64u;
    const uint32_t fd = hub ? 0u : cd;
    const uint32_t incl = wave_scan_add(fd), off = incl - fd;
    s_off[wv][lane] = off;
    s_st[wv][lane] = cst;
    if (lane == 63) s_off[wv][64] = incl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t n_cand = s_off[wv][64], n_steps = (n_cand + 63u) / 64u;  // <= 64 x 64 candidates: <= 64 steps
    uint64_t mystep = 0;
    for (uint32_t st = 0; st < n_steps; st++) {
        const uint32_t qf = st * 64u + lane;
        bool keep = false;
        if (qf < n_cand) {
            uint32_t lo = 0;  // last segment whose first candidate is <= qf (skips the empty ones)
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (s_off[wv][lo + step] <= qf) lo += step;
            const uint32_t r = nbr_rank[s_st[wv][lo] + (qf - s_off[wv][lo])];
            keep = r > thr && r != rb;
        }
        const uint64_t m = __ballot(keep);
        if (lane == st) mystep = m;
    }
    s_step[wv][lane] = mystep;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint64_t mask = 0;
    if (fd) {
        const uint32_t a = off >> 6, sh = off & 63u;
        mask = s_step[wv][a] >> sh;
        if (sh && a + 1u < 64u) mask |= s_step[wv][a + 1u] << (64u - sh);
        if (fd < 64u) mask &= (1ull << fd) - 1ull;
    }
    uint32_t cnt = (uint32_t)__popcll(mask);
    // rows longer than 64 entries, first time: how many they keep
    const uint64_t hubs = __ballot(hub);
    for (uint64_t hm = hubs; hm; hm &= hm - 1ull) {
        const int h = __builtin_ctzll(hm);
        const uint32_t hst = (uint32_t)__builtin_amdgcn_readlane((int)cst, h), hd = (uint32_t)__builtin_amdgcn_readlane((int)cd, h);
        uint32_t t = 0;
        for (uint32_t p0 = 0; p0 < hd; p0 += 64) {
            const uint32_t p = p0 + lane;
            bool keep = false;
            if (p < hd) {
                const uint32_t r = nbr_rank[hst + p];
                keep = r > thr && r != rb;
            }
            t += (uint32_t)__popcll(__ballot(keep));
        }
        if ((int)lane == h) cnt = t;
    }
    const uint32_t incl_c = wave_scan_add(cnt);
    const uint64_t slot = uoff_u + (incl_c - cnt);
    const uint32_t son0 = (uint32_t)((int64_t)slot + padj_i);
    const uint64_t ubase = toff_w + cpre_q;
    // key: partition and the words of s and b are the wave's, c is the lane's
    const uint64_t lmask = (1ull << lb) - 1ull;
    const uint64_t lab = ((((((uint64_t)part << lb) | ((ws >> 32) & lmask)) << lb) | ((wb >> 32) & lmask)) << lb) | ((wc >> 32) & lmask);
    const uint64_t z = ((ws & 0xffffffffull) << (2u * e)) | ((wb & 0xffffffffull) << e) | (wc & 0xffffffffull);
    const KeyT key = (KeyT)((lab << zbits) | z);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (valid && !hub && cnt) {
        const u32x4 p0 = {s, b, c, cst};
        const u32x4 p1 = {(uint32_t)mask, (uint32_t)(mask >> 32), son0, 0u};
        u32x4 *dst = reinterpret_cast<u32x4 *>(recs + ubase);
        dst[0] = p0;
        dst[1] = p1;
        keys[ubase] = key;
    }
    // rows longer than 64 entries, second time: one unit per piece
    for (uint64_t hm = hubs; hm; hm &= hm - 1ull) {
        const int h = __builtin_ctzll(hm);
        const uint32_t hst = (uint32_t)__builtin_amdgcn_readlane((int)cst, h), hd = (uint32_t)__builtin_amdgcn_readlane((int)cd, h);
        const uint32_t hc = (uint32_t)__builtin_amdgcn_readlane((int)c, h), hson = (uint32_t)__builtin_amdgcn_readlane((int)son0, h);
        const uint64_t hub_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(ubase >> 32), h) << 32) |
                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)ubase, h);
        const uint64_t hkey_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)key, h);
        const uint64_t hkey_hi = sizeof(KeyT) == 8 ? (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)key >> 32), h) : 0u;
        const KeyT hkey = (KeyT)((hkey_hi << 32) | hkey_lo);
        uint32_t running = 0;
        for (uint32_t p0 = 0, j = 0; p0 < hd; p0 += 64, j++) {
            const uint32_t p = p0 + lane;
            bool keep = false;
            if (p < hd) {
                const uint32_t r = nbr_rank[hst + p];
                keep = r > thr && r != rb;
            }
            const uint64_t m = __ballot(keep);
            if (m && lane == 0) {
                const u32x4 p0v = {s, b, hc, hst + p0};
                const u32x4 p1v = {(uint32_t)m, (uint32_t)(m >> 32), hson + running, 0u};
                u32x4 *dst = reinterpret_cast<u32x4 *>(recs + hub_base + j);
                dst[0] = p0v;
                dst[1] = p1v;
                keys[hub_base + j] = hkey;
            }
            running += (uint32_t)__popcll(m);
        }
    }
}

// records in sorted order and their path counts (units behind the last partition: none)
template <typename KeyT>
__global__ void k_tx_gather(uint64_t nu, uint32_t n_parts, uint32_t shift, const KeyT *__restrict__ sorted_keys,
                            const uint32_t *__restrict__ order, const TripX *__restrict__ recs, TripX *__restrict__ out,
                            uint64_t *__restrict__ cnt)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k <= nu; k += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t v = 0;
        if (k < nu && ((uint64_t)sorted_keys[k] >> shift) < n_parts) {
            const uint4 *src = reinterpret_cast<const uint4 *>(recs + order[k]);
            const uint4 a = src[0], b = src[1];
            uint4 *dst = reinterpret_cast<uint4 *>(out + k);
            dst[0] = a;
            dst[1] = b;
            v = (uint64_t)(__popc(b.x) + __popc(b.y));
        }
        cnt[k] = v;
    }
}

// position of the r-th set bit (r counted from 0; r < popcount(m))
__device__ __forceinline__ uint32_t select_bit(uint64_t m, uint32_t r)
{
    uint32_t base = 0, w = (uint32_t)m;
    const uint32_t c0 = (uint32_t)__popc(w);
    if (r >= c0) {
        r -= c0;
        w = (uint32_t)(m >> 32);
        base = 32;
    }
#pragma unroll
    for (uint32_t width = 16; width > 0; width >>= 1) {
        const uint32_t c = (uint32_t)__popc(w & ((1u << width) - 1u));
        if (r >= c) {
            r -= c;
            w >>= width;
            base += width;
        }
    }
    return base;
}

// One wave per leaf, workgroups in launch order, exit.  Leaf j holds points [j F, j F + ne) of the partition's sorted units.
// Lane t < ne is point j F + t: the units that start inside the leaf are marked in LDS, a ballot of the marks turns "my unit" into
// a popcount; the unit's record names s, b, c and the piece of c's row, the r-th set bit of its mask the fourth vertex.  Then the
// ne x 4E coordinates are spread over ALL lanes (at E = 8 a leaf has six entries of 32 doubles: entry lanes alone would leave
// nine tenths of the wave idle): double t of the leaf = entry t / D, dimension t % D -> one 8-byte load from the vertex table and
// four dwords of the window (bounces lo = hi, custom.h:244-248).  Window layout, MBR columns and the shifted store of the used
// prefix are k_pack_leaves_paths' and k_pack_leaves_pairs'.
template <int E>
__global__ __launch_bounds__(64 * kLeafWaves) void k_tx_leaves(uint64_t n_pts, uint64_t n_leaves, uint64_t r1, uint64_t pts0,
                                                               const uint64_t *__restrict__ pref, const uint32_t *__restrict__ first,
                                                               const TripX *__restrict__ units, const uint32_t *__restrict__ nbrs,
                                                               const double *__restrict__ vde, char *__restrict__ image,
                                                               double *__restrict__ node_mbr)
{
    constexpr int D = 4 * E, F = (int)index_fanout(D), kEnt = 4 * D + 1;
    constexpr int kStoreU4 = (5 + F * (16 * D + 4) + 15) / 16;  // 16-byte pieces of a leaf's used prefix
    constexpr int kWin = 4 * kStoreU4 + 4;                      // the window ends where the stored prefix ends (+ the dword its last piece shifts in)
    constexpr int kRounds = (F * D + 63) / 64;
    constexpr int kVid = F * 4 > 64 ? F * 4 : 64;  // the entries' vertex ids; the unit-start flags of the first phase live in the same words
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kLeafWaves][kWin];
    __shared__ uint32_t s_vid[kLeafWaves][kVid];
    uint32_t(*s_flag)[kVid] = s_vid;  // (read by the ballot below before any lane writes an id: one wave, LDS in program order)
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t j = blockIdx.x * (uint64_t)kLeafWaves + wv;
    if (j >= n_leaves) return;
    uint32_t *w = s_win[wv];
    const uint64_t q0 = j * (uint64_t)F;
    const uint32_t ne = (uint32_t)min((uint64_t)F, n_pts - q0);
    const uint64_t u0 = first[j], u_last = j + 1 < n_leaves ? (uint64_t)first[j + 1] : r1;  // the leaf's units: u0 .. u_last
    // where the units from u0 on start, relative to the leaf's first point (u0 itself: <= 0)
    const uint64_t ut = min(min(u0 + lane, u_last + 1), r1);  // (the lanes behind the leaf's last unit re-read the one after it: not flagged)
    // lane t holds unit u0 + t (record and first point) IN REGISTERS, both loads in flight together; a point lane fetches its unit's
    // fields from that lane through the crossbar below -- one memory round trip less than loading the record once the unit is known
    const uint4 *usrc = reinterpret_cast<const uint4 *>(units + min(ut, r1 - 1));
    const uint4 ua = usrc[0], ub = usrc[1];
    const int64_t rel64 = (int64_t)(pref[ut] - pts0) - (int64_t)q0;
    const int32_t rel = (int32_t)max((int64_t)-0x40000000, min((int64_t)0x40000000, rel64));
    s_flag[wv][lane] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane >= 1u && u0 + lane < r1 && rel >= 1 && rel < (int32_t)ne) s_flag[wv][rel] = 1u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint64_t starts = __ballot(s_flag[wv][lane] != 0u);
    const uint32_t ui = (uint32_t)__popcll(starts & ((2ull << lane) - 1ull));
    const int32_t rel_u = __shfl(rel, (int)ui, 64);
    uint4 a, b;
    a.x = (uint32_t)__shfl((int)ua.x, (int)ui, 64);
    a.y = (uint32_t)__shfl((int)ua.y, (int)ui, 64);
    a.z = (uint32_t)__shfl((int)ua.z, (int)ui, 64);
    a.w = (uint32_t)__shfl((int)ua.w, (int)ui, 64);
    b.x = (uint32_t)__shfl((int)ub.x, (int)ui, 64);
    b.y = (uint32_t)__shfl((int)ub.y, (int)ui, 64);
    b.z = (uint32_t)__shfl((int)ub.z, (int)ui, 64);
    if (lane < ne) {
        const uint32_t r = (uint32_t)((int32_t)lane - rel_u);
        const uint64_t mask = ((uint64_t)b.y << 32) | b.x;
        const uint32_t d = nbrs[a.w + select_bit(mask, r)];
        uint32_t *v = s_vid[wv] + lane * 4u;
        v[0] = a.x;
        v[1] = a.y;
        v[2] = a.z;
        v[3] = d;
        w[2 + lane * kEnt + 4 * D] = b.z + r;  // the path's index inside the partition (custom.h:243)
    }
    if (lane == 0) {
        w[0] = 0u;  // level 0 = leaf (byte 3 of the window)
        w[1] = ne;
    }
    for (uint32_t t = 2u + ne * kEnt + lane; t < (uint32_t)(4 * kStoreU4 + 1); t += 64) w[t] = 0u;  // (a last leaf that is not full)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // coordinates: all loads of the leaf first, then the window
    const uint32_t n_dbl = ne * (uint32_t)D;
    double x[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t t = min(lane + 64u * r, n_dbl - 1u);
        const uint32_t ent = t / (uint32_t)D, dim = t % (uint32_t)D;
        const uint32_t vid = s_vid[wv][ent * 4u + dim / (uint32_t)E];
        x[r] = vde[(uint64_t)vid * E + dim % (uint32_t)E];
    }
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t t = lane + 64u * r;
        if (t < n_dbl) {
            const uint32_t ent = t / (uint32_t)D, dim = t % (uint32_t)D;
            const uint64_t bits64 = (uint64_t)__double_as_longlong(x[r]);
            uint32_t *o = w + 2 + ent * kEnt + 4 * dim;
            o[0] = (uint32_t)bits64;  // bounces[2k]   (custom.h:246)
            o[1] = (uint32_t)(bits64 >> 32);
            o[2] = (uint32_t)bits64;  // bounces[2k+1] (custom.h:247)
            o[3] = (uint32_t)(bits64 >> 32);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the node's box for the level above: lane k scans dimension k of the assembled entries
    if (lane < (unsigned)D) {
        double lo = 1e300, hi = -1e300;
        for (uint32_t i = 0; i < ne; i++) {
            const uint32_t *p = w + 2 + i * kEnt + 4 * lane;
            const double v = __longlong_as_double((long long)(((uint64_t)p[1] << 32) | p[0]));
            lo = fmin(lo, v);
            hi = fmax(hi, v);
        }
        node_mbr[(j * D + lane) * 2] = lo;
        node_mbr[(j * D + lane) * 2 + 1] = hi;
    }
    // node j -> file block j + 1: the used prefix, 16 bytes per lane per round, past the caches
    uint4 *dst = reinterpret_cast<uint4 *>(image + (j + 1) * (uint64_t)kBlockLen);
#pragma unroll
    for (int r = 0; r < (kStoreU4 + 63) / 64; r++) {
        const int cidx = (int)lane + 64 * r;
        if (cidx < kStoreU4) {
            const uint4 lo4 = *reinterpret_cast<const uint4 *>(w + 4 * cidx);
            const uint32_t nx = w[4 * cidx + 4];
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 o;
            o.x = (lo4.x >> 24) | (lo4.y << 8);
            o.y = (lo4.y >> 24) | (lo4.z << 8);
            o.z = (lo4.z >> 24) | (lo4.w << 8);
            o.w = (lo4.w >> 24) | (nx << 8);
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst + cidx));
        }
    }
}

// Inner node j of a level for the wide entries of l = 3 (D = 4E: at E = 8 a node holds six entries of 516 bytes).  k_pack_inner gives a
// lane a whole child entry -- six busy lanes of 64 and a DPP reduction per dimension: 3.4 ms for the 0.74 M level-1 nodes of a 22 GB
// image, three quarters of the leaf kernel's time.  Here the node's ne x D (lo, hi) pairs are spread over all lanes (one 16-byte load and
// four window dwords each), the node's own box is a column scan of the window like a leaf's, and only the used prefix is stored.
// One wave per node, workgroups in launch order.
template <int E>
__global__ __launch_bounds__(64 * kLeafWaves) void k_tx_inner(uint64_t n_nodes, uint64_t n_child, uint64_t child0, uint64_t node0, int level,
                                                              const double *__restrict__ child_mbr, char *__restrict__ image,
                                                              double *__restrict__ node_mbr)
{
    constexpr int D = 4 * E, F = (int)index_fanout(D), kEnt = 4 * D + 1;
    constexpr int kStoreU4 = (5 + F * (16 * D + 4) + 15) / 16;
    constexpr int kWin = 4 * kStoreU4 + 4;
    constexpr int kRounds = (F * D + 63) / 64;
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kLeafWaves][kWin];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t j = blockIdx.x * (uint64_t)kLeafWaves + wv;
    if (j >= n_nodes) return;
    uint32_t *w = s_win[wv];
    const uint64_t c0 = j * (uint64_t)F;
    const uint32_t ne = (uint32_t)min((uint64_t)F, n_child - c0);
    const uint32_t n_pair = ne * (uint32_t)D;
    const uint4 *src = reinterpret_cast<const uint4 *>(child_mbr + c0 * 2 * D);  // (lo, hi) pairs of the node's children, consecutive
    uint4 v[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) v[r] = src[min(lane + 64u * r, n_pair - 1u)];
    if (lane == 0) {
        w[0] = (uint32_t)level << 24;  // byte 3 of the window = byte 0 of the block
        w[1] = ne;
    }
    if (lane < ne) w[2 + lane * kEnt + 4 * D] = (uint32_t)(child0 + c0 + lane);  // the child's block id
    for (uint32_t t = 2u + ne * kEnt + lane; t < (uint32_t)(4 * kStoreU4 + 1); t += 64) w[t] = 0u;
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t t = lane + 64u * r;
        if (t < n_pair) {
            uint32_t *o = w + 2 + (t / (uint32_t)D) * kEnt + 4 * (t % (uint32_t)D);
            o[0] = v[r].x;
            o[1] = v[r].y;
            o[2] = v[r].z;
            o[3] = v[r].w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < (unsigned)D) {
        double lo = 1e300, hi = -1e300;
        for (uint32_t i = 0; i < ne; i++) {
            const uint32_t *p = w + 2 + i * kEnt + 4 * lane;
            lo = fmin(lo, __longlong_as_double((long long)(((uint64_t)p[1] << 32) | p[0])));
            hi = fmax(hi, __longlong_as_double((long long)(((uint64_t)p[3] << 32) | p[2])));
        }
        node_mbr[(j * D + lane) * 2] = lo;
        node_mbr[(j * D + lane) * 2 + 1] = hi;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(image + (node0 + j + 1) * (uint64_t)kBlockLen);
#pragma unroll
    for (int r = 0; r < (kStoreU4 + 63) / 64; r++) {
        const int cidx = (int)lane + 64 * r;
        if (cidx < kStoreU4) {
            const uint4 lo4 = *reinterpret_cast<const uint4 *>(w + 4 * cidx);
            const uint32_t nx = w[4 * cidx + 4];
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 o;
            o.x = (lo4.x >> 24) | (lo4.y << 8);
            o.y = (lo4.y >> 24) | (lo4.z << 8);
            o.z = (lo4.z >> 24) | (lo4.w << 8);
            o.w = (lo4.w >> 24) | (nx << 8);
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(dst + cidx));
        }
    }
}

}  // namespace gnnpe
