// gnnpe_kernels.hip.h -- hand-written gfx950 kernels of the offline path (included by gnnpe_engine.hip).
//
// All kernels are HBM/L2-bound integer + fp64 gather/scatter work (the reference has no
// dense contraction, so there is nothing for MFMA).  Wave = 64 lanes; blocks are 256 threads.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace gnnpe {

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned wave_id() { return threadIdx.x >> 6; }

// ------------------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------------------
__global__ void k_invert_order(uint32_t n, const uint32_t *__restrict__ sorted, uint32_t *__restrict__ rank)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        rank[sorted[i]] = (uint32_t)i;
}

__global__ void k_offsets_to_start_deg(uint32_t n, const uint32_t *__restrict__ offs, uint32_t *__restrict__ start,
                                       uint32_t *__restrict__ deg, uint8_t *__restrict__ present)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t a = offs[i], b = offs[i + 1];
        start[i] = a;
        deg[i] = b - a;
        present[i] = 1;
    }
}

// rows loaded through gnnpe_load_rows / gnnpe_rows_append: row k (vertex ids[k]) occupies
// [base + roff[k], base + roff[k+1]) of the neighbour buffer.
__global__ void k_install_rows(uint64_t n_rows, const uint32_t *__restrict__ ids, const uint64_t *__restrict__ roff,
                               uint64_t base, uint32_t *__restrict__ start, uint32_t *__restrict__ deg,
                               uint8_t *__restrict__ present)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_rows; k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = ids[k];
        start[v] = (uint32_t)(base + roff[k]);
        deg[v] = (uint32_t)(roff[k + 1] - roff[k]);
        present[v] = 1;
    }
}

__global__ void k_gather_u32(uint64_t cnt, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ table,
                             uint32_t *__restrict__ out)
{
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < cnt; q += (uint64_t)gridDim.x * blockDim.x)
        out[q] = table[idx[q]];
}

// slab row degrees: pdeg[i] = deg(sorted[slab_begin + i]); pdeg[len] = 0 (scan sentinel)
__global__ void k_slab_degrees(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted,
                               const uint32_t *__restrict__ deg, uint32_t *__restrict__ pdeg)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= len; i += (uint64_t)gridDim.x * blockDim.x)
        pdeg[i] = i < len ? deg[sorted[slab_begin + i]] : 0u;
}

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
// R4 gen_vde (custom.h:513-544).  One thread per held row; the block's adjacency range is contiguous
// in the neighbour buffer, so it is streamed once with coalesced loads, turned into neighbour LABELS
// in LDS, and every thread then adds its own row's features in ascending-neighbour order -- the same
// operation order as the reference loop (:527-534), hence bit-identical fp64 results.
// The label table (|Sigma| x e doubles) also sits in LDS when it fits.
// ------------------------------------------------------------------------------------------------
constexpr int kVdeStage = 6144;        // neighbour labels staged per pass (24 KiB)
constexpr int kVdeTabMax = 4096;       // table doubles kept in LDS (32 KiB)

template <int E>
__global__ __launch_bounds__(256) void k_vde(uint32_t n_rows, const uint32_t *__restrict__ rows,
                                             const uint32_t *__restrict__ adj_start,
                                             const uint32_t *__restrict__ adj_deg,
                                             const uint32_t *__restrict__ nbrs,
                                             const uint32_t *__restrict__ labels,
                                             const double *__restrict__ xtab, uint32_t n_labels, uint32_t e_rt,
                                             double *__restrict__ nx, double *__restrict__ vde)
{
    const int e = E ? E : (int)e_rt;
    __shared__ uint32_t s_lab[kVdeStage];
    __shared__ double s_tab[kVdeTabMax];
    const bool tab_in_lds = (uint64_t)n_labels * e <= (uint64_t)kVdeTabMax;
    if (tab_in_lds)
        for (uint32_t i = threadIdx.x; i < n_labels * e; i += blockDim.x) s_tab[i] = xtab[i];

    const uint32_t r0 = blockIdx.x * blockDim.x;
    const uint32_t r = r0 + threadIdx.x;
    const uint32_t rlast = min(r0 + blockDim.x, n_rows) - 1;
    const uint32_t v0 = rows ? rows[r0] : r0;
    const uint32_t vl = rows ? rows[rlast] : rlast;
    const uint32_t q_begin = adj_start[v0];
    const uint32_t q_end = adj_start[vl] + adj_deg[vl];

    uint32_t v = 0, my_b = 0, my_e = 0;
    if (r < n_rows) {
        v = rows ? rows[r] : r;
        my_b = adj_start[v];
        my_e = my_b + adj_deg[v];
    }
    double acc[E ? E : 32];
#pragma unroll
    for (int k = 0; k < (E ? E : 32); k++) acc[k] = 0.0;

    __syncthreads();
    for (uint32_t c0 = q_begin; c0 < q_end; c0 += kVdeStage) {
        const uint32_t c1 = min(c0 + (uint32_t)kVdeStage, q_end);
        for (uint32_t q = c0 + threadIdx.x; q < c1; q += blockDim.x) s_lab[q - c0] = labels[nbrs[q]];
        __syncthreads();
        const uint32_t lo = max(my_b, c0), hi = min(my_e, c1);
        for (uint32_t q = lo; q < hi; q++) {
            const uint32_t lb = s_lab[q - c0];
            if (E) {
#pragma unroll
                for (int k = 0; k < E; k++) acc[k] += tab_in_lds ? s_tab[lb * E + k] : xtab[(uint64_t)lb * E + k];
            } else {
                for (int k = 0; k < e; k++) acc[k] += tab_in_lds ? s_tab[lb * e + k] : xtab[(uint64_t)lb * e + k];
            }
        }
        __syncthreads();
    }
    if (r < n_rows) {
        const uint32_t lv = labels[v];
        for (int k = 0; k < e; k++) {
            const double xv = tab_in_lds ? s_tab[lv * e + k] : xtab[(uint64_t)lv * e + k];
            nx[(uint64_t)v * e + k] = acc[k];
            vde[(uint64_t)v * e + k] = xv + acc[k];
        }
    }
}

// x[v] = table[label[v]] for every vertex (labels are replicated on every rank)
__global__ void k_x_from_labels(uint32_t n, uint32_t e, const uint32_t *__restrict__ labels,
                                const double *__restrict__ xtab, double *__restrict__ x)
{
    const uint64_t tot = (uint64_t)n * e;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = (uint32_t)(i / e), k = (uint32_t)(i % e);
        x[i] = xtab[(uint64_t)labels[v] * e + k];
    }
}

__global__ void k_vde_pack(uint32_t begin, uint32_t end, uint32_t e, const uint32_t *__restrict__ sorted,
                           const double *__restrict__ vde, double *__restrict__ buf)
{
    const uint64_t tot = (uint64_t)(end - begin) * e;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t row = (uint32_t)(i / e), k = (uint32_t)(i % e);
        buf[i] = vde[(uint64_t)sorted[begin + row] * e + k];
    }
}

__global__ void k_vde_unpack(uint32_t begin, uint32_t end, uint32_t e, const uint32_t *__restrict__ sorted,
                             const double *__restrict__ buf, double *__restrict__ vde)
{
    const uint64_t tot = (uint64_t)(end - begin) * e;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t row = (uint32_t)(i / e), k = (uint32_t)(i % e);
        vde[(uint64_t)sorted[begin + row] * e + k] = buf[i];
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

__global__ void k_per_start_counts(uint32_t len, const uint32_t *__restrict__ poffs,
                                   const uint64_t *__restrict__ eoff, uint64_t *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = eoff[poffs[i + 1]] - eoff[poffs[i]];
}

// tile k starts inside pair e iff eoff[e] <= k*T < eoff[e+1]
__global__ void k_tile_edges(uint64_t n_edges, uint32_t T, const uint64_t *__restrict__ eoff,
                             uint32_t *__restrict__ tile_edge)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n_edges; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = eoff[e], b = eoff[e + 1];
        if (b > a) {
            for (uint64_t k = (a + T - 1) / T; k * T < b; k++) tile_edge[k] = (uint32_t)e;
        }
    }
}

struct FillParams {
    const uint32_t *erow, *pnbr, *adj_start, *adj_deg, *nbrs, *nbr_rank, *sorted, *tile_edge, *member;
    const uint64_t *eoff;
    const double *vde, *x, *nbr_vde;  // nbr_vde[q] = vde[nbrs[q]] (e doubles per adjacency entry)
    uint64_t n_edges, begin, end, total;
    uint32_t slab_begin, e, tile0;
    uint32_t *out_ids;
    double *out_pde, *out_pdl;
    uint32_t *out_part;
};

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

// ------------------------------------------------------------------------------------------------
// R2 + R5 fill, variant 0 (default): OUTPUT-TILED.  Block k owns paths [k*T, (k+1)*T): it walks the
// (s, b) pairs that cover that range (tile_edge[k] .. tile_edge[k+1]), flattens their candidate
// lists across all 256 lanes (full lane use whatever the degrees are), keeps candidates with
// rank[c] > rank[s], turns the keep flags into output slots with a block-wide prefix count (the
// scanned pair offsets make slots contiguous across pairs), stages ids + embeddings in LDS, and
// finally streams the whole tile to HBM with 16-byte-per-lane stores -- every output byte is
// written exactly once, in full lines.  All reads in the candidate loop are contiguous segments
// (neighbour ids, their ranks, their embeddings); the start / middle embeddings are fetched once
// per pair into LDS.
// ------------------------------------------------------------------------------------------------
template <int E, int T, int CH, bool PDL>
__global__ __launch_bounds__(256) void k_fill_tiled(FillParams P)
{
    constexpr int D = 3 * E;
    static_assert(CH <= 256 && (CH & (CH - 1)) == 0, "chunk must be a power of two <= block size");
    __shared__ __attribute__((aligned(16))) uint32_t s_ids[T * 3];
    __shared__ __attribute__((aligned(16))) double s_pde[T * D];
    __shared__ __attribute__((aligned(16))) double s_pdl[PDL ? T * D : 1];
    __shared__ __attribute__((aligned(16))) double s_vs[CH * E], s_vb[CH * E];
    __shared__ uint32_t s_thr[CH], s_s[CH], s_b[CH], s_st[CH];
    __shared__ uint32_t s_cstart[CH + 1];
    __shared__ uint32_t s_wsum[2][4];

    const unsigned tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint64_t k = (uint64_t)P.tile0 + blockIdx.x;
    const uint64_t tlo = k * T;
    const uint64_t lo = max(tlo, P.begin), hi = min(tlo + T, P.end);
    if (lo >= hi) return;
    const int64_t olo = (int64_t)(lo - tlo), ohi = (int64_t)(hi - tlo);
    const bool want_pde = P.out_pde != nullptr;

    uint64_t e0 = P.tile_edge[k];
    // last pair that can contribute to this tile: the one holding the next tile's first path
    const uint64_t e_last = (tlo + T < P.total) ? (uint64_t)P.tile_edge[k + 1] : P.n_edges - 1;
    // slot (relative to the tile) of the first kept candidate of pair e0; <= 0
    const int64_t pos_base = (int64_t)P.eoff[e0] - (int64_t)tlo;
    int64_t running = 0;
    int parity = 0;
    bool done = false;

    while (!done && e0 <= e_last) {
        __syncthreads();  // previous chunk's rounds are done reading the metadata arrays
        // ---- stage metadata of up to CH pairs, scan their degrees ----
        const uint64_t ee = e0 + tid;
        uint32_t d = 0;
        if (tid < (unsigned)CH && ee <= e_last) {
            const uint32_t i = P.erow[ee], b = P.pnbr[ee];
            const uint32_t s = P.sorted[P.slab_begin + i];
            s_thr[tid] = P.slab_begin + i;
            s_s[tid] = s;
            s_b[tid] = b;
            s_st[tid] = P.adj_start[b];
            d = P.adj_deg[b];
            if (want_pde) {
#pragma unroll
                for (int kk = 0; kk < E; kk++) {
                    s_vs[tid * E + kk] = P.vde[(uint64_t)s * E + kk];
                    s_vb[tid * E + kk] = P.vde[(uint64_t)b * E + kk];
                }
            }
        }
        // inclusive wave scan of d
        uint32_t incl = d;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= (unsigned)off) incl += t;
        }
        if (lane == 63) s_wsum[parity][wv] = incl;
        __syncthreads();
        uint32_t wbase = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++)
            if ((unsigned)w2 < wv) wbase += s_wsum[parity][w2];
        const uint32_t C = s_wsum[parity][0] + s_wsum[parity][1] + s_wsum[parity][2] + s_wsum[parity][3];
        if (tid < (unsigned)CH) s_cstart[tid] = wbase + incl - d;
        if (tid == 0) s_cstart[CH] = C;
        parity ^= 1;
        __syncthreads();

        // ---- flattened candidates of this chunk, 256 per round ----
        for (uint32_t q0 = 0; q0 < C; q0 += 256) {
            const uint32_t q = q0 + tid;
            bool keep = false;
            uint32_t j = 0, c = 0, idx = 0;
            if (q < C) {
                // largest j with cstart[j] <= q
                uint32_t a = 0, bnd = CH;
                while (bnd - a > 1) {
                    const uint32_t mid = (a + bnd) >> 1;
                    if (s_cstart[mid] <= q) a = mid; else bnd = mid;
                }
                j = a;
                idx = s_st[j] + (q - s_cstart[j]);
                c = P.nbrs[idx];
                keep = P.nbr_rank[idx] > s_thr[j];
            }
            const uint64_t mask = __ballot(keep);
            if (lane == 0) s_wsum[parity][wv] = (uint32_t)__popcll(mask);
            __syncthreads();
            uint32_t before = 0, tot = 0;
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) {
                const uint32_t t = s_wsum[parity][w2];
                if ((unsigned)w2 < wv) before += t;
                tot += t;
            }
            parity ^= 1;
            const int64_t slot = pos_base + running + before + (int64_t)__popcll(mask & lt);
            if (keep && slot >= olo && slot < ohi) {
                const uint32_t s = s_s[j], b = s_b[j];
                s_ids[slot * 3 + 0] = s;
                s_ids[slot * 3 + 1] = b;
                s_ids[slot * 3 + 2] = c;
                if (want_pde) {
#pragma unroll
                    for (int kk = 0; kk < E; kk++) {
                        s_pde[slot * D + kk] = s_vs[j * E + kk];
                        s_pde[slot * D + E + kk] = s_vb[j * E + kk];
                        s_pde[slot * D + 2 * E + kk] = P.nbr_vde[(uint64_t)idx * E + kk];
                    }
                }
                if (PDL && P.out_pdl) {
#pragma unroll
                    for (int kk = 0; kk < E; kk++) {
                        s_pdl[slot * D + kk] = P.x[(uint64_t)s * E + kk];
                        s_pdl[slot * D + E + kk] = P.x[(uint64_t)b * E + kk];
                        s_pdl[slot * D + 2 * E + kk] = P.x[(uint64_t)c * E + kk];
                    }
                }
            }
            running += tot;
            if (pos_base + running >= ohi) {  // block-uniform: the tile is complete
                done = true;
                break;
            }
        }
        e0 += CH;
    }
    __syncthreads();

    // ---- stream the staged tile to HBM ----
    const uint64_t obase = lo - P.begin;  // first output row of this block
    const uint32_t nout = (uint32_t)(hi - lo);
    const bool full = (nout == (uint32_t)T) && (olo == 0) && ((obase & 3ull) == 0);
    if (P.out_ids) {
        uint32_t *dst = P.out_ids + obase * 3;
        if (full && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
            const uint4 *src4 = reinterpret_cast<const uint4 *>(s_ids);
            uint4 *dst4 = reinterpret_cast<uint4 *>(dst);
            for (uint32_t i = tid; i < (uint32_t)(T * 3 / 4); i += 256) dst4[i] = src4[i];
        } else {
            for (uint32_t i = tid; i < nout * 3; i += 256) dst[i] = s_ids[olo * 3 + i];
        }
    }
    if (want_pde) {
        double *dst = P.out_pde + obase * D;
        if (full && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
            const double2 *src2 = reinterpret_cast<const double2 *>(s_pde);
            double2 *dst2 = reinterpret_cast<double2 *>(dst);
            for (uint32_t i = tid; i < (uint32_t)(T * D / 2); i += 256) dst2[i] = src2[i];
        } else {
            for (uint32_t i = tid; i < nout * D; i += 256) dst[i] = s_pde[olo * D + i];
        }
    }
    if (PDL && P.out_pdl) {
        double *dst = P.out_pdl + obase * D;
        if (full && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
            const double2 *src2 = reinterpret_cast<const double2 *>(s_pdl);
            double2 *dst2 = reinterpret_cast<double2 *>(dst);
            for (uint32_t i = tid; i < (uint32_t)(T * D / 2); i += 256) dst2[i] = src2[i];
        } else {
            for (uint32_t i = tid; i < nout * D; i += 256) dst[i] = s_pdl[olo * D + i];
        }
    }
    if (P.out_part) {
        for (uint32_t i = tid; i < nout; i += 256) P.out_part[obase + i] = P.member[s_ids[(olo + i) * 3]];
    }
}

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
constexpr uint32_t kNoEdge = 0xFFFFFFFFu;
constexpr uint64_t kNoOff = ~0ull;

__device__ __forceinline__ uint32_t rl32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint64_t rl64(uint64_t v, int l)
{
    return ((uint64_t)rl32((uint32_t)(v >> 32), l) << 32) | rl32((uint32_t)v, l);
}
__device__ __forceinline__ double rlf64(double v, int l)
{
    return __hiloint2double((int)rl32((uint32_t)__double2hiint(v), l), (int)rl32((uint32_t)__double2loint(v), l));
}

// revpos[q] for adjacency entry q = (b -> u): position of b inside N(u), or kNoEdge when row u is not
// an OWNED row of this device (only owned rows start paths here).  Depends on the graph only -- not on
// the processing order -- so it is built when rows are loaded / appended, like the loader's sort of
// the adjacency lists (graph.cpp:231-233).  16 lanes per row, one binary search per entry.
__global__ void k_revpos(uint32_t n_rows, const uint32_t *__restrict__ rows, const uint8_t *__restrict__ owned,
                         const uint32_t *__restrict__ adj_start, const uint32_t *__restrict__ adj_deg,
                         const uint32_t *__restrict__ nbrs, uint32_t *__restrict__ revpos,
                         uint32_t *__restrict__ nbr_row)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; g < n_rows; g += ng) {
        const uint32_t b = rows ? rows[g] : (uint32_t)g;
        const uint32_t st = adj_start[b], d = adj_deg[b];
        for (uint32_t i = sub; i < d; i += 16) {
            const uint32_t u = nbrs[st + i];
            uint32_t r = kNoEdge;
            if (owned[u]) {
                const uint32_t lo0 = adj_start[u], du = adj_deg[u];
                uint32_t lo = 0, hi = du;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (nbrs[lo0 + mid] < b) lo = mid + 1; else hi = mid;
                }
                if (lo < du && nbrs[lo0 + lo] == b) r = lo;
            }
            revpos[st + i] = r;
            nbr_row[st + i] = b;  // row of every adjacency entry (lets kernels run one thread per entry)
        }
    }
}

// emission index of the directed pair (s = u, b) for adjacency entry q = (b -> u), or kNoEdge
__device__ __forceinline__ uint32_t pair_index(uint32_t revpos, uint32_t rank_u, uint32_t slab_begin, uint32_t slab_end,
                                               const uint32_t *__restrict__ poffs)
{
    return (revpos != kNoEdge && rank_u >= slab_begin && rank_u < slab_end) ? poffs[rank_u - slab_begin] + revpos
                                                                            : kNoEdge;
}

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

struct __attribute__((packed, aligned(4))) Triple {
    uint32_t s, b, c;
};

template <int E, class PT>
__device__ __forceinline__ void emit_path(const PT &P, uint64_t pos, uint32_t s, uint32_t b, uint32_t c,
                                          const double *vs, const double *vb, const double *vc)
{
    constexpr int D = 3 * E;
    const uint64_t o = pos - P.begin;
    if (P.out_ids) {
        Triple t = {s, b, c};
        *reinterpret_cast<Triple *>(P.out_ids + o * 3) = t;
    }
    if (P.out_pde) {
        double *dst = P.out_pde + o * D;
        if ((E & 1) == 0) {
            double2 *d2 = reinterpret_cast<double2 *>(dst);
#pragma unroll
            for (int k = 0; k < E / 2; k++) {
                d2[k] = make_double2(vs[2 * k], vs[2 * k + 1]);
                d2[E / 2 + k] = make_double2(vb[2 * k], vb[2 * k + 1]);
                d2[E + k] = make_double2(vc[2 * k], vc[2 * k + 1]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < E; k++) {
                dst[k] = vs[k];
                dst[E + k] = vb[k];
                dst[2 * E + k] = vc[k];
            }
        }
    }
    if (P.out_pdl) {
#pragma unroll
        for (int k = 0; k < E; k++) {
            P.out_pdl[o * D + k] = P.x[(uint64_t)s * E + k];
            P.out_pdl[o * D + E + k] = P.x[(uint64_t)b * E + k];
            P.out_pdl[o * D + 2 * E + k] = P.x[(uint64_t)c * E + k];
        }
    }
    if (P.out_part) P.out_part[o] = P.member[s];
}

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

// ------------------------------------------------------------------------------------------------
// R2 + R5 fill, variant 3: START-VERTEX-CENTRIC, one wave per start vertex s.
//
// All paths of s occupy ONE contiguous run of the output (count(s) rows starting at the scanned
// offset of its first pair), so a wave that emits them in order writes its region front to back:
// sequential in memory (DRAM-page and TLB friendly), every byte written once.  The candidates of s
// (the neighbour lists of its neighbours b_k) are flattened over the 64 lanes; a batch of R rounds
// first issues every id / rank load, then compacts (ballot + popcount against rank[s]), gathers the
// kept endpoints' embeddings from the n x e table (cache resident) and stores the rows.  No block
// barrier, no dependent load inside a batch beyond the endpoint gather.
// ------------------------------------------------------------------------------------------------
template <int E, int R>
__global__ __launch_bounds__(256) void k_fill_s(FillParams P, const uint32_t *__restrict__ poffs, uint32_t slab_len)
{
    __shared__ uint32_t s_cs[4][65], s_st[4][64], s_b[4][64];
    __shared__ __attribute__((aligned(16))) double s_vb[4][64 * E];
    const unsigned lane = lane_id(), wv = wave_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const bool want_pde = P.out_pde != nullptr;

    for (; w < slab_len; w += nw) {
        const uint32_t e0 = poffs[w], ds = poffs[w + 1] - e0;
        if (ds == 0) continue;
        uint64_t run = P.eoff[e0];
        const uint64_t run_end = P.eoff[e0 + ds];
        if (run_end == run || run >= P.end || run_end <= P.begin) continue;
        const uint32_t thr = P.slab_begin + (uint32_t)w;
        const uint32_t s = P.sorted[thr];
        double vs[E];
#pragma unroll
        for (int k = 0; k < E; k++) vs[k] = want_pde ? P.vde[(uint64_t)s * E + k] : 0.0;

        for (uint32_t k0 = 0; k0 < ds; k0 += 64) {
            // ---- up to 64 middle vertices: where their neighbour lists start, how long they are ----
            const uint32_t k = k0 + lane;
            uint32_t b = 0, st = 0, dg = 0;
            if (k < ds) {
                b = P.pnbr[e0 + k];
                st = P.adj_start[b];
                dg = P.adj_deg[b];
            }
            uint32_t incl = dg;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= (unsigned)off) incl += t;
            }
            const uint32_t C = rl32(incl, 63);
            s_cs[wv][lane] = incl - dg;
            s_st[wv][lane] = st;
            s_b[wv][lane] = b;
            if (lane == 0) s_cs[wv][64] = C;
            if (want_pde && k < ds) {
#pragma unroll
                for (int kk = 0; kk < E; kk++) s_vb[wv][lane * E + kk] = P.vde[(uint64_t)b * E + kk];
            }
            __builtin_amdgcn_wave_barrier();

            for (uint32_t q0 = 0; q0 < C; q0 += 64 * R) {
                uint32_t cc[R], rc[R], kk[R];
                // ---- phase A: every id / rank load of the batch ----
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t q = q0 + r * 64 + lane;
                    cc[r] = 0;
                    rc[r] = 0;
                    kk[r] = 0;
                    if (q < C) {
                        uint32_t a = 0, bnd = 64;  // largest a with cs[a] <= q
#pragma unroll
                        for (int it = 0; it < 6; it++) {
                            const uint32_t mid = (a + bnd) >> 1;
                            if (s_cs[wv][mid] <= q) a = mid; else bnd = mid;
                        }
                        const uint32_t idx = s_st[wv][a] + (q - s_cs[wv][a]);
                        kk[r] = a;
                        cc[r] = P.nbrs[idx];
                        rc[r] = P.nbr_rank[idx];
                    }
                }
                // ---- phase B: compact, gather the endpoint embedding, store ----
                uint64_t pos[R];
                bool kp[R];
                double vc[R][E];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const uint32_t q = q0 + r * 64 + lane;
                    const bool keep = q < C && rc[r] > thr;
                    const uint64_t mask = __ballot(keep);
                    pos[r] = run + (uint64_t)__popcll(mask & lt);
                    run += (uint64_t)__popcll(mask);
                    kp[r] = keep && pos[r] >= P.begin && pos[r] < P.end;
#pragma unroll
                    for (int k2 = 0; k2 < E; k2++) vc[r][k2] = (want_pde && kp[r]) ? P.vde[(uint64_t)cc[r] * E + k2] : 0.0;
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (kp[r]) {
                        double vb[E];
#pragma unroll
                        for (int k2 = 0; k2 < E; k2++) vb[k2] = want_pde ? s_vb[wv][kk[r] * E + k2] : 0.0;
                        emit_path<E, FillParams>(P, pos[r], s, s_b[wv][kk[r]], cc[r], vs, vb, vc[r]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();  // strip reused by the next chunk / start vertex
        }
    }
}

// ------------------------------------------------------------------------------------------------
// R2 + R5 fill, variants 4 / 5: variant 3 plus a per-wave LDS staging strip.  Kept rows are parked in
// the strip (up to 64 rows) and flushed with consecutive lanes on consecutive 16-byte (pde) /
// 4-byte (ids) pieces, so the L2 sees whole 64-byte write requests instead of one request per lane
// (PMC: variant 3 issued 2.6 write requests per 64 bytes written).  NV = true streams the endpoint
// embeddings from the per-adjacency array (contiguous with the rank stream) instead of gathering
// them from the n x e table.
// ------------------------------------------------------------------------------------------------
template <int E, int R, bool NV>
__global__ __launch_bounds__(256) void k_fill_s_staged(FillParams P, const uint32_t *__restrict__ poffs,
                                                       uint32_t slab_len)
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
        const uint32_t e0 = poffs[w], ds = poffs[w + 1] - e0;
        if (ds == 0) continue;
        uint64_t run = P.eoff[e0];
        const uint64_t run_end = P.eoff[e0 + ds];
        if (run_end == run || run >= P.end || run_end <= P.begin) continue;
        const uint32_t thr = P.slab_begin + (uint32_t)w;
        const uint32_t s = P.sorted[thr];
        const uint32_t part = P.out_part ? P.member[s] : 0u;
        double vs[E];
#pragma unroll
        for (int k = 0; k < E; k++) vs[k] = want_pde ? P.vde[(uint64_t)s * E + k] : 0.0;
        uint64_t fbase = run;  // output slot of the strip's first row
        uint32_t fill = 0;     // rows parked in the strip

        // write rows [fbase, fbase + fill) of the strip to global memory, contiguous pieces per lane
        auto flush = [&]() {
            __builtin_amdgcn_wave_barrier();
            // clip to [begin, end)
            const uint64_t lo = max(fbase, P.begin), hi = min(fbase + fill, P.end);
            if (hi > lo) {
                const uint32_t r0 = (uint32_t)(lo - fbase), nr = (uint32_t)(hi - lo);
                const uint64_t o = lo - P.begin;
                if (P.out_ids)
                    for (uint32_t g = lane; g < nr * 3; g += 64) P.out_ids[o * 3 + g] = my_ids[r0 * 3 + g];
                if (want_pde) {
                    if ((D & 1) == 0) {
                        const double2 *src = reinterpret_cast<const double2 *>(my_pde + (size_t)r0 * D);
                        double2 *dst = reinterpret_cast<double2 *>(P.out_pde + o * D);
                        for (uint32_t g = lane; g < nr * (D / 2); g += 64) dst[g] = src[g];
                    } else {
                        for (uint32_t g = lane; g < nr * D; g += 64) P.out_pde[o * D + g] = my_pde[(size_t)r0 * D + g];
                    }
                }
                if (P.out_part)
                    for (uint32_t g = lane; g < nr; g += 64) P.out_part[o + g] = part;
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
            uint32_t b = 0, st = 0, dg = 0;
            if (k < ds) {
                b = P.pnbr[e0 + k];
                st = P.adj_start[b];
                dg = P.adj_deg[b];
            }
            uint32_t incl = dg;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off);
                if (lane >= (unsigned)off) incl += t;
            }
            const uint32_t C = rl32(incl, 63);
            s_cs[wv][lane] = incl - dg;
            s_st[wv][lane] = st;
            s_b[wv][lane] = b;
            if (lane == 0) s_cs[wv][64] = C;
            if (want_pde && k < ds) {
#pragma unroll
                for (int kk = 0; kk < E; kk++) s_vb[wv][lane * E + kk] = P.vde[(uint64_t)b * E + kk];
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
                        if (NV && want_pde) {
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
                    if (!NV) {
#pragma unroll
                        for (int k2 = 0; k2 < E; k2++)
                            vc[r][k2] = (want_pde && kp[r]) ? P.vde[(uint64_t)cc[r] * E + k2] : 0.0;
                    }
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

// ------------------------------------------------------------------------------------------------
// R2 + R5 fill, variants 6 / 7: variants 4 / 5 with the dependent-load chain cut to three levels.
// PMC on variant 3-5 showed waves parked on memory ~95 % of the time with ~5 dependent global loads
// per start vertex (poffs -> eoff/sorted/pnbr -> adj_start/adj_deg/vde -> candidates -> endpoint
// embedding).  Here one 32-byte record per start vertex (StartRec) and one 16-byte record per
// (s, b) pair (PairRec), both written by small kernels of the count phase, give the chain
// record -> pair records -> candidates [-> endpoint embedding].
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) PairRec {
    uint32_t b, st, dg, pad;
};
struct __attribute__((aligned(16))) StartRec {
    uint64_t base, end;  // first / one-past-last output slot of this start vertex
    uint32_t e0, ds, s, part;
    uint32_t a_s, pad0, pad1, pad2;  // adj_start[s]: N(s) = the middle vertices of its pairs, in pair order
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

__global__ void k_start_recs(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted,
                             const uint32_t *__restrict__ member, const uint32_t *__restrict__ adj_start,
                             const uint32_t *__restrict__ poffs,
                             const uint64_t *__restrict__ eoff, StartRec *__restrict__ recs)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t e0 = poffs[i], e1 = poffs[i + 1];
        const uint32_t s = sorted[slab_begin + i];
        StartRec r = {eoff[e0], eoff[e1], e0, e1 - e0, s, member[s], adj_start[s], 0u, 0u, 0u};
        recs[i] = r;
    }
}

template <int E, int R, bool NV, bool NT>
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
                        if (NT) __builtin_nontemporal_store(my_ids[r0 * 3 + g], &P.out_ids[o * 3 + g]);
                        else P.out_ids[o * 3 + g] = my_ids[r0 * 3 + g];
                    }
                }
                if (want_pde) {
                    if ((D & 1) == 0) {
                        typedef double dbl2 __attribute__((ext_vector_type(2)));
                        const dbl2 *src = reinterpret_cast<const dbl2 *>(my_pde + (size_t)r0 * D);
                        dbl2 *dst = reinterpret_cast<dbl2 *>(P.out_pde + o * D);
                        for (uint32_t g = lane; g < nr * (D / 2); g += 64) {
                            if (NT) __builtin_nontemporal_store(src[g], &dst[g]);
                            else dst[g] = src[g];
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
                        if (NV && want_pde) {
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
                    if (!NV) {
#pragma unroll
                        for (int k2 = 0; k2 < E; k2++)
                            vc[r][k2] = (want_pde && kp[r]) ? P.vde[(uint64_t)cc[r] * E + k2] : 0.0;
                    }
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

// ------------------------------------------------------------------------------------------------
// halo helpers
// ------------------------------------------------------------------------------------------------
__global__ void k_mark_needed(uint64_t cnt, const uint32_t *__restrict__ nbrs, uint8_t *__restrict__ mark)
{
    for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < cnt; q += (uint64_t)gridDim.x * blockDim.x)
        mark[nbrs[q]] = 1;
}

// flag[v] = 1 iff v is needed, not held, and owned by rank r (slab bounds over the processing order)
__global__ void k_flag_owner(uint32_t n, const uint8_t *__restrict__ mark, const uint8_t *__restrict__ present,
                             const uint32_t *__restrict__ rank, uint32_t lo, uint32_t hi, uint8_t *__restrict__ flag)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t rk = rank[v];
        flag[v] = (mark[v] && !present[v] && rk >= lo && rk < hi) ? 1 : 0;
    }
}

// back to owned rows only: present := owned, halo degrees := 0
__global__ void k_drop_halo(uint32_t n, const uint8_t *__restrict__ owned, uint8_t *__restrict__ present,
                            uint32_t *__restrict__ deg)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        if (!owned[v]) deg[v] = 0;
        present[v] = owned[v];
    }
}

__global__ void k_rows_pack(uint64_t n_req, const uint32_t *__restrict__ ids, const uint64_t *__restrict__ roff,
                            const uint32_t *__restrict__ adj_start, const uint32_t *__restrict__ nbrs,
                            uint32_t *__restrict__ out)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; g < n_req; g += ng) {
        const uint32_t a = adj_start[ids[g]];
        const uint64_t o = roff[g];
        const uint32_t d = (uint32_t)(roff[g + 1] - o);
        for (uint32_t j = sub; j < d; j += 16) out[o + j] = nbrs[a + j];
    }
}

}  // namespace gnnpe
