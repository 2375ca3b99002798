// gnnpe_kernels.hip.h -- hand-written gfx950 kernels of the offline path (included by gnnpe_engine.hip):
// utilities, R4 (gen_vde), halo helpers and the pieces shared by the enumeration variants in
// gnnpe_fill_{pairwave,middle,start,ranked}.hip.h.
//
// All kernels are HBM/L2-bound integer + fp64 gather/scatter work (the reference has no
// dense contraction, so there is nothing for MFMA).  Wave = 64 lanes; blocks are 256 threads.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "gnnpe_records.h"

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

// ------------------------------------------------------------------------------------------------
// non-simple graphs (gnnpe_set_multigraph_rows): the rows as graph.cpp:211-233 leaves them -- repeats kept -- beside the
// simple rows the enumeration runs on.
// ------------------------------------------------------------------------------------------------
// row k of the multigraph form, stripped of its repeats, must be the loaded row of the same vertex; *bad = smallest row position
// where it is not (or where the list is not ascending / holds an id >= n or the vertex itself)
__global__ void k_check_multi_rows(uint32_t n, uint32_t n_rows, const uint32_t *__restrict__ rows, const uint64_t *__restrict__ moff,
                                   const uint32_t *__restrict__ mnbr, const uint32_t *__restrict__ adj_start,
                                   const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ nbrs, uint32_t *__restrict__ bad)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_rows; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = rows ? rows[k] : (uint32_t)k;
        const uint32_t st = adj_start[v], d = adj_deg[v];
        uint32_t j = 0, prev = 0;
        bool ok = true, have_prev = false;
        for (uint64_t i = moff[k]; i < moff[k + 1] && ok; i++) {
            const uint32_t u = mnbr[i];
            if (u >= n || u == v || (have_prev && u < prev)) ok = false;
            else if (!(have_prev && u == prev)) {
                if (j < d && nbrs[st + j] == u) j++;
                else ok = false;
            }
            prev = u;
            have_prev = true;
        }
        if (!ok || j != d) atomicMin(bad, (uint32_t)k);
    }
}

// gen_vde over the multigraph rows (custom.h:527-540): one thread per (row, dimension), the row's entries in their stored
// (ascending) order from 0.0 -- the reference's operation order; a repeated entry is summed as often as it is stored
__global__ void k_vde_multi(uint32_t n_rows, uint32_t e, const uint32_t *__restrict__ rows, const uint64_t *__restrict__ moff,
                            const uint32_t *__restrict__ mlabel, const uint32_t *__restrict__ labels, const double *__restrict__ xtab,
                            double *__restrict__ nx, double *__restrict__ vde)
{
    for (uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; t < (uint64_t)n_rows * e; t += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = t / e;
        const uint32_t d = (uint32_t)(t % e);
        const uint32_t v = rows ? rows[k] : (uint32_t)k;
        double s = 0.0;
        for (uint64_t i = moff[k]; i < moff[k + 1]; i++) s += xtab[(uint64_t)mlabel[i] * e + d];
        nx[(uint64_t)v * e + d] = s;
        vde[(uint64_t)v * e + d] = xtab[(uint64_t)labels[v] * e + d] + s;
    }
}

__global__ void k_row_lengths_u64(uint32_t n_rows, const uint64_t *__restrict__ off, uint32_t *__restrict__ len)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_rows; k += (uint64_t)gridDim.x * blockDim.x)
        len[k] = (uint32_t)(off[k + 1] - off[k]);
}

// slab row degrees: pdeg[i] = deg(sorted[slab_begin + i]); pdeg[len] = 0 (scan sentinel)
__global__ void k_slab_degrees(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted,
                               const uint32_t *__restrict__ deg, uint32_t *__restrict__ pdeg)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= len; i += (uint64_t)gridDim.x * blockDim.x)
        pdeg[i] = i < len ? deg[sorted[slab_begin + i]] : 0u;
}

// ------------------------------------------------------------------------------------------------
// R4 gen_vde (custom.h:513-544).  One thread per owned row; the block's adjacency range is contiguous, so the
// neighbours' LABELS -- stored next to the neighbour ids when the rows are loaded: graph structure, labels come with
// the graph file -- are streamed once with coalesced loads into LDS, and every thread then adds its own row's
// features in ascending-neighbour order -- the same
// operation order as the reference loop (:527-534), hence bit-identical fp64 results.
// The label table (|Sigma| x e doubles) also sits in LDS when it fits.
// ------------------------------------------------------------------------------------------------
// what k_vde needs to write the count kernel's per-vertex gather records beside the embeddings (vinfo == nullptr: it does not)
struct VinfoPack {
    double *vinfo;
    const uint32_t *rank, *poffs;
    uint32_t slab_begin, slab_end;
};
constexpr int kVdeStage = 6144;        // neighbour labels staged per pass (24 KiB)
constexpr int kVdeTabMax = 4096;       // table doubles kept in LDS (32 KiB)

template <int E>
__global__ __launch_bounds__(256) void k_vde(uint32_t n_rows, const uint32_t *__restrict__ rows,
                                             const uint32_t *__restrict__ adj_start,
                                             const uint32_t *__restrict__ adj_deg,
                                             const uint32_t *__restrict__ nbr_label,
                                             const uint32_t *__restrict__ labels,
                                             const double *__restrict__ xtab, uint32_t n_labels, uint32_t e_rt,
                                             double *__restrict__ nx, double *__restrict__ vde, bool hub_split,
                                             double *__restrict__ x_out, VinfoPack vp)
{
    const int e = E ? E : (int)e_rt;
    __shared__ uint32_t s_lab[kVdeStage];
    // the table's LDS copy is sized by the launch (n_labels x e doubles, nothing when it does not fit): a fixed 32 KiB
    // array next to the 24 KiB stage left two workgroups per CU, and the kernel waits on LDS latency (vde phase at config 3:
    // 0.153 -> 0.099 ms)
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
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
    bool hub = false;
    if (r < n_rows) {
        v = rows ? rows[r] : r;
        my_b = adj_start[v];
        my_e = my_b + adj_deg[v];
        // a row longer than kHubDegree is summed by k_vde_hubs (one lane per (row, dimension)): a single thread of this
        // block walking 4 500 entries while the other 255 wait made k_vde<8> 12.5 ms at BASELINE config 5
        hub = hub_split && adj_deg[v] > kHubDegree;
        if (hub) my_e = my_b;
    }
    double acc[E ? E : 32];
#pragma unroll
    for (int k = 0; k < (E ? E : 32); k++) acc[k] = 0.0;

    __syncthreads();
    // (one loop per address space of the table: a pointer that may be LDS or global makes every lookup a flat load)
    auto sum_rows = [&](auto tab) {
        for (uint32_t c0 = q_begin; c0 < q_end; c0 += kVdeStage) {
            const uint32_t c1 = min(c0 + (uint32_t)kVdeStage, q_end);
            for (uint32_t q = c0 + threadIdx.x; q < c1; q += blockDim.x) s_lab[q - c0] = nbr_label[q];
            __syncthreads();
            const uint32_t lo = max(my_b, c0), hi = min(my_e, c1);
            for (uint32_t q = lo; q < hi; q++) {
                const uint32_t lb = s_lab[q - c0];
                if (E) {
#pragma unroll
                    for (int k = 0; k < E; k++) acc[k] += tab[lb * E + k];
                } else {
                    for (int k = 0; k < e; k++) acc[k] += tab[lb * e + k];
                }
            }
            __syncthreads();
        }
    };
    if (tab_in_lds) sum_rows(s_tab); else sum_rows(xtab);
    if (r < n_rows) {
        const uint32_t lv = labels[v];
        auto finish = [&](auto tab) {
            for (int k = 0; k < e; k++) {
                const double xv = tab[lv * e + k];
                // x = the label's feature (custom.h:521): written here when every vertex is a row of this launch, so that the step
                // needs no k_x_from_labels launch (round 6); hub rows too -- only their sums belong to k_vde_hubs
                if (x_out) x_out[(uint64_t)v * e + k] = xv;
                if (!hub) {
                    nx[(uint64_t)v * e + k] = acc[k];
                    vde[(uint64_t)v * e + k] = xv + acc[k];
                }
            }
            // the count kernel's gather record {vde[v], rank[v] | first pair slot of v as a start vertex} (k_pack_vinfo), when the
            // caller knows the slab's pair offsets are current: one launch less per step.  Graphs with hub rows keep the launch.
            if constexpr (E > 0) {
                if (vp.vinfo) {
                    constexpr int S = E + 2;
#pragma unroll
                    for (int k = 0; k < E; k++) vp.vinfo[(uint64_t)v * S + k] = tab[lv * E + k] + acc[k];
                    const uint32_t rk = vp.rank[v];
                    const uint32_t po = (rk >= vp.slab_begin && rk < vp.slab_end) ? vp.poffs[rk - vp.slab_begin] : 0xFFFFFFFFu;  // kNoEdge
                    reinterpret_cast<uint64_t *>(vp.vinfo)[(uint64_t)v * S + E] = ((uint64_t)po << 32) | rk;
                    vp.vinfo[(uint64_t)v * S + E + 1] = 0.0;
                }
            }
        };
        if (tab_in_lds) finish(s_tab); else finish(xtab);
    }
}

// gen_vde for the rows longer than kHubDegree (custom.h:523-541).  The sum of a row is one fp64 chain in ascending
// neighbour order (bit-exactness), so the parallelism is across rows and across the e dimensions: one LANE per (hub row,
// dimension), 64 / E rows per wave; the lanes of a row read the same label (one address, broadcast), every lane looks its
// own dimension up in the table and adds.  Four entries are fetched ahead of the chain's adds.
template <int E>
__global__ __launch_bounds__(256) void k_vde_hubs(uint32_t n_hub, const uint32_t *__restrict__ hub_rows,
                                                  const uint8_t *__restrict__ owned, const uint32_t *__restrict__ adj_start,
                                                  const uint32_t *__restrict__ adj_deg,
                                                  const uint32_t *__restrict__ nbr_label,
                                                  const uint32_t *__restrict__ labels, const double *__restrict__ xtab,
                                                  uint32_t n_labels, double *__restrict__ nx, double *__restrict__ vde)
{
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    const bool tab_in_lds = (uint64_t)n_labels * E <= (uint64_t)kVdeTabMax;
    if (tab_in_lds)
        for (uint32_t i = threadIdx.x; i < n_labels * E; i += blockDim.x) s_tab[i] = xtab[i];
    __syncthreads();
    constexpr uint32_t RPW = 64 / E;  // rows per wave
    const unsigned lane = lane_id();
    const uint32_t k = lane % E, rw = lane / E;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w * RPW < n_hub; w += nw) {
        const uint64_t h = w * RPW + rw;
        if (rw >= RPW || h >= n_hub) continue;
        const uint32_t v = hub_rows[h];
        if (owned && !owned[v]) continue;  // halo rows are summed by their owners (their labels are not held here)
        const uint32_t st = adj_start[v], d = adj_deg[v];
        double acc = 0.0;
        // (two loops, one per address space of the table: a pointer that may be either makes every lookup a flat load)
        auto sum_row = [&](auto tab) {
            // the next kPf labels are in flight while this group's table values are added.  (Round 4, first version: four in
            // flight -- the longest row of config 5, 4 517 entries, then walks 1 130 dependent load round trips on its own,
            // which was the whole kernel's 0.50 ms; sixteen: 0.19 ms, thirty-two: 0.17 -- profiles/r04_deep_kernel_stats.csv.)
            constexpr uint32_t kPf = 16;
            const uint32_t *lp = nbr_label + st;
            uint32_t j = 0, lc[kPf], ln[kPf];
#pragma unroll
            for (uint32_t i = 0; i < kPf; i++) lc[i] = 0;
            if (d >= kPf) {
#pragma unroll
                for (uint32_t i = 0; i < kPf; i++) lc[i] = lp[i];
            }
            for (; j + 2 * kPf <= d; j += kPf) {
#pragma unroll
                for (uint32_t i = 0; i < kPf; i++) ln[i] = lp[j + kPf + i];
                double a[kPf];
#pragma unroll
                for (uint32_t i = 0; i < kPf; i++) a[i] = tab[lc[i] * E + k];
#pragma unroll
                for (uint32_t i = 0; i < kPf; i++) acc += a[i];  // ascending entry order: the reference's sum, bit for bit
#pragma unroll
                for (uint32_t i = 0; i < kPf; i++) lc[i] = ln[i];
            }
            if (j + kPf <= d) {
#pragma unroll
                for (uint32_t i = 0; i < kPf; i++) acc += tab[lc[i] * E + k];
                j += kPf;
            }
            for (; j < d; j++) acc += tab[lp[j] * E + k];
            return tab[labels[v] * E + k];
        };
        const double xv = tab_in_lds ? sum_row(s_tab) : sum_row(xtab);
        nx[(uint64_t)v * E + k] = acc;
        vde[(uint64_t)v * E + k] = xv + acc;
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

// every rank's slab of an all-gathered table [n_ranks][stride rows][e] back into the vde table, one launch: position p
// of the processing order belongs to the slab r with bounds[r] <= p < bounds[r + 1] (bounds: n_ranks + 1 device words)
__global__ void k_vde_unpack_all(uint32_t n, uint32_t e, uint32_t n_ranks, const uint32_t *__restrict__ bounds,
                                 uint32_t stride, uint32_t skip_rank, const uint32_t *__restrict__ sorted,
                                 const double *__restrict__ buf, double *__restrict__ vde)
{
    const uint64_t tot = (uint64_t)n * e;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t p = (uint32_t)(i / e), k = (uint32_t)(i % e);
        uint32_t lo = 0, hi = n_ranks;  // largest r with bounds[r] <= p
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (bounds[mid] <= p) lo = mid; else hi = mid;
        }
        if (lo == skip_rank) continue;
        vde[(uint64_t)sorted[p] * e + k] = buf[((uint64_t)lo * stride + (p - bounds[lo])) * e + k];
    }
}

// ------------------------------------------------------------------------------------------------
// shared pieces of the enumeration kernels (gnnpe_fill_*.hip.h)
// ------------------------------------------------------------------------------------------------
__global__ void k_per_start_counts(uint32_t len, const uint32_t *__restrict__ poffs,
                                   const uint64_t *__restrict__ eoff, uint64_t *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = eoff[poffs[i + 1]] - eoff[poffs[i]];
}

struct FillParams {
    const uint32_t *erow, *pnbr, *adj_start, *adj_deg, *nbrs, *nbr_rank, *sorted, *member;
    const uint64_t *eoff;
    const double *vde, *x, *nbr_vde;  // nbr_vde[q] = vde[nbrs[q]] (e doubles per adjacency entry)
    uint64_t n_edges, begin, end;
    uint32_t slab_begin, e;
    uint32_t *out_ids;
    double *out_pde, *out_pdl;
    uint32_t *out_part;
};

constexpr uint64_t kNoOff = ~0ull;

__device__ __forceinline__ uint32_t rl32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// v_writelane_b32: `old` with lane `l` (wave-uniform) replaced by the wave-uniform `v`.  This clang has no builtin for it;
// the LLVM intrinsic is bound by name.
extern "C" __device__ int gnnpe_llvm_writelane(int v, int l, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t writelane32(uint32_t v, uint32_t l, uint32_t old)
{
    return (uint32_t)gnnpe_llvm_writelane((int)v, (int)l, (int)old);
}

// revpos[q] for adjacency entry q = (b -> u): position of b inside N(u), or kNoEdge when row u is not
// an OWNED row of this device (only owned rows start paths here).  Depends on the graph only -- not on
// the processing order -- so it is built when rows are loaded / appended, like the loader's sort of
// the adjacency lists (graph.cpp:231-233).  16 lanes per row, one binary search per entry.
__global__ void k_revpos(uint32_t n_rows, const uint32_t *__restrict__ rows, const uint8_t *__restrict__ owned,
                         const uint32_t *__restrict__ adj_start, const uint32_t *__restrict__ adj_deg,
                         const uint32_t *__restrict__ nbrs, uint32_t *__restrict__ revpos)
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
        }
    }
}

// Load-time validation of adjacency rows handed over through the C-ABI (the loader's own output is valid by
// construction, graph.cpp:231-233 sorts every list): neighbour ids < n, strictly ascending (sorted, no duplicate edge),
// no self-loop.  Every kernel indexes per-vertex tables by neighbour id and binary-searches rows, and the closed-form
// count assumes a simple graph.  bad[0] = smallest offending row (atomicMin), untouched when all rows are fine.
__global__ void k_validate_rows(uint32_t n, uint64_t n_rows, const uint32_t *__restrict__ rows,
                                const uint32_t *__restrict__ adj_start, const uint32_t *__restrict__ adj_deg,
                                const uint32_t *__restrict__ nbrs, uint32_t *__restrict__ bad)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; g < n_rows; g += ng) {
        const uint32_t b = rows ? rows[g] : (uint32_t)g;
        const uint32_t st = adj_start[b], d = adj_deg[b];
        bool ok = true;
        for (uint32_t i = sub; i < d; i += 16) {
            const uint32_t u = nbrs[st + i];
            if (u >= n || u == b || (i > 0 && nbrs[st + i - 1] >= u)) ok = false;
        }
        if (!ok) atomicMin(bad, b);
    }
}

// Rows longer than `limit` (hub rows of the enumeration): their ids, and [begin, end) of their adjacency for the
// per-row sorts.  Order of the list is arbitrary (atomic append); nothing downstream depends on it.
__global__ void k_hub_list(uint64_t n_rows, const uint32_t *__restrict__ rows, const uint32_t *__restrict__ adj_start,
                           const uint32_t *__restrict__ adj_deg, uint32_t limit, uint32_t *__restrict__ counter,
                           uint32_t *__restrict__ hub_rows, uint32_t *__restrict__ hub_beg, uint32_t *__restrict__ hub_end)
{
    for (uint64_t g = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; g < n_rows; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = rows ? rows[g] : (uint32_t)g;
        const uint32_t d = adj_deg[b];
        if (d > limit) {
            const uint32_t k = atomicAdd(counter, 1u);
            if (hub_rows) {
                hub_rows[k] = b;
                hub_beg[k] = adj_start[b];
                hub_end[k] = adj_start[b] + d;
            }
        }
    }
}

__global__ void k_hub_sort_keys(uint32_t n_hub, const uint32_t *__restrict__ hub_rows, const uint32_t *__restrict__ adj_deg,
                                uint32_t *__restrict__ keys)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_hub; k += (uint64_t)gridDim.x * blockDim.x)
        keys[k] = adj_deg[hub_rows[k]];
}
__global__ void k_hub_bounds(uint32_t n_hub, const uint32_t *__restrict__ hub_rows, const uint32_t *__restrict__ adj_start,
                             const uint32_t *__restrict__ adj_deg, uint32_t *__restrict__ hub_beg, uint32_t *__restrict__ hub_end)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n_hub; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = hub_rows[k];
        hub_beg[k] = adj_start[b];
        hub_end[k] = adj_start[b] + adj_deg[b];
    }
}

// emission index of the directed pair (s = u, b) for adjacency entry q = (b -> u), or kNoEdge
__device__ __forceinline__ uint32_t pair_index(uint32_t revpos, uint32_t rank_u, uint32_t slab_begin, uint32_t slab_end,
                                               const uint32_t *__restrict__ poffs)
{
    return (revpos != kNoEdge && rank_u >= slab_begin && rank_u < slab_end) ? poffs[rank_u - slab_begin] + revpos
                                                                            : kNoEdge;
}

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

// The start vertices' output offsets WITHOUT the scan over the pairs (round 5): the emit kernel of the start-vertex shape needs one
// number per START vertex -- its first output slot -- not one per pair.  One pass: a workgroup takes a tile of 256 start vertices
// (tiles in order from a ticket counter), sums every start's pair counts (16 lanes per start over its consecutive pair records),
// scans the 256 sums, learns the tile's prefix by decoupled look-back over the tiles before it (status word = value << 2 | state:
// 1 = the tile's own sum, 2 = its inclusive prefix; one 64-bit atomic word, so value and state arrive together) and writes the
// start records.  It replaces the two scan launches over 2.0e7 pairs (read 320 MB, write 160 MB) and k_start_recs (0.18 ms together at
// BASELINE config 3) by one launch that reads the pair counts once; the per-pair offsets (eoff) are built only for who needs them
// (tile table, index build, per-start counts: ensure_eoff in gnnpe_engine.hip).  total_out = eoff[n_edges], where the total lives.
constexpr int kStartTile = 256;
__global__ __launch_bounds__(kStartTile) void k_start_scan(uint32_t len, uint32_t slab_begin, const uint32_t *__restrict__ sorted,
                                                           const uint32_t *__restrict__ member, const uint32_t *__restrict__ adj_start,
                                                           const uint32_t *__restrict__ poffs, const RankedPair *__restrict__ pairs,
                                                           unsigned long long *__restrict__ status, uint32_t *__restrict__ ticket,
                                                           StartRec *__restrict__ recs, uint64_t *__restrict__ total_out,
                                                           uint32_t *__restrict__ clear_words, uint32_t n_clear)
{
    // (the emit kernel's ticket heads, zeroed here instead of by a memset in front of the fill: a launch less per step)
    for (uint32_t i = blockIdx.x * kStartTile + threadIdx.x; i < n_clear; i += gridDim.x * kStartTile) clear_words[i] = 0u;
    typedef hipcub::BlockScan<uint64_t, kStartTile> Scan;
    __shared__ typename Scan::TempStorage s_scan;
    __shared__ uint64_t s_cnt[kStartTile];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_tile;
    const uint32_t n_tiles = (len + kStartTile - 1) / kStartTile, tid = threadIdx.x, sub = tid & 15u, grp = tid >> 4;
    for (;;) {
        if (tid == 0) s_tile = __builtin_amdgcn_atomic_inc32(ticket, 0xFFFFFFFFu, __ATOMIC_RELAXED, "agent");
        __syncthreads();
        const uint32_t tile = s_tile;
        if (tile >= n_tiles) break;
        // 1. the starts' own counts: 16 lanes per start vertex, sixteen start vertices per lane group -- all their row bounds first,
        // then all their pair counts (two per lane and start: 32 pairs cover a start of BASELINE's graphs), so that a tile is two
        // round trips, not thirty-two; longer rows finish in a loop
        // (in two halves of eight start vertices per lane group: 48 registers of row bounds and sums held the kernel to five
        // workgroups per CU)
        constexpr int NPASS = kStartTile / 16, HALF = NPASS / 2;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint32_t e0[HALF], e1[HALF], sum[HALF];
#pragma unroll
            for (int pp = 0; pp < HALF; pp++) {
                const int pass = h * HALF + pp;
                const uint32_t i = min(tile * kStartTile + pass * 16 + grp, len - 1u);  // (past the last start: its bounds again, dropped below)
                e0[pp] = poffs[i];
                e1[pp] = poffs[i + 1];
            }
#pragma unroll
            for (int pp = 0; pp < HALF; pp++) {
                const int pass = h * HALF + pp;
                const bool in = tile * kStartTile + pass * 16 + grp < len;
                const uint32_t k0 = e0[pp] + sub, k1 = k0 + 16, last = e1[pp] ? e1[pp] - 1u : 0u;
                const uint32_t c0 = pairs[min(k0, last)].cnt, c1 = pairs[min(k1, last)].cnt;
                sum[pp] = (in && k0 < e1[pp] ? c0 & ~kHubFlag : 0u) + (in && k1 < e1[pp] ? c1 & ~kHubFlag : 0u);
                if (!in) e1[pp] = 0;
            }
#pragma unroll
            for (int pp = 0; pp < HALF; pp++) {
                const int pass = h * HALF + pp;
                for (uint32_t k = e0[pp] + sub + 32; k < e1[pp]; k += 16) sum[pp] += pairs[k].cnt & ~kHubFlag;
                unsigned long long v = sum[pp];  // (64-bit from here: 16 lanes x 2^32 would not fit -- ADVICE r5; pairs[ne] is the zero sentinel)
                v += __shfl_xor(v, 8, 16);
                v += __shfl_xor(v, 4, 16);
                v += __shfl_xor(v, 2, 16);
                v += __shfl_xor(v, 1, 16);
                if (sub == 0) s_cnt[pass * 16 + grp] = v;
            }
        }
        __syncthreads();
        // 2. inside the tile
        const uint64_t mine = s_cnt[tid];
        uint64_t excl = 0, aggregate = 0;
        Scan(s_scan).ExclusiveSum(mine, excl, aggregate);
        // 3. the tile's prefix: look back over the tiles before it (taken in ticket order, so every one of them is running), 64 of
        // them per step -- one lane each; a single thread walking back one tile per round trip made the pass 0.15 ms
        if (tid < 64) {
            if (tid == 0)
                __hip_atomic_store(&status[tile], (unsigned long long)((aggregate << 2) | (tile ? 1ull : 2ull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t prefix = 0;
            int64_t top = (int64_t)tile - 1;  // nearest predecessor of the window
            while (top >= 0) {
                const int64_t idx = top - (int64_t)tid;
                // (in front of tile 0: an inclusive prefix of zero)
                const unsigned long long w = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 2ull;
                const uint64_t inc = __ballot((w & 3ull) == 2ull), none = __ballot((w & 3ull) == 0ull);
                const uint32_t first_inc = inc ? (uint32_t)__builtin_ctzll(inc) : 64u;
                const uint64_t need = first_inc >= 63u ? ~0ull : ((1ull << (first_inc + 1)) - 1ull);  // lanes up to the first inclusive prefix
                if (none & need) {  // one of them has not published yet
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                uint64_t v = tid <= first_inc ? (uint64_t)(w >> 2) : 0ull;
                for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
                prefix += v;
                if (first_inc < 64u) break;
                top -= 64;
            }
            if (tid == 0) {
                if (tile) __hip_atomic_store(&status[tile], (unsigned long long)(((prefix + aggregate) << 2) | 2ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = prefix;
                if (tile == n_tiles - 1) *total_out = prefix + aggregate;
            }
        }
        __syncthreads();
        // 4. the start records
        const uint32_t i = tile * kStartTile + tid;
        if (i < len) {
            const uint32_t e0 = poffs[i], e1 = poffs[i + 1];
            const uint32_t s = sorted[slab_begin + i];
            const uint64_t base = s_prefix + excl;
            StartRec r = {base, base + mine, e0, e1 - e0, s, member[s], adj_start[s], 0u, 0u, 0u};
            recs[i] = r;
        }
        __syncthreads();  // s_tile, s_cnt and s_prefix are rewritten by the next tile
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

// key[v] = owning rank of v (slab bounds over the processing order) if v is needed and not held, else 255;
// ids[v] = v.  A stable 8-bit radix sort of (key, id) then yields the request lists grouped by owner with
// ascending ids; k_key_counts reads the group sizes off the sorted keys.
__global__ void k_need_owner(uint32_t n, uint32_t n_ranks, const uint32_t *__restrict__ bounds,
                             const uint8_t *__restrict__ mark, const uint8_t *__restrict__ present,
                             const uint32_t *__restrict__ rank, uint8_t *__restrict__ key, uint32_t *__restrict__ ids)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t k = 255;
        if (mark[v] && !present[v]) {
            const uint32_t rk = rank[v];
            uint32_t lo = 0, hi = n_ranks;  // largest r with bounds[r] <= rk
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (bounds[mid] <= rk) lo = mid; else hi = mid;
            }
            k = (uint8_t)lo;
        }
        key[v] = k;
        ids[v] = (uint32_t)v;
    }
}

// hist[r] = number of sorted keys equal to r (one thread per rank, two binary searches)
__global__ void k_key_counts(uint32_t n, uint32_t n_ranks, const uint8_t *__restrict__ sorted_key,
                             unsigned long long *__restrict__ hist)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_ranks) return;
    uint32_t b[2];
    for (int t = 0; t < 2; t++) {
        const uint32_t want = r + t;  // first position with key >= want
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (sorted_key[mid] < want) lo = mid + 1; else hi = mid;
        }
        b[t] = lo;
    }
    hist[r] = b[1] - b[0];
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

// Truncated halo rows (multi-GPU): a rank whose slab starts at processing position min_rank never emits a path through
// a neighbour ranked before it (kept iff rank[c] > rank[s] >= min_rank), so the entries with rank < min_rank of a halo
// row are dropped when the row is installed.  kept[k] = entries of packed row k that stay; 16 lanes per row.
// (an id >= n is KEPT, never used as an index: the row validation that follows the install rejects it by name)
__global__ void k_rows_kept_counts(uint64_t n_rows, const uint64_t *__restrict__ src_off, const uint32_t *__restrict__ src,
                                   const uint32_t *__restrict__ rank, uint32_t n, uint32_t min_rank, uint32_t *__restrict__ kept)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    const uint64_t g_end = (n_rows + 3) & ~(uint64_t)3;  // whole waves take part in the shuffles
    for (; g < g_end; g += ng) {
        uint32_t cnt = 0;
        if (g < n_rows) {
            const uint64_t a = src_off[g], b = src_off[g + 1];
            for (uint64_t q = a + sub; q < b; q += 16) {
                const uint32_t u = src[q];
                cnt += (u >= n || rank[u] >= min_rank) ? 1u : 0u;
            }
        }
        cnt += __shfl_xor(cnt, 8);
        cnt += __shfl_xor(cnt, 4);
        cnt += __shfl_xor(cnt, 2);
        cnt += __shfl_xor(cnt, 1);
        if (sub == 0 && g < n_rows) kept[g] = cnt;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) kept[n_rows] = 0;
}

// copy the kept entries of packed row k, in order, to dst[dst_off[k] ...): one wave per row
__global__ void k_rows_compact(uint64_t n_rows, const uint64_t *__restrict__ src_off, const uint32_t *__restrict__ src,
                               const uint32_t *__restrict__ rank, uint32_t n, uint32_t min_rank,
                               const uint64_t *__restrict__ dst_off, uint32_t *__restrict__ dst)
{
    const unsigned lane = lane_id();
    const uint64_t lt = (1ull << lane) - 1ull;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < n_rows; w += nw) {
        const uint64_t a = src_off[w], b = src_off[w + 1];
        uint64_t o = dst_off[w];
        for (uint64_t q0 = a; q0 < b; q0 += 64) {
            const uint64_t q = q0 + lane;
            uint32_t u = 0;
            bool keep = false;
            if (q < b) {
                u = src[q];
                keep = u >= n || rank[u] >= min_rank;
            }
            const uint64_t mask = __ballot(keep);
            if (keep) dst[o + __popcll(mask & lt)] = u;
            o += __popcll(mask);
        }
    }
}

}  // namespace gnnpe

