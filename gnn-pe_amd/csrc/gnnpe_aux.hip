// gnnpe_aux.hip -- SURVEY 8(f) row 3, second half: the auxiliary index the online side keeps next to every R-tree
// (Partition::build_auxiliary_index, GNN-PE/include/custom.h:268-364; struct Auxiliary_Index custom.h:152-165).
//
// Per node block of a partition's index.dat, indexed by block id:
//   degrees[j]    = max over the paths below the node of the degree of their j-th vertex          (custom.h:276-291, 331-344)
//   label_mbr     = min / max over those paths of pde_label, laid out lo0, hi0, lo1, hi1, ...     (custom.h:293-311, 346-360)
//   key           = 0 - hi_0 - hi_1 - ... - hi_{D-1} of the node's entry in its PARENT, subtracted in that order
//                   (custom.h:324-328); the root has no parent and keeps 0 (custom.h:159)
// The reference computes it on every start of `-m online` by a recursive walk that re-reads every block from the file
// (one `new RTNode` per node).  Here it is a bottom-up pass over the image on the device: one wave per node block, one
// launch per tree level.  It reads the TREE from the image alone (level byte, entry count, entries: rtnode.cpp:1099-1117,
// entry.cpp:127-136), so it serves the bulk-loaded images of gnnpe_index.hip and an index.dat the reference's own
// insert loop wrote alike -- which is how the tests pin it: the same file through the compiled reference and through
// this pass must give the same arrays, bit for bit.
#include "gnnpe_common.h"
#include "gnnpe_dpp.hip.h"

#include <algorithm>
#include <cstring>

namespace gnnpe {

constexpr uint32_t kAuxBlockLen = 4096;  // blk_file.cpp:38: the only block length the reference writes

__device__ __forceinline__ int32_t ld_i32(const char *p)
{
    int32_t v;
    __builtin_memcpy(&v, p, 4);  // entries start at byte 5 of a block: nothing in them is aligned
    return v;
}
__device__ __forceinline__ double ld_f64(const char *p)
{
    double v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
// err[0] = first problem seen (0 none, 1 leaf son outside the partition's paths, 2 child block id outside the file,
// 3 vertex id outside the graph, 4 entry count beyond the block), err[1] = the block it was seen in
__device__ __forceinline__ void aux_fail(uint32_t *err, uint32_t code, uint32_t blk)
{
    if (atomicCAS(&err[0], 0u, code) == 0u) err[1] = blk;
}

// {degree, label} of every vertex side by side: one 8-byte gather per path vertex instead of two 4-byte ones (the leaf
// level issues 1.2e9 of them at config 3 and is bound by their number)
// vmax[0..1]: the largest degree and the largest label (the pair-major build packs both into a record's id bits when they fit)
__global__ void k_aux_pack_vertex(uint32_t n, const uint32_t *__restrict__ degree, const uint32_t *__restrict__ labels,
                                  uint64_t *__restrict__ vdl, uint32_t *__restrict__ vmax)
{
    uint32_t md = 0, ml = 0;
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = degree[v], lab = labels[v];
        vdl[v] = (uint64_t)d | ((uint64_t)lab << 32);
        md = max(md, d);
        ml = max(ml, lab);
    }
    md = dpp_max_u32(md);
    ml = dpp_max_u32(ml);
    if ((threadIdx.x & 63u) == 63u) {  // the reductions end in the last lane; (30 000 atomics on two words took 0.3 ms: look first)
        if (md > __atomic_load_n(vmax, __ATOMIC_RELAXED)) atomicMax(vmax, md);
        if (ml > __atomic_load_n(vmax + 1, __ATOMIC_RELAXED)) atomicMax(vmax + 1, ml);
    }
}

// The leaf level for any path length and embedding width, one wave per node block: entry -> son = index of a path of the
// partition -> its vertices -> their degrees and label features.  It visits every block and lists the inner nodes it passes
// (a few per thousand blocks) for the upper levels' launches (k_aux_upper), which then walk that list instead of the whole
// image (five launches over 5.4 M block headers were 2.5 ms of the pass at config 3).
__global__ __launch_bounds__(256) void k_aux_leaves_any(const char *__restrict__ image, uint32_t n_nodes, uint32_t D, uint32_t L,
                                                        uint32_t e, uint64_t cnt, const uint32_t *__restrict__ tuples, uint32_t n,
                                                        const uint64_t *__restrict__ vdl, const double *__restrict__ xtab,
                                                        uint32_t *__restrict__ adeg, double *__restrict__ ambr,
                                                        uint32_t *__restrict__ err, uint32_t *__restrict__ upper,
                                                        uint32_t *__restrict__ n_upper)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t w0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t esz = 16 * D + 4, cap = (kAuxBlockLen - 5) / esz;
    for (uint64_t b = w0; b < n_nodes; b += nw) {
        const char *blk = image + (b + 1) * (uint64_t)kAuxBlockLen;  // block 0 of the file is the header (blk_file.cpp:110)
        if (blk[0] != 0) {
            if (lane == 0) upper[atomicAdd(n_upper, 1u)] = (uint32_t)b;
            continue;
        }
        const int32_t ne_raw = ld_i32(blk + 1);
        if (ne_raw < 0 || (uint32_t)ne_raw > cap) {
            if (lane == 0) aux_fail(err, 4u, (uint32_t)b);
            continue;
        }
        const uint32_t ne = (uint32_t)ne_raw;
        // One entry per lane; a node holds more than 64 entries only when D <= 3 (cap = 4091 / (16 D + 4)), then the
        // chunks of 64 are combined through lane 0's own earlier stores.  `son` is read once per entry, a path's vertex
        // once per position: degrees[j] and the e label-feature dimensions of position j come from the same gather.
        // (a node without entries -- the empty tree's root leaf -- keeps the constructor's zeros, custom.h:159-164: the
        // arrays were cleared before the first launch)
        for (uint32_t t0 = 0; t0 < ne; t0 += 64) {
            const uint32_t t = t0 + lane;
            bool ok = t < ne;
            uint32_t son = 0;
            if (ok) {
                son = (uint32_t)ld_i32(blk + 5 + (uint64_t)t * esz + 16 * D);
                if (son >= cnt) {
                    aux_fail(err, 1u, (uint32_t)b);
                    ok = false;
                }
            }
            for (uint32_t j = 0; j < L; j++) {
                uint32_t d = 0, lab = 0;
                bool okv = ok;
                if (ok) {
                    const uint32_t v = tuples[(uint64_t)son * L + j];
                    if (v >= n) {
                        aux_fail(err, 3u, (uint32_t)b);
                        okv = false;
                    } else {
                        const uint64_t w = vdl[v];  // {degree, label}: an 8 MB table, unlike the n x e table of x rows
                        d = (uint32_t)w;
                        lab = (uint32_t)(w >> 32);
                    }
                }
                uint32_t dmax = wave_max_u32(d);
                if (lane == 0) {
                    if (t0) dmax = max(dmax, adeg[b * L + j]);
                    adeg[b * L + j] = dmax;
                }
                for (uint32_t kk = 0; kk < e; kk++) {
                    const uint32_t k = j * e + kk;
                    // pde_label (custom.h:561-567) = x of the path's vertices = the label's row of the table
                    // (gen_vde_x, custom.h:492-511): a 1 KB table instead of a random 8-byte gather per dimension
                    double lo = __builtin_huge_val(), hi = -__builtin_huge_val();
                    if (okv) lo = hi = xtab[(uint64_t)lab * e + kk];
                    lo = wave_min(lo);
                    hi = wave_max(hi);
                    if (lane == 0) {
                        if (t0) {
                            lo = fmin(lo, ambr[(b * D + k) * 2]);
                            hi = fmax(hi, ambr[(b * D + k) * 2 + 1]);
                        }
                        ambr[(b * D + k) * 2] = lo;
                        ambr[(b * D + k) * 2 + 1] = hi;
                    }
                }
            }
        }
    }
}

// The upper levels (level >= 1, walking the list the leaf-level launch left), bottom-up, one launch per level: entry -> son =
// child block (already done) -> the child's arrays; the entry's upper bounds give the child's key.  Round 2 reduced one value
// at a time -- load the child's degree or bound, reduce, store, next -- and a node cost a resident wave ~25 us, every load
// waiting for the reduction before it (0.40 ms for the 131 K level-1 nodes of config 3).  Here a lane fetches eight of its child's
// values before the first reduction (all L degrees, then the bounds four dimensions at a time as 16-byte pairs, then the
// entry's own upper bounds for the child's key), so a node is three round trips instead of L + 2D + D.
__global__ __launch_bounds__(256) void k_aux_upper(const char *__restrict__ image, uint32_t n_nodes, int level, uint32_t D,
                                                   uint32_t L, double *__restrict__ key, uint32_t *__restrict__ adeg,
                                                   double *__restrict__ ambr, uint32_t *__restrict__ err,
                                                   const uint32_t *__restrict__ upper, const uint32_t *__restrict__ n_upper)
{
    typedef double dbl2 __attribute__((ext_vector_type(2)));
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t w0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t esz = 16 * D + 4, cap = (kAuxBlockLen - 5) / esz;
    const uint64_t n_visit = (uint64_t)*n_upper;
    for (uint64_t it = w0; it < n_visit; it += nw) {
        const uint64_t b = (uint64_t)upper[it];
        const char *blk = image + (b + 1) * (uint64_t)kAuxBlockLen;
        if ((int)blk[0] != level) continue;
        const int32_t ne_raw = ld_i32(blk + 1);
        if (ne_raw < 0 || (uint32_t)ne_raw > cap) {
            if (lane == 0) aux_fail(err, 4u, (uint32_t)b);
            continue;
        }
        const uint32_t ne = (uint32_t)ne_raw;
        for (uint32_t t0 = 0; t0 < ne; t0 += 64) {  // more than 64 entries only when D <= 3: chunks combine through lane 0's stores
            const uint32_t t = t0 + lane;
            bool ok = t < ne;
            const char *ent = blk + 5 + (uint64_t)(ok ? t : 0) * esz;
            uint32_t son = 0;
            if (ok) {
                son = (uint32_t)ld_i32(ent + 16 * D);
                if (son >= n_nodes) {
                    aux_fail(err, 2u, (uint32_t)b);
                    ok = false;
                    son = 0;
                }
            }
            for (uint32_t j0 = 0; j0 < L; j0 += 8) {
                uint32_t dv[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) dv[i] = (ok && j0 + i < L) ? adeg[(uint64_t)son * L + j0 + i] : 0u;
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    if (j0 + i >= L) break;
                    uint32_t dmax = wave_max_u32(dv[i]);
                    if (lane == 0) {
                        if (t0) dmax = max(dmax, adeg[b * L + j0 + i]);
                        adeg[b * L + j0 + i] = dmax;
                    }
                }
            }
            const dbl2 *cm = reinterpret_cast<const dbl2 *>(ambr) + (uint64_t)son * D;
            for (uint32_t k0 = 0; k0 < D; k0 += 8) {
                dbl2 mv[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    mv[i].x = __builtin_huge_val();
                    mv[i].y = -__builtin_huge_val();
                    if (ok && k0 + i < D) mv[i] = cm[k0 + i];
                }
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    if (k0 + i >= D) break;
                    double lo = wave_min(mv[i].x), hi = wave_max(mv[i].y);
                    if (lane == 0) {
                        const uint64_t at = (b * D + k0 + i) * 2;
                        if (t0) {
                            lo = fmin(lo, ambr[at]);
                            hi = fmax(hi, ambr[at + 1]);
                        }
                        ambr[at] = lo;
                        ambr[at + 1] = hi;
                    }
                }
            }
            double kv = 0.0;
            for (uint32_t k0 = 0; k0 < D; k0 += 8) {
                double h[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) h[i] = k0 + i < D ? ld_f64(ent + (2 * (k0 + i) + 1) * 8) : 0.0;
#pragma unroll
                for (uint32_t i = 0; i < 8; i++)
                    if (k0 + i < D) kv -= h[i];  // custom.h:324-328, same order
            }
            if (ok) key[son] = kv;
        }
    }
}

// The leaf level again, for compile-time path length and embedding width (what gnnpe_count_paths produces: L = 3 or 4,
// e with a specialised emit kernel).  The generic kernel above walks a leaf as twelve dependent steps -- header, entry
// count, son, then per path position vertex -> degree / label -> feature row -> a wave-wide reduction -- and a leaf cost a
// resident wave ~18 us (5.3 M leaves: 11.7 of the pass's 14.4 ms at config 3).  Here the header is one load, the vertices
// of all positions are fetched together, then their {degree, label} words, and the L + 2 L e reductions are unrolled
// side by side on DPP.  Measured at config 3 (bench.py index_build.aux_index_ms): 14.4 -> 13.4 (label table instead of x
// gathers) -> 12.7 (this kernel) -> 11.7 (one gather per vertex) -> 10.3 ms (DPP instead of ds_bpermute) -> 8.2 (upper
// levels from a list); staging the block through LDS with coalesced loads changed nothing (10.4).
template <int LC, int EC>
__global__ __launch_bounds__(256) void k_aux_leaves(const char *__restrict__ image, uint32_t n_nodes, uint64_t cnt,
                                                    const uint32_t *__restrict__ tuples, uint32_t n,
                                                    const uint64_t *__restrict__ vdl,
                                                    const double *__restrict__ xtab, uint32_t *__restrict__ adeg,
                                                    double *__restrict__ ambr, uint32_t *__restrict__ err,
                                                    uint32_t *__restrict__ upper, uint32_t *__restrict__ n_upper)
{
    constexpr int D = LC * EC;
    constexpr uint32_t esz = 16 * D + 4, cap = (kAuxBlockLen - 5) / esz;
    static_assert(cap <= 64, "one entry per lane");
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t w0 = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t b = w0; b < n_nodes; b += nw) {
        const char *blk = image + (b + 1) * (uint64_t)kAuxBlockLen;
        const uint64_t h = *reinterpret_cast<const uint64_t *>(blk);  // level (1 byte), entry count (4 bytes): blocks are 4 KiB aligned
        if ((int)(int8_t)(h & 0xFFu) != 0) {  // an inner node: listed for the launches of the upper levels
            if (lane == 0) upper[atomicAdd(n_upper, 1u)] = (uint32_t)b;
            continue;
        }
        const uint32_t ne = (uint32_t)(h >> 8);
        if (ne > cap) {
            if (lane == 0) aux_fail(err, 4u, (uint32_t)b);
            continue;
        }
        if (ne == 0) continue;  // the empty tree's root leaf keeps the constructor's zeros (custom.h:159-164)
        bool ok = lane < ne;
        uint32_t son = 0;
        if (ok) {
            son = (uint32_t)ld_i32(blk + 5 + (uint64_t)lane * esz + 16 * D);
            if (son >= cnt) {
                aux_fail(err, 1u, (uint32_t)b);
                ok = false;
            }
        }
        uint32_t v[LC], d[LC], lab[LC];
#pragma unroll
        for (int j = 0; j < LC; j++) v[j] = ok ? tuples[(uint64_t)son * LC + j] : 0u;
#pragma unroll
        for (int j = 0; j < LC; j++) {
            if (ok && v[j] >= n) {
                aux_fail(err, 3u, (uint32_t)b);
                ok = false;
            }
        }
#pragma unroll
        for (int j = 0; j < LC; j++) {
            const uint64_t w = ok ? vdl[v[j]] : 0ull;
            d[j] = (uint32_t)w;
            lab[j] = (uint32_t)(w >> 32);
        }
        double lo[D], hi[D];
#pragma unroll
        for (int k = 0; k < D; k++) {
            const double pl = ok ? xtab[(uint64_t)lab[k / EC] * EC + k % EC] : 0.0;  // pde_label (custom.h:561-567)
            lo[k] = ok ? pl : __builtin_huge_val();
            hi[k] = ok ? pl : -__builtin_huge_val();
        }
#pragma unroll
        for (int j = 0; j < LC; j++) d[j] = dpp_max_u32(d[j]);
#pragma unroll
        for (int k = 0; k < D; k++) {
            lo[k] = dpp_min_f64(lo[k]);
            hi[k] = dpp_max_f64(hi[k]);
        }
        if (lane == 63) {  // the reductions end in the last lane
#pragma unroll
            for (int k = 0; k < D; k++) {
                ambr[(b * D + k) * 2] = lo[k];
                ambr[(b * D + k) * 2 + 1] = hi[k];
            }
#pragma unroll
            for (int j = 0; j < LC; j++) adeg[b * LC + j] = d[j];
        }
    }
}

int ensure_vertex_words(gnnpe_ctx *c)
{
    GNNPE_REQUIRE(c->rows_identity || c->have_deg_all, GNNPE_ERR_UNSUPPORTED,
                  "the auxiliary index needs every vertex' degree: load the whole graph (gnnpe_load_csr) or call gnnpe_set_degrees");
    int rc;
    if (c->aux_vdl_valid) return GNNPE_OK;  // degrees and labels belong to the graph: once per load (or gnnpe_set_degrees)
    if ((rc = c->aux_vdl.reserve(((size_t)c->n + 1) * 8 + 8))) return rc;
    const uint32_t *deg = c->have_deg_all ? c->deg_all.as<uint32_t>() : c->adj_deg.as<uint32_t>();
    uint32_t *vmax = reinterpret_cast<uint32_t *>(c->aux_vdl.as<uint64_t>() + c->n);  // {largest degree, largest label} behind the table
    GNNPE_HIP_TRY(hipMemsetAsync(vmax, 0, 8, c->stream));
    if (c->n)
        hipLaunchKernelGGL(k_aux_pack_vertex, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, c->n, deg, c->labels.as<uint32_t>(),
                           c->aux_vdl.as<uint64_t>(), vmax);
    GNNPE_HIP_TRY(hipGetLastError());
    c->aux_vdl_valid = true;
    return GNNPE_OK;
}

}  // namespace gnnpe

using namespace gnnpe;

extern "C" {

int gnnpe_aux_index_device(gnnpe_ctx *c, const void *dev_image, uint64_t nbytes, uint64_t cnt, uint32_t L,
                           const void *dev_tuples, void **dev_key, void **dev_degrees, void **dev_label_mbr,
                           uint32_t *n_nodes_out, uint32_t *dim_out)
{
    GNNPE_REQUIRE(c && dev_image && dev_key && dev_degrees && dev_label_mbr && n_nodes_out, GNNPE_ERR_ARG, "null argument");
    GNNPE_REQUIRE(c->have_graph && c->have_table && c->have_vde, GNNPE_ERR_ARG,
                  "gnnpe_aux_index_device: need the graph, the label table and gnnpe_vde (label features x)");
    GNNPE_REQUIRE(c->rows_identity || c->have_deg_all, GNNPE_ERR_UNSUPPORTED,
                  "the auxiliary index needs every vertex' degree: load the whole graph (gnnpe_load_csr) or call gnnpe_set_degrees");
    GNNPE_REQUIRE(L >= 1 && (cnt == 0 || dev_tuples), GNNPE_ERR_ARG, "null path tuples");
    GNNPE_REQUIRE(nbytes >= 2 * (uint64_t)kAuxBlockLen && nbytes % kAuxBlockLen == 0, GNNPE_ERR_ARG,
                  "index image of %llu bytes is not a whole number (>= 2) of %u-byte blocks", (unsigned long long)nbytes, kAuxBlockLen);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    // header block: blocklength, number of node blocks (blk_file.cpp:38-39), then dimension ... root (rtree.cpp:341-362)
    char h[32];
    GNNPE_HIP_TRY(hipMemcpyAsync(h, dev_image, 32, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    int32_t blen, nblk, dim, root;
    memcpy(&blen, h, 4);
    memcpy(&nblk, h + 4, 4);
    memcpy(&dim, h + 8, 4);
    memcpy(&root, h + 25, 4);
    GNNPE_REQUIRE(blen == (int32_t)kAuxBlockLen, GNNPE_ERR_ARG, "index image: block length %d, expected %u", blen, kAuxBlockLen);
    GNNPE_REQUIRE(nblk >= 1 && ((uint64_t)nblk + 1) * kAuxBlockLen <= nbytes, GNNPE_ERR_ARG,
                  "index image: header counts %d node blocks, the image holds %llu", nblk,
                  (unsigned long long)(nbytes / kAuxBlockLen - 1));
    GNNPE_REQUIRE(dim == (int32_t)(L * c->e), GNNPE_ERR_ARG, "index image: dimension %d, paths of %u vertices x e = %u give %u", dim, L,
                  c->e, L * c->e);
    GNNPE_REQUIRE(root >= 0 && root < nblk, GNNPE_ERR_ARG, "index image: root block %d outside [0, %d)", root, nblk);
    const uint32_t D = (uint32_t)dim, N = (uint32_t)nblk;
    char root_level = 0;
    GNNPE_HIP_TRY(hipMemcpyAsync(&root_level, (const char *)dev_image + ((uint64_t)root + 1) * kAuxBlockLen, 1,
                                 hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    GNNPE_REQUIRE(root_level >= 0 && root_level < 32, GNNPE_ERR_ARG, "index image: root level %d", (int)root_level);

    int rc;
    c->img_aux_valid = false;  // the arrays are about to hold this image's index
    if ((rc = c->aux_key.reserve(((size_t)N + 1) * 8)) || (rc = c->aux_deg.reserve(((size_t)N * L + 1) * 4)) ||
        (rc = c->aux_mbr.reserve(((size_t)N * 2 * D + 1) * 8)))
        return rc;
    GNNPE_HIP_TRY(hipMemsetAsync(c->aux_key.p, 0, (size_t)N * 8, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(c->aux_deg.p, 0, (size_t)N * L * 4, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(c->aux_mbr.p, 0, (size_t)N * 2 * D * 8, c->stream));
    uint32_t *d_err = c->small.as<uint32_t>() + 600;  // bytes 2400..2411 of the context's small buffer: error code, block, list length
    uint32_t *d_nup = d_err + 2;
    GNNPE_HIP_TRY(hipMemsetAsync(d_err, 0, 12, c->stream));
    if ((rc = c->aux_upper.reserve(((size_t)N + 1) * 4))) return rc;
    if ((rc = ensure_vertex_words(c))) return rc;
    // the leaf level -- almost all of the nodes -- through the specialised kernel where there is one (D >= 4: one entry per lane)
    int first_generic = 0;
#define GNNPE_AUXL(LL, EE)                                                                                               \
    if (first_generic == 0 && L == LL && c->e == EE) {                                                                   \
        hipLaunchKernelGGL((k_aux_leaves<LL, EE>), dim3(grid_for((uint64_t)N * 64)), dim3(kBlock), 0, c->stream,           \
                           (const char *)dev_image, N, cnt, (const uint32_t *)dev_tuples, c->n, c->aux_vdl.as<uint64_t>(),    \
                           c->xtab.as<double>(), c->aux_deg.as<uint32_t>(), c->aux_mbr.as<double>(), d_err,                \
                           c->aux_upper.as<uint32_t>(), d_nup);                                                          \
        first_generic = 1;                                                                                               \
    }
    GNNPE_AUXL(3, 2) GNNPE_AUXL(3, 3) GNNPE_AUXL(3, 4) GNNPE_AUXL(3, 8) GNNPE_AUXL(4, 1) GNNPE_AUXL(4, 2) GNNPE_AUXL(4, 3) GNNPE_AUXL(4, 4) GNNPE_AUXL(4, 8)
#undef GNNPE_AUXL
    if (first_generic == 0)  // the leaf level of any other (L, e), one value at a time; it lists the inner nodes as well
        hipLaunchKernelGGL(k_aux_leaves_any, dim3(grid_for((uint64_t)N * 64)), dim3(kBlock), 0, c->stream, (const char *)dev_image, N, D,
                           L, c->e, cnt, (const uint32_t *)dev_tuples, c->n, c->aux_vdl.as<uint64_t>(), c->xtab.as<double>(),
                           c->aux_deg.as<uint32_t>(), c->aux_mbr.as<double>(), d_err, c->aux_upper.as<uint32_t>(), d_nup);
    for (int level = 1; level <= (int)root_level; level++)
        hipLaunchKernelGGL(k_aux_upper, dim3(grid_for((uint64_t)N * 64 / 32 + 64)), dim3(kBlock), 0, c->stream, (const char *)dev_image, N,
                           level, D, L, c->aux_key.as<double>(), c->aux_deg.as<uint32_t>(), c->aux_mbr.as<double>(), d_err,
                           c->aux_upper.as<uint32_t>(), d_nup);
    GNNPE_HIP_TRY(hipGetLastError());
    uint32_t err[2] = {0, 0};
    GNNPE_HIP_TRY(hipMemcpyAsync(err, d_err, 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    static const char *const what[] = {"", "a leaf entry points outside the partition's paths", "a child block id lies outside the file",
                                       "a path holds a vertex id outside the graph", "an entry count exceeds the block's capacity"};
    GNNPE_REQUIRE(err[0] == 0, GNNPE_ERR_ARG, "index image, node block %u: %s", err[1], what[std::min(err[0], 4u)]);
    *dev_key = c->aux_key.p;
    *dev_degrees = c->aux_deg.p;
    *dev_label_mbr = c->aux_mbr.p;
    *n_nodes_out = N;
    if (dim_out) *dim_out = D;
    return GNNPE_OK;
}

}  // extern "C"
