// gnnpe_index.hip -- R6: bulk-loaded R-tree over a partition's path embeddings, emitted directly in
// the reference's on-disk format (index.dat).
//
// The reference builds the tree with one RTree::insert per path (custom.h:235-257 ->
// libsrc/rtree/rtree.cpp:198-284): ~96 us per insert, each walking root->leaf through 4 KiB block
// reads.  The online consumer (custom.h:258-266, 366-489) only needs a VALID tree
// in the same file format: dense node block ids, an internal root, child MBRs enclosed by their
// parent entries, leaf `son` = the path's index inside the partition.  Tree SHAPE is free, so here it
// is bulk-loaded: Z-order key of the (quantised) embedding -> one device radix sort -> leaves packed
// from consecutive runs -> upper levels packed from consecutive children, every node block
// assembled in LDS and written as one aligned 4 KiB store.
//
// File layout (little endian, packed) restated from the reference:
//   block 0 : int32 blocklength, int32 n_node_blocks                      blk_file.cpp:38-39,51-52
//             @8: int32 dim, n_data, n_dnodes, n_inodes, bool root_is_data, int32 root   rtree.cpp:341-362
//   block k+1 = node k: char level, int32 n_entries, entries               rtnode.cpp:1099-1117
//   entry   : 2*dim doubles (lo0,hi0,lo1,hi1,...), int32 son              entry.cpp:127-136
//   capacity = (4096 - 5) / (16*dim + 4)                                   rtnode.cpp:27-28
#include <hipcub/hipcub.hpp>

#include <fcntl.h>
#include <unistd.h>

#include <cstdio>  // rename

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "gnnpe_common.h"
#include "gnnpe_dpp.hip.h"
#include "gnnpe_records.h"

namespace gnnpe {

constexpr int kBlockLen = 4096;

// Where the leaf rectangles come from: path tuples + the vde table (GNN-PE: point MBRs lo = hi =
// pde row, custom.h:244-248) or an explicit array of boxes (GNN-PGE: a vertex' path_group,
// GNN-PGE/include/custom.h:171-177), cnt x 2D doubles laid out (lo0, hi0, lo1, hi1, ...).
struct LeafSrc {
    const uint32_t *vids;
    const double *vde;
    const double *boxes;
    uint32_t L, e, D;
    __device__ __forceinline__ void get(uint64_t p, uint32_t k, double *lo, double *hi) const
    {
        if (boxes) {
            *lo = boxes[(p * D + k) * 2];
            *hi = boxes[(p * D + k) * 2 + 1];
        } else {
            *lo = *hi = vde[(uint64_t)vids[p * L + k / e] * e + k % e];
        }
    }
};

// ---- keys ---------------------------------------------------------------------------------------
// per-dimension min / max of the partition's points (point p, dim k = vde[vids[p][k / e]][k % e])
__global__ void k_point_minmax(uint64_t cnt, LeafSrc S, double *__restrict__ mn, double *__restrict__ mx)
{
    // one block per dimension-slice of the input; partial results combined with atomics on the
    // ordered-integer image of the doubles (all embeddings are positive finite numbers)
    const uint32_t D = S.D;
    __shared__ double s_mn[256], s_mx[256];
    for (uint32_t k = 0; k < D; k++) {
        double a = 1e300, b = -1e300;
        for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < cnt; p += (uint64_t)gridDim.x * blockDim.x) {
            double lo, hi;
            S.get(p, k, &lo, &hi);
            const double v = 0.5 * (lo + hi);  // key on the rectangle's centre
            a = fmin(a, v);
            b = fmax(b, v);
        }
        s_mn[threadIdx.x] = a;
        s_mx[threadIdx.x] = b;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                s_mn[threadIdx.x] = fmin(s_mn[threadIdx.x], s_mn[threadIdx.x + s]);
                s_mx[threadIdx.x] = fmax(s_mx[threadIdx.x], s_mx[threadIdx.x + s]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            // non-negative doubles order like their bit patterns
            atomicMin(reinterpret_cast<unsigned long long *>(mn) + k, (unsigned long long)__double_as_longlong(s_mn[0]));
            atomicMax(reinterpret_cast<unsigned long long *>(mx) + k, (unsigned long long)__double_as_longlong(s_mx[0]));
        }
        __syncthreads();
    }
}

// Z-order key: `bits` bits per dimension, most significant bit plane first; within a plane dimension 0
// is the most significant.  Bit t of dimension k lands at position t*D + (D-1-k).
__global__ void k_zorder_keys(uint64_t cnt, LeafSrc S, uint32_t bits, const double *__restrict__ mn,
                              const double *__restrict__ mx, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t D = S.D;
    const uint32_t qmax = (1u << bits) - 1u;
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < cnt; p += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t key = 0;
        for (uint32_t k = 0; k < D; k++) {
            double lo, hi;
            S.get(p, k, &lo, &hi);
            const double v = 0.5 * (lo + hi);
            const double span = mx[k] - mn[k];
            const uint32_t q = span > 0.0 ? (uint32_t)fmin((v - mn[k]) / span * (double)qmax, (double)qmax) : 0u;
            for (uint32_t t = 0; t < bits; t++) key |= (uint64_t)((q >> t) & 1u) << (t * D + (D - 1 - k));
        }
        keys[p] = key;
        vals[p] = (uint32_t)p;
    }
}

// ---- node assembly --------------------------------------------------------------------------------
__device__ __forceinline__ void lds_put(char *dst, const void *src, int n)
{
    const char *s = reinterpret_cast<const char *>(src);
    for (int i = 0; i < n; i++) dst[i] = s[i];
}

// leaf j holds sorted points [j*F, min((j+1)*F, cnt)); block id = j.  One workgroup per leaf.
__global__ __launch_bounds__(64) void k_pack_leaves(uint64_t cnt, uint32_t F, LeafSrc S,
                                                    const uint32_t *__restrict__ order, char *__restrict__ image,
                                                    double *__restrict__ node_mbr)
{
    __shared__ __attribute__((aligned(16))) char s_blk[kBlockLen];
    __shared__ double s_lo[64], s_hi[64];
    const uint32_t D = S.D, esz = 16 * D + 4;
    const uint64_t j = blockIdx.x;
    const uint64_t p0 = j * F;
    const uint32_t ne = (uint32_t)min((uint64_t)F, cnt - p0);
    for (uint32_t i = threadIdx.x; i < kBlockLen / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(s_blk)[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_blk[0] = 0;  // level 0 = leaf
        const int32_t n32 = (int32_t)ne;
        lds_put(s_blk + 1, &n32, 4);
    }
    if (threadIdx.x < ne) {
        const uint32_t son = order[p0 + threadIdx.x];  // the path's index inside the partition (custom.h:243)
        char *ent = s_blk + 5 + threadIdx.x * esz;
        for (uint32_t k = 0; k < D; k++) {
            double lo, hi;
            S.get(son, k, &lo, &hi);
            lds_put(ent + 16 * k, &lo, 8);      // bounces[2k]   (custom.h:246)
            lds_put(ent + 16 * k + 8, &hi, 8);  // bounces[2k+1] (custom.h:247)
        }
        const int32_t s32 = (int32_t)son;
        lds_put(ent + 16 * D, &s32, 4);
    }
    // node MBR for the parent level
    for (uint32_t k = 0; k < D; k++) {
        double lo = 1e300, hi = -1e300;
        if (threadIdx.x < ne) S.get(order[p0 + threadIdx.x], k, &lo, &hi);
        s_lo[threadIdx.x] = lo;
        s_hi[threadIdx.x] = hi;
        __syncthreads();
        for (int s = 32; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                s_lo[threadIdx.x] = fmin(s_lo[threadIdx.x], s_lo[threadIdx.x + s]);
                s_hi[threadIdx.x] = fmax(s_hi[threadIdx.x], s_hi[threadIdx.x + s]);
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            node_mbr[(j * D + k) * 2] = s_lo[0];
            node_mbr[(j * D + k) * 2 + 1] = s_hi[0];
        }
        __syncthreads();
    }
    uint32_t *dst = reinterpret_cast<uint32_t *>(image + (j + 1) * (uint64_t)kBlockLen);  // node j -> file block j+1
    for (uint32_t i = threadIdx.x; i < kBlockLen / 4; i += blockDim.x) dst[i] = reinterpret_cast<uint32_t *>(s_blk)[i];
}

// internal node j of a level: children = nodes [child0 + j*F, child0 + min((j+1)*F, n_child)) of the
// level below; block id = node0 + j.  One wave per node (a resident grid of four-wave workgroups), one lane per child entry.  An entry (16D bytes of bounces + the
// son's block id) starts at byte 5 + t * (16D + 4) of the block: always one byte past a dword boundary, so the lane shifts
// its dwords by three bytes on the way into the LDS image (a byte and a half-word in front, whole dwords, a byte behind;
// round 2 moved every entry a byte at a time: 0.42-0.54 ms for the 131 K level-1 nodes of config 3); the node's own
// bounding box is a DPP min / max over the lanes.
// With `adeg` (the pair-major build with the auxiliary index): also the node's rows of Partition::build_auxiliary_index
// (custom.h:268-364) from its children's -- max degree per path position, label-feature box, and every child's key = minus the
// sum of its entry's upper bounds, dimensions in ascending order (custom.h:324-328) -- so that no pass reads the image back.
__global__ __launch_bounds__(256) void k_pack_inner(uint64_t n_nodes, uint64_t n_child, uint64_t child0, uint64_t node0, uint32_t F,
                                                    uint32_t D, int level, const double *__restrict__ child_mbr,
                                                    char *__restrict__ image, double *__restrict__ node_mbr, uint32_t L,
                                                    double *__restrict__ key, uint32_t *__restrict__ adeg, double *__restrict__ ambr)
{
    __shared__ __attribute__((aligned(16))) char s_all[4][kBlockLen];
    char *s_blk = s_all[threadIdx.x >> 6];
    const uint32_t esz = 16 * D + 4, t = threadIdx.x & 63u;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t j = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6; j < n_nodes; j += nw) {  // a wave per node
        const uint64_t c0 = j * F;
        const uint32_t ne = (uint32_t)min((uint64_t)F, n_child - c0);
        for (uint32_t i = t; i < kBlockLen / 16; i += 64) reinterpret_cast<uint4 *>(s_blk)[i] = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t == 0) {
            s_blk[0] = (char)level;
            const int32_t n32 = (int32_t)ne;
            lds_put(s_blk + 1, &n32, 4);
        }
        // the entry's (lo, hi) pairs, eight dimensions' loads in flight at a time; every dword but the first goes out joined to
        // the top byte of the one before it; the node's own box is reduced from the same registers
        const bool live = t < ne;
        const uint4 *src = reinterpret_cast<const uint4 *>(child_mbr + (c0 + (live ? t : 0)) * 2 * D);
        const uint32_t son = (uint32_t)(child0 + c0 + t);  // the child's block id
        char *ent = s_blk + 5 + t * esz;
        uint32_t *out = reinterpret_cast<uint32_t *>(ent + 3);  // dword-aligned
        uint32_t prev = 0, m = 0;
        double kv = 0.0;
        auto join = [&](uint32_t w) {
            if (live) out[m] = (prev >> 24) | (w << 8);
            m++;
            prev = w;
        };
        for (uint32_t q0 = 0; q0 < D; q0 += 8) {
            uint4 buf[8];
#pragma unroll
            for (uint32_t i = 0; i < 8; i++)
                if (q0 + i < D) buf[i] = src[q0 + i];
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) {
                if (q0 + i >= D) break;
                const uint4 v = buf[i];
                if (q0 + i == 0) {
                    if (live) {
                        ent[0] = (char)(v.x & 0xFFu);
                        *reinterpret_cast<uint16_t *>(ent + 1) = (uint16_t)(v.x >> 8);
                    }
                    prev = v.x;
                } else {
                    join(v.x);
                }
                join(v.y);
                join(v.z);
                join(v.w);
                const double my_hi = __longlong_as_double((long long)(((uint64_t)v.w << 32) | v.z));
                kv -= my_hi;
                const double lo = wave_min(live ? __longlong_as_double((long long)(((uint64_t)v.y << 32) | v.x)) : 1e300);
                const double hi = wave_max(live ? my_hi : -1e300);
                if (t == 0) {
                    node_mbr[(j * D + q0 + i) * 2] = lo;
                    node_mbr[(j * D + q0 + i) * 2 + 1] = hi;
                }
            }
        }
        join(son);
        if (live) ent[esz - 1] = (char)(son >> 24);
        if (adeg) {
            typedef double dbl2 __attribute__((ext_vector_type(2)));
            const uint64_t b = node0 + j, sn = live ? son : 0u;
            if (live) key[son] = kv;
            for (uint32_t j0 = 0; j0 < L; j0 += 8) {
                uint32_t dv[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) dv[i] = (live && j0 + i < L) ? adeg[sn * L + j0 + i] : 0u;
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    if (j0 + i >= L) break;
                    const uint32_t dmax = wave_max_u32(dv[i]);
                    if (t == 0) adeg[b * L + j0 + i] = dmax;
                }
            }
            const dbl2 *cm = reinterpret_cast<const dbl2 *>(ambr) + sn * D;
            for (uint32_t k0 = 0; k0 < D; k0 += 8) {
                dbl2 mv[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    mv[i].x = __builtin_huge_val();
                    mv[i].y = -__builtin_huge_val();
                    if (live && k0 + i < D) mv[i] = cm[k0 + i];
                }
#pragma unroll
                for (uint32_t i = 0; i < 8; i++) {
                    if (k0 + i >= D) break;
                    const double lo = wave_min(mv[i].x), hi = wave_max(mv[i].y);
                    if (t == 0) {
                        ambr[(b * D + k0 + i) * 2] = lo;
                        ambr[(b * D + k0 + i) * 2 + 1] = hi;
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *dst = reinterpret_cast<u32x4 *>(image + (node0 + j + 1) * (uint64_t)kBlockLen);
        for (uint32_t i = t; i < kBlockLen / 16; i += 64) __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(s_blk)[i], dst + i);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// ---- fast path: 3-vertex paths with a compile-time embedding width (D = 3E) -------------------------
// Same keys, same leaves and same bytes as the generic kernels above (tests/test_gpu_index.py builds
// one partition both ways), but each point's coordinates are gathered once into registers, the two
// passes over the points are single passes, and one wave assembles a leaf with dword LDS traffic only.
template <int E> struct PathPoints {
    static constexpr int D = 3 * E;
    const uint32_t *vids;
    const double *vde;
    __device__ __forceinline__ uint3 ids(uint64_t p) const { return make_uint3(vids[p * 3], vids[p * 3 + 1], vids[p * 3 + 2]); }
    __device__ __forceinline__ void coords(uint3 t, double (&v)[D]) const
    {
#pragma unroll
        for (int k = 0; k < E; k++) {
            v[k] = vde[(uint64_t)t.x * E + k];
            v[E + k] = vde[(uint64_t)t.y * E + k];
            v[2 * E + k] = vde[(uint64_t)t.z * E + k];
        }
    }
};

__global__ void k_minmax_init(double *__restrict__ mn, double *__restrict__ mx)
{
    mn[threadIdx.x] = 1e300;
    mx[threadIdx.x] = 0.0;
}

// ---- label-major sort key for 3-vertex paths ------------------------------------------------------
// The online traversal opens a node only if the query path's label embedding lies inside the node's
// label MBR AND the node's upper corner dominates the query's pde (custom.h:441-462); a leaf entry
// matches only on identical labels (custom.h:408-413).  So leaves should hold ONE label triple where
// possible, and be compact in pde inside it (scripts/index_key_study.py: 27 instead of 13 000 leaves
// opened per query at 64 labels, 76 instead of 14 700 at 8):
//   key = [label(a) | label(b) | label(c)]  (3 x lb bits)  then  Z-order of the path's 3E coordinates,
//         each quantised to zb bits by its RANK among the vertices' values (equi-depth: skew-proof).
// Both parts are per-vertex: vkey[v] = {low 32: the vertex' zb*E quantised bits spread to their
// Z-order positions for the LAST path position, high 32: label}; position j shifts the low part left
// by (2-j)*E.  A path key is three 8-byte gathers from an n x 8 byte table and a few shifts.
__global__ void k_vkey_labels(uint32_t n, const uint32_t *__restrict__ labels, uint64_t *__restrict__ vkey)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x)
        vkey[v] = (uint64_t)labels[v] << 32;
}
__global__ void k_vkey_column(uint32_t n, uint32_t e, uint32_t comp, const double *__restrict__ vde,
                              uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x) {
        keys[v] = (uint64_t)__double_as_longlong(vde[v * e + comp]);  // embeddings are positive: bit order = value order
        vals[v] = (uint32_t)v;
    }
}
// sorted position r of component `comp` -> zb-bit bucket -> bits spread at stride D = 3e
__global__ void k_vkey_spread(uint32_t n, uint32_t e, uint32_t comp, uint32_t zb, const uint32_t *__restrict__ sorted_v,
                              uint64_t *__restrict__ vkey)
{
    const uint32_t D = 3 * e;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t q = (uint32_t)((r << zb) / n);
        uint64_t s = 0;
        for (uint32_t t = 0; t < zb; t++) s |= (uint64_t)((q >> t) & 1u) << (t * D + (e - 1 - comp));
        vkey[sorted_v[r]] |= s;  // one writer per vertex per launch
    }
}
// The table is narrowed to 32-bit words {label << sbits | spread bits} when both parts fit (n x 4 bytes
// then stays L2-resident at a million vertices).
__global__ void k_vkey_narrow(uint32_t n, uint32_t sbits, const uint64_t *__restrict__ vkey, uint32_t *__restrict__ vkey32)
{
    for (uint64_t v = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; v < n; v += (uint64_t)gridDim.x * blockDim.x)
        vkey32[v] = (uint32_t)((vkey[v] >> 32) << sbits) | (uint32_t)vkey[v];
}
template <typename KeyT, typename WordT>
__global__ __launch_bounds__(256) void k_path_keys(uint64_t cnt, const uint32_t *__restrict__ vids,
                                                   const WordT *__restrict__ vkey, uint32_t e, uint32_t lb, uint32_t sbits,
                                                   uint32_t zbits, KeyT *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint64_t lmask = (1ull << lb) - 1ull, smask = (1ull << sbits) - 1ull;
    for (uint64_t p = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; p < cnt; p += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t ka = vkey[vids[p * 3]], kb = vkey[vids[p * 3 + 1]], kc = vkey[vids[p * 3 + 2]];
        const uint64_t lab = ((((ka >> sbits) & lmask) << lb | ((kb >> sbits) & lmask)) << lb) | ((kc >> sbits) & lmask);
        const uint64_t z = ((ka & smask) << (2 * e)) | ((kb & smask) << e) | (kc & smask);
        keys[p] = (KeyT)((lab << zbits) | z);
        vals[p] = (uint32_t)p;
    }
}

// One wave per leaf, kLeafWaves leaves in flight per workgroup, every wave looping over leaves.
// The node block is assembled in a wave-private LDS window shifted by 3 bytes, so that the packed
// format's odd offsets (entries start at byte 5, rtnode.cpp:1099-1117) fall on dword boundaries:
//   window byte 3 = level, dword 1 = n_entries, entry i = dwords [2 + i*(4D+1), 2 + (i+1)*(4D+1))
// and file dword w = (window dword w >> 24) | (window dword w+1 << 8).
// A point is three dependent random reads (sort order -> path tuple -> vde rows); the loop keeps three
// leaves in flight per wave, one per stage, so an iteration costs one memory latency instead of three.
// Lanes beyond a leaf's entry count (and waves past the last leaf) read element 0: harmless, branch-free.
constexpr int kLeafWaves = 4;
template <int E>
__global__ __launch_bounds__(64 * kLeafWaves) void k_pack_leaves_paths(uint64_t cnt, uint64_t n_leaves, uint32_t F,
                                                                        PathPoints<E> S,
                                                                        const uint32_t *__restrict__ order,
                                                                        char *__restrict__ image,
                                                                        double *__restrict__ node_mbr)
{
    constexpr int D = 3 * E;
    constexpr int kWin = kBlockLen / 4 + 4;  // dwords; the last window dwords feed the shifted reads of dword 1023
    constexpr int kEnt = 4 * D + 1;          // dwords per entry
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kLeafWaves][kWin];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t *w = s_win[wv];
    const uint64_t stride = (uint64_t)gridDim.x * kLeafWaves;
    auto entries = [&](uint64_t jj) -> uint32_t { return jj < n_leaves ? (uint32_t)min((uint64_t)F, cnt - jj * F) : 0u; };
    auto son_of = [&](uint64_t jj) -> uint32_t { return (uint32_t)lane < entries(jj) ? order[jj * F + lane] : 0u; };

    uint64_t j = blockIdx.x * (uint64_t)kLeafWaves + wv;
    uint32_t son_a = son_of(j), son_b = son_of(j + stride), son_c = son_of(j + 2 * stride);
    uint3 ids_b = S.ids(son_b);
    double v[D];
    S.coords(S.ids(son_a), v);
    for (; j < n_leaves; j += stride) {
        // issue the next stage of the two leaves behind this one before touching this leaf's data
        const uint32_t son_d = son_of(j + 3 * stride);
        const uint3 ids_c = S.ids(son_c);
        double v_b[D];
        S.coords(ids_b, v_b);

        const uint32_t ne = entries(j);
        if (lane == 0) {
            w[0] = 0u;  // level 0 = leaf (byte 3 of the window)
            w[1] = ne;
        }
        for (uint32_t i = 2 + ne * kEnt + lane; i < (uint32_t)kWin; i += 64) w[i] = 0u;  // zero tail
        if ((uint32_t)lane < ne) {
            uint32_t *ent = w + 2 + lane * kEnt;
#pragma unroll
            for (int k = 0; k < D; k++) {
                const uint64_t bits64 = (uint64_t)__double_as_longlong(v[k]);
                const uint32_t x = (uint32_t)bits64, y = (uint32_t)(bits64 >> 32);
                ent[4 * k] = x;      // bounces[2k]   (custom.h:246)
                ent[4 * k + 1] = y;
                ent[4 * k + 2] = x;  // bounces[2k+1] (custom.h:247)
                ent[4 * k + 3] = y;
            }
            ent[4 * D] = son_a;  // the path's index inside the partition (custom.h:243)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // node MBR for the parent level: lane k scans dimension k of the assembled entries (cross-lane
        // reductions of 2D doubles cost ~20k cycles per leaf in ds_bpermute round trips; this is ~40 LDS reads)
        if (lane < D) {
            double lo = 1e300, hi = -1e300;
#pragma unroll 8
            for (uint32_t i = 0; i < ne; i++) {
                const uint32_t *q = w + 2 + i * kEnt + 4 * lane;
                const double x = __longlong_as_double((long long)(((uint64_t)q[1] << 32) | q[0]));
                lo = fmin(lo, x);
                hi = fmax(hi, x);
            }
            node_mbr[(j * D + lane) * 2] = lo;
            node_mbr[(j * D + lane) * 2 + 1] = hi;
        }
        // node j -> file block j+1, 16 bytes per lane per round, streamed past the caches
        uint4 *dst = reinterpret_cast<uint4 *>(image + (j + 1) * (uint64_t)kBlockLen);
#pragma unroll
        for (int r = 0; r < kBlockLen / 16 / 64; r++) {
            const int c = lane + 64 * r;
            const uint4 lo4 = *reinterpret_cast<const uint4 *>(w + 4 * c);
            const uint32_t nx = w[4 * c + 4];
            uint4 o;
            o.x = (lo4.x >> 24) | (lo4.y << 8);
            o.y = (lo4.y >> 24) | (lo4.z << 8);
            o.z = (lo4.z >> 24) | (lo4.w << 8);
            o.w = (lo4.w >> 24) | (nx << 8);
            __builtin_nontemporal_store(o.x, &dst[c].x);
            __builtin_nontemporal_store(o.y, &dst[c].y);
            __builtin_nontemporal_store(o.z, &dst[c].z);
            __builtin_nontemporal_store(o.w, &dst[c].w);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // rotate the pipeline
        son_a = son_b;
        son_b = son_c;
        son_c = son_d;
        ids_b = ids_c;
#pragma unroll
        for (int k = 0; k < D; k++) v[k] = v_b[k];
    }
}

// ---- pair-major build: the partition's index straight from the enumeration's records ---------------------------------
// The builder above starts from a tuple array and pays one random 12-byte read (a 128-byte line fill) and three vde row
// gathers per point: 81 GB of line fills for 2.0e8 points (profiles/r02_pmc_index.txt).  A context that has just
// enumerated the partition already holds everything in streams: the paths of a pair (s, b) are the first cnt records of
// b's row block {c, id-position, vde[c]}, vde[b] is the block's header, and the path's index inside the partition is
// the pair's first index + popcount(G below the id-position).  So the unit that is SORTED is the pair, not the path
// (20 M instead of 200 M keys at config 3), with key = [partition | label(s) | label(b) | z-order of vde[s], vde[b]]
// (per-vertex quantised parts, ensure_vkey), ties in path order; leaves take F consecutive points of the sorted pairs
// (index_fanout: 39 at e = 2).
// Which leaves a query opens: labels of s and b exact, their embeddings in one quantisation cell, c free --
// scripts/index_key_study.py counts 55 leaves per query against 34 for the per-path label-major key and 10 087 in
// path order (64 labels), at the same number of level-1 nodes.
// The record says nothing about s itself (until round 5 it carried vde[s], degree and label of s: 48 bytes at e = 2, 96 at
// e = 8): b's row block is in DESCENDING rank order, so the cnt neighbours ranked after s are its first cnt records and the
// record at position cnt is s' own -- {s | id-position, vde[s]}, in the aux copy of the block with s' {degree, label} word --
// inside the lines the leaf reads anyway.  A hub row keeps id order: its units carry the position of s (spos).
struct __attribute__((aligned(16))) PairX {
    uint32_t block, cnt;  // row block of b (kRowAlign units); paths of the unit (| kUnitHub)
    uint64_t G, son0;     // son0: index inside the partition of the unit's first path (hub unit: first row entry << 32)
    uint32_t spos, plo;   // hub unit: position of s' record in b's id-ordered row; sorted records: low 32 bits of the points in front
                          // of the unit (k_px_permute_scan) -- a partition holds fewer than 2^31, so the leaves need no more
};
static_assert(sizeof(PairX) == 32, "two 16-byte pieces");

// leaf fill: entries per node.  The reference's node capacity is (4096 - 5) / (16 D + 4) (rtnode.cpp:27-28) and its own
// insert path splits a node that reaches capacity - 1 (rtnode.cpp:528), so capacity - 1 is the fullest node it ever
// holds itself; round 2 filled to capacity - 2.
__host__ __device__ constexpr uint32_t index_fanout(uint32_t D)
{
    return (kBlockLen - 5) / (16 * D + 4) - 1 < 64 ? (kBlockLen - 5) / (16 * D + 4) - 1 : 64;
}

// partition-local first path index of every start vertex: starts sorted by partition, then an exclusive scan
__global__ void k_px_start_parts(uint32_t len, const StartRec *__restrict__ srec, uint32_t *__restrict__ part, uint32_t *__restrict__ idx)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        part[i] = srec[i].part;
        idx[i] = (uint32_t)i;
    }
}
__global__ void k_px_sorted_counts(uint32_t len, const StartRec *__restrict__ srec, const uint32_t *__restrict__ sorted_idx,
                                   uint64_t *__restrict__ cnt)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k <= len; k += (uint64_t)gridDim.x * blockDim.x)
        cnt[k] = k < len ? srec[sorted_idx[k]].end - srec[sorted_idx[k]].base : 0ull;
}
// pos[k] = paths before the k-th start of the partition-sorted list; the first start of a partition carries that
// partition's offset, which is subtracted: pbase[start] = paths of the same partition before this start
__global__ void k_px_pbase(uint32_t len, const uint32_t *__restrict__ sorted_part, const uint32_t *__restrict__ sorted_idx,
                           const uint64_t *__restrict__ pos, uint64_t *__restrict__ pbase)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < len; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t pt = sorted_part[k];
        uint64_t lo = 0, hi = k;  // first position of this partition in the sorted list
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (sorted_part[mid] < pt) lo = mid + 1; else hi = mid;
        }
        pbase[sorted_idx[k]] = pos[k] - pos[lo];
    }
}

__global__ void k_px_pbase_single(uint32_t len, const StartRec *__restrict__ srec, uint64_t *__restrict__ pbase)
{
    const uint64_t b0 = srec[0].base;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        pbase[i] = srec[i].base - b0;
}

// Sorted items are UNITS: a pair whose middle row has at most 64 entries is one unit (its records are addressed through
// G), a hub pair is cut into units of 64 consecutive entries of the id-ordered hub row, each with the 64-bit mask of the
// entries ranked after s (the r-th path of the unit is the r-th set bit).  PairX.cnt bit 31 marks a hub unit, whose
// first record index rides in the high half of son0.
constexpr uint32_t kUnitHub = 0x80000000u;

// units of every pair (hub pairs: one per 64 row entries), and the list of hub pairs {pair, start}; 16 lanes per start
__global__ void k_px_unit_counts(uint32_t len, const StartRec *__restrict__ srec, const RankedPair *__restrict__ pairs,
                                 uint32_t *__restrict__ ucount, uint32_t *__restrict__ hub_counter, uint2 *__restrict__ hub_list)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; g < len; g += ng) {
        const uint32_t e0 = srec[g].e0, ds = srec[g].ds;
        for (uint32_t k = sub; k < ds; k += 16) {
            const RankedPair pr = pairs[e0 + k];
            uint32_t u = 1;
            if (pr.cnt & kHubFlag) {
                u = ((uint32_t)pr.G + 63u) / 64u;
                const uint32_t at = atomicAdd(hub_counter, 1u);
                if (hub_list) hub_list[at] = make_uint2(e0 + k, (uint32_t)g);
            }
            ucount[e0 + k] = u;
        }
    }
}

// Sort key of a unit: [partition | label(s) | label(b) | per level of the quantisation: E bits of s, E bits of b].  The vertex
// words hold a vertex' spread bits D apart (a path key interleaves three vertices, k_path_keys); a pair has two, and the
// third vertex' always-zero slots are squeezed out here -- the order is the same, the key is zb * E bits shorter (config 3:
// 25 -> 21 bits = three radix passes instead of four).
// WordT: the wide table's {label << 32 | spread bits} or the narrow copy's {label << sbits | spread bits} (k_vkey_narrow).
template <typename WordT> __device__ __forceinline__ uint32_t vword_label(WordT w, uint32_t sbits, uint32_t lb)
{
    const uint32_t l = sizeof(WordT) == 8 ? (uint32_t)((uint64_t)w >> 32) : (uint32_t)((uint64_t)w >> sbits);
    return l & ((1u << lb) - 1u);
}
template <typename WordT>
__device__ __forceinline__ uint64_t px_key(uint64_t part, WordT ws, WordT wb, uint32_t e, uint32_t lb, uint32_t sbits, uint32_t zb)
{
    const uint64_t smask = (1ull << sbits) - 1ull, emask = (1ull << e) - 1ull;
    const uint64_t lab = (((part << lb) | vword_label(ws, sbits, lb)) << lb) | vword_label(wb, sbits, lb);
    const uint64_t zs = (uint64_t)ws & smask, zv = (uint64_t)wb & smask;
    uint64_t z = 0;
    for (uint32_t l = 0; l < zb; l++)
        z |= ((((zs >> (l * 3u * e)) & emask) << e) | ((zv >> (l * 3u * e)) & emask)) << (l * 2u * e);
    return (lab << (zb * 2u * e)) | z;
}

// one record + one sort key per ordinary pair, at the pair's unit slot; 16 lanes per start vertex
template <int E, typename KeyT, typename WordT>
__global__ void k_px_pairs(uint32_t len, uint32_t n_parts, const StartRec *__restrict__ srec, const RankedPair *__restrict__ pairs,
                           const uint64_t *__restrict__ eoff, const uint32_t *__restrict__ nbrs, const WordT *__restrict__ vkey,
                           const uint64_t *__restrict__ pbase, const uint64_t *__restrict__ ufirst, uint32_t lb, uint32_t sbits,
                           uint32_t zb, PairX *__restrict__ px, KeyT *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const unsigned sub = threadIdx.x & 15u;
    uint64_t g = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t ng = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (; g < len; g += ng) {
        const StartRec sr = srec[g];
        const WordT ks = vkey[sr.s];
        const uint64_t pb = pbase[g];
        uint64_t before = 0;  // paths of this start vertex' pairs in front of the strip of 16 (eoff == nullptr: the offsets are not built)
        for (uint32_t k0 = 0; k0 < sr.ds; k0 += 16) {
            const uint32_t k = k0 + sub;
            const bool valid = k < sr.ds;
            const uint32_t q = sr.e0 + (valid ? k : sr.ds - 1u);
            const RankedPair pr = pairs[q];
            const uint32_t b = nbrs[sr.a_s + (valid ? k : sr.ds - 1u)];
            const WordT kb = vkey[b];
            // the pair's first path inside its start vertex: eoff[q] - sr.base, or -- graphs without hub rows, whose index build then
            // never needs the per-pair offsets -- a scan over the strip's 16 lanes
            uint64_t within;
            if (eoff) {
                within = eoff[q] - sr.base;
            } else {
                const uint32_t cnt = valid ? (pr.cnt & ~kHubFlag) : 0u;
                uint32_t incl = cnt;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const uint32_t t = __shfl_up(incl, o, 16);
                    if ((int)sub >= o) incl += t;
                }
                within = before + (incl - cnt);
                before += __shfl(incl, 15, 16);
            }
            if (!valid || (pr.cnt & kHubFlag)) continue;  // hub pairs: k_px_hub_units
            const uint64_t at = ufirst ? ufirst[q] : (uint64_t)q;
            // {block, cnt, G | son0, spos, pad}: two 16-byte stores, consecutive lanes 32 bytes apart
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const uint64_t son0 = pb + within;
            const u32x4 p0 = {pr.block, pr.cnt, (uint32_t)pr.G, (uint32_t)(pr.G >> 32)};
            const u32x4 p1 = {(uint32_t)son0, (uint32_t)(son0 >> 32), 0u, 0u};
            u32x4 *dst = reinterpret_cast<u32x4 *>(px + at);
            dst[0] = p0;
            dst[1] = p1;
            // pairs without paths sort behind every partition (partition field = n_parts)
            keys[at] = (KeyT)px_key<WordT>(pr.cnt ? sr.part : n_parts, ks, kb, E, lb, sbits, zb);
            vals[at] = (uint32_t)at;
        }
    }
}

// the units of the hub pairs: one wave per hub pair, 64 row entries per unit, kept mask by ballot
template <int E, typename KeyT>
__global__ __launch_bounds__(256) void k_px_hub_units(uint32_t n_hub_pairs, uint32_t n_parts, uint32_t slab_begin,
                                                      const uint2 *__restrict__ hub_list, const StartRec *__restrict__ srec,
                                                      const RankedPair *__restrict__ pairs, const uint64_t *__restrict__ eoff,
                                                      const uint32_t *__restrict__ nbrs, const char *__restrict__ recs,
                                                      const uint64_t *__restrict__ vkey, const uint64_t *__restrict__ pbase,
                                                      const uint64_t *__restrict__ ufirst, uint32_t lb, uint32_t sbits, uint32_t zb,
                                                      PairX *__restrict__ px, KeyT *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const unsigned lane = threadIdx.x & 63u;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < n_hub_pairs; w += nw) {
        const uint32_t q = hub_list[w].x, g = hub_list[w].y;
        const StartRec sr = srec[g];
        const RankedPair pr = pairs[q];
        const uint32_t d = (uint32_t)pr.G, thr = slab_begin + g, b = nbrs[sr.a_s + (q - sr.e0)];
        const RecWide<E> *row = reinterpret_cast<const RecWide<E> *>(recs + (uint64_t)pr.block * kRowAlign + 8 * E);
        const uint64_t ks = vkey[sr.s];
        const uint64_t key_ok = px_key<uint64_t>(sr.part, ks, vkey[b], E, lb, sbits, zb);
        const uint64_t key_no = px_key<uint64_t>(n_parts, ks, vkey[b], E, lb, sbits, zb);
        uint64_t son = pbase[g] + (eoff[q] - sr.base);
        const uint64_t at0 = ufirst[q];
        // s' own entry of the row: the one whose rank is s' (ranks are a permutation) -- a pass of its own over the row's ranks
        // in front of the units, which all carry it (hub pairs are few)
        uint32_t spos = 0;
        for (uint32_t j0 = 0; j0 < d; j0 += 64u) {
            const uint32_t j = j0 + lane;
            const uint64_t me = __ballot(j < d && row[j].aux == thr);
            if (me) spos = j0 + (uint32_t)__ffsll((long long)me) - 1u;
        }
        for (uint32_t u = 0; u * 64u < d; u++) {
            const uint32_t j = u * 64u + lane;
            const bool keep = j < d && row[j].aux > thr;
            const uint64_t mask = __ballot(keep);
            const uint32_t cnt = (uint32_t)__popcll(mask);
            if (lane == 0) {
                PairX x;
                x.block = pr.block;
                x.cnt = cnt | kUnitHub;
                x.G = mask;
                x.son0 = son | ((uint64_t)(u * 64u) << 32);
                x.spos = spos;
                x.plo = 0u;
                px[at0 + u] = x;
                keys[at0 + u] = (KeyT)(cnt ? key_ok : key_no);
                vals[at0 + u] = (uint32_t)(at0 + u);
            }
            son += cnt;
        }
    }
}
// The units in sorted order AND the points in front of each (pref[i]; pref[ne] = all points) in one pass: a workgroup takes a
// tile of kPermTile sorted units (tiles in order from a ticket counter), gathers their records -- one 16-byte piece per thread
// and step, consecutive lanes storing consecutive pieces (a record per thread stored its pieces a record apart: 0.70 ms for
// 2.0e7 48-byte records at config 3) -- scans the tile's counts and learns the tile's prefix by decoupled look-back (status word =
// value << 2 | state, as in k_start_scan).  Until round 5 the counts went to an array of their own and a rocPRIM scan followed
// (0.13 ms + 160 MB at config 3).
constexpr int kPermBlock = 256, kPermTile = 1024;
__global__ __launch_bounds__(kPermBlock) void k_px_permute_scan(uint64_t ne, const uint32_t *__restrict__ order, const PairX *__restrict__ px,
                                                                PairX *__restrict__ out, uint64_t *__restrict__ pref,
                                                                unsigned long long *__restrict__ status, uint32_t *__restrict__ ticket)
{
    constexpr int PC = (int)(sizeof(PairX) / 16), kSteps = kPermTile * PC / kPermBlock, kPer = kPermTile / kPermBlock;
    static_assert(PC == 2 && kPermBlock % PC == 0, "a thread keeps to one piece of its records");
    typedef hipcub::BlockScan<uint64_t, kPermBlock> Scan;
    __shared__ typename Scan::TempStorage s_scan;
    __shared__ uint32_t s_cnt[kPermTile];
    __shared__ uint64_t s_prefix;
    __shared__ uint32_t s_tile;
    const uint4 *src = reinterpret_cast<const uint4 *>(px);
    uint4 *dst = reinterpret_cast<uint4 *>(out);
    const uint32_t tid = threadIdx.x;
    const uint64_t n_tiles = (ne + kPermTile - 1) / kPermTile;
    for (;;) {
        if (tid == 0) s_tile = __builtin_amdgcn_atomic_inc32(ticket, 0xFFFFFFFFu, __ATOMIC_RELAXED, "agent");
        __syncthreads();
        const uint64_t tile = s_tile;
        if (tile >= n_tiles) break;
        const uint64_t r0 = tile * kPermTile;
        // 1. the tile's records: every step's source index first, then every step's piece; the counts into LDS
        uint32_t o[kSteps];
        uint4 v[kSteps];
#pragma unroll
        for (int i = 0; i < kSteps; i++) o[i] = order[min(r0 + (uint64_t)(i * kPermBlock + tid) / PC, ne - 1)];
#pragma unroll
        for (int i = 0; i < kSteps; i++) v[i] = src[(uint64_t)o[i] * PC + (tid % PC)];
#pragma unroll
        for (int i = 0; i < kSteps; i++) {
            const uint32_t piece = i * kPermBlock + tid;
            if (tid % PC == 0) s_cnt[piece / PC] = r0 + piece / PC < ne ? (v[i].y & 0x7FFFFFFFu) : 0u;  // {block, cnt, G}: the first piece
        }
        __syncthreads();
        // 2. inside the tile: kPer consecutive units per thread
        uint32_t c[kPer];
        uint64_t mine = 0;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            c[j] = s_cnt[tid * kPer + j];
            mine += c[j];
        }
        uint64_t excl = 0, aggregate = 0;
        Scan(s_scan).ExclusiveSum(mine, excl, aggregate);
        // 3. the tile's prefix: look back over the tiles before it (taken in ticket order, so every one of them is running)
        if (tid < 64) {
            if (tid == 0)
                __hip_atomic_store(&status[tile], (unsigned long long)((aggregate << 2) | (tile ? 1ull : 2ull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t prefix = 0;
            int64_t top = (int64_t)tile - 1;
            while (top >= 0) {
                const int64_t idx = top - (int64_t)tid;
                const unsigned long long w = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 2ull;
                const uint64_t inc = __ballot((w & 3ull) == 2ull), none = __ballot((w & 3ull) == 0ull);
                const uint32_t first_inc = inc ? (uint32_t)__builtin_ctzll(inc) : 64u;
                const uint64_t need = first_inc >= 63u ? ~0ull : ((1ull << (first_inc + 1)) - 1ull);
                if (none & need) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                uint64_t x = tid <= first_inc ? (uint64_t)(w >> 2) : 0ull;
                for (int sh = 32; sh; sh >>= 1) x += __shfl_xor(x, sh, 64);
                prefix += x;
                if (first_inc < 64u) break;
                top -= 64;
            }
            if (tid == 0) {
                if (tile) __hip_atomic_store(&status[tile], (unsigned long long)(((prefix + aggregate) << 2) | 2ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = prefix;
                if (tile == n_tiles - 1) pref[ne] = prefix + aggregate;
            }
        }
        __syncthreads();
        // 4. the points in front of every unit of the tile: all 64 bits to `pref`, the low 32 back into LDS for the records
        uint64_t run = s_prefix + excl;
        const uint64_t rt = r0 + (uint64_t)tid * kPer;
        {
            static_assert(kPer == 4, "two 16-byte stores per thread");
            const uint64_t p4[4] = {run, run + c[0], run + c[0] + c[1], run + c[0] + c[1] + c[2]};
#pragma unroll
            for (int j = 0; j < kPer; j++) s_cnt[tid * kPer + j] = (uint32_t)p4[j];
            if (rt + kPer <= ne) {  // (16-byte stores: pref is 8-byte aligned times an even index)
                typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 a = {p4[0], p4[1]}, b = {p4[2], p4[3]};
                reinterpret_cast<u64x2 *>(pref + rt)[0] = a;
                reinterpret_cast<u64x2 *>(pref + rt)[1] = b;
            } else {
#pragma unroll
                for (int j = 0; j < kPer; j++)
                    if (rt + j < ne) pref[rt + j] = p4[j];
            }
        }
        __syncthreads();
        // 5. the records, the second piece with the low word of its unit's prefix (PairX::plo)
#pragma unroll
        for (int i = 0; i < kSteps; i++) {
            const uint32_t piece = i * kPermBlock + tid;
            if (tid % PC == 1) v[i].w = s_cnt[piece / PC];  // {son0, spos, plo}
            if (r0 + piece / PC < ne) dst[r0 * PC + piece] = v[i];
        }
        __syncthreads();  // s_tile, s_cnt and s_prefix are rewritten by the next tile
    }
}
struct U32ToU64 {
    __host__ __device__ uint64_t operator()(uint32_t v) const { return (uint64_t)v; }
};
// first sorted pair of every partition (bounds[n_parts] = first pair without paths) and the points in front of it
// (bounds[n_parts + 1 + pt]): launched behind the prefix scan, one copy to the host for both
template <typename KeyT>
__global__ void k_px_bounds(uint64_t ne, uint32_t n_parts, const KeyT *__restrict__ sorted_keys, uint32_t shift,
                            const uint64_t *__restrict__ pref, uint64_t *__restrict__ bounds)
{
    const uint32_t pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt > n_parts) return;
    uint64_t lo = 0, hi = ne;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (((uint64_t)sorted_keys[mid] >> shift) < pt) lo = mid + 1; else hi = mid;
    }
    bounds[pt] = lo;
    bounds[n_parts + 1 + pt] = pref[lo];
}
// first[j] = sorted pair that holds the first point of leaf j (the largest k with pref[k] <= point index).  Pair-driven: a
// pair with points [a, z) of the partition names the leaves whose first point j * F falls inside -- none or one for most
// pairs -- so the pass streams the prefix once (round 2 searched it once per leaf: 24 dependent loads each, 0.15 ms).
__global__ void k_px_leaf_first(uint64_t n_leaves, uint32_t F, uint64_t r0, uint64_t r1, const uint64_t *__restrict__ pref,
                                uint32_t *__restrict__ first)
{
    const uint64_t base = pref[r0];
    for (uint64_t q = r0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < r1; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t a = pref[q] - base, z = pref[q + 1] - base;
        for (uint64_t j = (a + F - 1) / F; j * F < z && j < n_leaves; j++) first[j] = (uint32_t)q;
    }
}

// The row blocks once more, with every vertex' {degree, label} word inside the lines the leaf kernel reads anyway (round 4;
// round 3 kept the words in strips of their own beside the blocks: one more line per pair, 2.5 ms on a 5.1 ms kernel).
// Aux block of the row at unit `blk` = kAuxScale units from unit kAuxScale * blk: header {vde[b], word of b}, then per record
//   COMPACT  the record with its vertex ID REPLACED by the word degree | label << dbits (the leaf kernel never needs the id of
//            the third vertex: the path's index comes from the id-POSITION, which stays): records keep their size, so the
//            auxiliary rows cost the leaves no line at all.  Needs dbits + bits(label) <= kPackedIdBits (build_raux).
//   else     the record itself followed by the 8-byte word -- it always fits: 8E + 8 + d (R + 8) <= 2 (8E + d R) for R >= 8;
//            records of 28 instead of 20 bytes: one more line per pair on average (+1.9 ms at config 3).
// One wave per held row, once per count (the record order is the count's): 2m gathers from an n x 8 byte table.
constexpr uint32_t kAuxScale = 2;
template <int E, bool PACKED, bool COMPACT>
__global__ __launch_bounds__(256) void k_px_aux_blocks(uint32_t n_held, const uint32_t *__restrict__ held,
                                                       const uint32_t *__restrict__ adj_deg, const uint32_t *__restrict__ rblock,
                                                       const char *__restrict__ recs, const uint64_t *__restrict__ vdl,
                                                       uint32_t dbits, char *__restrict__ aux)
{
    typedef typename RecOf<E, PACKED>::type Rec;
    const unsigned lane = threadIdx.x & 63u;
    uint64_t w = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (; w < n_held; w += nw) {
        const uint32_t b = held ? held[w] : (uint32_t)w;
        const uint32_t d = adj_deg[b];
        if (d == 0) continue;
        const uint32_t blk = rblock[b];
        const char *src = recs + (uint64_t)blk * kRowAlign;
        char *dst = aux + (uint64_t)blk * kAuxScale * kRowAlign;
        if (lane < (unsigned)(2 * E)) reinterpret_cast<uint32_t *>(dst)[lane] = reinterpret_cast<const uint32_t *>(src)[lane];
        if (lane == 0) {
            const uint64_t wv = vdl[b];
            reinterpret_cast<uint32_t *>(dst + 8 * E)[0] = (uint32_t)wv;
            reinterpret_cast<uint32_t *>(dst + 8 * E)[1] = (uint32_t)(wv >> 32);
        }
        const bool wide = d > kHubDegree || !PACKED;
        const uint32_t rw = d > kHubDegree ? (uint32_t)(sizeof(RecWide<E>) / 4) : (uint32_t)(sizeof(Rec) / 4);  // dwords per record
        const uint32_t ow = COMPACT ? rw : rw + 2;
        for (uint32_t r = lane; r < d; r += 64) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(src + 8 * E) + (uint64_t)r * rw;
            uint32_t *o = reinterpret_cast<uint32_t *>(dst + 8 * E + 8) + (uint64_t)r * ow;
            const uint32_t first = q[0];
            const uint32_t id = wide ? first : (first & ((1u << kPackedIdBits) - 1u));
            const uint64_t wv = vdl[id];
            if constexpr (COMPACT) {
                const uint32_t cw = (uint32_t)wv | ((uint32_t)(wv >> 32) << dbits);
                o[0] = wide ? cw : (cw | (first & ~((1u << kPackedIdBits) - 1u)));
                for (uint32_t z = 1; z < rw; z++) o[z] = q[z];
            } else {
                for (uint32_t z = 0; z < rw; z++) o[z] = q[z];
                o[rw] = (uint32_t)wv;
                o[rw + 1] = (uint32_t)(wv >> 32);
            }
        }
    }
}

}  // namespace gnnpe
#include "gnnpe_index_deep.hip.h"
namespace gnnpe {

// zero the bytes [from, 4096) of every block of the image: the leaf kernel below stores only a leaf's used prefix
__global__ void k_scrub_tails(char *__restrict__ image, uint64_t n_blocks, uint32_t from)
{
    const uint32_t per = (kBlockLen - from) / 16;  // 16-byte pieces per block
    const uint64_t tot = n_blocks * per;
    const uint4 z = {0u, 0u, 0u, 0u};
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x)
        *reinterpret_cast<uint4 *>(image + (i / per) * (uint64_t)kBlockLen + from + (i % per) * 16) = z;
}

// One wave per leaf.  The leaf's points are F consecutive points of the sorted pairs: lane t first holds sorted pair
// first[j] + t (its record and its first point) IN REGISTERS -- the entry lanes fetch their pair's fields from that lane
// through the crossbar (ds_bpermute), so the wave's LDS is the leaf window alone and eight waves per SIMD still fit at
// F = 39 -- then entry t: pair by binary search over the pairs' first entries, record r of the pair's row block, vde[s]
// from s' own record in b's row block (position cnt of the rank-ordered block: PairX), vde[b] from the block header.
// Auxiliary index of the leaf (Partition::build_auxiliary_index, custom.h:276-311), when `adeg` is given: the entry's
// three vertices' {degree, label} come from the aux copy of b's row block (header: b; records: s and c),
// the label features from the label table, the L + 2D reductions run side by side on DPP -- what round 2 computed in a
// second pass that re-read the whole image and gathered every path's tuple (7.6 ms at config 3).
// Stores: a leaf's used prefix only (kStoreU4 16-byte pieces); the image's tails are zeroed once per buffer (k_scrub_tails).
template <int E, bool PACKED, int AUX, int NL>  // AUX: 0 = image only; 1 = aux blocks with 8-byte words behind the records; 2 = compact
__global__ __launch_bounds__(64 * kLeafWaves) void k_pack_leaves_pairs(uint64_t n_pts, uint64_t n_leaves, uint64_t r0, uint64_t r1,
                                                                        const uint64_t *__restrict__ pref,
                                                                        const uint32_t *__restrict__ first,
                                                                        const PairX *__restrict__ px, const char *__restrict__ recs_plain,
                                                                        const char *__restrict__ recs_aux,
                                                                        const uint16_t *__restrict__ xrank,
                                                                        const double *__restrict__ xsorted, uint32_t n_labels,
                                                                        uint32_t xrank_lds, uint32_t dbits, char *__restrict__ image,
                                                                        double *__restrict__ node_mbr, uint32_t *__restrict__ adeg,
                                                                        double *__restrict__ ambr, uint32_t xcd_chunk)
{
    typedef typename RecOf<E, PACKED>::type Rec;
    constexpr int D = 3 * E;
    constexpr int kEnt = 4 * D + 1;
    constexpr int F = (int)index_fanout(D);
    constexpr int kWin = (2 + F * kEnt + 4 + 3) / 4 * 4;  // dwords, a multiple of 4; the last 4 feed the shifted reads
    constexpr int kStrip = F + 1 < 64 ? (F + 1 + 7) / 8 * 8 : 64;
    constexpr int kStoreU4 = (5 + F * (16 * D + 4) + 15) / 16;  // 16-byte pieces of a full leaf's used prefix
    __shared__ __attribute__((aligned(16))) uint32_t s_win[kLeafWaves][kWin];
    __shared__ uint8_t s_pp[kLeafWaves][NL][kStrip];
    // the label table by rank (gen_vde_x rows, custom.h:492-511; gnnpe_common.h: xrank / xsorted) for the leaves' label MBR:
    // the 16-bit ranks in LDS when the table is small (xrank_lds entries; 64 labels x e = 2: 256 bytes), global otherwise
    // (round 4: and the table's doubles by rank, xsorted, in front of them -- the aux epilogue's 2D lookups were global loads
    // between the leaf's last record and its stores: one more dependent round trip in a wave that lives for three)
    extern __shared__ __attribute__((aligned(8))) unsigned char s_dyn[];
    double *const s_xsorted = reinterpret_cast<double *>(s_dyn);
    uint16_t *const s_xrank = reinterpret_cast<uint16_t *>(s_dyn + (size_t)xrank_lds * 8);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // with the auxiliary index the records come from the aux blocks (k_px_aux_blocks): header and every record carry the
    // {degree, label} word of their vertex behind them
    // (AUX = 2: the record's id bits ARE the word, k_px_aux_blocks<.., COMPACT>: records as long as the plain ones)
    constexpr uint32_t kHX = AUX ? 8u : 0u;       // bytes of b's word behind the header
    constexpr uint32_t kXW = AUX == 1 ? 8u : 0u;  // bytes of the word behind every record
    constexpr uint32_t kUnitBytes = AUX ? kAuxScale * kRowAlign : kRowAlign;
    const char *const recs = AUX ? recs_aux : recs_plain;
    const uint16_t *const xr = (AUX && xrank_lds) ? s_xrank : xrank;
    const double *const xs = (AUX && xrank_lds) ? s_xsorted : xsorted;
    uint32_t *w = s_win[wv];
    const uint64_t pbase = pref[r0];
    // One leaf per wave (NL = 1), no grid-stride loop: the leaves in flight are then one contiguous window of the image and of
    // the sorted pairs, and the block scheduler balances the tail (round 2, scripts/index_ab.py: 2048 resident-sized blocks
    // walking strided leaves 8.34 ms, 24576 blocks 7.42, one leaf per wave 7.11).
    // Round 3 tried the resident grid again WITH the next leaf's pairs and the bounds of the one after it fetched ahead (two of
    // a leaf's three dependent round trips hidden): 64 VGPRs with 7 spilled at 8 waves per SIMD, 6.2 ms against 5.1 -- in a loop
    // the next leaf's loads queue behind this leaf's stores, which a wave that simply ends never waits for.
    // (Held to 4 waves per SIMD the one-leaf-per-wave kernel takes 7.1 ms against 5.1 at 8: T = B + L / occupancy puts the
    // line-traffic floor B near 3.1 ms and the dependent round trips at 2.0 ms of today's time.)
    // Touching ahead -- every wave also loading a word of the pairs and of the row blocks' first lines of the leaf 2 048 ...
    // 16 384 leaves on, to leave them in the L2 -- cost 0.5 ms at every distance (scripts/index_ab.py: 5.95 -> 6.43-6.53 ms per
    // further partition): the kernel is short of request slots, not of patience.
    // NL = 2 (round 4, measured and NOT used): a wave takes TWO consecutive leaves and walks their three round trips side by
    // side -- both leaves' bounds, then both leaves' pairs, then both leaves' records, every load of both before the first
    // store of either (checked in the ISA) -- and assembles them one after the other in the same LDS window: twice the useful
    // requests in flight per wave at the hardware's eight waves per SIMD.  Same process, same buffers (scripts/
    // index_aux_ab.py): image 5.27-5.29 ms against 5.17-5.18 with one leaf per wave, with the auxiliary rows 6.57 against
    // 6.39.  So the kernel does not wait for latency at eight waves per SIMD: it moves the lines it touches (29.6 GB by the
    // counters) at 5.7 TB/s, above what a plain 71 % write / 29 % read stream reaches on these boxes (4.9-5.2 TB/s).
    // xcd_chunk (A/B aid, GNNPE_LEAF_XCD_CHUNK; scripts/leaf_xcd_ab.py): workgroups go to the eight XCDs in turn; inside every run
    // of 8 * xcd_chunk workgroups the one on XCD x takes the x-th run of xcd_chunk consecutive leaf groups, so that neighbours in
    // the sorted pairs and in `first` meet in one L2.  Within +-2 % for every chunk (profiles/r05_leaf_xcd_ab.txt): off.
    uint32_t bid = blockIdx.x;
    if (xcd_chunk) {
        const uint32_t span = 8u * xcd_chunk, base = bid / span * span;
        if (base + span <= gridDim.x) bid = base + ((bid - base) & 7u) * xcd_chunk + ((bid - base) >> 3);
    }
    const uint64_t jw = (uint64_t)__builtin_amdgcn_readfirstlane((int)(bid * kLeafWaves + wv)) * NL;
    bool have_leaf[NL];
    uint64_t g0[NL];
    uint32_t ne[NL], rel_cur[NL];  // (rel_cur: points of the partition in front of this lane's pair; all ones: no pair)
    PairX x_cur[NL];
    // the pairs of leaf j are first[j] .. first[j + 1] (the last one may continue in the next leaf): only those lanes load
    uint32_t fj[NL + 1];
#pragma unroll
    for (int q = 0; q <= NL; q++) fj[q] = jw + q < n_leaves ? first[jw + q] : (uint32_t)r1;
    // (no load under a lane mask, and both leaves' loads issued before either's values are touched -- hipcc had put leaf 0's
    // `pref - pbase` in front of leaf 1's loads: a lane without a pair of its own asks for the leaf's first pair again, the
    // very address lane 0 asks for)
    bool have_pair[NL];
    constexpr int kPxW = (int)(sizeof(PairX) / 4);
    uint32_t xw[NL][kPxW];
#pragma unroll
    for (int q = 0; q < NL; q++) {
        const uint64_t j = jw + q;
        have_leaf[q] = j < n_leaves;  // (wave-uniform; false only in the last workgroup)
        g0[q] = j * F;
        ne[q] = have_leaf[q] ? (uint32_t)min((uint64_t)F, n_pts - g0[q]) : 0u;
        const uint64_t f_end = (uint64_t)fj[q + 1];
        const uint64_t kk = (uint64_t)fj[q] + lane;
        have_pair[q] = have_leaf[q] && kk < r1 && kk <= f_end;
        const uint64_t kc = have_pair[q] ? kk : (have_leaf[q] ? (uint64_t)fj[q] : r0);
        const uint32_t *src = reinterpret_cast<const uint32_t *>(px + kc);
#pragma unroll
        for (int z = 0; z < kPxW; z++) xw[q][z] = src[z];
    }
#pragma unroll
    for (int q = 0; q < NL; q++) {
#pragma unroll
        for (int z = 0; z < kPxW; z++) asm volatile("" : "+v"(xw[q][z]));
    }
#pragma unroll
    for (int q = 0; q < NL; q++) {
        // (the record's low word of the points in front of it, minus the partition's: exact below 2^32 points a partition)
        rel_cur[q] = have_pair[q] ? xw[q][7] - (uint32_t)pbase : 0xFFFFFFFFu;
#pragma unroll
        for (int z = 0; z < kPxW; z++)
            if (!have_pair[q]) xw[q][z] = 0u;
        // PairX: {block, cnt, G, son0, spos, plo} (dwords 0, 1, 2-3, 4-5, 6, 7)
        x_cur[q].block = xw[q][0];
        x_cur[q].cnt = xw[q][1];
        x_cur[q].G = ((uint64_t)xw[q][3] << 32) | xw[q][2];
        x_cur[q].son0 = ((uint64_t)xw[q][5] << 32) | xw[q][4];
        x_cur[q].spos = xw[q][6];
    }
    // the label tables go into LDS BEHIND the pairs' loads (round 4): filled and fenced by a workgroup barrier at the kernel's
    // start they were a round trip of their own in front of everything; here their wait is the wait for the pairs
    if (AUX && xrank_lds) {
        for (uint32_t i = threadIdx.x; i < xrank_lds; i += 64 * kLeafWaves) {
            s_xrank[i] = xrank[i];
            s_xsorted[i] = xsorted[i];
        }
        __syncthreads();
    }
    // ---- per entry of either leaf: its pair (from the lane that holds it), then the loads of its row block's header and of
    // its record -- issued for every leaf of the wave before any of them is used
    constexpr int kHdrW = 2 * E + (int)(kHX / 4);
    constexpr int RWd = (int)(sizeof(Rec) / 4), RWw = (int)(sizeof(RecWide<E>) / 4);
    constexpr int kRecW = RWw + (int)(kXW / 4);  // dwords of the longest record (a hub unit's) with its word
    uint32_t e_pp[NL], fh[NL], rr_[NL];
    uint64_t e_G[NL], sw[NL];
    uint32_t hv[NL][kHdrW], wq[NL][kRecW];  // block header, the entry's own record (c)
    // s' own record is fetched by the lane that HOLDS the pair (a handful of lanes per leaf), as soon as the pair record is there --
    // its vde words and, with AUX, its {degree, label} word -- and handed to the pair's entries through the crossbar behind the
    // wait.  (Every entry lane fetching it itself, 4-5 more dword loads per lane: +0.32 ms on the 5.3 ms kernel.)
    constexpr int kSW = 2 * E + (AUX == 1 ? 2 : AUX == 2 ? 1 : 0);
    uint32_t hs[NL][kSW], e_a[NL];
#pragma unroll
    for (int q = 0; q < NL; q++) {
        const bool hubh = (x_cur[q].cnt & kUnitHub) != 0u;
        const uint32_t sath = hubh ? x_cur[q].spos : x_cur[q].cnt;  // behind the cnt records ranked after s; a hub row: where the unit says
        const uint32_t strideh = (hubh ? (uint32_t)(sizeof(RecWide<E>) / 4) : (uint32_t)(sizeof(Rec) / 4)) + kXW / 4;
        const uint32_t *rsh = reinterpret_cast<const uint32_t *>(recs + (uint64_t)x_cur[q].block * kUnitBytes + 8 * E + kHX) + (uint64_t)sath * strideh;
        const uint32_t voff = hubh ? 2u : (PACKED ? 1u : 2u);  // first dword of the record's vde
        {
            // (one load per four words: the record is dword-aligned, the hardware takes the unaligned 16 bytes)
            typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
            typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(4)));
            int z = 0;
#pragma unroll
            for (; z + 4 <= 2 * E; z += 4) {
                const u32x4a v4 = *reinterpret_cast<const u32x4a *>(rsh + voff + z);
                hs[q][z] = v4.x;
                hs[q][z + 1] = v4.y;
                hs[q][z + 2] = v4.z;
                hs[q][z + 3] = v4.w;
            }
            if (z + 2 <= 2 * E) {
                const u32x2a v2 = *reinterpret_cast<const u32x2a *>(rsh + voff + z);
                hs[q][z] = v2.x;
                hs[q][z + 1] = v2.y;
            }
        }
        if constexpr (AUX == 1) {
            hs[q][2 * E] = rsh[strideh - 2];
            hs[q][2 * E + 1] = rsh[strideh - 1];
        }
        if constexpr (AUX == 2) hs[q][2 * E] = rsh[0];
    }
#pragma unroll
    for (int q = 0; q < NL; q++) {
        // this lane's pair as the entry lanes will ask for it
        uint32_t pp = 0xFFu;
        uint32_t p_first = 0;
        uint64_t p_son = 0;
        const uint32_t g032 = (uint32_t)g0[q];
        if (rel_cur[q] < g032 + ne[q] && lane < kStrip) {
            pp = rel_cur[q] >= g032 ? rel_cur[q] - g032 : 0u;  // first entry of the pair inside this leaf
            // hub unit: its first record inside the id-ordered hub row, flagged in bit 31
            p_first = (x_cur[q].cnt & kUnitHub) ? ((uint32_t)(x_cur[q].son0 >> 32) | kUnitHub) : 0u;
            // low byte: points of the unit before this leaf's first point (the unit may have begun in the previous leaf)
            p_son = ((x_cur[q].son0 & 0xFFFFFFFFull) << 8) | (uint64_t)(rel_cur[q] >= g032 ? 0u : g032 - rel_cur[q]);
        }
        if (lane < kStrip) s_pp[wv][q][lane] = (uint8_t)pp;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // No load below sits under a lane mask (hipcc may wait for such a load where the lanes join, and a wait between the two
        // leaves' loads is what this kernel must not have): an idle lane acts as entry 0 -- the very addresses lane 0 asks for
        const uint32_t le = (uint32_t)lane < ne[q] ? (uint32_t)lane : 0u;
        uint32_t a = 0;  // largest a with s_pp[a] <= le (unused slots hold 0xFF)
        {
            uint32_t bnd = kStrip;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const uint32_t mid = (a + bnd) >> 1;
                if ((uint32_t)s_pp[wv][q][mid] <= le) a = mid; else bnd = mid;
            }
        }
        // the pair's fields from lane a (wave-wide shuffles: every lane takes part, idle lanes read lane 0)
        e_pp[q] = (uint32_t)__shfl((int)pp, (int)a);
        const uint32_t e_blk = (uint32_t)__shfl((int)x_cur[q].block, (int)a);
        fh[q] = (uint32_t)__shfl((int)p_first, (int)a);
        e_G[q] = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(x_cur[q].G >> 32), (int)a) << 32) | (uint32_t)__shfl((int)(uint32_t)x_cur[q].G, (int)a);
        sw[q] = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(p_son >> 32), (int)a) << 32) | (uint32_t)__shfl((int)(uint32_t)p_son, (int)a);
        e_a[q] = a;
        // (a wave's missing leaf: the first record of the buffer's first block)
        const uint32_t r = have_leaf[q] ? le - e_pp[q] + (uint32_t)(sw[q] & 0xFFu) : 0u;  // point inside the unit
        rr_[q] = r;
        const char *const blk = recs + (uint64_t)(have_leaf[q] ? e_blk : 0u) * kUnitBytes;
        // the block's header (vde[b] and, with AUX, b's {degree, label} word) is requested BEFORE the record: hipcc sinks a
        // load to its first use, which put this one behind the wait for the record -- a fourth dependent round trip per leaf
#pragma unroll
        for (int z = 0; z < kHdrW; z++) hv[q][z] = reinterpret_cast<const uint32_t *>(blk)[z];
        const bool hub = have_leaf[q] && (fh[q] & kUnitHub);
        uint32_t rec_at = r;
        if (hub) {
            // hub unit: the r-th entry ranked after s = the r-th set bit of the mask; paths follow in id order
            uint64_t m = e_G[q];
            uint32_t rr = r, pos = 0;
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) {
                const uint32_t cbits = (uint32_t)__popcll(m & ((1ull << sh) - 1ull));
                if (rr >= cbits) {
                    rr -= cbits;
                    m >>= sh;
                    pos += sh;
                }
            }
            rec_at = (fh[q] & ~kUnitHub) + pos;
        }
        // a hub row's records are wide ones {id, rank, vde}; read once by this leaf: non-temporal (round 2 A/B: 6.80 -> 6.63 ms
        // per further partition).  Dword loads: a 16-byte + a 4-byte load of the dword-aligned record were 5 % slower.
        const uint32_t stride = (hub ? (uint32_t)RWw : (uint32_t)RWd) + kXW / 4;
        const uint32_t *rq = reinterpret_cast<const uint32_t *>(blk + 8 * E + kHX) + (uint64_t)rec_at * stride;
#pragma unroll
        for (int z = 0; z < RWd + (int)(kXW / 4); z++) wq[q][z] = __builtin_nontemporal_load(rq + z);
        if constexpr (RWw > RWd) {  // (packed ids: a hub unit's record is one dword longer; a wave-uniform branch)
            static_assert(RWw - RWd <= 1, "one dword more");
            uint32_t extra = 0u;
            if (__ballot(hub)) extra = rq[hub ? kRecW - 1 : 0];
            wq[q][kRecW - 1] = extra;
        }
    }
    // every load of the wave is in flight: ONE wait, in front of the first leaf's assembly
#pragma unroll
    for (int q = 0; q < NL; q++) {
#pragma unroll
        for (int z = 0; z < kHdrW; z++) asm volatile("" : "+v"(hv[q][z]));
#pragma unroll
        for (int z = 0; z < kRecW; z++) asm volatile("" : "+v"(wq[q][z]));
#pragma unroll
        for (int z = 0; z < kSW; z++) asm volatile("" : "+v"(hs[q][z]));
    }
    // s' words from the lane that holds the entry's pair (wave-wide shuffles: idle lanes read lane 0's)
    uint32_t es[NL][kSW];
#pragma unroll
    for (int q = 0; q < NL; q++)
#pragma unroll
        for (int z = 0; z < kSW; z++) es[q][z] = (uint32_t)__shfl((int)hs[q][z], (int)e_a[q]);
#pragma unroll
    for (int q = 0; q < NL; q++) {
        if (!have_leaf[q]) continue;  // (wave-uniform)
        const uint64_t j = jw + q;
        const bool act = (uint32_t)lane < ne[q];
        if (lane == 0) {
            w[0] = 0u;  // level 0 = leaf (byte 3 of the window)
            w[1] = ne[q];
        }
        for (uint32_t i = 2 + ne[q] * kEnt + lane; i < (uint32_t)kWin; i += 64) w[i] = 0u;  // zero the window's tail
        // per entry: degrees of its three vertices, and per dimension the rank of its label feature twice -- as it is (the
        // wave's max gives the MBR's upper bound) and complemented (max of the complement = the lower bound); idle lanes
        // hold zeros, the identity of max
        constexpr int kRk = (2 * D + 1) / 2;  // dwords of packed 16-bit ranks
        uint32_t dg[3] = {0u, 0u, 0u};
        uint32_t rk[kRk];
#pragma unroll
        for (int k = 0; k < kRk; k++) rk[k] = 0u;
        if (act) {
            const bool hub = (fh[q] & kUnitHub) != 0u;
            double vc[E], vs[E];
            uint32_t son;
            uint64_t wc = 0, wsv = 0;  // {degree, label} of the entry's third and first vertex (AUX)
            uint32_t first_dw = wq[q][0], first_s = AUX == 2 ? es[q][kSW - 1] : 0u;
            // (records decoded dword by dword, by selects -- an index that depends on the lane would put the array into scratch:
            // wide {id, aux, vde}, packed {id | id-position << 26, vde} -- gnnpe_records.h)
            constexpr int kV = PACKED ? 1 : 2;  // first dword of an ordinary record's vde (a hub unit's: 2)
#pragma unroll
            for (int k = 0; k < E; k++) {
                const uint32_t lo32 = hub ? wq[q][2 + 2 * k] : wq[q][kV + 2 * k];
                const uint32_t hi32 = hub ? wq[q][2 + 2 * k + 1] : wq[q][kV + 2 * k + 1];
                vc[k] = __longlong_as_double((long long)(((uint64_t)hi32 << 32) | lo32));
                vs[k] = __longlong_as_double((long long)(((uint64_t)es[q][2 * k + 1] << 32) | es[q][2 * k]));
            }
            if constexpr (AUX == 1) {
                const uint32_t w0 = hub ? wq[q][RWw] : wq[q][RWd], w1 = hub ? wq[q][RWw + 1] : wq[q][RWd + 1];
                wc = ((uint64_t)w1 << 32) | w0;
                wsv = ((uint64_t)es[q][2 * E + 1] << 32) | es[q][2 * E];
            }
            uint32_t ip;
            if constexpr (PACKED) ip = first_dw >> kPackedIdBits; else ip = wq[q][1];
            if (PACKED && !hub) {
                first_dw &= (1u << kPackedIdBits) - 1u;
                first_s &= (1u << kPackedIdBits) - 1u;
            }
            son = hub ? (uint32_t)(sw[q] >> 8) + rr_[q]
                      : (uint32_t)((sw[q] >> 8) + (uint64_t)__popcll(e_G[q] & ((1ull << (ip & 63u)) - 1ull)));
            if constexpr (AUX == 2) {
                wc = (uint64_t)(first_dw & ((1u << (dbits & 31u)) - 1u)) | ((uint64_t)(first_dw >> (dbits & 31u)) << 32);
                wsv = (uint64_t)(first_s & ((1u << (dbits & 31u)) - 1u)) | ((uint64_t)(first_s >> (dbits & 31u)) << 32);
            }
            double vb[E];
#pragma unroll
            for (int k = 0; k < E; k++) vb[k] = __longlong_as_double((long long)(((uint64_t)hv[q][2 * k + 1] << 32) | hv[q][2 * k]));
            uint32_t *ent = w + 2 + lane * kEnt;
#pragma unroll
            for (int k = 0; k < D; k++) {
                const double val = k < E ? vs[k] : (k < 2 * E ? vb[k - E] : vc[k - 2 * E]);
                const uint64_t bits64 = (uint64_t)__double_as_longlong(val);
                const uint32_t x = (uint32_t)bits64, y = (uint32_t)(bits64 >> 32);
                ent[4 * k] = x;      // bounces[2k]   (custom.h:246)
                ent[4 * k + 1] = y;
                ent[4 * k + 2] = x;  // bounces[2k+1] (custom.h:247)
                ent[4 * k + 3] = y;
            }
            ent[4 * D] = son;  // the path's index inside the partition (custom.h:243)
            if constexpr (AUX != 0) {
                const uint64_t wb = ((uint64_t)hv[q][2 * E + 1] << 32) | hv[q][2 * E];
                dg[0] = (uint32_t)wsv;
                dg[1] = (uint32_t)wb;
                dg[2] = (uint32_t)wc;
                const uint32_t lab[3] = {(uint32_t)(wsv >> 32), (uint32_t)(wb >> 32), (uint32_t)(wc >> 32)};
                uint16_t h[2 * D + 1];
                h[2 * D] = 0;
#pragma unroll
                for (int k = 0; k < D; k++) {  // pde_label (custom.h:561-567), by rank
                    const uint16_t r16 = xr[(uint64_t)lab[k / E] * E + k % E];
                    h[2 * k] = r16;
                    h[2 * k + 1] = (uint16_t)~r16;
                }
#pragma unroll
                for (int k = 0; k < kRk; k++) rk[k] = (uint32_t)h[2 * k] | ((uint32_t)h[2 * k + 1] << 16);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // node MBR for the parent level: kMbrParts lanes per dimension, each scanning every kMbrParts-th assembled entry,
        // then a butterfly over the dimension's lanes (one lane per dimension walking all entries was 0.6 of the
        // kernel's time)
        constexpr int kMbrParts = D * 16 <= 64 ? 16 : D * 8 <= 64 ? 8 : D * 4 <= 64 ? 4 : 2;
        static_assert(D * kMbrParts <= 64, "embedding width too large for the leaf kernel's MBR lanes");
        if (lane < D * kMbrParts) {
            const int k = lane / kMbrParts, part = lane % kMbrParts;
            double lo = 1e300, hi = -1e300;
            {
                // a fixed number of reads, all issued before the first is used (entries past the leaf's last are zeros of the
                // window's tail and are skipped)
                constexpr int kIt = (F + kMbrParts - 1) / kMbrParts;
                uint32_t x0[kIt], x1[kIt];
#pragma unroll
                for (int it = 0; it < kIt; it++) {
                    const uint32_t *e4 = w + 2 + min((uint32_t)(part + it * kMbrParts), (uint32_t)(F - 1)) * kEnt + 4 * k;
                    x0[it] = e4[0];
                    x1[it] = e4[1];
                }
#pragma unroll
                for (int it = 0; it < kIt; it++) {
                    const double x = __longlong_as_double((long long)(((uint64_t)x1[it] << 32) | x0[it]));
                    const bool in = (uint32_t)(part + it * kMbrParts) < ne[q];
                    lo = in ? fmin(lo, x) : lo;
                    hi = in ? fmax(hi, x) : hi;
                }
            }
#pragma unroll
            for (int m = 1; m < kMbrParts; m <<= 1) {
                lo = fmin(lo, __shfl_xor(lo, m));
                hi = fmax(hi, __shfl_xor(hi, m));
            }
            if (part == 0) {  // {lo, hi}: one 16-byte store
                typedef double dbl2v __attribute__((ext_vector_type(2)));
                dbl2v lh = {lo, hi};
                *reinterpret_cast<dbl2v *>(node_mbr + (j * D + k) * 2) = lh;
            }
        }
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 *dst = reinterpret_cast<u32x4 *>(image + (j + 1) * (uint64_t)kBlockLen);  // node j -> file block j+1
#pragma unroll
        for (int rr = 0; rr < (kStoreU4 + 63) / 64; rr++) {
            const int c = lane + 64 * rr;
            if (c < kStoreU4) {
                u32x4 o = {0u, 0u, 0u, 0u};  // beyond the window: zeros
                if (4 * c + 4 < kWin) {
                    const uint4 lo4 = *reinterpret_cast<const uint4 *>(w + 4 * c);
                    const uint32_t nx = w[4 * c + 4];
                    o.x = (lo4.x >> 24) | (lo4.y << 8);
                    o.y = (lo4.y >> 24) | (lo4.z << 8);
                    o.z = (lo4.z >> 24) | (lo4.w << 8);
                    o.w = (lo4.w >> 24) | (nx << 8);
                }
                __builtin_nontemporal_store(o, dst + c);  // (one 16-byte store or four dword stores: the same time)
            }
        }
        if constexpr (AUX != 0) {
            // The leaf's auxiliary rows: maxima over its entries of D rank words and three degrees.  Through the LDS window,
            // which is free once the image's stores have read it: every entry lane parks its D + 3 dwords, kParts lanes per
            // column scan the entries, a DPP step or two joins the parts, and the results go out as ONE store per array,
            // consecutive lanes on consecutive words.  (Until round 4: nine full-wave DPP reductions -- 160 VALU instructions
            // and 90 cycles of DPP hazard nops in a kernel whose budget is ~600 VALU instructions per leaf at config 3: the
            // kernel runs out of instruction issue as well as of requests, 536 VALU instructions with the auxiliary rows
            // against 272 without -- and the last lane stored its 2D doubles and three degrees itself: 27 single-lane store
            // instructions = 27 write requests per leaf beside the image's 62.)  With this epilogue knocked out the build
            // took 5.71 ms against 6.12 with it and 5.35 for the image alone (profiles/r04_index_aux_ab.txt).
            constexpr int kCols = D + 3;
            constexpr int kParts = kCols * 8 <= 64 ? 8 : kCols * 4 <= 64 ? 4 : kCols * 2 <= 64 ? 2 : 1;
            static_assert(kCols * kParts <= 64 && kRk == D, "one lane group per column");
            static_assert((F + 1) * kCols <= F * kEnt, "the parked columns fit the window's entry area");
            uint32_t *sc = w + 2;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the stores' reads of the window are done
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (act) {
#pragma unroll
                for (int k = 0; k < D; k++) sc[lane * kCols + k] = rk[k];
#pragma unroll
                for (int t = 0; t < 3; t++) sc[lane * kCols + D + t] = dg[t];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int col = lane / kParts, part = lane % kParts;
            const bool ranks = col < D;  // packed 16-bit maxima; the degree columns are whole dwords
            uint32_t red = 0;
            {
                // a fixed number of reads, all issued before the first is used (entries past the leaf's last read parked
                // columns of an earlier use of the window or entry bytes -- inside the window -- and count as zero)
                constexpr int kIt = (F + kParts - 1) / kParts;
                const uint32_t cc = min((uint32_t)col, (uint32_t)(kCols - 1));
                uint32_t vv[kIt];
#pragma unroll
                for (int it = 0; it < kIt; it++) vv[it] = sc[min((uint32_t)(part + it * kParts), (uint32_t)(F - 1)) * kCols + cc];
#pragma unroll
                for (int it = 0; it < kIt; it++) {
                    const uint32_t v = (uint32_t)(part + it * kParts) < ne[q] ? vv[it] : 0u;
                    red = ranks ? pk_max_u16(red, v) : max(red, v);
                }
            }
            if constexpr (kParts >= 2) {
                const uint32_t o = dpp_u32_zero<0xB1, 0xF>(red);  // quad_perm [1,0,3,2]
                red = ranks ? pk_max_u16(red, o) : max(red, o);
            }
            if constexpr (kParts >= 4) {
                const uint32_t o = dpp_u32_zero<0x4E, 0xF>(red);  // quad_perm [2,3,0,1]
                red = ranks ? pk_max_u16(red, o) : max(red, o);
            }
            if constexpr (kParts >= 8) {
                const uint32_t o = dpp_u32_zero<0x141, 0xF>(red);  // row_half_mirror
                red = ranks ? pk_max_u16(red, o) : max(red, o);
            }
            if (col < kCols && part == 0) sc[F * kCols + col] = red;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            static_assert(2 * D <= 64, "one lane per bound of the leaf's label MBR");
            if (lane < 2 * D) {  // even lanes the lower bound, odd lanes the upper: ranks back to the table's doubles
                const uint32_t word = sc[F * kCols + (lane >> 1)];  // {rank max, complemented rank max} of dimension lane / 2
                const uint32_t r16 = (lane & 1) ? (word & 0xFFFFu) : ((~(word >> 16)) & 0xFFFFu);
                ambr[j * (2 * D) + lane] = xs[(uint64_t)((lane >> 1) % E) * n_labels + r16];
            }
            if (lane < 3) adeg[j * 3 + lane] = sc[F * kCols + D + lane];
        }
        if (q + 1 < NL) {  // the window is reused by the wave's next leaf
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// rows of `src` (n x L uint32) selected by idx -> dst
__global__ void k_gather_rows_u32(uint64_t n, uint32_t L, const uint64_t *__restrict__ idx, uint64_t idx_base,
                                  const uint32_t *__restrict__ src, uint32_t *__restrict__ dst)
{
    const uint64_t tot = n * L;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / L;
        dst[i] = src[(idx[r] - idx_base) * L + i % L];
    }
}

}  // namespace gnnpe

using namespace gnnpe;

extern "C" {

// Per-vertex key parts of the label-major path key (see k_path_keys); rebuilt when the vde table changed.
static int ensure_vkey(gnnpe_ctx *c)
{
    const uint32_t n = c->n, e = c->e, D = 3 * e;
    uint32_t lb = 1;
    while (lb < 21 && (1ull << lb) < c->n_labels) lb++;  // more than 2^21 labels: the low 21 bits still group well
    uint32_t zb_cap = 0;
    while (zb_cap < 4 && zb_cap * D + e <= 32) zb_cap++;  // the spread bits of one vertex live in 32 bits
    uint32_t zb = std::min(2u, zb_cap);                   // 2 bits per dimension is where the study saturates
    while (zb > 0 && 3 * lb + zb * D > 64) zb--;
    const uint32_t passes = (3 * lb + zb * D + 7) / 8, width = 3 * lb + zb * D <= 32 ? 32 : 64;
    while (zb < zb_cap && (3 * lb + (zb + 1) * D + 7) / 8 == passes && 3 * lb + (zb + 1) * D <= width) zb++;  // free bits
    if (c->vkey_valid && c->vkey_zb == zb && c->vkey_lb == lb) return GNNPE_OK;
    int rc;
    if ((rc = c->vkey.reserve(((size_t)n + 1) * 12))) return rc;  // wide table + narrow copy
    // the component sorts below run in the index scratch buffers
    if ((rc = c->idx_keys.reserve(((size_t)n + 1) * 8 * 2)) || (rc = c->idx_vals.reserve(((size_t)n + 1) * 4 * 2))) return rc;
    uint64_t *vkey = c->vkey.as<uint64_t>();
    hipLaunchKernelGGL(k_vkey_labels, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, n, c->labels.as<uint32_t>(), vkey);
    if (zb && n) {
        uint64_t *k_in = c->idx_keys.as<uint64_t>(), *k_out = k_in + n;
        uint32_t *v_in = c->idx_vals.as<uint32_t>(), *v_out = v_in + n;
        size_t tb = 0;
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)n, 0, 64, c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        for (uint32_t comp = 0; comp < e; comp++) {
            hipLaunchKernelGGL(k_vkey_column, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, n, e, comp, c->vde.as<double>(), k_in,
                               v_in);
            tb = c->cub_tmp.bytes;
            GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, k_in, k_out, v_in, v_out, (int)n, 0, 64, c->stream));
            hipLaunchKernelGGL(k_vkey_spread, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, n, e, comp, zb, v_out, vkey);
        }
    }
    // narrow copy behind the wide one when label and spread bits share 32 bits
    const uint32_t sbits = zb ? (zb - 1) * D + e : 0;
    c->vkey_sbits = lb + sbits <= 32 ? sbits : 32;
    if (c->vkey_sbits != 32)
        hipLaunchKernelGGL(k_vkey_narrow, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, n, sbits, vkey,
                           reinterpret_cast<uint32_t *>(vkey + n + 1));
    GNNPE_HIP_TRY(hipGetLastError());
    c->vkey_valid = true;
    c->vkey_zb = zb;
    c->vkey_lb = lb;
    return GNNPE_OK;
}

// one launch per upper level: parents packed from consecutive children
static int pack_upper_levels(gnnpe_ctx *c, const std::vector<uint64_t> &level_n, uint32_t F, uint32_t D, char *image, double *mbr_a,
                             double *mbr_b, uint32_t L = 0, double *key = nullptr, uint32_t *adeg = nullptr, double *ambr = nullptr)
{
    uint64_t child0 = 0, node0 = level_n[0];
    for (size_t lv = 1; lv < level_n.size(); lv++) {
        hipLaunchKernelGGL(k_pack_inner, dim3(grid_for(level_n[lv] * 64)), dim3(kBlock), 0, c->stream, level_n[lv], level_n[lv - 1],
                           child0, node0, F, D, (int)lv, mbr_a, image, mbr_b, L, key, adeg, ambr);
        std::swap(mbr_a, mbr_b);
        child0 = node0;
        node0 += level_n[lv];
    }
    GNNPE_HIP_TRY(hipGetLastError());
    return GNNPE_OK;
}

// header block (blk_file.cpp:38-39 + rtree.cpp:341-362): root_is_data is ONE byte, root follows at byte 25
static int write_header(gnnpe_ctx *c, char *image, const int32_t hdr[8])
{
    char h[64];
    memset(h, 0, sizeof(h));
    memcpy(h, &hdr[0], 4);
    memcpy(h + 4, &hdr[1], 4);
    memcpy(h + 8, &hdr[2], 4);
    memcpy(h + 12, &hdr[3], 4);
    memcpy(h + 16, &hdr[4], 4);
    memcpy(h + 20, &hdr[5], 4);
    h[24] = (char)hdr[6];
    memcpy(h + 25, &hdr[7], 4);
    GNNPE_HIP_TRY(hipMemcpyAsync(image, h, 64, hipMemcpyHostToDevice, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    return GNNPE_OK;
}

// level sizes: leaves, then parents until a single node remains; root must be internal (an empty tree is one empty leaf
// that is the root, rtree.cpp:11-32)
static uint64_t plan_levels(uint64_t cnt, uint32_t F, std::vector<uint64_t> &level_n)
{
    level_n.clear();
    if (cnt == 0) {
        level_n.push_back(1);
    } else {
        level_n.push_back((cnt + F - 1) / F);
        do level_n.push_back((level_n.back() + F - 1) / F); while (level_n.back() > 1);
    }
    uint64_t n_nodes = 0;
    for (uint64_t v : level_n) n_nodes += v;
    return n_nodes;
}

// entries per node of the two builders: the pair-major build fills to capacity - 1 (index_fanout), the tuple-array build to
// capacity - 2; one lane assembles one entry, so small entries fill a node to 64 at most
static uint32_t builder_fanout(uint32_t D, int builder)
{
    const uint32_t cap = (kBlockLen - 5) / (16 * D + 4);  // rtnode.cpp:27-28
    if (builder == 0) return index_fanout(D);
    return cap >= 3 ? std::max(2u, std::min(cap - 2, 64u)) : 0u;  // (capacity 3: two entries, as the pair-major build; a fan-out of one never reaches a root)
}

extern "C" uint64_t gnnpe_index_file_bytes(uint64_t points, uint32_t D, int builder)
{
    // a node must hold three entries (build_image and the pair-major build REQUIRE the same): from D = 85 on none does, and
    // a fan-out below two would never reach a single root
    if (D == 0 || (kBlockLen - 5) / (16ull * D + 4) < 3) return 0;
    const uint32_t F = builder_fanout(D, builder);
    if (F < 2) return 0;
    std::vector<uint64_t> level_n;
    return (plan_levels(points, F, level_n) + 1) * (uint64_t)kBlockLen;
}

static int build_image(gnnpe_ctx *c, uint64_t cnt, LeafSrc S, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8])
{
    GNNPE_REQUIRE(c && dev_image && nbytes, GNNPE_ERR_ARG, "null argument");
    GNNPE_REQUIRE(cnt < (1ull << 31), GNNPE_ERR_RANGE, "index over %llu entries exceeds the format's int32 counts",
                  (unsigned long long)cnt);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint32_t D = S.D;
    const uint32_t cap = (kBlockLen - 5) / (16 * D + 4);  // rtnode.cpp:27-28
    GNNPE_REQUIRE(cap >= 3, GNNPE_ERR_UNSUPPORTED, "entry size for dim %u gives node capacity %u", D, cap);
    // the reference splits a node on reaching capacity-1 (rtnode.cpp:528,576): keep <= capacity-2; one lane
    // assembles one entry, so very small entries (dim <= 3) fill a node to 64 instead
    const uint32_t F = builder_fanout(D, 1);
    int rc;

    std::vector<uint64_t> level_n;
    const uint64_t n_nodes = plan_levels(cnt, F, level_n);
    GNNPE_REQUIRE(n_nodes < (1ull << 31), GNNPE_ERR_RANGE, "too many index nodes");
    const uint64_t image_bytes = (n_nodes + 1) * (uint64_t)kBlockLen;
    c->img_valid = false;  // the image buffer is about to hold something else
    if ((rc = c->index_image.reserve(image_bytes))) return rc;
    char *image = c->index_image.as<char>();
    c->img_scrub_ptr = nullptr;  // this builder fills whole blocks: the pair-major build scrubs the tails again before its next use
    c->img_aux_valid = false;
    GNNPE_HIP_TRY(hipMemsetAsync(image, 0, kBlockLen, c->stream));

    int32_t hdr[8] = {kBlockLen, (int32_t)n_nodes, (int32_t)D, (int32_t)cnt, (int32_t)level_n[0],
                      (int32_t)(n_nodes - level_n[0]), cnt == 0 ? 1 : 0, (int32_t)(n_nodes - 1)};
    if (cnt == 0) {
        GNNPE_HIP_TRY(hipMemsetAsync(image + kBlockLen, 0, kBlockLen, c->stream));  // level 0, 0 entries
    } else {
        GNNPE_REQUIRE(D <= 64, GNNPE_ERR_UNSUPPORTED, "dimension %u > 64", D);
        const bool paths3 = !S.boxes && S.L == 3;  // GNN-PE's 3-vertex paths: label-major keys
        const bool fast = paths3 && (S.e == 1 || S.e == 2 || S.e == 3 || S.e == 4 || S.e == 8);
        const uint32_t wide_grid = 256 * 8;  // grid-stride kernels: 8 workgroups per CU
        const uint64_t sort_n = std::max<uint64_t>(cnt, paths3 ? c->n : 0);
        if ((rc = c->idx_keys.reserve(sort_n * 8 * 2)) || (rc = c->idx_vals.reserve(sort_n * 4 * 2))) return rc;
        uint32_t *v_in = c->idx_vals.as<uint32_t>(), *v_out = v_in + cnt;
        if (paths3) {
            // 1. keys: per-vertex parts (once per vde table), then three gathers per path
            if ((rc = ensure_vkey(c))) return rc;
            const uint32_t kbits = 3 * c->vkey_lb + c->vkey_zb * D;
            const uint32_t g = (uint32_t)std::min<uint64_t>(wide_grid, (cnt + 255) / 256);
            size_t tb = 0;
#define GNNPE_PATH_KEYS(KT)                                                                                        \
    do {                                                                                                           \
        if (c->vkey_sbits != 32)                                                                                   \
            hipLaunchKernelGGL((k_path_keys<KT, uint32_t>), dim3(g), dim3(256), 0, c->stream, cnt, S.vids,         \
                               reinterpret_cast<const uint32_t *>(c->vkey.as<uint64_t>() + c->n + 1), S.e,         \
                               c->vkey_lb, c->vkey_sbits, c->vkey_zb * D, k_in, v_in);                             \
        else                                                                                                       \
            hipLaunchKernelGGL((k_path_keys<KT, uint64_t>), dim3(g), dim3(256), 0, c->stream, cnt, S.vids,         \
                               c->vkey.as<uint64_t>(), S.e, c->vkey_lb, 32u, c->vkey_zb * D, k_in, v_in);          \
    } while (0)
            // 2. one radix sort over the used key bits, 32-bit keys when they fit
            if (kbits <= 32) {
                uint32_t *k_in = c->idx_keys.as<uint32_t>(), *k_out = k_in + cnt;
                GNNPE_PATH_KEYS(uint32_t);
                GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)cnt, 0, (int)kbits,
                                                                c->stream));
                if ((rc = c->cub_tmp.reserve(tb))) return rc;
                tb = c->cub_tmp.bytes;
                GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, k_in, k_out, v_in, v_out, (int)cnt, 0,
                                                                (int)kbits, c->stream));
            } else {
                uint64_t *k_in = c->idx_keys.as<uint64_t>(), *k_out = k_in + cnt;
                GNNPE_PATH_KEYS(uint64_t);
                GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)cnt, 0, (int)kbits,
                                                                c->stream));
                if ((rc = c->cub_tmp.reserve(tb))) return rc;
                tb = c->cub_tmp.bytes;
                GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, k_in, k_out, v_in, v_out, (int)cnt, 0,
                                                                (int)kbits, c->stream));
            }
#undef GNNPE_PATH_KEYS
        } else {
            // boxes (GNN-PGE): Z-order of the box centres on a uniform grid over their range
            if ((rc = c->small.reserve(256 + 2 * 64 * 8))) return rc;
            double *mn = reinterpret_cast<double *>(c->small.as<char>() + 256), *mx = mn + 64;
            uint64_t *k_in = c->idx_keys.as<uint64_t>(), *k_out = k_in + cnt;
            hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(64), 0, c->stream, mn, mx);
            const uint32_t bits = std::max(1u, std::min(16u, 64u / D));
            hipLaunchKernelGGL(k_point_minmax, dim3(std::min<uint64_t>(1024, (cnt + 255) / 256)), dim3(256), 0, c->stream, cnt,
                               S, mn, mx);
            hipLaunchKernelGGL(k_zorder_keys, dim3(grid_for(cnt)), dim3(kBlock), 0, c->stream, cnt, S, bits, mn, mx, k_in, v_in);
            size_t tb = 0;
            GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)cnt, 0,
                                                            (int)(bits * D), c->stream));
            if ((rc = c->cub_tmp.reserve(tb))) return rc;
            tb = c->cub_tmp.bytes;
            GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, k_in, k_out, v_in, v_out, (int)cnt, 0,
                                                            (int)(bits * D), c->stream));
        }
        GNNPE_HIP_TRY(hipGetLastError());
        // 3. leaves, then one launch per upper level
        uint64_t max_level = 0;
        for (uint64_t v : level_n) max_level = std::max(max_level, v);
        if ((rc = c->idx_mbr.reserve(2 * max_level * 2 * D * 8))) return rc;
        double *mbr_a = c->idx_mbr.as<double>(), *mbr_b = mbr_a + max_level * 2 * D;
#define GNNPE_IDX_LEAVES(EE)                                                                                       \
    do {                                                                                                           \
        PathPoints<EE> P = {S.vids, S.vde};                                                                        \
        const uint32_t g = (uint32_t)std::min<uint64_t>(wide_grid, (level_n[0] + kLeafWaves - 1) / kLeafWaves);    \
        hipLaunchKernelGGL((k_pack_leaves_paths<EE>), dim3(g), dim3(64 * kLeafWaves), 0, c->stream, cnt,           \
                           level_n[0], F, P, v_out, image, mbr_a);                                                 \
    } while (0)
        if (fast) {
            switch (S.e) {
            case 1: GNNPE_IDX_LEAVES(1); break;
            case 2: GNNPE_IDX_LEAVES(2); break;
            case 3: GNNPE_IDX_LEAVES(3); break;
            case 4: GNNPE_IDX_LEAVES(4); break;
            default: GNNPE_IDX_LEAVES(8); break;
            }
        } else {
            hipLaunchKernelGGL(k_pack_leaves, dim3((uint32_t)level_n[0]), dim3(64), 0, c->stream, cnt, F, S, v_out, image, mbr_a);
        }
#undef GNNPE_IDX_LEAVES
        if ((rc = pack_upper_levels(c, level_n, F, D, image, mbr_a, mbr_b))) return rc;
    }
    if ((rc = write_header(c, image, hdr))) return rc;
    *dev_image = image;
    *nbytes = image_bytes;
    if (hdr_out) memcpy(hdr_out, hdr, sizeof(hdr));
    return GNNPE_OK;
}

// Device image -> file: two pinned staging buffers, the copy-back of piece k+1 overlaps the write() of piece k
// (the 22 GB of index.dat at config 3 move at ~9 GB/s: buffered writes to ONE file serialise on its inode lock, so
// the slices below mostly help filesystems without that lock; the device build itself is 3-5 ms per partition).
// Files are written under "<path>.tmp" and renamed when complete: a failure half way (disk full, a copy-back error) must
// not leave a truncated index.dat behind -- the reference's Partition constructor only tests that the file exists
// (custom.h:222-235) and would read it.
static std::string tmp_name(const char *path) { return std::string(path) + ".tmp"; }
static int commit_file(const char *path, int rc)
{
    const std::string tmp = tmp_name(path);
    if (rc == GNNPE_OK && rename(tmp.c_str(), path) != 0) {
        set_error("cannot rename %s to %s", tmp.c_str(), path);
        rc = GNNPE_ERR_IO;
    }
    if (rc != GNNPE_OK) (void)unlink(tmp.c_str());
    return rc;
}

static int write_device_image(gnnpe_ctx *c, const char *image, uint64_t nbytes, const char *final_path)
{
    constexpr uint64_t kPiece = 64ull << 20;
    const std::string tmp_path = tmp_name(final_path);
    const char *path = tmp_path.c_str();
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) {
        set_error("cannot open %s for writing", path);
        return GNNPE_ERR_IO;
    }
    char *stage[2] = {nullptr, nullptr};
    int rc = GNNPE_OK;
    for (int k = 0; k < 2 && !rc; k++)
        if (hipHostMalloc((void **)&stage[k], std::min<uint64_t>(kPiece, std::max<uint64_t>(nbytes, 1))) != hipSuccess) {
            set_error("index copy-back: cannot allocate pinned staging memory");
            rc = GNNPE_ERR_HIP;
        }
    std::mutex mu;
    std::condition_variable cv;
    uint64_t ready[2] = {0, 0};  // bytes waiting in stage[k] (0 = free)
    bool done = false;
    std::atomic<bool> io_failed{false};
    constexpr int kWriters = 4;
    std::thread writer([&] {
        uint64_t file_off = 0;
        for (int k = 0;; k ^= 1) {
            uint64_t nb;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready[k] || done; });
                if (!ready[k]) return;
                nb = ready[k];
            }
            // the copy into the page cache is the slow part: kWriters threads, one slice of the piece each
            auto slice = [&](int t) {
                const uint64_t lo = nb * t / kWriters, hi = nb * (t + 1) / kWriters;
                for (uint64_t o = lo; o < hi && !io_failed;) {
                    const ssize_t w = pwrite(fd, stage[k] + o, hi - o, (off_t)(file_off + o));
                    if (w <= 0) io_failed = true;
                    else o += (uint64_t)w;
                }
            };
            std::thread helpers[kWriters - 1];
            for (int t = 1; t < kWriters; t++) helpers[t - 1] = std::thread(slice, t);
            slice(0);
            for (auto &h : helpers) h.join();
            file_off += nb;
            {
                std::lock_guard<std::mutex> lk(mu);
                ready[k] = 0;
            }
            cv.notify_all();
        }
    });
    int k = 0;
    for (uint64_t o = 0; o < nbytes && !rc; o += kPiece, k ^= 1) {
        const uint64_t nb = std::min(kPiece, nbytes - o);
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready[k] == 0; });
        }
        hipError_t he = hipMemcpyAsync(stage[k], image + o, nb, hipMemcpyDeviceToHost, c->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
        if (he != hipSuccess) {
            set_error("index copy-back: %s", hipGetErrorString(he));
            rc = GNNPE_ERR_HIP;
            break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            ready[k] = nb;
        }
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ready[0] == 0 && ready[1] == 0; });
        done = true;
    }
    cv.notify_all();
    writer.join();
    if (close(fd) != 0 || io_failed) {
        if (!rc) set_error("write failed on %s", path);
        if (!rc) rc = GNNPE_ERR_IO;
    }
    for (int q = 0; q < 2; q++)
        if (stage[q]) (void)hipHostFree(stage[q]);
    return commit_file(final_path, rc);
}

// Several device images -> several files at once: buffered writes to ONE file serialise (above), writes to different
// files do not, so every file gets its own writer thread and pair of pinned pieces and the copy-back deals pieces to the
// files round-robin.  p = 8 at config 3 (8 x 2.8 GB): 2.7 s one file after the other -> see DESIGN.md section 4.
static int write_device_images(gnnpe_ctx *c, size_t n, const char *const *images, const uint64_t *nbytes, const char *const *final_paths)
{
    constexpr uint64_t kPiece = 32ull << 20;
    std::vector<std::string> tmp_paths(n);
    std::vector<const char *> paths(n);
    for (size_t f = 0; f < n; f++) {
        tmp_paths[f] = tmp_name(final_paths[f]);
        paths[f] = tmp_paths[f].c_str();
    }
    struct Lane {
        int fd = -1;
        char *stage[2] = {nullptr, nullptr};
        uint64_t ready[2] = {0, 0};  // bytes waiting in stage[k] (0 = free)
        uint64_t next = 0;           // first byte of the image not yet copied back
        int k = 0;
        bool done = false;
        std::thread writer;
    };
    std::vector<Lane> lanes(n);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<bool> io_failed{false};
    int rc = GNNPE_OK;
    for (size_t f = 0; f < n && !rc; f++) {
        Lane &ln = lanes[f];
        ln.fd = open(paths[f], O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (ln.fd < 0) {
            set_error("cannot open %s for writing", paths[f]);
            rc = GNNPE_ERR_IO;
            break;
        }
        for (int k = 0; k < 2 && !rc; k++)
            if (hipHostMalloc((void **)&ln.stage[k], std::min<uint64_t>(kPiece, std::max<uint64_t>(nbytes[f], 1))) != hipSuccess) {
                set_error("index copy-back: cannot allocate pinned staging memory");
                rc = GNNPE_ERR_HIP;
            }
    }
    if (!rc)
        for (size_t f = 0; f < n; f++)
            lanes[f].writer = std::thread([&, f] {
                Lane &ln = lanes[f];
                uint64_t file_off = 0;
                for (int k = 0;; k ^= 1) {
                    uint64_t nb;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return ln.ready[k] || ln.done; });
                        if (!ln.ready[k]) return;
                        nb = ln.ready[k];
                    }
                    for (uint64_t o = 0; o < nb && !io_failed;) {
                        const ssize_t w = pwrite(ln.fd, ln.stage[k] + o, nb - o, (off_t)(file_off + o));
                        if (w <= 0) io_failed = true;
                        else o += (uint64_t)w;
                    }
                    file_off += nb;
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        ln.ready[k] = 0;
                    }
                    cv.notify_all();
                }
            });
    bool more = !rc;
    while (more && !rc) {
        more = false;
        for (size_t f = 0; f < n && !rc; f++) {
            Lane &ln = lanes[f];
            if (ln.next >= nbytes[f]) continue;
            const uint64_t nb = std::min(kPiece, nbytes[f] - ln.next);
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ln.ready[ln.k] == 0; });
            }
            hipError_t he = hipMemcpyAsync(ln.stage[ln.k], images[f] + ln.next, nb, hipMemcpyDeviceToHost, c->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(c->stream);
            if (he != hipSuccess) {
                set_error("index copy-back: %s", hipGetErrorString(he));
                rc = GNNPE_ERR_HIP;
                break;
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                ln.ready[ln.k] = nb;
            }
            cv.notify_all();
            ln.next += nb;
            ln.k ^= 1;
            more = more || ln.next < nbytes[f];
        }
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        for (auto &ln : lanes) {
            cv.wait(lk, [&] { return (ln.ready[0] == 0 && ln.ready[1] == 0) || !ln.writer.joinable(); });
            ln.done = true;
        }
    }
    cv.notify_all();
    for (size_t f = 0; f < n; f++) {
        Lane &ln = lanes[f];
        if (ln.writer.joinable()) ln.writer.join();
        if (ln.fd >= 0 && close(ln.fd) != 0 && !rc) {
            set_error("write failed on %s", paths[f]);
            rc = GNNPE_ERR_IO;
        }
        for (int q = 0; q < 2; q++)
            if (ln.stage[q]) (void)hipHostFree(ln.stage[q]);
    }
    if (io_failed && !rc) {
        set_error("index.dat: write failed");
        rc = GNNPE_ERR_IO;
    }
    // all or nothing: the files appear under their names only when every one of them is complete
    for (size_t f = 0; f < n; f++) {
        const int rf = commit_file(final_paths[f], rc);
        if (!rc) rc = rf;
    }
    return rc;
}

int gnnpe_build_index_device(gnnpe_ctx *c, uint64_t cnt, uint32_t L, const void *dev_vids, void **dev_image,
                             uint64_t *nbytes, int32_t hdr_out[8])
{
    GNNPE_REQUIRE(c && c->have_vde, GNNPE_ERR_ARG, "gnnpe_build_index: call gnnpe_vde first");
    GNNPE_REQUIRE(L >= 1 && (cnt == 0 || dev_vids), GNNPE_ERR_ARG, "null path ids");
    LeafSrc S = {(const uint32_t *)dev_vids, c->vde.as<double>(), nullptr, L, c->e, L * c->e};
    return build_image(c, cnt, S, dev_image, nbytes, hdr_out);
}

int gnnpe_build_box_index_device(gnnpe_ctx *c, uint64_t cnt, uint32_t dim, const void *dev_boxes, void **dev_image,
                                 uint64_t *nbytes, int32_t hdr_out[8])
{
    GNNPE_REQUIRE(c && dim >= 1 && (cnt == 0 || dev_boxes), GNNPE_ERR_ARG, "gnnpe_build_box_index_device: bad argument");
    LeafSrc S = {nullptr, nullptr, (const double *)dev_boxes, 1, dim, dim};
    return build_image(c, cnt, S, dev_image, nbytes, hdr_out);
}

// ---- pair-major build, host side -----------------------------------------------------------------------------------
static bool fast_dim(uint32_t e) { return e == 1 || e == 2 || e == 3 || e == 4 || e == 8; }

// The image buffer (grow-only).  The leaf kernel is 80 % stores into it, and the rate a buffer takes them at is a property of
// the allocation, so a long-lived context may place it by a draw: GNNPE_IMAGE_CANDIDATES=k allocates k
// candidates, streams into each and keeps the fastest (measured at config 3: candidates at 4.69 / 3.79 / 3.83 ms per 24 GB,
// leaf kernel 5.9 -> 5.1 ms).  Off by default: three 24 GB allocations cost 1.2 s, more than a one-shot build ever gets back.
static int reserve_image(gnnpe_ctx *c, uint64_t need)
{
    if (need <= c->index_image.bytes) return GNNPE_OK;
    const uint32_t cands = (uint32_t)std::max<long>(1, diag_int("GNNPE_IMAGE_CANDIDATES", 1));  // (diagnostic builds)
    if (need < (1ull << 30) || cands < 2) return c->index_image.reserve(need);
    c->index_image.release();
    const uint64_t want = need + need / 8 + 256;
    void *q = nullptr;
    int rc = draw_device_buffer(c, want, cands, &q);
    if (rc) return rc;
    c->index_image.p = q;
    c->index_image.bytes = want;
    return GNNPE_OK;
}

// the enumeration state this build reads: ranked records of an l = 2 count carrying the current vde table
static bool pair_major_ok(const gnnpe_ctx *c)
{
    return c->counted && c->l == 2 && c->counted_variant == 4 && c->have_vde && c->ranked_vde_valid &&
           fast_dim(c->e) && c->total_paths < (1ull << 40);
}

static uint32_t bits_for(uint64_t max_value)
{
    uint32_t b = 0;
    while (b < 63 && (1ull << b) <= max_value) b++;
    return b;
}

}  // extern "C" (templates need C++ linkage)

// once per count: units (pairs; hub pairs cut into 64-entry units) sorted by [partition | label(s) | label(b) | z(s, b)],
// their records in that order, the prefix of their path counts and every partition's range
template <int E> static int build_pair_order(gnnpe_ctx *c)
{
    typedef PairX PX;
    int rc;
    if ((rc = ensure_vkey(c))) return rc;
    // the pair records hold a path index.  Hub pairs take it from the per-pair offsets; a graph without hub rows computes it inside
    // k_px_pairs (a scan over a start vertex' pairs), so its index build leaves eoff unbuilt
    if (c->n_hub && (rc = gnnpe_ensure_eoff(c))) return rc;
    const uint64_t *eoff_or_null = c->n_hub ? c->eoff.as<uint64_t>() : nullptr;
    const uint32_t len = c->slab_end - c->slab_begin, D = 3 * E, p = c->p;
    const uint64_t ne = c->n_edges;
    const StartRec *srec = c->srec.as<StartRec>();
    const RankedPair *pairs = c->rpairs.as<RankedPair>();
    size_t tb = 0;
    // 0. units: one per ordinary pair; hub pairs (if the graph has hub rows) one per 64 row entries
    uint64_t nu = ne, n_hub_pairs = 0;
    const uint64_t *ufirst = nullptr;
    if (c->n_hub && ne) {
        if ((rc = c->px_units.reserve((ne + 2) * 12 + 64))) return rc;
        uint64_t *uf = c->px_units.as<uint64_t>();
        uint32_t *ucount = reinterpret_cast<uint32_t *>(uf + ne + 1);
        uint32_t *d_hubs = c->small.as<uint32_t>() + 520;  // byte 2080 of the context's small buffer
        for (int pass = 0; pass < 2; pass++) {
            GNNPE_HIP_TRY(hipMemsetAsync(d_hubs, 0, 4, c->stream));
            GNNPE_HIP_TRY(hipMemsetAsync(ucount + ne, 0, 4, c->stream));
            hipLaunchKernelGGL(k_px_unit_counts, dim3(grid_for((uint64_t)len * 16)), dim3(kBlock), 0, c->stream, len, srec, pairs,
                               ucount, d_hubs, pass ? c->px_hubs.as<uint2>() : (uint2 *)nullptr);
            GNNPE_HIP_TRY(hipGetLastError());
            if (pass == 0) {
                uint32_t h = 0;
                GNNPE_HIP_TRY(hipMemcpyAsync(&h, d_hubs, 4, hipMemcpyDeviceToHost, c->stream));
                GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
                n_hub_pairs = h;
                if (!n_hub_pairs) break;
                if ((rc = c->px_hubs.reserve((n_hub_pairs + 1) * 8))) return rc;
            }
        }
        hipcub::TransformInputIterator<uint64_t, U32ToU64, const uint32_t *> it(ucount, U32ToU64());
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, it, uf, (int64_t)(ne + 1), c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        tb = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, it, uf, (int64_t)(ne + 1), c->stream));
        GNNPE_HIP_TRY(hipMemcpyAsync(&nu, uf + ne, 8, hipMemcpyDeviceToHost, c->stream));
        GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
        ufirst = uf;
    }
    GNNPE_REQUIRE(nu < (1ull << 31), GNNPE_ERR_RANGE, "%llu sort units exceed the 32-bit unit ids", (unsigned long long)nu);
    // scratch layout (px_tmp): start sort {part_in, part_out, idx_in, idx_out: u32 x len}, counts / positions {u64 x (len + 1)},
    // unit keys {u64 x nu x 2}, unit values {u32 x nu x 2}, device bounds {u64 x (p + 1)}
    const size_t o_part = 0, o_idx = o_part + ((size_t)len + 1) * 8, o_cnt = o_idx + ((size_t)len + 1) * 8,
                 o_pos = o_cnt + ((size_t)len + 2) * 8, o_keys = o_pos + ((size_t)len + 2) * 8, o_vals = o_keys + (nu + 1) * 16,
                 o_bnd = o_vals + (nu + 1) * 8, o_stat = o_bnd + ((size_t)p + 2) * 16,
                 o_end = o_stat + (nu / kPermTile + 4) * 8;  // (look-back words of k_px_permute_scan, then its ticket)
    if ((rc = c->px_tmp.reserve(o_end + 64)) || (rc = c->px_recs.reserve((nu + 1) * sizeof(PX))) ||
        (rc = c->px_sorted.reserve((nu + 1) * sizeof(PX))) || (rc = c->px_pref.reserve((nu + 2) * 8)) ||
        (rc = c->px_pbase.reserve(((size_t)len + 1) * 8)))
        return rc;
    char *tmp = c->px_tmp.as<char>();
    uint32_t *part_in = reinterpret_cast<uint32_t *>(tmp + o_part), *part_out = part_in + len + 1;
    uint32_t *idx_in = reinterpret_cast<uint32_t *>(tmp + o_idx), *idx_out = idx_in + len + 1;
    uint64_t *cnt = reinterpret_cast<uint64_t *>(tmp + o_cnt), *pos = reinterpret_cast<uint64_t *>(tmp + o_pos);
    uint32_t *v_in = reinterpret_cast<uint32_t *>(tmp + o_vals), *v_out = v_in + nu + 1;
    uint64_t *d_bounds = reinterpret_cast<uint64_t *>(tmp + o_bnd);
    // 1. partition-local index of every start vertex' first path
    if (len && p == 1) {
        // one partition: the paths before a start vertex are its first output slot minus the slab's (no sort, no scan: 30
        // launches less)
        hipLaunchKernelGGL(k_px_pbase_single, dim3(grid_for(len)), dim3(kBlock), 0, c->stream, len, srec, c->px_pbase.as<uint64_t>());
    } else if (len) {
        hipLaunchKernelGGL(k_px_start_parts, dim3(grid_for(len)), dim3(kBlock), 0, c->stream, len, srec, part_in, idx_in);
        const int pb = (int)std::max(1u, bits_for(p));
        tb = 0;
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, part_in, part_out, idx_in, idx_out, (int)len, 0, pb, c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        tb = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, part_in, part_out, idx_in, idx_out, (int)len, 0, pb, c->stream));
        hipLaunchKernelGGL(k_px_sorted_counts, dim3(grid_for((uint64_t)len + 1)), dim3(kBlock), 0, c->stream, len, srec, idx_out, cnt);
        tb = 0;
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, pos, (int64_t)len + 1, c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        tb = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, cnt, pos, (int64_t)len + 1, c->stream));
        hipLaunchKernelGGL(k_px_pbase, dim3(grid_for(len)), dim3(kBlock), 0, c->stream, len, part_out, idx_out, pos, c->px_pbase.as<uint64_t>());
    }
    // 2. unit records + keys, sorted, permuted, scanned
    const uint32_t lb = c->vkey_lb, zb = c->vkey_zb, sbits = zb ? (zb - 1) * D + E : 0;
    const uint32_t shift = 2 * lb + zb * 2 * E, kbits = bits_for(p) + shift;  // (px_key: the pair key has no slots for a third vertex)
    const bool narrow = c->vkey_sbits != 32;  // the vertex words as 32-bit {label << sbits | spread bits}: half the table to gather from
    GNNPE_REQUIRE(kbits <= 64, GNNPE_ERR_UNSUPPORTED, "pair key needs %u bits", kbits);
    PX *px = c->px_recs.as<PX>(), *pxs = c->px_sorted.as<PX>();
    GNNPE_HIP_TRY(hipMemsetAsync(pxs + nu, 0, sizeof(PX), c->stream));  // sentinel of the scan
#define GNNPE_PX_SORT(KT)                                                                                               \
    do {                                                                                                                \
        KT *k_in = reinterpret_cast<KT *>(tmp + o_keys), *k_out = k_in + nu + 1;                                        \
        if (len && narrow)                                                                                              \
            hipLaunchKernelGGL((k_px_pairs<E, KT, uint32_t>), dim3(grid_for((uint64_t)len * 16)), dim3(kBlock), 0, c->stream, len, p, \
                               srec, pairs, eoff_or_null, c->nbrs.as<uint32_t>(),                                         \
                               reinterpret_cast<const uint32_t *>(c->vkey.as<uint64_t>() + c->n + 1),                     \
                               c->px_pbase.as<uint64_t>(), ufirst, lb, sbits, zb, px, k_in, v_in);                      \
        else if (len)                                                                                                   \
            hipLaunchKernelGGL((k_px_pairs<E, KT, uint64_t>), dim3(grid_for((uint64_t)len * 16)), dim3(kBlock), 0, c->stream, len, p, \
                               srec, pairs, eoff_or_null, c->nbrs.as<uint32_t>(), c->vkey.as<uint64_t>(),                  \
                               c->px_pbase.as<uint64_t>(), ufirst, lb, sbits, zb, px, k_in, v_in);                      \
        if (n_hub_pairs)                                                                                                \
            hipLaunchKernelGGL((k_px_hub_units<E, KT>), dim3(grid_for(n_hub_pairs * 64)), dim3(kBlock), 0, c->stream,     \
                               (uint32_t)n_hub_pairs, p, c->slab_begin, c->px_hubs.as<uint2>(), srec, pairs, c->eoff.as<uint64_t>(), \
                               c->nbrs.as<uint32_t>(), c->rrecs.as<char>(), c->vkey.as<uint64_t>(),                        \
                               c->px_pbase.as<uint64_t>(), ufirst, lb, sbits, zb, px, k_in, v_in);                      \
        tb = 0;                                                                                                         \
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)nu, 0, (int)kbits, c->stream)); \
        if ((rc = c->cub_tmp.reserve(tb))) return rc;                                                                   \
        tb = c->cub_tmp.bytes;                                                                                          \
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, k_in, k_out, v_in, v_out, (int)nu, 0, (int)kbits, c->stream)); \
    } while (0)
    if (nu) {
        if (kbits <= 32) GNNPE_PX_SORT(uint32_t); else GNNPE_PX_SORT(uint64_t);
        unsigned long long *stat = reinterpret_cast<unsigned long long *>(tmp + o_stat);
        const uint64_t n_tiles = (nu + kPermTile - 1) / kPermTile;
        GNNPE_HIP_TRY(hipMemsetAsync(stat, 0, (n_tiles + 1) * 8, c->stream));
        hipLaunchKernelGGL(k_px_permute_scan, dim3((unsigned)std::min<uint64_t>(n_tiles, 256 * 8)), dim3(kPermBlock), 0, c->stream, nu,
                           v_out, px, pxs, c->px_pref.as<uint64_t>(), stat, reinterpret_cast<uint32_t *>(stat + n_tiles));
    } else {
        GNNPE_HIP_TRY(hipMemsetAsync(c->px_pref.p, 0, 8, c->stream));
    }
#undef GNNPE_PX_SORT
    c->px_raux_valid = false;  // the {degree, label} strips are built by the first build that asks for the auxiliary index
    // 3. partition ranges and their first points, to the host: one copy, one wait (until round 5 the bounds, a wait, then a copy per
    // partition and another wait: 0.17 ms of the build at p = 1)
    {
        const void *k_sorted = tmp + o_keys + (nu + 1) * (kbits <= 32 ? 4 : 8);  // k_out of the sort above
        if (kbits <= 32)
            hipLaunchKernelGGL((k_px_bounds<uint32_t>), dim3(((p + 1) + 63) / 64), dim3(64), 0, c->stream, nu, p, (const uint32_t *)k_sorted,
                               shift, c->px_pref.as<uint64_t>(), d_bounds);
        else
            hipLaunchKernelGGL((k_px_bounds<uint64_t>), dim3(((p + 1) + 63) / 64), dim3(64), 0, c->stream, nu, p, (const uint64_t *)k_sorted,
                               shift, c->px_pref.as<uint64_t>(), d_bounds);
    }
    GNNPE_HIP_TRY(hipGetLastError());
    std::vector<uint64_t> both(2 * ((size_t)p + 1));
    GNNPE_HIP_TRY(hipMemcpyAsync(both.data(), d_bounds, both.size() * 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    c->px_bounds.assign(both.begin(), both.begin() + p + 1);
    c->px_points.assign(both.begin() + p + 1, both.end());
    c->px_valid = true;
    c->px_gen = c->count_gen;
    return GNNPE_OK;
}

// The {degree, label} strips beside the row blocks (the leaves' auxiliary index reads them): record order = this count's, so
// they are valid for one count like the pair order, and built by the first build of that count that wants the auxiliary index
template <int E> static int build_raux(gnnpe_ctx *c)
{
    int rc;
    GNNPE_REQUIRE(c->rows_identity || c->have_deg_all, GNNPE_ERR_UNSUPPORTED,
                  "the auxiliary index needs every vertex' degree: load the whole graph (gnnpe_load_csr) or call gnnpe_set_degrees");
    if ((rc = ensure_vertex_words(c))) return rc;
    if ((rc = c->px_raux.reserve((c->rblock_units + 1) * (uint64_t)kAuxScale * kRowAlign))) return rc;
    const uint32_t *held = c->rows_identity ? nullptr : c->held.as<uint32_t>();
    const dim3 grid(grid_for((uint64_t)c->n_held * 64)), block(kBlock);
    // the largest degree and label decide the record form: both inside a record's id bits when they fit (once per count: one
    // 8-byte copy and a stream synchronisation; GNNPE_AUX_WIDE=1, read when the context is created, forces the 8-byte words: tests run both)
    uint32_t vmax[2] = {0u, 0u};
    GNNPE_HIP_TRY(hipMemcpyAsync(vmax, c->aux_vdl.as<uint64_t>() + c->n, 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t dbits = bits_for(vmax[0]), lbits = bits_for(vmax[1]);
    c->px_raux_compact = dbits + lbits <= kPackedIdBits && dbits >= 1 && !c->sw.aux_wide;
    c->px_raux_dbits = dbits;
    if (c->n_held) {
#define GNNPE_AXB(PK, CP)                                                                                                  \
    hipLaunchKernelGGL((k_px_aux_blocks<E, PK, CP>), grid, block, 0, c->stream, c->n_held, held, c->adj_deg.as<uint32_t>(), \
                       c->rblock.as<uint32_t>(), c->rrecs.as<char>(), c->aux_vdl.as<uint64_t>(), dbits, c->px_raux.as<char>())
        if (c->n <= (1u << kPackedIdBits)) {
            if (c->px_raux_compact) GNNPE_AXB(true, true); else GNNPE_AXB(true, false);
        } else {
            if (c->px_raux_compact) GNNPE_AXB(false, true); else GNNPE_AXB(false, false);
        }
#undef GNNPE_AXB
    }
    GNNPE_HIP_TRY(hipGetLastError());
    c->px_raux_valid = true;
    return GNNPE_OK;
}

extern "C" {

static int ensure_raux(gnnpe_ctx *c)
{
    if (c->px_raux_valid) return GNNPE_OK;
    switch (c->e) {
    case 1: return build_raux<1>(c);
    case 2: return build_raux<2>(c);
    case 3: return build_raux<3>(c);
    case 4: return build_raux<4>(c);
    default: return build_raux<8>(c);
    }
}

static int ensure_pair_order(gnnpe_ctx *c)
{
    if (c->px_valid && c->px_gen == c->count_gen) return GNNPE_OK;
    switch (c->e) {
    case 1: return build_pair_order<1>(c);
    case 2: return build_pair_order<2>(c);
    case 3: return build_pair_order<3>(c);
    case 4: return build_pair_order<4>(c);
    default: return build_pair_order<8>(c);
    }
}

// with_aux: also the auxiliary arrays (c->aux_deg / c->aux_mbr rows of the leaves from the leaf kernel, of the inner nodes
// and every node's key from k_pack_inner) -- the whole Partition::build_auxiliary_index of this image
static int build_partition_image(gnnpe_ctx *c, uint32_t pid, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8], bool with_aux = false)
{
    int rc;
    if ((rc = ensure_pair_order(c))) return rc;
    const uint64_t r0 = c->px_bounds[pid], r1 = c->px_bounds[pid + 1], cnt = c->px_points[pid + 1] - c->px_points[pid];
    GNNPE_REQUIRE(cnt < (1ull << 31), GNNPE_ERR_RANGE, "index over %llu entries exceeds the format's int32 counts", (unsigned long long)cnt);
    const uint32_t e = c->e, D = 3 * e;
    const uint32_t F = index_fanout(D);
    c->img_aux_valid = false;
    if (cnt == 0) {  // the reference's empty tree
        LeafSrc S = {nullptr, c->vde.as<double>(), nullptr, 3, e, D};
        return build_image(c, 0, S, dev_image, nbytes, hdr_out);
    }
    if (with_aux && (rc = ensure_raux(c))) return rc;
    std::vector<uint64_t> level_n;
    const uint64_t n_nodes = plan_levels(cnt, F, level_n);
    GNNPE_REQUIRE(n_nodes < (1ull << 31), GNNPE_ERR_RANGE, "too many index nodes");
    const uint64_t image_bytes = (n_nodes + 1) * (uint64_t)kBlockLen;
    uint64_t max_level = 0;
    for (uint64_t v : level_n) max_level = std::max(max_level, v);
    if ((rc = reserve_image(c, image_bytes)) || (rc = c->px_first.reserve((level_n[0] + 1) * 4)) ||
        (rc = c->idx_mbr.reserve(2 * max_level * 2 * D * 8)))
        return rc;
    char *image = c->index_image.as<char>();
    // the leaf kernel stores a leaf's used prefix only: the tails of the buffer's blocks are zeroed once per buffer (and again
    // after a builder that fills blocks further, see mark_image_tails_dirty)
    const uint32_t tail_from = (5 + F * (16 * D + 4) + 15) / 16 * 16;
    if (c->img_scrub_ptr != image || c->img_scrub_bytes != c->index_image.bytes || c->img_scrub_from != tail_from) {
        if (tail_from < (uint32_t)kBlockLen)
            hipLaunchKernelGGL(k_scrub_tails, dim3(kMaxGrid), dim3(kBlock), 0, c->stream, image, (uint64_t)(c->index_image.bytes / kBlockLen),
                               tail_from);
        c->img_scrub_ptr = image;
        c->img_scrub_bytes = c->index_image.bytes;
        c->img_scrub_from = tail_from;
    }
    double *mbr_a = c->idx_mbr.as<double>(), *mbr_b = mbr_a + max_level * 2 * D;
    GNNPE_HIP_TRY(hipMemsetAsync(image, 0, kBlockLen, c->stream));
    int32_t hdr[8] = {kBlockLen, (int32_t)n_nodes, (int32_t)D, (int32_t)cnt, (int32_t)level_n[0], (int32_t)(n_nodes - level_n[0]), 0,
                      (int32_t)(n_nodes - 1)};
    const uint64_t nl = level_n[0];
    uint32_t *adeg = nullptr;
    double *ambr = nullptr;
    if (with_aux) {
        if ((rc = c->aux_key.reserve((n_nodes + 1) * 8)) || (rc = c->aux_deg.reserve((n_nodes * 3 + 1) * 4)) ||
            (rc = c->aux_mbr.reserve((n_nodes * 2 * D + 1) * 8)))
            return rc;
        GNNPE_HIP_TRY(hipMemsetAsync(c->aux_key.p, 0, n_nodes * 8, c->stream));  // the root keeps 0 (custom.h:159)
        adeg = c->aux_deg.as<uint32_t>();
        ambr = c->aux_mbr.as<double>();
    }
    hipLaunchKernelGGL(k_px_leaf_first, dim3(grid_for(r1 - r0)), dim3(kBlock), 0, c->stream, nl, F, r0, r1, c->px_pref.as<uint64_t>(),
                       c->px_first.as<uint32_t>());
    const bool packed = c->n <= (1u << kPackedIdBits);
    GNNPE_REQUIRE((nl + kLeafWaves - 1) / kLeafWaves < (1ull << 31), GNNPE_ERR_UNSUPPORTED, "too many leaves for one launch");
    // one leaf per wave (the kernel's NL = 2 form, two leaves per wave, is not instantiated: see its comment)
    const uint32_t g = (uint32_t)((nl + kLeafWaves - 1) / kLeafWaves);
    // the label tables (16-bit ranks and the doubles by rank: 10 bytes per entry) ride in LDS when they fit beside the windows
    // without costing a wave of occupancy (eight workgroups of 15 904 + 4 480 bytes fill a CU's 160 KB)
    const uint32_t xrank_lds = (with_aux && (uint64_t)c->n_labels * e <= 448) ? c->n_labels * e : 0u;
    // GNNPE_LEAF_LDS_PAD (A/B aid): this much more dynamic LDS nobody touches = fewer resident workgroups per CU
    const uint32_t leaf_pad = (uint32_t)std::max<long>(0, std::min<long>(40000, diag_int("GNNPE_LEAF_LDS_PAD", 0)));
    const uint32_t xcd_chunk = (uint32_t)std::max<long>(0, diag_int("GNNPE_LEAF_XCD_CHUNK", 0));  // (both: diagnostic builds)
#define GNNPE_PXL(EE, PK, AX) GNNPE_PXN(EE, PK, AX, 1)
#define GNNPE_PXN(EE, PK, AX, NN)                                                                                       \
    hipLaunchKernelGGL((k_pack_leaves_pairs<EE, PK, AX, NN>), dim3(g), dim3(64 * kLeafWaves), xrank_lds * 10 + leaf_pad, c->stream, cnt, nl, r0, r1, \
                       c->px_pref.as<uint64_t>(), c->px_first.as<uint32_t>(), c->px_sorted.as<PairX>(), c->rrecs.as<char>(), \
                       c->px_raux.as<char>(), c->xrank.as<uint16_t>(), c->xsorted.as<double>(), c->n_labels, xrank_lds,        \
                       c->px_raux_dbits, image, mbr_a, adeg, ambr, xcd_chunk)
#define GNNPE_PXE(EE)                                                           \
    do {                                                                        \
        if (with_aux && c->px_raux_compact) {                                   \
            if (packed) GNNPE_PXL(EE, true, 2); else GNNPE_PXL(EE, false, 2);   \
        } else if (with_aux) {                                                  \
            if (packed) GNNPE_PXL(EE, true, 1); else GNNPE_PXL(EE, false, 1);   \
        } else {                                                                \
            if (packed) GNNPE_PXL(EE, true, 0); else GNNPE_PXL(EE, false, 0);   \
        }                                                                       \
    } while (0)
    switch (e) {
    case 1: GNNPE_PXE(1); break;
    case 2: GNNPE_PXE(2); break;
    case 3: GNNPE_PXE(3); break;
    case 4: GNNPE_PXE(4); break;
    default: GNNPE_PXE(8); break;
    }
#undef GNNPE_PXE
#undef GNNPE_PXL
#undef GNNPE_PXN
    GNNPE_HIP_TRY(hipGetLastError());
    // (with_aux: the upper levels' auxiliary rows and all keys come out of the same launches, bottom-up)
    if ((rc = pack_upper_levels(c, level_n, F, D, image, mbr_a, mbr_b, 3, with_aux ? c->aux_key.as<double>() : nullptr, adeg, ambr))) return rc;
    if ((rc = write_header(c, image, hdr))) return rc;
    if (with_aux) {
        c->img_aux_valid = true;
        c->img_aux_nodes = (uint32_t)n_nodes;
    }
    *dev_image = image;
    *nbytes = image_bytes;
    if (hdr_out) memcpy(hdr_out, hdr, sizeof(hdr));
    return GNNPE_OK;
}


// ---- triple-major build (l = 3), host side: gnnpe_index_deep.hip.h ------------------------------------------------------
// the enumeration state this build reads: the work units and output slots of an l = 3 count, the current vde table
static bool triple_major_ok(const gnnpe_ctx *c)
{
    return c->counted && c->l == 3 && c->counted_variant == 5 && c->have_vde && c->eoff_valid && fast_dim(c->e) &&
           c->total_paths < (1ull << 40);
}

// once per count: units sorted by [partition | label(s) | label(b) | label(c) | z(s, b, c)], their records in that order, the
// prefix of their path counts and every partition's range
static int build_triple_order(gnnpe_ctx *c)
{
    int rc;
    if ((rc = ensure_vkey(c))) return rc;
    const uint32_t len = c->slab_end - c->slab_begin, e = c->e, p = c->p, n = c->n;
    const uint64_t ne = c->n_edges, n_wu = c->n_units;
    size_t tb = 0;
    c->tx_valid = false;
    c->px_valid = false;  // (the buffers below are the pair order's)
    // 0. pieces of the rows behind every adjacency entry, units of every pair
    if ((rc = c->tx_cpre.reserve(((size_t)c->nbr_used + 1) * 4)) || (rc = c->tx_row_units.reserve(((size_t)n + 1) * 4)) ||
        (rc = c->tx_toff.reserve((ne + 2) * 8)))
        return rc;
    uint64_t *toff = c->tx_toff.as<uint64_t>();
    uint64_t nu = 0;
    if (n)
        hipLaunchKernelGGL(k_tx_row_pieces, dim3(grid_for((uint64_t)n * 64)), dim3(kBlock), 0, c->stream, n,
                           c->rows_identity ? (const uint8_t *)nullptr : c->present.as<uint8_t>(), c->adj_start.as<uint32_t>(),
                           c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(), c->tx_cpre.as<uint32_t>(), c->tx_row_units.as<uint32_t>());
    hipLaunchKernelGGL(k_tx_pair_units, dim3(grid_for(ne + 1)), dim3(kBlock), 0, c->stream, ne, c->pnbr.as<uint32_t>(),
                       c->tx_row_units.as<uint32_t>(), toff);
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, toff, toff, (int64_t)(ne + 1), c->stream));
    if ((rc = c->cub_tmp.reserve(tb))) return rc;
    tb = c->cub_tmp.bytes;
    GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, toff, toff, (int64_t)(ne + 1), c->stream));
    GNNPE_HIP_TRY(hipMemcpyAsync(&nu, toff + ne, 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    // Units are counted with the empty ones (half of them on G(n, m)); a graph whose partitions fit the format's int32 counts can still
    // have more units than a 32-bit id names, or than memory holds at ~100 bytes each: such a count keeps the tuple-array build
    size_t free_b = 0, total_b = 0;
    GNNPE_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t held = c->px_tmp.bytes + c->px_recs.bytes + c->px_sorted.bytes + c->px_pref.bytes;
    if (nu >= std::min<uint64_t>(1ull << 31, c->sw.index_max_units) || nu * 100 > free_b + held) {
        c->tx_refused_gen = c->count_gen;
        return GNNPE_ERR_RANGE;
    }
    // scratch layout (px_tmp): start sort {part_in, part_out, idx_in, idx_out: u32 x len}, counts / positions / pbase {u64 x (len + 2)},
    // unit keys {u64 x (nu + 1) x 2}, unit values {u32 x (nu + 1) x 2}, device bounds {u64 x 2 (p + 1)}
    const size_t o_part = 0, o_idx = o_part + ((size_t)len + 1) * 8, o_cnt = o_idx + ((size_t)len + 1) * 8,
                 o_pos = o_cnt + ((size_t)len + 2) * 8, o_pb = o_pos + ((size_t)len + 2) * 8, o_keys = o_pb + ((size_t)len + 2) * 8,
                 o_vals = o_keys + (nu + 1) * 16, o_bnd = o_vals + (nu + 1) * 8, o_end = o_bnd + ((size_t)p + 2) * 16;
    if ((rc = c->px_tmp.reserve(o_end + 64)) || (rc = c->px_recs.reserve((nu + 1) * sizeof(TripX))) ||
        (rc = c->px_sorted.reserve((nu + 1) * sizeof(TripX))) || (rc = c->px_pref.reserve((nu + 2) * 8)) ||
        (rc = c->tx_padj.reserve(((size_t)len + 1) * 8)))
        return rc;
    char *tmp = c->px_tmp.as<char>();
    uint32_t *part_in = reinterpret_cast<uint32_t *>(tmp + o_part), *part_out = part_in + len + 1;
    uint32_t *idx_in = reinterpret_cast<uint32_t *>(tmp + o_idx), *idx_out = idx_in + len + 1;
    uint64_t *cnt = reinterpret_cast<uint64_t *>(tmp + o_cnt), *pos = reinterpret_cast<uint64_t *>(tmp + o_pos);
    uint64_t *pbase = reinterpret_cast<uint64_t *>(tmp + o_pb);
    uint32_t *v_in = reinterpret_cast<uint32_t *>(tmp + o_vals), *v_out = v_in + nu + 1;
    uint64_t *d_bounds = reinterpret_cast<uint64_t *>(tmp + o_bnd);
    const uint32_t *poffs = c->poffs.as<uint32_t>();
    const uint64_t *eoff = c->eoff.as<uint64_t>();
    // 1. partition-local index of every start vertex' first path (minus its first output slot)
    if (len && p > 1) {
        hipLaunchKernelGGL(k_tx_start_parts, dim3(grid_for(len)), dim3(kBlock), 0, c->stream, len, c->slab_begin, c->sorted.as<uint32_t>(),
                           c->member.as<uint32_t>(), part_in, idx_in);
        const int pb = (int)std::max(1u, bits_for(p));
        tb = 0;
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, part_in, part_out, idx_in, idx_out, (int)len, 0, pb, c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        tb = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, part_in, part_out, idx_in, idx_out, (int)len, 0, pb, c->stream));
        hipLaunchKernelGGL(k_tx_sorted_counts, dim3(grid_for((uint64_t)len + 1)), dim3(kBlock), 0, c->stream, len, poffs, eoff, idx_out, cnt);
        tb = 0;
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, cnt, pos, (int64_t)len + 1, c->stream));
        if ((rc = c->cub_tmp.reserve(tb))) return rc;
        tb = c->cub_tmp.bytes;
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, cnt, pos, (int64_t)len + 1, c->stream));
        hipLaunchKernelGGL(k_px_pbase, dim3(grid_for(len)), dim3(kBlock), 0, c->stream, len, part_out, idx_out, pos, pbase);
    }
    if (len)
        hipLaunchKernelGGL(k_tx_padj, dim3(grid_for(len)), dim3(kBlock), 0, c->stream, len, poffs, eoff,
                           p > 1 ? pbase : (const uint64_t *)nullptr, c->tx_padj.as<int64_t>());
    // 2. unit records + keys, sorted, gathered, scanned
    const uint32_t lb = c->vkey_lb, zbits = c->vkey_zb * 3 * e, shift = 3 * lb + zbits, kbits = bits_for(p) + shift;
    GNNPE_REQUIRE(kbits <= 64, GNNPE_ERR_UNSUPPORTED, "triple key needs %u bits", kbits);
    TripX *recs = c->px_recs.as<TripX>(), *sorted = c->px_sorted.as<TripX>();
    uint64_t *pref = c->px_pref.as<uint64_t>();
#define GNNPE_TX_SORT(KT)                                                                                                   \
    do {                                                                                                                    \
        KT *k_in = reinterpret_cast<KT *>(tmp + o_keys), *k_out = k_in + nu + 1;                                            \
        hipLaunchKernelGGL((k_tx_fill_empty<KT>), dim3(grid_for(nu)), dim3(kBlock), 0, c->stream, nu, (KT)((uint64_t)p << shift), k_in, v_in); \
        hipLaunchKernelGGL((k_tx_units<KT>), dim3((unsigned)((n_wu + 3) / 4)), dim3(256), 0, c->stream, n_wu, c->upair.as<uint32_t>(), \
                           c->ufirst.as<uint64_t>(), c->uoff.as<uint64_t>(), c->erow.as<uint32_t>(), c->pnbr.as<uint32_t>(),  \
                           c->sorted.as<uint32_t>(), c->slab_begin, c->member.as<uint32_t>(), c->adj_start.as<uint32_t>(),     \
                           c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(), c->nbr_rank.as<uint32_t>(), c->rank.as<uint32_t>(), \
                           c->tx_cpre.as<uint32_t>(), toff, c->tx_padj.as<int64_t>(), c->vkey.as<uint64_t>(), e, lb, zbits, recs, k_in); \
        tb = 0;                                                                                                             \
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k_in, k_out, v_in, v_out, (int)nu, 0, (int)kbits, c->stream)); \
        if ((rc = c->cub_tmp.reserve(tb))) return rc;                                                                       \
        tb = c->cub_tmp.bytes;                                                                                              \
        GNNPE_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(c->cub_tmp.p, tb, k_in, k_out, v_in, v_out, (int)nu, 0, (int)kbits, c->stream)); \
        hipLaunchKernelGGL((k_tx_gather<KT>), dim3(grid_for(nu + 1)), dim3(kBlock), 0, c->stream, nu, p, shift, k_out, v_out, recs, sorted, pref); \
        tb = 0;                                                                                                             \
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, pref, pref, (int64_t)(nu + 1), c->stream));             \
        if ((rc = c->cub_tmp.reserve(tb))) return rc;                                                                       \
        tb = c->cub_tmp.bytes;                                                                                              \
        GNNPE_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(c->cub_tmp.p, tb, pref, pref, (int64_t)(nu + 1), c->stream));        \
        hipLaunchKernelGGL((k_px_bounds<KT>), dim3(((p + 1) + 63) / 64), dim3(64), 0, c->stream, nu, p, (const KT *)k_out, shift, pref, d_bounds); \
    } while (0)
    GNNPE_REQUIRE(n_wu < (1ull << 33), GNNPE_ERR_RANGE, "too many work units for one launch");
    if (nu && n_wu) {
        if (kbits <= 32) GNNPE_TX_SORT(uint32_t); else GNNPE_TX_SORT(uint64_t);
    } else {
        GNNPE_HIP_TRY(hipMemsetAsync(pref, 0, 16, c->stream));
        GNNPE_HIP_TRY(hipMemsetAsync(d_bounds, 0, 2 * ((size_t)p + 1) * 8, c->stream));
    }
#undef GNNPE_TX_SORT
    GNNPE_HIP_TRY(hipGetLastError());
    std::vector<uint64_t> both(2 * ((size_t)p + 1));
    GNNPE_HIP_TRY(hipMemcpyAsync(both.data(), d_bounds, both.size() * 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    c->tx_bounds.assign(both.begin(), both.begin() + p + 1);
    c->tx_points.assign(both.begin() + p + 1, both.end());
    c->tx_nu = nu;
    c->tx_valid = true;
    c->tx_gen = c->count_gen;
    return GNNPE_OK;
}

static int build_triple_partition_image(gnnpe_ctx *c, uint32_t pid, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8])
{
    int rc;
    if (!(c->tx_valid && c->tx_gen == c->count_gen) && (rc = build_triple_order(c))) return rc;
    const uint64_t r0 = c->tx_bounds[pid], r1 = c->tx_bounds[pid + 1], cnt = c->tx_points[pid + 1] - c->tx_points[pid];
    GNNPE_REQUIRE(cnt < (1ull << 31), GNNPE_ERR_RANGE, "index over %llu entries exceeds the format's int32 counts", (unsigned long long)cnt);
    const uint32_t e = c->e, D = 4 * e;
    const uint32_t F = index_fanout(D);
    c->img_aux_valid = false;
    if (cnt == 0) {  // the reference's empty tree
        LeafSrc S = {nullptr, c->vde.as<double>(), nullptr, 4, e, D};
        return build_image(c, 0, S, dev_image, nbytes, hdr_out);
    }
    std::vector<uint64_t> level_n;
    const uint64_t n_nodes = plan_levels(cnt, F, level_n);
    GNNPE_REQUIRE(n_nodes < (1ull << 31), GNNPE_ERR_RANGE, "too many index nodes");
    const uint64_t image_bytes = (n_nodes + 1) * (uint64_t)kBlockLen;
    uint64_t max_level = 0;
    for (uint64_t v : level_n) max_level = std::max(max_level, v);
    if ((rc = reserve_image(c, image_bytes)) || (rc = c->px_first.reserve((level_n[0] + 1) * 4)) ||
        (rc = c->idx_mbr.reserve(2 * max_level * 2 * D * 8)))
        return rc;
    char *image = c->index_image.as<char>();
    // the leaf kernel stores a leaf's used prefix only (see build_partition_image)
    const uint32_t tail_from = (5 + F * (16 * D + 4) + 15) / 16 * 16;
    if (c->img_scrub_ptr != image || c->img_scrub_bytes != c->index_image.bytes || c->img_scrub_from != tail_from) {
        if (tail_from < (uint32_t)kBlockLen)
            hipLaunchKernelGGL(k_scrub_tails, dim3(kMaxGrid), dim3(kBlock), 0, c->stream, image, (uint64_t)(c->index_image.bytes / kBlockLen),
                               tail_from);
        c->img_scrub_ptr = image;
        c->img_scrub_bytes = c->index_image.bytes;
        c->img_scrub_from = tail_from;
    }
    double *mbr_a = c->idx_mbr.as<double>(), *mbr_b = mbr_a + max_level * 2 * D;
    GNNPE_HIP_TRY(hipMemsetAsync(image, 0, kBlockLen, c->stream));
    int32_t hdr[8] = {kBlockLen, (int32_t)n_nodes, (int32_t)D, (int32_t)cnt, (int32_t)level_n[0], (int32_t)(n_nodes - level_n[0]), 0,
                      (int32_t)(n_nodes - 1)};
    const uint64_t nl = level_n[0];
    hipLaunchKernelGGL(k_px_leaf_first, dim3(grid_for(r1 - r0)), dim3(kBlock), 0, c->stream, nl, F, r0, r1, c->px_pref.as<uint64_t>(),
                       c->px_first.as<uint32_t>());
    GNNPE_REQUIRE((nl + kLeafWaves - 1) / kLeafWaves < (1ull << 31), GNNPE_ERR_UNSUPPORTED, "too many leaves for one launch");
    const uint32_t g = (uint32_t)((nl + kLeafWaves - 1) / kLeafWaves);
#define GNNPE_TXL(EE)                                                                                                     \
    hipLaunchKernelGGL((k_tx_leaves<EE>), dim3(g), dim3(64 * kLeafWaves), 0, c->stream, cnt, nl, r1, c->tx_points[pid],    \
                       c->px_pref.as<uint64_t>(), c->px_first.as<uint32_t>(), c->px_sorted.as<TripX>(), c->nbrs.as<uint32_t>(), \
                       c->vde.as<double>(), image, mbr_a)
    switch (e) {
    case 1: GNNPE_TXL(1); break;
    case 2: GNNPE_TXL(2); break;
    case 3: GNNPE_TXL(3); break;
    case 4: GNNPE_TXL(4); break;
    default: GNNPE_TXL(8); break;
    }
#undef GNNPE_TXL
    GNNPE_HIP_TRY(hipGetLastError());
    // the upper levels: one launch per level, parents packed from consecutive children (k_tx_inner: the entries spread over the lanes)
    {
        uint64_t child0 = 0, node0 = level_n[0];
        for (size_t lv = 1; lv < level_n.size(); lv++) {
            const uint32_t gi = (uint32_t)((level_n[lv] + kLeafWaves - 1) / kLeafWaves);
#define GNNPE_TXI(EE)                                                                                                      \
    hipLaunchKernelGGL((k_tx_inner<EE>), dim3(gi), dim3(64 * kLeafWaves), 0, c->stream, level_n[lv], level_n[lv - 1], child0, node0, (int)lv, \
                       (const double *)mbr_a, image, mbr_b)
            switch (e) {
            case 1: GNNPE_TXI(1); break;
            case 2: GNNPE_TXI(2); break;
            case 3: GNNPE_TXI(3); break;
            case 4: GNNPE_TXI(4); break;
            default: GNNPE_TXI(8); break;
            }
#undef GNNPE_TXI
            std::swap(mbr_a, mbr_b);
            child0 = node0;
            node0 += level_n[lv];
        }
        GNNPE_HIP_TRY(hipGetLastError());
    }
    if ((rc = write_header(c, image, hdr))) return rc;
    *dev_image = image;
    *nbytes = image_bytes;
    if (hdr_out) memcpy(hdr_out, hdr, sizeof(hdr));
    return GNNPE_OK;
}

int gnnpe_write_device_file(gnnpe_ctx *c, const void *dev_src, uint64_t nbytes, const char *path)
{
    GNNPE_REQUIRE(c && path && (dev_src || !nbytes), GNNPE_ERR_ARG, "gnnpe_write_device_file: null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    return write_device_image(c, (const char *)dev_src, nbytes, path);
}

int gnnpe_gather_rows_device(gnnpe_ctx *c, uint64_t k, uint32_t L, const void *dev_sel, uint64_t sel_base, const void *dev_rows,
                             void *dev_out)
{
    GNNPE_REQUIRE(c && L >= 1 && (k == 0 || (dev_sel && dev_rows && dev_out)), GNNPE_ERR_ARG, "gnnpe_gather_rows_device: null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    if (k)
        hipLaunchKernelGGL(k_gather_rows_u32, dim3(grid_for(k * L)), dim3(kBlock), 0, c->stream, k, L, (const uint64_t *)dev_sel,
                           sel_base, (const uint32_t *)dev_rows, (uint32_t *)dev_out);
    GNNPE_HIP_TRY(hipGetLastError());
    return GNNPE_OK;
}

// the partition's paths as vertex tuples in path-id order (two passes over the slab in chunks): the tuple-array build's input
static int collect_partition_tuples(gnnpe_ctx *c, uint32_t pid, DevBuf &mine, uint64_t *cnt_out)
{
    const uint32_t L = c->l + 1;
    const uint64_t total = c->total_paths;
    const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(total, 1), 64ull << 20);
    DevBuf ids, part, sel;
    int rc;
    if ((rc = ids.reserve(chunk * L * 4)) || (rc = part.reserve(chunk * 4)) || (rc = sel.reserve(chunk * 8))) return rc;
    uint64_t cnt = 0;
    for (int pass = 0; pass < 2 && !rc; pass++) {
        uint64_t at = 0;
        for (uint64_t b = 0; b < total && !rc; b += chunk) {
            const uint64_t e = std::min(total, b + chunk);
            uint64_t k = 0;
            if ((rc = gnnpe_path_partitions_device(c, b, e, part.p))) break;
            if ((rc = gnnpe_select_partition(c, e - b, part.p, pid, 0, sel.p, &k))) break;
            if (pass == 0) {
                cnt += k;
            } else if (k) {
                if ((rc = gnnpe_fill_paths_device(c, b, e, ids.p, nullptr, nullptr))) break;
                hipLaunchKernelGGL(k_gather_rows_u32, dim3(grid_for(k * L)), dim3(kBlock), 0, c->stream, k, L,
                                   sel.as<uint64_t>(), (uint64_t)0, ids.as<uint32_t>(), mine.as<uint32_t>() + at * L);
                at += k;
            }
        }
        if (pass == 0 && !rc) rc = mine.reserve(std::max<uint64_t>(cnt, 1) * L * 4);
    }
    (void)hipStreamSynchronize(c->stream);  // the chunk buffers are released on return
    *cnt_out = cnt;
    return rc;
}

static bool fused_aux_ok(const gnnpe_ctx *c)
{
    return pair_major_ok(c) && (c->rows_identity || c->have_deg_all) && c->n_labels <= 65536;  // 16-bit label ranks
}

static int build_partition(gnnpe_ctx *c, uint32_t pid, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8], bool with_aux)
{
    GNNPE_REQUIRE(c && dev_image && nbytes, GNNPE_ERR_ARG, "null argument");
    GNNPE_REQUIRE(c->counted && c->have_vde && c->have_order, GNNPE_ERR_ARG,
                  "gnnpe_build_index: need gnnpe_vde and gnnpe_count_paths first");
    if (int rc_t = gnnpe::resolve_total(c)) return rc_t;  // the last count may have been enqueue-only
    GNNPE_REQUIRE(pid < c->p, GNNPE_ERR_ARG, "partition %u >= %u", pid, c->p);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc;
    c->img_valid = false;
    c->img_aux_valid = false;
    if (pair_major_ok(c)) {
        rc = build_partition_image(c, pid, dev_image, nbytes, hdr_out, with_aux && fused_aux_ok(c));
    } else if (triple_major_ok(c) && c->tx_refused_gen != c->count_gen &&
               ((rc = build_triple_partition_image(c, pid, dev_image, nbytes, hdr_out)) == GNNPE_OK || c->tx_refused_gen != c->count_gen)) {
        // (built, or failed for a reason of its own; a count whose units do not fit falls through to the tuple-array build below)
    } else {
        // the generic enumeration (embedding widths without a specialised kernel): the partition's tuples, then the tuple-array build
        DevBuf mine;
        uint64_t cnt = 0;
        rc = collect_partition_tuples(c, pid, mine, &cnt);
        if (!rc) rc = gnnpe_build_index_device(c, cnt, c->l + 1, mine.p, dev_image, nbytes, hdr_out);
        (void)hipStreamSynchronize(c->stream);
    }
    if (!rc && *dev_image == c->index_image.p) {  // remembered for gnnpe_build_aux_index
        c->img_valid = true;
        c->img_pid = pid;
        c->img_gen = c->count_gen;
        c->img_bytes = *nbytes;
    }
    return rc;
}

int gnnpe_build_index_partition_device(gnnpe_ctx *c, uint32_t pid, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8])
{
    return build_partition(c, pid, dev_image, nbytes, hdr_out, false);
}

// the partition's tuples -> the generic pass of gnnpe_aux.hip (images the leaf kernel did not annotate: l = 3, slab-only
// contexts without the degree table, foreign trees)
static int aux_generic(gnnpe_ctx *c, uint32_t pid, const void *image, uint64_t nbytes, uint32_t *N, uint32_t *D)
{
    int rc;
    DevBuf mine;
    uint64_t cnt = 0;
    if ((rc = collect_partition_tuples(c, pid, mine, &cnt))) return rc;
    void *d_key = nullptr, *d_deg = nullptr, *d_mbr = nullptr;
    rc = gnnpe_aux_index_device(c, image, nbytes, cnt, c->l + 1, mine.p, &d_key, &d_deg, &d_mbr, N, D);
    (void)hipStreamSynchronize(c->stream);  // `mine` is released on return
    return rc;
}

int gnnpe_build_index_partition_aux_device(gnnpe_ctx *c, uint32_t pid, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8],
                                           void **dev_key, void **dev_degrees, void **dev_label_mbr, uint32_t *n_nodes)
{
    GNNPE_REQUIRE(dev_key && dev_degrees && dev_label_mbr && n_nodes, GNNPE_ERR_ARG, "null argument");
    int rc = build_partition(c, pid, dev_image, nbytes, hdr_out, true);
    if (rc) return rc;
    uint32_t N = c->img_aux_nodes, D = 0;
    if (!c->img_aux_valid && (rc = aux_generic(c, pid, *dev_image, *nbytes, &N, &D))) return rc;
    *dev_key = c->aux_key.p;
    *dev_degrees = c->aux_deg.p;
    *dev_label_mbr = c->aux_mbr.p;
    *n_nodes = N;
    return GNNPE_OK;
}

int gnnpe_build_index(gnnpe_ctx *c, uint32_t pid, const char *path)
{
    GNNPE_REQUIRE(c && path, GNNPE_ERR_ARG, "null argument");
    void *image = nullptr;
    uint64_t nbytes = 0;
    int rc = gnnpe_build_index_partition_device(c, pid, &image, &nbytes, nullptr);
    if (!rc) rc = write_device_image(c, (const char *)image, nbytes, path);
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

// auxiliary index of partition pid over `image` (an index.dat image of that partition on the device) -> aux_index.bin:
// magic "GNNPEAUX", uint32 version = 1, uint32 L, uint32 D, uint32 reserved, uint64 n_nodes, then key[n_nodes] (double),
// degrees[n_nodes x L] (uint32), label_mbr[n_nodes x 2D] (double), all indexed by node block id
static int aux_to_file(gnnpe_ctx *c, uint32_t pid, const void *image, uint64_t nbytes, const char *path)
{
    int rc;
    const uint32_t L = c->l + 1;
    uint32_t N = c->img_aux_nodes, D = L * c->e;
    // the arrays of the image in index_image when the leaf kernel annotated it, the generic pass otherwise
    if (!(c->img_aux_valid && image == c->index_image.p) && (rc = aux_generic(c, pid, image, nbytes, &N, &D))) return rc;
    const void *d_key = c->aux_key.p, *d_deg = c->aux_deg.p, *d_mbr = c->aux_mbr.p;
    std::vector<char> host((size_t)N * (8 + 4 * (size_t)L + 16 * (size_t)D));
    char *hk = host.data(), *hd = hk + (size_t)N * 8, *hm = hd + (size_t)N * L * 4;
    GNNPE_HIP_TRY(hipMemcpyAsync(hk, d_key, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipMemcpyAsync(hd, d_deg, (size_t)N * L * 4, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipMemcpyAsync(hm, d_mbr, (size_t)N * 2 * D * 8, hipMemcpyDeviceToHost, c->stream));
    GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    FILE *f = fopen(path, "wb");
    if (!f) {
        set_error("cannot open %s for writing", path);
        return GNNPE_ERR_IO;
    }
    const uint32_t h32[4] = {1u, L, D, 0u};
    const uint64_t n64 = N;
    const bool ok = fwrite("GNNPEAUX", 1, 8, f) == 8 && fwrite(h32, 4, 4, f) == 4 && fwrite(&n64, 8, 1, f) == 1 &&
                    fwrite(host.data(), 1, host.size(), f) == host.size();
    if (fclose(f) != 0 || !ok) {
        set_error("%s: write failed", path);
        return GNNPE_ERR_IO;
    }
    return GNNPE_OK;
}

int gnnpe_build_index_files(gnnpe_ctx *c, uint32_t n_parts, const char *const *paths, const char *const *aux_paths)
{
    GNNPE_REQUIRE(c && paths && n_parts >= 1 && n_parts <= c->p, GNNPE_ERR_ARG, "gnnpe_build_index_files: bad argument");
    for (uint32_t pid = 0; pid < n_parts; pid++)
        GNNPE_REQUIRE(paths[pid] && (!aux_paths || aux_paths[pid]), GNNPE_ERR_ARG, "null path for partition %u", pid);
    int rc = GNNPE_OK;
    if (n_parts == 1) {
        void *image = nullptr;
        uint64_t nbytes = 0;
        if ((rc = build_partition(c, 0, &image, &nbytes, nullptr, aux_paths != nullptr))) return rc;
        if (aux_paths && (rc = aux_to_file(c, 0, image, nbytes, aux_paths[0]))) return rc;
        rc = write_device_image(c, (const char *)image, nbytes, paths[0]);
        (void)hipStreamSynchronize(c->stream);
        return rc;
    }
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    // The images are built first (milliseconds each) and kept, then their files are written side by side (one writer per
    // file: 22 GB at 38 GB/s instead of 10).  Kept copies cost device memory, so they are collected in WAVES whose sum stays
    // inside what hipMemGetInfo reports free (with a quarter left for the next partition's scratch); an image that does not
    // fit beside anything is written straight from the build buffer, one file at a time, as before.
    struct Kept {
        void *p = nullptr;
        uint64_t bytes = 0;
        uint32_t pid = 0;
    };
    std::vector<Kept> wave;
    auto flush = [&]() -> int {
        int r = GNNPE_OK;
        if (!wave.empty()) {
            std::vector<const char *> images, wpaths;
            std::vector<uint64_t> sizes;
            for (const Kept &k : wave) {
                images.push_back((const char *)k.p);
                sizes.push_back(k.bytes);
                wpaths.push_back(paths[k.pid]);
            }
            if (hipStreamSynchronize(c->stream) != hipSuccess) r = GNNPE_ERR_HIP;
            if (!r) r = write_device_images(c, wave.size(), images.data(), sizes.data(), wpaths.data());
        }
        for (Kept &k : wave) (void)hipFree(k.p);
        wave.clear();
        return r;
    };
    // GNNPE_TESTING=index_keep_bytes=<n> (testing aid): an upper bound on the bytes of kept copies, to exercise the waves on a large device
    const uint64_t keep_cap = c->sw.index_keep_bytes;
    auto room_for = [&](uint64_t nbytes) {
        size_t free_b = 0, tot_b = 0;
        if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess) return false;
        uint64_t kept = 0;
        for (const Kept &k : wave) kept += k.bytes;
        return nbytes + (256ull << 20) <= (uint64_t)free_b / 4 * 3 && kept + nbytes <= keep_cap;
    };
    for (uint32_t pid = 0; pid < n_parts && !rc; pid++) {
        void *image = nullptr;
        uint64_t nbytes = 0;
        if ((rc = build_partition(c, pid, &image, &nbytes, nullptr, aux_paths != nullptr))) break;
        if (aux_paths && (rc = aux_to_file(c, pid, image, nbytes, aux_paths[pid]))) break;
        if (!room_for(nbytes) && !wave.empty()) {  // this wave is full: write it out, then look again
            if ((rc = flush())) break;
        }
        Kept k;
        k.bytes = nbytes;
        k.pid = pid;
        if (room_for(nbytes) && hipMalloc(&k.p, std::max<uint64_t>(nbytes, 16)) == hipSuccess) {
            hipError_t he = hipMemcpyAsync(k.p, image, nbytes, hipMemcpyDeviceToDevice, c->stream);
            wave.push_back(k);
            if (he != hipSuccess) {
                set_error("index image copy: %s", hipGetErrorString(he));
                rc = GNNPE_ERR_HIP;
            }
        } else {
            (void)hipGetLastError();  // a failed hipMalloc is not an error of the build
            if ((rc = flush())) break;
            rc = write_device_image(c, (const char *)image, nbytes, paths[pid]);
        }
    }
    const int rf = flush();  // also releases the kept copies after an error
    if (!rc) rc = rf;
    (void)hipStreamSynchronize(c->stream);
    return rc;
}

int gnnpe_build_aux_index(gnnpe_ctx *c, uint32_t pid, const char *path)
{
    GNNPE_REQUIRE(c && path, GNNPE_ERR_ARG, "null argument");
    GNNPE_REQUIRE(c->counted && c->have_vde && c->have_order, GNNPE_ERR_ARG,
                  "gnnpe_build_aux_index: need gnnpe_vde and gnnpe_count_paths first");
    if (int rc_t = gnnpe::resolve_total(c)) return rc_t;  // the last count may have been enqueue-only
    GNNPE_REQUIRE(pid < c->p, GNNPE_ERR_ARG, "partition %u >= %u", pid, c->p);
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    int rc;
    void *image = c->index_image.p;
    uint64_t nbytes = c->img_bytes;
    // (an image built without its auxiliary arrays is built again with them where the leaf kernel can supply them: cheaper
    // than the generic pass over the finished image)
    const bool have = c->img_valid && c->img_pid == pid && c->img_gen == c->count_gen;
    if ((!have || (!c->img_aux_valid && fused_aux_ok(c))) && (rc = build_partition(c, pid, &image, &nbytes, nullptr, true))) return rc;
    return aux_to_file(c, pid, image, nbytes, path);
}

}  // extern "C"
