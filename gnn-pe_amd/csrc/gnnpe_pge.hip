// gnnpe_pge.hip -- GNN-PGE offline (SURVEY 8(f) "next" row 1): per-vertex path groups and the R-tree
// over vertices.
//
// Reference: GNN-PGE/src/main.cpp:91-195.  For every vertex v the 1-hop paths (v, u), u in N(v)
// ascending (dfs to depth 2, GNN-PGE/include/custom.h:52-71) have the embedding [vde[v], vde[u]]
// (D = 2e dims); path_group[v] is the per-dimension [min, max] over those paths and
// path_label_group[v] the same over [x[v], x[u]] -- a segmented min/max over the neighbour list.
// Vertices without neighbours get [vde, vde] / [x, x] in the first e dims and zeros after
// (main.cpp:104-121).  Each partition's R-tree holds its vertices' path_group rectangles
// (custom.h:165-186), son = position in the partition's vertex list.
#include <vector>

#include "gnnpe_common.h"

namespace gnnpe {

// Sixteen lanes per held row (a DPP row): lane t takes the row's entries t, t + 16, ... -- neighbour ids and their labels
// arrive coalesced, vde[u] is ONE gather of e doubles per entry and x[u] comes from the label table (the reference reads
// both per dimension: 2e gathers per entry) -- and keeps the running [min, max] of every dimension; four DPP steps inside
// the row of sixteen leave the row's result in every lane.  min / max are order-independent, so the result is bit-identical
// to the reference's sequential compare-and-replace loop (main.cpp:151-176).
template <int CTRL> __device__ __forceinline__ double dpp16_f64(double v)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)b, (int)(uint32_t)b, CTRL, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(b >> 32), (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
#define GNNPE_ROW16(OP, v)                                             \
    v = OP(v, dpp16_f64<0xB1>(v));  /* quad_perm [1,0,3,2] */          \
    v = OP(v, dpp16_f64<0x4E>(v));  /* quad_perm [2,3,0,1] */          \
    v = OP(v, dpp16_f64<0x141>(v)); /* row_half_mirror */              \
    v = OP(v, dpp16_f64<0x140>(v)); /* row_mirror: all sixteen lanes hold the row's result */

template <int E>
__global__ __launch_bounds__(256) void k_pge_groups(uint32_t n_rows, const uint32_t *__restrict__ rows,
                                                    const uint32_t *__restrict__ adj_start,
                                                    const uint32_t *__restrict__ adj_deg,
                                                    const uint32_t *__restrict__ nbrs, const uint32_t *__restrict__ nbr_label,
                                                    const uint32_t *__restrict__ labels, const double *__restrict__ xtab,
                                                    const double *__restrict__ vde, double *__restrict__ pg,
                                                    double *__restrict__ plg)
{
    constexpr int D = 2 * E;
    const unsigned sub = threadIdx.x & 15u;
    uint64_t r = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 4;
    const uint64_t nr16 = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    for (; r < n_rows; r += nr16) {  // (n_rows is rounded up by the launch so that whole waves iterate together)
        const uint32_t v = rows ? rows[r] : (uint32_t)r;
        const uint32_t st = adj_start[v], d = adj_deg[v];
        double lo[E], hi[E], llo[E], lhi[E];
#pragma unroll
        for (int k = 0; k < E; k++) {
            lo[k] = llo[k] = inf;
            hi[k] = lhi[k] = -inf;
        }
        for (uint32_t j = sub; j < d; j += 16) {
            const uint32_t u = nbrs[st + j], lb = nbr_label[st + j];
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double a = vde[(uint64_t)u * E + k], b = xtab[(uint64_t)lb * E + k];
                lo[k] = fmin(lo[k], a);
                hi[k] = fmax(hi[k], a);
                llo[k] = fmin(llo[k], b);
                lhi[k] = fmax(lhi[k], b);
            }
        }
#pragma unroll
        for (int k = 0; k < E; k++) {
            GNNPE_ROW16(fmin, lo[k])
            GNNPE_ROW16(fmax, hi[k])
            GNNPE_ROW16(fmin, llo[k])
            GNNPE_ROW16(fmax, lhi[k])
        }
        if (sub == 0) {
            double *g = pg + (uint64_t)v * 2 * D, *lg = plg + (uint64_t)v * 2 * D;
            const uint32_t lv = labels[v];
#pragma unroll
            for (int k = 0; k < E; k++) {
                const double a = vde[(uint64_t)v * E + k], b = xtab[(uint64_t)lv * E + k];
                g[2 * k] = g[2 * k + 1] = a;
                lg[2 * k] = lg[2 * k + 1] = b;
                // a vertex without neighbours keeps zeros in the second half (main.cpp:104-121)
                g[2 * (E + k)] = d ? lo[k] : 0.0;
                g[2 * (E + k) + 1] = d ? hi[k] : 0.0;
                lg[2 * (E + k)] = d ? llo[k] : 0.0;
                lg[2 * (E + k) + 1] = d ? lhi[k] : 0.0;
            }
        }
    }
}

// any other embedding width: one thread per held row (round 1's form)
__global__ __launch_bounds__(256) void k_pge_groups_any(uint32_t n_rows, const uint32_t *__restrict__ rows,
                                                        const uint32_t *__restrict__ adj_start,
                                                        const uint32_t *__restrict__ adj_deg,
                                                        const uint32_t *__restrict__ nbrs, const double *__restrict__ x,
                                                        const double *__restrict__ vde, uint32_t e,
                                                        double *__restrict__ pg, double *__restrict__ plg)
{
    const uint32_t D = 2 * e;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t v = rows ? rows[r] : (uint32_t)r;
        const uint32_t st = adj_start[v], d = adj_deg[v];
        double *g = pg + (uint64_t)v * 2 * D, *lg = plg + (uint64_t)v * 2 * D;
        for (uint32_t k = 0; k < e; k++) {
            const double a = vde[(uint64_t)v * e + k], b = x[(uint64_t)v * e + k];
            g[2 * k] = g[2 * k + 1] = a;
            lg[2 * k] = lg[2 * k + 1] = b;
        }
        for (uint32_t k = 0; k < e; k++) {
            double lo = 0.0, hi = 0.0, llo = 0.0, lhi = 0.0;
            for (uint32_t j = 0; j < d; j++) {
                const uint32_t u = nbrs[st + j];
                const double a = vde[(uint64_t)u * e + k], b = x[(uint64_t)u * e + k];
                if (j == 0) {
                    lo = hi = a;
                    llo = lhi = b;
                } else {
                    lo = fmin(lo, a);
                    hi = fmax(hi, a);
                    llo = fmin(llo, b);
                    lhi = fmax(lhi, b);
                }
            }
            g[2 * (e + k)] = lo;
            g[2 * (e + k) + 1] = hi;
            lg[2 * (e + k)] = llo;
            lg[2 * (e + k) + 1] = lhi;
        }
    }
}

// boxes[i] = path_group[vertices[i]]
__global__ void k_gather_boxes(uint64_t cnt, uint32_t width, const uint32_t *__restrict__ vertices,
                               const double *__restrict__ pg, double *__restrict__ boxes)
{
    const uint64_t tot = cnt * width;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * blockDim.x)
        boxes[i] = pg[(uint64_t)vertices[i / width] * width + i % width];
}

}  // namespace gnnpe

using namespace gnnpe;

extern "C" {

int gnnpe_pge_groups(gnnpe_ctx *c, double *host_path_group, double *host_path_label_group)
{
    GNNPE_REQUIRE(c && c->have_vde, GNNPE_ERR_ARG, "gnnpe_pge_groups: call gnnpe_vde first");
    GNNPE_REQUIRE(!c->multigraph, GNNPE_ERR_UNSUPPORTED, "gnnpe_pge_groups: simple graphs only (gnnpe_set_multigraph_rows was called)");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = c->n, e = c->e;
    const size_t bytes = (size_t)n * 4 * e * 8;
    int rc;
    if ((rc = c->pge_pg.reserve(bytes + 16)) || (rc = c->pge_plg.reserve(bytes + 16))) return rc;
    GNNPE_HIP_TRY(hipMemsetAsync(c->pge_pg.p, 0, bytes, c->stream));
    GNNPE_HIP_TRY(hipMemsetAsync(c->pge_plg.p, 0, bytes, c->stream));
    if (c->n_rows) {
        const uint32_t *rows = c->rows_identity ? nullptr : c->rows.as<uint32_t>();
        const dim3 block(kBlock);
#define GNNPE_PG(EE)                                                                                                     \
    hipLaunchKernelGGL((k_pge_groups<EE>), dim3(grid_for((uint64_t)c->n_rows * 16)), block, 0, c->stream, c->n_rows, rows, \
                       c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(),                    \
                       c->nbr_label.as<uint32_t>(), c->labels.as<uint32_t>(), c->xtab.as<double>(), c->vde.as<double>(),  \
                       c->pge_pg.as<double>(), c->pge_plg.as<double>())
        switch (e) {
        case 1: GNNPE_PG(1); break;
        case 2: GNNPE_PG(2); break;
        case 4: GNNPE_PG(4); break;
        case 8: GNNPE_PG(8); break;
        default:
            hipLaunchKernelGGL(k_pge_groups_any, dim3(grid_for(c->n_rows)), block, 0, c->stream, c->n_rows, rows,
                               c->adj_start.as<uint32_t>(), c->adj_deg.as<uint32_t>(), c->nbrs.as<uint32_t>(), c->x.as<double>(),
                               c->vde.as<double>(), e, c->pge_pg.as<double>(), c->pge_plg.as<double>());
            break;
        }
#undef GNNPE_PG
        GNNPE_HIP_TRY(hipGetLastError());
    }
    c->have_pge = true;
    if (host_path_group) GNNPE_HIP_TRY(hipMemcpyAsync(host_path_group, c->pge_pg.p, bytes, hipMemcpyDeviceToHost, c->stream));
    if (host_path_label_group)
        GNNPE_HIP_TRY(hipMemcpyAsync(host_path_label_group, c->pge_plg.p, bytes, hipMemcpyDeviceToHost, c->stream));
    if (host_path_group || host_path_label_group) GNNPE_HIP_TRY(hipStreamSynchronize(c->stream));
    return GNNPE_OK;
}

int gnnpe_pge_device_ptr(gnnpe_ctx *c, void **dev_path_group, void **dev_path_label_group)
{
    GNNPE_REQUIRE(c && c->have_pge, GNNPE_ERR_ARG, "gnnpe_pge_device_ptr: call gnnpe_pge_groups first");
    if (dev_path_group) *dev_path_group = c->pge_pg.p;
    if (dev_path_label_group) *dev_path_label_group = c->pge_plg.p;
    return GNNPE_OK;
}

int gnnpe_pge_build_index(gnnpe_ctx *c, uint64_t n_sel, const uint32_t *host_vertices, const char *path)
{
    GNNPE_REQUIRE(c && c->have_pge && path && (n_sel == 0 || host_vertices), GNNPE_ERR_ARG,
                  "gnnpe_pge_build_index: call gnnpe_pge_groups first / null argument");
    GNNPE_HIP_TRY(hipSetDevice(c->device));
    const uint32_t D = 2 * c->e, width = 2 * D;
    for (uint64_t i = 0; i < n_sel; i++) GNNPE_REQUIRE(host_vertices[i] < c->n, GNNPE_ERR_ARG, "vertex %u out of range", host_vertices[i]);
    DevBuf dv, boxes;
    int rc = GNNPE_OK;
    if ((rc = dv.reserve((n_sel + 1) * 4)) || (rc = boxes.reserve((n_sel + 1) * width * 8))) return rc;
    if (n_sel) {
        GNNPE_HIP_TRY(hipMemcpyAsync(dv.p, host_vertices, n_sel * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_gather_boxes, dim3(grid_for(n_sel * width)), dim3(kBlock), 0, c->stream, n_sel, width,
                           dv.as<uint32_t>(), c->pge_pg.as<double>(), boxes.as<double>());
        GNNPE_HIP_TRY(hipGetLastError());
    }
    void *image = nullptr;
    uint64_t nbytes = 0;
    rc = gnnpe_build_box_index_device(c, n_sel, D, boxes.p, &image, &nbytes, nullptr);
    if (!rc) {
        std::vector<char> host(nbytes);
        rc = gnnpe_copy_to_host(c, host.data(), image, nbytes);
        if (!rc) {
            FILE *f = fopen(path, "wb");
            if (!f || fwrite(host.data(), 1, nbytes, f) != nbytes || fclose(f) != 0) {
                set_error("cannot write %s", path);
                rc = GNNPE_ERR_IO;
            }
        }
    }
    (void)hipStreamSynchronize(c->stream);
    dv.release();
    boxes.release();
    return rc;
}

}  // extern "C"
