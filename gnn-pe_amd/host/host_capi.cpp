// host_capi.cpp -- C-ABI wrappers around the host loader (R0 / R1) so that non-C++ callers (the Python
// multi-GPU driver) use the SAME loader code as gnnpe_main.  Compiled into libgnnpe_hip.so.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gnnpe_hip.h"
#include "graph_loader.h"
#include "query_plan.h"
#include "refine.h"

namespace gnnpe {
void set_error(const char *fmt, ...);
}

extern "C" {

int gnnpe_host_load_graph(const char *path, uint32_t *n, uint32_t *m, uint32_t **offsets, uint32_t **nbrs, uint32_t **labels,
                          uint32_t meta[3])
{
    if (!path || !n || !m || !offsets || !nbrs || !labels) {
        gnnpe::set_error("gnnpe_host_load_graph: null argument");
        return GNNPE_ERR_ARG;
    }
    gnnpe_host::StaticGraph g;
    std::string err;
    const int rc = g.load(path, &err);
    if (rc != 0) {
        gnnpe::set_error("%s", err.c_str());
        return rc;
    }
    auto dup = [](const std::vector<uint32_t> &v) {
        uint32_t *p = (uint32_t *)malloc((v.size() + 1) * sizeof(uint32_t));
        if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(uint32_t));
        return p;
    };
    *n = g.n;
    *m = g.m;
    *offsets = dup(g.offsets);
    *nbrs = dup(g.neighbors);
    *labels = dup(g.labels);
    if (meta) {
        meta[0] = g.labels_count;
        meta[1] = g.max_degree;
        meta[2] = g.max_label_frequency;
    }
    return 0;
}

int gnnpe_host_read_membership(const char *path, uint32_t n, uint32_t p, uint32_t *sorted_nodes, uint32_t *membership)
{
    std::vector<uint32_t> sn, mem;
    std::string err;
    const int rc = gnnpe_host::read_membership(path ? path : "", n, p, &sn, &mem, &err);
    if (rc != 0) {
        gnnpe::set_error("%s", err.c_str());
        return rc;
    }
    if (n) {
        memcpy(sorted_nodes, sn.data(), (size_t)n * 4);
        memcpy(membership, mem.data(), (size_t)n * 4);
    }
    return 0;
}

int gnnpe_host_query_plan(const char *query_graph_path, uint32_t e, uint32_t *n_query_vertices, uint32_t *n_paths,
                          uint32_t **vids, uint32_t **labels, uint32_t **degrees, double **pde)
{
    if (!query_graph_path || !n_query_vertices || !n_paths || !vids || !labels || !degrees || !pde) {
        gnnpe::set_error("gnnpe_host_query_plan: null argument");
        return GNNPE_ERR_ARG;
    }
    gnnpe_host::StaticGraph q;
    std::string err;
    int rc = q.load(query_graph_path, &err);
    if (rc != 0) {
        gnnpe::set_error("%s", err.c_str());
        return rc;
    }
    gnnpe_host::QueryPlan plan;
    if ((rc = gnnpe_host::build_query_plan(q, e, &plan, &err)) != 0) {
        gnnpe::set_error("%s", err.c_str());
        return GNNPE_ERR_ARG;
    }
    auto dup32 = [](const std::vector<uint32_t> &v) {
        uint32_t *p = (uint32_t *)malloc((v.size() + 1) * sizeof(uint32_t));
        if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(uint32_t));
        return p;
    };
    *n_query_vertices = plan.n_vertices;
    *n_paths = plan.n_paths();
    *vids = dup32(plan.vids);
    *labels = dup32(plan.labels);
    *degrees = dup32(plan.degrees);
    *pde = (double *)malloc((plan.pde.size() + 1) * sizeof(double));
    if (*pde && !plan.pde.empty()) memcpy(*pde, plan.pde.data(), plan.pde.size() * sizeof(double));
    return 0;
}

int gnnpe_host_refine(uint32_t n, const uint32_t *offsets, const uint32_t *nbrs, const uint32_t *labels,
                      const char *query_graph_path, const uint32_t *candidate_bitmap, uint64_t limit, uint64_t *answers)
{
    if (!offsets || !nbrs || !labels || !query_graph_path || !candidate_bitmap || !answers) {
        gnnpe::set_error("gnnpe_host_refine: null argument");
        return GNNPE_ERR_ARG;
    }
    gnnpe_host::StaticGraph q, g;
    std::string err;
    int rc = q.load(query_graph_path, &err);
    if (rc != 0) {
        gnnpe::set_error("%s", err.c_str());
        return rc;
    }
    g.n = n;
    g.offsets.assign(offsets, offsets + n + 1);
    g.neighbors.assign(nbrs, nbrs + offsets[n]);
    g.labels.assign(labels, labels + n);
    const uint64_t words = ((uint64_t)n + 31) / 32;
    std::vector<std::vector<uint32_t>> cand(q.n);
    for (uint32_t u = 0; u < q.n; u++)
        for (uint64_t w = 0; w < words; w++)
            for (uint32_t bits = candidate_bitmap[(size_t)u * words + w]; bits; bits &= bits - 1)
                cand[u].push_back((uint32_t)(w * 32 + __builtin_ctz(bits)));
    if (gnnpe_host::refine_count(g, q, cand, limit, answers, &err) != 0) {
        gnnpe::set_error("%s", err.c_str());
        return GNNPE_ERR_ARG;
    }
    return 0;
}

void gnnpe_host_free(void *ptr) { free(ptr); }

}  // extern "C"
