// online_capi.cpp -- C-ABI of the host refinement (include/gnnpe_online.h).  Compiled into libgnnpe_online.so, which links
// against libgnnpe_hip.so for the loader and the error text.  Out of SURVEY section 8's scope (frozen).
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/gnnpe_online.h"
#include "graph_loader.h"
#include "refine.h"

namespace gnnpe {
void set_error(const char *fmt, ...);
}

extern "C" {

int gnnpe_host_refine(uint32_t n, const uint32_t *offsets, const uint32_t *nbrs, const uint32_t *labels,
                      const char *query_graph_path, const uint32_t *candidate_bitmap, uint64_t limit, uint64_t *answers)
{
    if (!offsets || !nbrs || !labels || !query_graph_path || !candidate_bitmap || !answers) {
        gnnpe::set_error("gnnpe_host_refine: null argument");
        return GNNPE_ERR_ARG;
    }
    gnnpe_host::StaticGraph q, g;
    std::string err;
    int rc = q.load(query_graph_path, &err, true);
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

}  // extern "C"
