// query_plan.cpp -- see query_plan.h
#include "query_plan.h"

#include <algorithm>
#include <set>

#include "../../include/gnnpe_hip.h"

namespace gnnpe_host {

namespace {

struct PlanPath {
    uint32_t v[3];
    uint32_t weight;
};

// dfs_query (custom.h:94-119) for 3-vertex paths
void dfs(const StaticGraph &q, std::vector<uint32_t> &path, std::set<std::vector<uint32_t>> &seen,
         std::vector<std::vector<uint32_t>> &out)
{
    if (path.size() == 3) {
        if (seen.count(path)) return;
        std::vector<uint32_t> rev(path.rbegin(), path.rend());
        if (seen.count(rev)) return;
        out.push_back(path);
        seen.insert(path);
        return;
    }
    const uint32_t node = path.back();
    for (uint32_t i = q.offsets[node]; i < q.offsets[node + 1]; i++) {
        const uint32_t nb = q.neighbors[i];
        if (std::find(path.begin(), path.end(), nb) != path.end()) continue;
        path.push_back(nb);
        dfs(q, path, seen, out);
        path.pop_back();
    }
}

}  // namespace

int build_query_plan(const StaticGraph &q, uint32_t e, QueryPlan *out, std::string *err)
{
    if (!out || e == 0) {
        if (err) *err = "build_query_plan: null output / e = 0";
        return -2;
    }
    const uint32_t n = q.n, L = 3;
    out->n_vertices = n;
    out->L = L;
    out->e = e;
    out->vids.clear();
    out->labels.clear();
    out->degrees.clear();
    out->pde.clear();
    out->pde_label.clear();

    // main.cpp:139-146
    std::vector<std::vector<uint32_t>> all_paths;
    std::set<std::vector<uint32_t>> seen;
    for (uint32_t node = 0; node < n; node++) {
        std::vector<uint32_t> path = {node};
        dfs(q, path, seen, all_paths);
    }

    // gen_vde (custom.h:513-544): x from the label, nx summed over ascending neighbours from 0.0, vde = x + nx
    const uint32_t n_labels = std::max<uint32_t>(q.labels_count, 1);
    std::vector<double> table((size_t)n_labels * e), x((size_t)n * e), vde((size_t)n * e);
    if (gnnpe_host_label_table(n_labels, e, table.data()) != 0) {
        if (err) *err = "label table failed";
        return -2;
    }
    for (uint32_t v = 0; v < n; v++)
        for (uint32_t k = 0; k < e; k++) x[(size_t)v * e + k] = table[(size_t)q.labels[v] * e + k];
    for (uint32_t v = 0; v < n; v++)
        for (uint32_t k = 0; k < e; k++) {
            double nx = 0.0;
            for (uint32_t i = q.offsets[v]; i < q.offsets[v + 1]; i++) nx += x[(size_t)q.neighbors[i] * e + k];
            vde[(size_t)v * e + k] = x[(size_t)v * e + k] + nx;
        }

    // gen_query_pde (custom.h:574-631): weight = sum of degrees; std::sort by weight, descending -- the same
    // library algorithm and comparator as the reference, so ties fall the same way
    std::vector<PlanPath> paths(all_paths.size());
    for (size_t i = 0; i < all_paths.size(); i++) {
        paths[i].weight = 0;
        for (uint32_t j = 0; j < L; j++) {
            paths[i].v[j] = all_paths[i][j];
            paths[i].weight += q.degree(all_paths[i][j]);
        }
    }
    std::sort(paths.begin(), paths.end(), [](const PlanPath &a, const PlanPath &b) { return a.weight > b.weight; });
    std::set<uint32_t> covered;
    for (const PlanPath &p : paths) {
        uint32_t hit = 0;
        for (uint32_t j = 0; j < L; j++) hit += covered.count(p.v[j]) ? 1u : 0u;
        if (hit != L) {
            for (uint32_t j = 0; j < L; j++) {
                covered.insert(p.v[j]);
                out->vids.push_back(p.v[j]);
                out->labels.push_back(q.labels[p.v[j]]);
                out->degrees.push_back(q.degree(p.v[j]));
                for (uint32_t k = 0; k < e; k++) {
                    out->pde.push_back(vde[(size_t)p.v[j] * e + k]);
                    out->pde_label.push_back(x[(size_t)p.v[j] * e + k]);
                }
            }
        }
        if (covered.size() == n) break;
    }
    return 0;
}

}  // namespace gnnpe_host
