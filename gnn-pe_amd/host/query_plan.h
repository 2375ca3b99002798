// query_plan.h -- query side of the online FILTER, host C++17.
//
// Restates what the reference does to a query graph before it touches the index (GNN-PE/src/main.cpp:136-151):
//   dfs_query        custom.h:94-119   every simple 3-vertex path of the query graph, from every vertex in id
//                                      order, neighbours ascending, kept unless it or its reverse was kept before
//   gen_vde          custom.h:513-544  x, nx, vde of the query vertices (same arithmetic as the data side)
//   gen_query_pde    custom.h:574-631  per path: vids, labels, degrees, weight = sum of degrees, pde; paths sorted by
//                                      weight (descending, std::sort) and taken greedily while they still cover a new
//                                      vertex, until every query vertex is covered
// The result is the query plan: the only query-side input of the filter (gnnpe_filter_candidates).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "graph_loader.h"

namespace gnnpe_host {

struct QueryPlan {
    uint32_t n_vertices = 0, L = 3, e = 0;
    std::vector<uint32_t> vids, labels, degrees;  // n_paths x L
    std::vector<double> pde, pde_label;           // n_paths x e*L
    uint32_t n_paths() const { return L ? (uint32_t)(vids.size() / L) : 0; }
};

// 0 = ok; <0 with *err set
int build_query_plan(const StaticGraph &query, uint32_t e, QueryPlan *out, std::string *err);

}  // namespace gnnpe_host
