// refine.h -- the REFINEMENT half of the reference's `-m online` (host C++17), so that the filter's candidate
// sets can be turned into the answer count without the reference binary.
//
// What the reference computes (GNN-PE/include/custom.h:634-932, QuickSI-style exploration): the number of
// embeddings f of the query graph into the data graph -- injective, label(f(u)) = label(u),
// degree(u) <= degree(f(u)), every query edge mapped onto a data edge -- whose image of the START vertex lies in
// that vertex' candidate set; the other query vertices are extended through the data graph's adjacency and are
// NOT checked against their candidate sets (generateValidCandidates, custom.h:755-797).  The start vertex is the
// query vertex with the fewest candidates, ties to the larger query degree, then to the smaller id
// (selectGQLStartVertex, custom.h:634-654).  Counting stops at the answer limit (-n, custom.h:842-846).
// The count does not depend on the matching order, so this is an independent backtracking search over the same
// definition, not a transcription of the reference's loop; tests pin it to the reference's answers.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "graph_loader.h"

namespace gnnpe_host {

// The matching order both refinements (host below, device in csrc/gnnpe_refine.hip) walk: order[0] = the start vertex
// (custom.h:634-654), then always the unvisited vertex with the most visited neighbours; pivot[i] = one earlier
// neighbour of order[i] (extension goes through the data neighbours of its image), back = the other earlier
// neighbours (edge checks), flattened with back_off[i]..back_off[i+1].
struct MatchOrder {
    std::vector<uint32_t> order, pivot, back, back_off;
};
int build_match_order(const StaticGraph &query, const std::vector<uint64_t> &candidate_counts, MatchOrder *out, std::string *err);

// candidates[u] = ascending data vertex ids of query vertex u.  Returns 0 and *answers, or <0 with *err
// (disconnected query graph, size mismatch).
int refine_count(const StaticGraph &data, const StaticGraph &query, const std::vector<std::vector<uint32_t>> &candidates,
                 uint64_t limit, uint64_t *answers, std::string *err);

}  // namespace gnnpe_host
