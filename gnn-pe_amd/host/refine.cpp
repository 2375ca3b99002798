// refine.cpp -- see refine.h
#include "refine.h"

#include <algorithm>

namespace gnnpe_host {

namespace {

struct Search {
    const StaticGraph &g, &q;
    std::vector<uint32_t> order, pivot;            // matching order; an earlier neighbour of order[i] for i >= 1
    std::vector<std::vector<uint32_t>> back;       // the other earlier neighbours of order[i]
    std::vector<uint32_t> image;                   // query vertex -> data vertex
    std::vector<uint8_t> used;                     // data vertex taken
    uint64_t count = 0, limit;

    bool edge(uint32_t a, uint32_t b) const
    {
        return std::binary_search(g.neighbors.begin() + g.offsets[a], g.neighbors.begin() + g.offsets[a + 1], b);
    }
    void extend(size_t depth)
    {
        if (depth == order.size()) {
            count++;
            return;
        }
        const uint32_t u = order[depth], lab = q.labels[u], deg = q.degree(u);
        const uint32_t p = image[pivot[depth]];
        for (uint32_t i = g.offsets[p]; i < g.offsets[p + 1] && count < limit; i++) {
            const uint32_t v = g.neighbors[i];
            if (used[v] || g.labels[v] != lab || g.degree(v) < deg) continue;
            bool ok = true;
            for (uint32_t w : back[depth])
                if (!edge(v, image[w])) {
                    ok = false;
                    break;
                }
            if (!ok) continue;
            image[u] = v;
            used[v] = 1;
            extend(depth + 1);
            used[v] = 0;
        }
    }
};

}  // namespace

int refine_count(const StaticGraph &data, const StaticGraph &query, const std::vector<std::vector<uint32_t>> &cand,
                 uint64_t limit, uint64_t *answers, std::string *err)
{
    const uint32_t nq = query.n;
    if (!answers || cand.size() != nq) {
        if (err) *err = "refine_count: one candidate list per query vertex expected";
        return -2;
    }
    *answers = 0;
    if (nq == 0 || limit == 0) return 0;
    // start vertex: fewest candidates, then larger degree, then smaller id (custom.h:634-654)
    uint32_t start = 0;
    for (uint32_t u = 1; u < nq; u++)
        if (cand[u].size() < cand[start].size() ||
            (cand[u].size() == cand[start].size() && query.degree(u) > query.degree(start)))
            start = u;
    Search s{data, query, {}, {}, {}, std::vector<uint32_t>(nq, 0), std::vector<uint8_t>(data.n, 0), 0, limit};
    // a connected order from the start: always take the unvisited vertex with the most visited neighbours
    std::vector<uint8_t> seen(nq, 0);
    s.order.push_back(start);
    s.pivot.push_back(start);
    s.back.emplace_back();
    seen[start] = 1;
    for (uint32_t step = 1; step < nq; step++) {
        uint32_t best = nq, best_links = 0;
        for (uint32_t u = 0; u < nq; u++) {
            if (seen[u]) continue;
            uint32_t links = 0;
            for (uint32_t i = query.offsets[u]; i < query.offsets[u + 1]; i++) links += seen[query.neighbors[i]];
            if (links > best_links) {
                best = u;
                best_links = links;
            }
        }
        if (best == nq) {
            if (err) *err = "refine_count: the query graph is not connected";
            return -2;
        }
        std::vector<uint32_t> earlier;
        for (uint32_t w : s.order)
            if (std::binary_search(query.neighbors.begin() + query.offsets[best], query.neighbors.begin() + query.offsets[best + 1], w))
                earlier.push_back(w);
        s.order.push_back(best);
        s.pivot.push_back(earlier[0]);
        s.back.emplace_back(earlier.begin() + 1, earlier.end());
        seen[best] = 1;
    }
    for (uint32_t v : cand[start]) {
        if (s.count >= limit) break;
        if (v >= data.n) {
            if (err) *err = "refine_count: candidate id out of range";
            return -2;
        }
        s.image[start] = v;
        s.used[v] = 1;
        s.extend(1);
        s.used[v] = 0;
    }
    *answers = std::min(s.count, limit);
    return 0;
}

}  // namespace gnnpe_host
