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

int build_match_order(const StaticGraph &query, const std::vector<uint64_t> &cnt, MatchOrder *out, std::string *err)
{
    const uint32_t nq = query.n;
    if (!out || cnt.size() != nq || nq == 0) {
        if (err) *err = "build_match_order: one candidate count per query vertex expected";
        return -2;
    }
    // start vertex: fewest candidates, then larger degree, then smaller id (custom.h:634-654)
    uint32_t start = 0;
    for (uint32_t u = 1; u < nq; u++)
        if (cnt[u] < cnt[start] || (cnt[u] == cnt[start] && query.degree(u) > query.degree(start))) start = u;
    out->order.assign(1, start);
    out->pivot.assign(1, start);
    out->back.clear();
    out->back_off.assign(2, 0);
    std::vector<uint8_t> seen(nq, 0);
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
            if (err) *err = "the query graph is not connected";
            return -2;
        }
        std::vector<uint32_t> earlier;
        for (uint32_t w : out->order)
            if (std::binary_search(query.neighbors.begin() + query.offsets[best], query.neighbors.begin() + query.offsets[best + 1], w))
                earlier.push_back(w);
        out->order.push_back(best);
        out->pivot.push_back(earlier[0]);
        out->back.insert(out->back.end(), earlier.begin() + 1, earlier.end());
        out->back_off.push_back((uint32_t)out->back.size());
        seen[best] = 1;
    }
    return 0;
}

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
    std::vector<uint64_t> cnt(nq);
    for (uint32_t u = 0; u < nq; u++) cnt[u] = cand[u].size();
    MatchOrder mo;
    if (build_match_order(query, cnt, &mo, err) != 0) return -2;
    Search s{data, query, mo.order, mo.pivot, {}, std::vector<uint32_t>(nq, 0), std::vector<uint8_t>(data.n, 0), 0, limit};
    for (uint32_t i = 0; i < nq; i++) s.back.emplace_back(mo.back.begin() + mo.back_off[i], mo.back.begin() + mo.back_off[i + 1]);
    const uint32_t start = mo.order[0];
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
