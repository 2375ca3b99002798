// graph_loader.cpp -- see graph_loader.h.  Single pass over the file bytes with a hand-rolled
// tokenizer (the reference uses iostream extraction, ~1.3 s per 100K/1M graph).
#include "graph_loader.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <unordered_map>

namespace gnnpe_host {

namespace {

struct Cursor {
    const char *p, *end;
    void skip_ws()
    {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
    }
    bool next_char(char *c)
    {
        skip_ws();
        if (p >= end) return false;
        *c = *p++;
        return true;
    }
    bool next_u32(uint32_t *v)
    {
        skip_ws();
        if (p >= end || *p < '0' || *p > '9') return false;
        uint64_t x = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            x = x * 10 + (uint64_t)(*p - '0');
            if (x > 0xFFFFFFFFull) return false;
            p++;
        }
        *v = (uint32_t)x;
        return true;
    }
};

bool read_file(const std::string &path, std::vector<char> *buf)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf->resize(sz > 0 ? (size_t)sz : 0);
    size_t got = sz > 0 ? fread(buf->data(), 1, (size_t)sz, f) : 0;
    fclose(f);
    buf->resize(got);
    return true;
}

}  // namespace

int StaticGraph::load(const std::string &path, std::string *err)
{
    std::vector<char> buf;
    if (!read_file(path, &buf)) {
        if (err) *err = "Can not open the graph file " + path + " .";  // graph.cpp:167
        return -1;
    }
    Cursor c{buf.data(), buf.data() + buf.size()};
    char type;
    if (!c.next_char(&type) || !c.next_u32(&n) || !c.next_u32(&m)) {  // "t n m", graph.cpp:172
        if (err) *err = "malformed header (expected `t <vertices> <edges>`)";
        return -2;
    }
    offsets.assign((size_t)n + 1, 0);
    neighbors.assign((size_t)m * 2, 0);
    labels.assign(n, 0);
    std::vector<uint32_t> cursor(n, 0);
    std::unordered_map<uint32_t, uint32_t> freq;
    uint32_t max_label = 0, next_vertex = 0;
    uint64_t filled = 0;
    max_degree = 0;
    while (c.next_char(&type)) {
        if (type == 'v') {  // graph.cpp:185-206
            uint32_t id, label, degree;
            if (!c.next_u32(&id) || !c.next_u32(&label) || !c.next_u32(&degree)) {
                if (err) *err = "malformed `v` line";
                return -2;
            }
            // offsets_[id+1] = offsets_[id] + degree (graph.cpp:192) only works for ascending dense ids
            if (id != next_vertex || id >= n) {
                if (err) *err = "`v` lines must list ids 0..n-1 in ascending order (got " + std::to_string(id) + ")";
                return -2;
            }
            next_vertex++;
            labels[id] = label;
            if ((uint64_t)offsets[id] + degree > (uint64_t)m * 2) {
                if (err) *err = "degree fields exceed 2*m";
                return -2;
            }
            offsets[id + 1] = offsets[id] + degree;
            max_degree = std::max(max_degree, degree);
            auto it = freq.find(label);
            if (it == freq.end()) {
                freq.emplace(label, 1);
                max_label = std::max(max_label, label);
            } else {
                it->second++;
            }
        } else if (type == 'e') {  // graph.cpp:207-219
            uint32_t a, b;
            if (!c.next_u32(&a) || !c.next_u32(&b)) {
                if (err) *err = "malformed `e` line";
                return -2;
            }
            if (next_vertex != n) {
                if (err) *err = "`e` line before all `v` lines";
                return -2;
            }
            if (a >= n || b >= n || cursor[a] >= degree(a) || cursor[b] >= degree(b)) {
                if (err) *err = "edge " + std::to_string(a) + " " + std::to_string(b) + " does not fit the declared degrees";
                return -2;
            }
            neighbors[offsets[a] + cursor[a]++] = b;
            neighbors[offsets[b] + cursor[b]++] = a;
            filled += 2;
        } else {
            if (err) *err = std::string("unexpected record type `") + type + "`";
            return -2;
        }
    }
    if (next_vertex != n || filled != (uint64_t)offsets[n]) {
        if (err) *err = "vertex / edge lines do not match the declared counts and degrees";
        return -2;
    }
    labels_count = std::max<uint32_t>((uint32_t)freq.size(), n ? max_label + 1 : 0);  // graph.cpp:223
    max_label_frequency = 0;
    for (auto &kv : freq) max_label_frequency = std::max(max_label_frequency, kv.second);
    for (uint32_t v = 0; v < n; v++) {  // graph.cpp:231-233
        std::sort(neighbors.begin() + offsets[v], neighbors.begin() + offsets[v + 1]);
        // the closed-form enumeration needs a simple graph (SURVEY 8(a) preconditions)
        for (uint32_t j = offsets[v]; j < offsets[v + 1]; j++) {
            if (neighbors[j] == v || (j > offsets[v] && neighbors[j] == neighbors[j - 1])) {
                if (err) *err = "self-loop or duplicate edge at vertex " + std::to_string(v) + " (simple graphs only)";
                return -2;
            }
        }
    }
    return 0;
}

std::string StaticGraph::metadata_text() const
{
    // graph.cpp:245-246
    return "|V|: " + std::to_string(n) + ", |E|: " + std::to_string(m) + ", |Σ|: " + std::to_string(labels_count) +
           "\nMax Degree: " + std::to_string(max_degree) + ", Max Label Frequency: " + std::to_string(max_label_frequency) +
           "\n";
}

int read_membership(const std::string &path, uint32_t n, uint32_t p, std::vector<uint32_t> *sorted_nodes,
                    std::vector<uint32_t> *membership, std::string *err)
{
    std::vector<char> buf;
    if (!read_file(path, &buf)) {
        if (err) *err = "cannot open " + path;
        return -1;
    }
    Cursor c{buf.data(), buf.data() + buf.size()};
    sorted_nodes->assign(n, 0);
    membership->assign(n, 0);
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t i = 0; i < n; i++) {  // main.cpp:81-84
        uint32_t v, part;
        if (!c.next_u32(&v) || !c.next_u32(&part)) {
            if (err) *err = path + ": line " + std::to_string(i + 1) + " missing (need one line per vertex)";
            return -2;
        }
        if (v >= n || seen[v]) {
            if (err) *err = path + ": vertex " + std::to_string(v) + " out of range or listed twice";
            return -2;
        }
        if (part >= p) {
            if (err) *err = path + ": partition " + std::to_string(part) + " >= -p " + std::to_string(p);
            return -2;
        }
        seen[v] = 1;
        (*sorted_nodes)[i] = v;
        (*membership)[v] = part;
    }
    return 0;
}

}  // namespace gnnpe_host
