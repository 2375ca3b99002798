// graph_loader.cpp -- see graph_loader.h.  Hand-rolled tokenizer over the file bytes; the `e` lines (the bulk) are
// parsed, scattered into the rows and sorted by a few host threads (the reference uses iostream extraction on one
// thread, ~1.3 s per 100K/1M graph).
#include "graph_loader.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
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

// whole file mapped read-only (no copy; the parser threads fault the pages in)
struct FileView {
    const char *data = nullptr;
    size_t size = 0;
    bool mapped = false;
    ~FileView()
    {
        if (mapped && data) munmap(const_cast<char *>(data), size);
    }
    size_t bytes() const { return size; }
};

bool read_file(const std::string &path, FileView *v)
{
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return false;
    }
    v->size = (size_t)st.st_size;
    if (v->size) {
        void *p = mmap(nullptr, v->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) {
            close(fd);
            return false;
        }
        v->data = (const char *)p;
        v->mapped = true;
    }
    close(fd);
    return true;
}

}  // namespace

int StaticGraph::load(const std::string &path, std::string *err, bool strict)
{
    FileView buf;
    if (!read_file(path, &buf)) {
        if (err) *err = "Can not open the graph file " + path + " .";  // graph.cpp:167
        return -1;
    }
    Cursor c{buf.data, buf.data + buf.size};
    char type;
    if (!c.next_char(&type) || !c.next_u32(&n) || !c.next_u32(&m)) {  // "t n m", graph.cpp:172
        if (err) *err = "malformed header (expected `t <vertices> <edges>`)";
        return -2;
    }
    offsets.assign((size_t)n + 1, 0);
    neighbors.assign((size_t)m * 2, 0);
    labels.assign(n, 0);
    std::unordered_map<uint32_t, uint32_t> freq;
    uint32_t max_label = 0, next_vertex = 0;
    max_degree = 0;
    // ---- `v` lines (graph.cpp:185-206): sequential, they build the offsets incrementally ----
    const char *edge_begin = c.end;
    while (true) {
        c.skip_ws();
        if (c.p >= c.end) break;
        if (*c.p != 'v') {
            edge_begin = c.p;
            break;
        }
        c.p++;
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
    }
    if (edge_begin < c.end && *edge_begin == 'e' && next_vertex != n) {
        if (err) *err = "`e` line before all `v` lines";
        return -2;
    }

    // ---- `e` lines (graph.cpp:207-219): the bulk of the file, parsed by byte ranges cut at line ends.  The lists are
    // sorted afterwards (graph.cpp:231-233), so the order in which the threads append to a row does not matter. ----
    const size_t edge_bytes = (size_t)(c.end - edge_begin);
    unsigned T = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (edge_bytes < (1u << 20)) T = 1;
    std::vector<const char *> cut(T + 1, c.end);
    cut[0] = edge_begin;
    for (unsigned t = 1; t < T; t++) {
        const char *q = edge_begin + edge_bytes * t / T;
        while (q < c.end && *q != '\n') q++;
        cut[t] = q < c.end ? q + 1 : c.end;
    }
    std::vector<std::vector<uint32_t>> pairs(T);
    std::vector<std::string> errs(T);
    const uint32_t nn = n;
    auto parse = [&](unsigned t) {
        Cursor k{cut[t], cut[t + 1]};
        std::vector<uint32_t> &out = pairs[t];
        out.reserve((size_t)(cut[t + 1] - cut[t]) / 6);
        char type;
        while (k.next_char(&type)) {
            if (type == 'e') {
                uint32_t a, b;
                if (!k.next_u32(&a) || !k.next_u32(&b)) {
                    errs[t] = "malformed `e` line";
                    return;
                }
                if (a >= nn || b >= nn) {
                    errs[t] = "edge " + std::to_string(a) + " " + std::to_string(b) + " does not fit the declared degrees";
                    return;
                }
                if (a == b) {  // see graph_loader.h: the reference's own result for such a line is undefined
                    errs[t] = "self-loop at vertex " + std::to_string(a) + ": the reference's loader writes both ends of `e v v` into one "
                              "slot and leaves the next one uninitialised (graph.cpp:211-218), so its own result for such a file is undefined";
                    return;
                }
                out.push_back(a);
                out.push_back(b);
            } else if (type == 'v') {
                uint32_t id = 0;
                k.next_u32(&id);
                errs[t] = "`v` lines must list ids 0..n-1 in ascending order (got " + std::to_string(id) + ")";
                return;
            } else {
                errs[t] = std::string("unexpected record type `") + type + "`";
                return;
            }
        }
    };
    auto run = [&](auto &&fn) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(fn, t);
        fn(0u);
        for (auto &x : th) x.join();
    };
    run(parse);
    for (unsigned t = 0; t < T; t++)
        if (!errs[t].empty()) {  // the earliest error in file order
            if (err) *err = errs[t];
            return -2;
        }
    // scatter into the rows: thread t owns the rows of a contiguous vertex range and scans all parsed pairs for them
    // (sequential reads, writes confined to its own slice of `neighbors`; no atomics, no shared cache lines)
    std::vector<uint32_t> cursor(n, 0);
    auto fill = [&](unsigned t) {
        const uint32_t lo = (uint32_t)((uint64_t)nn * t / T), hi = (uint32_t)((uint64_t)nn * (t + 1) / T);
        for (unsigned src = 0; src < T; src++) {
            const std::vector<uint32_t> &in = pairs[src];
            for (size_t i = 0; i + 1 < in.size(); i += 2) {
                const uint32_t a = in[i], b = in[i + 1];
                for (int side = 0; side < 2; side++) {
                    const uint32_t v = side ? b : a, w = side ? a : b;
                    if (v < lo || v >= hi) continue;
                    if (cursor[v] >= degree(v)) {
                        if (errs[t].empty())
                            errs[t] = "edge " + std::to_string(a) + " " + std::to_string(b) + " does not fit the declared degrees";
                        return;
                    }
                    neighbors[offsets[v] + cursor[v]++] = w;
                }
            }
        }
    };
    run(fill);
    for (unsigned t = 0; t < T; t++)
        if (!errs[t].empty()) {
            if (err) *err = errs[t];
            return -2;
        }
    uint64_t filled = 0;
    for (unsigned t = 0; t < T; t++) filled += pairs[t].size();
    if (next_vertex != n || filled != (uint64_t)offsets[n]) {
        if (err) *err = "vertex / edge lines do not match the declared counts and degrees";
        return -2;
    }
    pairs.clear();
    labels_count = std::max<uint32_t>((uint32_t)freq.size(), n ? max_label + 1 : 0);  // graph.cpp:223
    max_label_frequency = 0;
    for (auto &kv : freq) max_label_frequency = std::max(max_label_frequency, kv.second);
    // graph.cpp:231-233: every list ascending; repeats and self-loops are found on the way (see graph_loader.h)
    std::atomic<uint32_t> next_block{0};
    std::atomic<uint32_t> dup_vertex{0xFFFFFFFFu}, loop_vertex{0xFFFFFFFFu};
    auto note = [](std::atomic<uint32_t> &slot, uint32_t v) {
        uint32_t cur = slot.load();
        while (v < cur && !slot.compare_exchange_weak(cur, v)) {}
    };
    auto sort_rows = [&](unsigned) {
        for (;;) {
            const uint32_t v0 = next_block.fetch_add(4096, std::memory_order_relaxed);
            if (v0 >= nn) return;
            for (uint32_t v = v0; v < std::min(nn, v0 + 4096); v++) {
                std::sort(neighbors.begin() + offsets[v], neighbors.begin() + offsets[v + 1]);
                bool dup = false, loop = false;
                for (uint32_t j = offsets[v]; j < offsets[v + 1]; j++) {
                    loop |= neighbors[j] == v;
                    dup |= j > offsets[v] && neighbors[j] == neighbors[j - 1];
                }
                if (loop) note(loop_vertex, v);
                if (dup) note(dup_vertex, v);
            }
        }
    };
    run(sort_rows);
    simple_offsets.clear();
    simple_neighbors.clear();
    if (loop_vertex.load() != 0xFFFFFFFFu) {
        if (err)
            *err = "self-loop at vertex " + std::to_string(loop_vertex.load()) + ": the reference's loader writes both ends of `e v v` into one "
                   "slot and leaves the next one uninitialised (graph.cpp:211-218), so its own result for such a file is undefined";
        return -2;
    }
    simple = dup_vertex.load() == 0xFFFFFFFFu;
    if (!simple && strict) {
        if (err) *err = "duplicate edge at vertex " + std::to_string(dup_vertex.load()) + " (--strict: simple graphs only)";
        return -2;
    }
    if (!simple) {
        // the rows the reference's enumeration amounts to: every list without its repeats
        simple_offsets.assign((size_t)n + 1, 0);
        simple_neighbors.reserve(neighbors.size());
        for (uint32_t v = 0; v < nn; v++) {
            for (uint32_t j = offsets[v]; j < offsets[v + 1]; j++)
                if (j == offsets[v] || neighbors[j] != neighbors[j - 1]) simple_neighbors.push_back(neighbors[j]);
            simple_offsets[v + 1] = (uint32_t)simple_neighbors.size();
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
    FileView buf;
    if (!read_file(path, &buf)) {
        if (err) *err = "cannot open " + path;
        return -1;
    }
    Cursor c{buf.data, buf.data + buf.size};
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
