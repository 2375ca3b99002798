// graph_loader.h -- R0: the data-graph loader of the offline path (host C++17).
//
// Mirrors Static_Graph::loadGraphFromFile (GNN-PE/libsrc/graph/graph.cpp:163-242) and the
// accessors the offline step uses (include/graph/graph.h:154-176): text `.graph` -> CSR with
// ascending neighbour lists + labels, plus the metadata printGraphMetaData (graph.cpp:244-247)
// prints.  Unlike the reference it validates what it reads (SURVEY 8(a) preconditions) and reports
// errors instead of reading out of bounds.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace gnnpe_host {

struct StaticGraph {
    uint32_t n = 0, m = 0;                 // vertices_count_, edges_count_
    std::vector<uint32_t> offsets;         // n + 1
    std::vector<uint32_t> neighbors;       // 2 m, each list ascending
    std::vector<uint32_t> labels;          // n
    uint32_t labels_count = 0;             // max(#distinct labels, max label + 1)   (graph.cpp:223)
    uint32_t max_degree = 0;
    uint32_t max_label_frequency = 0;

    // Non-simple input: duplicate `e` lines.  The reference stores what it reads (graph.cpp:211-218: no de-duplication), so
    // `degree` and the neighbour sum of gen_vde count the repeats (graph.h:154-156, custom.h:527-534), while its hash set drops a
    // path met again (custom.h:68-77): the enumeration sees the de-duplicated rows.  `simple` is false for such a file; offsets /
    // neighbors are then the rows as the reference holds them and simple_offsets / simple_neighbors the rows its enumeration
    // amounts to (empty for a simple graph: both views are the same arrays).
    // A SELF-LOOP line is refused in every mode: for `e u u` the reference computes both slots before it advances either cursor
    // (graph.cpp:211-218), writes u into ONE slot twice, advances u's cursor by two and so leaves a slot of `new ui[2m]`
    // uninitialised -- what it enumerates then depends on what that memory held (profiles/r06_selfloop_reference.txt: in practice a
    // spurious one-directional neighbour 0).  There is no defined result to be identical to.
    bool simple = true;
    std::vector<uint32_t> simple_offsets, simple_neighbors;
    const std::vector<uint32_t> &enum_offsets() const { return simple ? offsets : simple_offsets; }
    const std::vector<uint32_t> &enum_neighbors() const { return simple ? neighbors : simple_neighbors; }

    // 0 = ok, -1 = cannot open (reference: message + exit(-1), graph.cpp:166-169), -2 = malformed.  strict: a file with a
    // duplicate edge is malformed too (rounds 1-5; `--strict`)
    int load(const std::string &path, std::string *err, bool strict = false);
    // the two lines of printGraphMetaData (graph.cpp:245-246)
    std::string metadata_text() const;
    uint32_t degree(uint32_t v) const { return offsets[v + 1] - offsets[v]; }
};

// R1: membership.txt (GNN-PE/src/main.cpp:77-85).  Line i = "<vertex> <partition>"; the line order
// is the processing order.  Returns 0, or <0 with *err set (missing file, short file, duplicate or
// out-of-range vertex, partition >= p) -- the reference checks none of these.
int read_membership(const std::string &path, uint32_t n, uint32_t p, std::vector<uint32_t> *sorted_nodes,
                    std::vector<uint32_t> *membership, std::string *err);

}  // namespace gnnpe_host
