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

    // 0 = ok, -1 = cannot open (reference: message + exit(-1), graph.cpp:166-169), -2 = malformed
    int load(const std::string &path, std::string *err);
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
