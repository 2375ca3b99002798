// ref_online.cpp -- harness around the UNMODIFIED reference headers (TEST INFRASTRUCTURE ONLY).
//
// Compiled by oracle/Makefile against the sources where they lie in /root/reference/GNN-PE (nothing is
// copied).  It drives the reference's own online filter the way GNN-PE/src/main.cpp:121-185 does -- gen_vde,
// gen_pde, Partition (R-tree + auxiliary index), dfs_query, gen_query_pde, Partition::query -- and stops
// before the refinement to dump what the filter produced, so that a filter built elsewhere can be compared
// with it; or it reads candidate sets from a file and runs the reference's own refinement on them.
//
// usage: ref_online <dataset dir/> <data.graph> <query.graph> <p> dump   <out.bin>
//        ref_online <dataset dir/> <data.graph> <query.graph> <p> refine <candidates.bin>
//        ref_online <dataset dir/> <data.graph> <query.graph> <p> aux    <out.bin> [e]
//   aux out.bin    : what the reference's Partition constructor leaves behind (custom.h:205-266, 268-364), per partition:
//                    uint32 n_paths, n_nodes, L, D; the partition's paths (vids[L] labels[L] degrees[L] uint32,
//                    pde[D] pde_label[D] double, each); then per node block id: key (double), degrees[L] (uint32),
//                    label_mbr[2D] (double).  Uses the index.dat found in the partition directory (or inserts one).
//   out.bin        : uint32 n_query_vertices, n_query_paths, L, e; per query path: vids[L] labels[L] degrees[L]
//                    (uint32), pde[eL] pde_label[eL] (double); then per query vertex: uint32 count, ids ascending
//   candidates.bin : uint32 n_query_vertices; per query vertex: uint32 count, ids
// Include order follows GNN-PE/src/main.cpp:1-16 (gendef.h defines min/max macros).
#include "./rtree/rtree.h"
#include "./rtree/rtnode.h"
#include "./rtree/entry.h"
#include "./blockfile/blk_file.h"
#include "./blockfile/cache.h"
#include "./linlist/linlist.h"
#include "./rtree/rtree_cmd.h"
#include "rand.h"
#include "cdf.h"

#include "./graph/graph.h"
#include "custom.h"

#undef min
#undef max

#include <cstdint>
#include <cstdio>

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: ref_online <dataset dir/> <data.graph> <query.graph> <p> dump|refine <file>\n");
        return 2;
    }
    const string dataset_path = argv[1], mode = argv[5];
    partition_num = (ui)atoi(argv[4]);
    if (argc > 7) vde_dim = (ui)atoi(argv[7]);  // optional embedding width (main.cpp:53 `-e`; default 2, custom.h:45)
    path_length = 3;  // main.cpp:58 with the only working -l 2
    pde_dim = vde_dim * path_length;
    MAX_LIMIT = UINT_MAX;

    Static_Graph *data = new Static_Graph(true);
    data->loadGraphFromFile(argv[2]);
    Static_Graph *query = new Static_Graph(true);
    query->loadGraphFromFile(argv[3]);
    const ui nq = query->getVerticesCount();

    if (mode == "refine") {  // the reference's refinement on candidate sets produced elsewhere (main.cpp:176-179)
        FILE *f = fopen(argv[6], "rb");
        if (!f) return 3;
        uint32_t n = 0;
        if (fread(&n, 4, 1, f) != 1 || n != nq) return 4;
        vector<set<ui>> cand(nq);
        for (ui i = 0; i < nq; i++) {
            uint32_t c = 0;
            if (fread(&c, 4, 1, f) != 1) return 4;
            vector<uint32_t> ids(c);
            if (c && fread(ids.data(), 4, c, f) != c) return 4;
            cand[i].insert(ids.begin(), ids.end());
        }
        fclose(f);
        ui answer_num = 0;
        refinement(data, query, cand, answer_num);
        cout << "Answer Number: " << answer_num << endl;
        return 0;
    }

    // main.cpp:123-134
    vector<Vertex> data_vertices = gen_vde(data);
    vector<Path> data_paths = gen_pde(data_vertices, dataset_path + "/gnn-pe/all_paths.txt");
    const string partitions_path = dataset_path + "gnn-pe/partitions/";
    vector<Partition> partitions;
    for (ui i = 0; i < partition_num; i++) {
        Partition partition(data_paths, partitions_path + "partition-" + to_string(i) + "/");
        partitions.push_back(partition);
    }
    if (mode == "aux") {  // the Partition constructor's outputs, nothing else
        FILE *fa = fopen(argv[6], "wb");
        if (!fa) return 3;
        for (ui pid = 0; pid < partition_num; pid++) {
            const Partition &P = partitions[pid];
            const uint32_t hdr[4] = {(uint32_t)P.paths.size(), (uint32_t)P.node_num, path_length, pde_dim};
            fwrite(hdr, 4, 4, fa);
            for (const auto &q : P.paths) {
                fwrite(q.vids.data(), 4, path_length, fa);
                fwrite(q.labels.data(), 4, path_length, fa);
                fwrite(q.degrees.data(), 4, path_length, fa);
                fwrite(q.pde.data(), 8, pde_dim, fa);
                fwrite(q.pde_label.data(), 8, pde_dim, fa);
            }
            for (ui k = 0; k < P.node_num; k++) {
                fwrite(&P.auxiliary_index[k].key, 8, 1, fa);
                fwrite(P.auxiliary_index[k].degrees.data(), 4, path_length, fa);
                fwrite(P.auxiliary_index[k].label_mbr.data(), 8, 2 * pde_dim, fa);
            }
        }
        fclose(fa);
        return 0;
    }
    // main.cpp:139-151
    vector<vector<ui>> all_paths;
    unordered_set<vector<ui>, VectorHash> all_paths_set;
    for (ui node = 0; node < nq; node++) {
        vector<ui> path = {node};
        dfs_query(node, path_length - 2, path, query, all_paths, all_paths_set);
    }
    vector<Vertex> query_vertices = gen_vde(query);
    vector<Query_Path> query_paths = gen_query_pde(query_vertices, all_paths);
    Query_Plan Q(query_paths);
    // main.cpp:153-171 (sequential: the dump does not time anything)
    vector<set<ui>> candidate_set(nq);
    for (ui pid = 0; pid < partition_num; pid++) {
        vector<set<ui>> part(nq);
        partitions[pid].query(nq, part, Q);
        for (ui i = 0; i < nq; i++) candidate_set[i].insert(part[i].begin(), part[i].end());
    }

    FILE *f = fopen(argv[6], "wb");
    if (!f) return 3;
    const uint32_t hdr[4] = {nq, (uint32_t)query_paths.size(), path_length, vde_dim};
    fwrite(hdr, 4, 4, f);
    for (const auto &q : query_paths) {
        fwrite(q.vids.data(), 4, path_length, f);
        fwrite(q.labels.data(), 4, path_length, f);
        fwrite(q.degrees.data(), 4, path_length, f);
        fwrite(q.pde.data(), 8, pde_dim, f);
        fwrite(q.pde_label.data(), 8, pde_dim, f);
    }
    for (ui i = 0; i < nq; i++) {
        const uint32_t c = (uint32_t)candidate_set[i].size();
        fwrite(&c, 4, 1, f);
        for (ui v : candidate_set[i]) fwrite(&v, 4, 1, f);
    }
    fclose(f);
    ui answer_num = 0;
    refinement(data, query, candidate_set, answer_num);
    cout << "Answer Number: " << answer_num << endl;
    return 0;
}
