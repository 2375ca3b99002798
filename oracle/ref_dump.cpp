// ref_dump.cpp -- harness around the UNMODIFIED reference headers (TEST INFRASTRUCTURE ONLY).
//
// Compiled by oracle/Makefile against the sources where they lie in /root/reference/GNN-PE
// (nothing is copied).  It calls the reference's own gen_vde / gen_pde / gen_vde_x
// (GNN-PE/include/custom.h:492-572) and dumps their outputs as raw little-endian arrays so the
// restatement in gnnpe_oracle.c and the golden fixtures can be pinned to the reference itself.
//
// usage: ref_dump <graph> <e> <out_vde.bin> [<all_paths.txt> <out_pde.bin>]
//   out_vde.bin : uint32 n, uint32 e, then x[n*e], nx[n*e], vde[n*e] doubles, then uint32 label[n], degree[n]
//   out_pde.bin : uint64 P, uint32 L, uint32 e, then per path: vids[L] labels[L] degrees[L] (uint32),
//                 pde[e*L], pde_label[e*L] (double)
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

#include <cstdio>
#include <cstdint>

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: ref_dump <graph> <e> <out_vde.bin> [<all_paths.txt> <out_pde.bin>]\n");
        return 2;
    }
    vde_dim = (ui)atoi(argv[2]);
    path_length = 3; // main.cpp:58 (path_length += 1 with the only working -l 2)
    pde_dim = vde_dim * path_length;

    Static_Graph *g = new Static_Graph(true);
    g->loadGraphFromFile(argv[1]);
    vector<Vertex> V = gen_vde(g);

    FILE *f = fopen(argv[3], "wb");
    uint32_t n = g->getVerticesCount(), e = vde_dim;
    fwrite(&n, 4, 1, f);
    fwrite(&e, 4, 1, f);
    for (uint32_t i = 0; i < n; i++) fwrite(V[i].x.data(), 8, e, f);
    for (uint32_t i = 0; i < n; i++) fwrite(V[i].nx.data(), 8, e, f);
    for (uint32_t i = 0; i < n; i++) fwrite(V[i].vde.data(), 8, e, f);
    for (uint32_t i = 0; i < n; i++) fwrite(&V[i].label, 4, 1, f);
    for (uint32_t i = 0; i < n; i++) fwrite(&V[i].degree, 4, 1, f);
    fclose(f);

    if (argc >= 6) {
        vector<Path> P = gen_pde(V, argv[4]);
        f = fopen(argv[5], "wb");
        uint64_t np = P.size();
        uint32_t L = path_length;
        fwrite(&np, 8, 1, f);
        fwrite(&L, 4, 1, f);
        fwrite(&e, 4, 1, f);
        for (uint64_t i = 0; i < np; i++) {
            fwrite(P[i].vids.data(), 4, L, f);
            fwrite(P[i].labels.data(), 4, L, f);
            fwrite(P[i].degrees.data(), 4, L, f);
            fwrite(P[i].pde.data(), 8, e * L, f);
            fwrite(P[i].pde_label.data(), 8, e * L, f);
        }
        fclose(f);
    }
    return 0;
}
