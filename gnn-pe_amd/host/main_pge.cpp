// main_pge.cpp -- `gnnpge_main`: drop-in for the GROUPED variant's `main -m offline`
// (GNN-PGE/src/main.cpp:38-245), the first "next" row of SURVEY 8(f).
//
// Same flags and defaults as the reference; reads <f>gnn-pge/membership.txt, writes
//   <f>gnn-pge/data_vertices.bin                         (main.cpp:179-194)
//   <f>gnn-pge/partitions/partition-i/index.dat          (Partition ctor, custom.h:141-195; the reference
//                                                          builds these at the end of its offline run too)
// Embeddings, path groups and the R-tree images come from the GPU through include/gnnpe_hip.h.
// The `key` double of every record is uninitialised memory in the reference for data vertices
// (SURVEY 8(f)); it is written as 0 here and never read by the reference's online code for them.
#include <string>
#include <vector>

#include "cli_common.h"
#include "graph_loader.h"

using namespace cli;

int main(int argc, char **argv)
{
    Options o = parse_args(argc, argv, "gnnpge_main");
    const auto t0 = Clock::now();
    // GNN-PGE/include/custom.h:47-49: path_length = 1 + 1 (vertices per path), pde_dim = vde_dim * path_length
    if (o.path_length != 2) die("-l " + std::to_string(o.path_length) + ": only the reference default (2) is supported");
    if (o.partition_num == 0) die("-p must be >= 1");
    if (o.mode == "online") die("-m online is the reference's own binary: run it on the files this tool wrote", 2);
    if (o.mode != "offline") return 0;

    gnnpe_host::StaticGraph g;
    std::string err;
    int rc = g.load(o.data_graph, &err, true);  // GNN-PGE: simple graphs only
    if (rc == -1) {  // graph.cpp:166-169
        printf("%s\n", err.c_str());
        exit(-1);
    }
    if (rc != 0) die(o.data_graph + ": " + err);
    std::vector<uint32_t> sorted_nodes, membership;
    if (gnnpe_host::read_membership(o.dataset_path + "gnn-pge/membership.txt", g.n, o.partition_num, &sorted_nodes, &membership,
                                    &err) != 0)
        die(err);
    const std::string partitions_path = o.dataset_path + "gnn-pge/partitions/";
    for (uint32_t i = 0; i < o.partition_num; i++)
        if (!is_dir(partitions_path + "partition-" + std::to_string(i)))
            die("missing directory " + partitions_path + "partition-" + std::to_string(i) + "/ (the prep step creates it)");
    // partition_vertices[membership[node]] in membership.txt order (main.cpp:86-89)
    std::vector<std::vector<uint32_t>> part(o.partition_num);
    for (uint32_t node : sorted_nodes) part[membership[node]].push_back(node);

    if (gnnpe_device_count() <= 0) die("no HIP device: this tool has no CPU fallback");
    gnnpe_ctx *ctx = gnnpe_create(0);
    if (!ctx) die(std::string("gnnpe_create: ") + gnnpe_last_error());
    const uint32_t e = o.vde_dim, nl = std::max<uint32_t>(g.labels_count, 1);
    std::vector<double> table((size_t)nl * e);
    check(gnnpe_host_label_table(nl, e, table.data()), "label table");
    check(gnnpe_load_csr(ctx, g.n, g.offsets.data(), g.neighbors.data(), g.labels.data()), "load_csr");
    check(gnnpe_set_label_table(ctx, nl, e, table.data()), "set_label_table");
    const size_t ne = (size_t)g.n * e;
    std::vector<double> x(ne), nx(ne), vde(ne), pg(ne * 4), plg(ne * 4);
    check(gnnpe_vde(ctx, x.data(), nx.data(), vde.data()), "vde");            // main.cpp:93
    check(gnnpe_pge_groups(ctx, pg.data(), plg.data()), "pge_groups");        // main.cpp:97-177
    const auto t1 = Clock::now();

    {  // main.cpp:179-194
        const std::string path = o.dataset_path + "gnn-pge/data_vertices.bin";
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) die("cannot open " + path + " for writing");
        const uint32_t D2 = 4 * e;
        std::vector<char> rec(12 + 8 + 8 * (3 * (size_t)e + 2 * D2));
        fwrite(&g.n, 4, 1, f);
        for (uint32_t v = 0; v < g.n; v++) {
            char *p = rec.data();
            const uint32_t deg = g.degree(v);
            const double key = 0.0;
            memcpy(p, &v, 4);
            memcpy(p + 4, &g.labels[v], 4);
            memcpy(p + 8, &deg, 4);
            memcpy(p + 12, &key, 8);
            p += 20;
            memcpy(p, &x[(size_t)v * e], 8 * e);
            memcpy(p + 8 * e, &nx[(size_t)v * e], 8 * e);
            memcpy(p + 16 * e, &vde[(size_t)v * e], 8 * e);
            p += 24 * e;
            memcpy(p, &pg[(size_t)v * D2], 8 * D2);
            memcpy(p + 8 * D2, &plg[(size_t)v * D2], 8 * D2);
            if (fwrite(rec.data(), 1, rec.size(), f) != rec.size()) die("write error on " + path);
        }
        if (fclose(f) != 0) die("write error on " + path);
    }
    const auto t2 = Clock::now();
    for (uint32_t i = 0; i < o.partition_num; i++)
        check(gnnpe_pge_build_index(ctx, part[i].size(), part[i].data(),
                                    (partitions_path + "partition-" + std::to_string(i) + "/index.dat").c_str()),
              "pge_build_index");
    const auto t3 = Clock::now();
    gnnpe_destroy(ctx);
    if (o.timing)
        fprintf(stderr, "{\"vertices\": %u, \"load_embed_group_s\": %.3f, \"write_bin_s\": %.3f, \"index_s\": %.3f, \"end_to_end_s\": %.3f}\n",
                g.n, secs(t0, t1), secs(t1, t2), secs(t2, t3), secs(t0, t3));
    return 0;
}
