// host_capi.cpp -- C-ABI wrappers around the host loader (R0 / R1) so that non-C++ callers (the Python
// multi-GPU driver) use the SAME loader code as gnnpe_main.  Compiled into libgnnpe_hip.so.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gnnpe_hip.h"
#include "graph_loader.h"
#include "query_plan.h"

namespace gnnpe {
void set_error(const char *fmt, ...);
}

extern "C" {

static int load_graph_file(const char *path, uint32_t *n, uint32_t *m, uint32_t **offsets, uint32_t **nbrs, uint32_t **labels,
                           uint32_t meta[3], uint32_t **simple_offsets, uint32_t **simple_nbrs);

int gnnpe_host_load_graph(const char *path, uint32_t *n, uint32_t *m, uint32_t **offsets, uint32_t **nbrs, uint32_t **labels,
                          uint32_t meta[3])
{
    return load_graph_file(path, n, m, offsets, nbrs, labels, meta, nullptr, nullptr);
}

int gnnpe_host_load_multigraph(const char *path, uint32_t *n, uint32_t *m, uint32_t **offsets, uint32_t **nbrs, uint32_t **labels,
                               uint32_t meta[3], uint32_t **simple_offsets, uint32_t **simple_nbrs)
{
    if (!simple_offsets || !simple_nbrs) {
        gnnpe::set_error("gnnpe_host_load_multigraph: null argument");
        return GNNPE_ERR_ARG;
    }
    return load_graph_file(path, n, m, offsets, nbrs, labels, meta, simple_offsets, simple_nbrs);
}

// simple_offsets == nullptr: strict (a non-simple file is malformed)
static int load_graph_file(const char *path, uint32_t *n, uint32_t *m, uint32_t **offsets, uint32_t **nbrs, uint32_t **labels,
                           uint32_t meta[3], uint32_t **simple_offsets, uint32_t **simple_nbrs)
{
    if (!path || !n || !m || !offsets || !nbrs || !labels) {
        gnnpe::set_error("gnnpe_host_load_graph: null argument");
        return GNNPE_ERR_ARG;
    }
    gnnpe_host::StaticGraph g;
    std::string err;
    const int rc = g.load(path, &err, simple_offsets == nullptr);
    if (rc != 0) {
        gnnpe::set_error("%s", err.c_str());
        return rc;
    }
    auto dup = [](const std::vector<uint32_t> &v) {
        uint32_t *p = (uint32_t *)malloc((v.size() + 1) * sizeof(uint32_t));
        if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(uint32_t));
        return p;
    };
    *n = g.n;
    *m = g.m;
    *offsets = dup(g.offsets);
    *nbrs = dup(g.neighbors);
    *labels = dup(g.labels);
    if (simple_offsets) {
        *simple_offsets = g.simple ? nullptr : dup(g.simple_offsets);
        *simple_nbrs = g.simple ? nullptr : dup(g.simple_neighbors);
    }
    if (meta) {
        meta[0] = g.labels_count;
        meta[1] = g.max_degree;
        meta[2] = g.max_label_frequency;
    }
    return 0;
}

int gnnpe_host_read_membership(const char *path, uint32_t n, uint32_t p, uint32_t *sorted_nodes, uint32_t *membership)
{
    std::vector<uint32_t> sn, mem;
    std::string err;
    const int rc = gnnpe_host::read_membership(path ? path : "", n, p, &sn, &mem, &err);
    if (rc != 0) {
        gnnpe::set_error("%s", err.c_str());
        return rc;
    }
    if (n) {
        memcpy(sorted_nodes, sn.data(), (size_t)n * 4);
        memcpy(membership, mem.data(), (size_t)n * 4);
    }
    return 0;
}

int gnnpe_host_query_plan(const char *query_graph_path, uint32_t e, uint32_t *n_query_vertices, uint32_t *n_paths,
                          uint32_t **vids, uint32_t **labels, uint32_t **degrees, double **pde)
{
    if (!query_graph_path || !n_query_vertices || !n_paths || !vids || !labels || !degrees || !pde) {
        gnnpe::set_error("gnnpe_host_query_plan: null argument");
        return GNNPE_ERR_ARG;
    }
    gnnpe_host::StaticGraph q;
    std::string err;
    int rc = q.load(query_graph_path, &err, true);
    if (rc != 0) {
        gnnpe::set_error("%s", err.c_str());
        return rc;
    }
    gnnpe_host::QueryPlan plan;
    if ((rc = gnnpe_host::build_query_plan(q, e, &plan, &err)) != 0) {
        gnnpe::set_error("%s", err.c_str());
        return GNNPE_ERR_ARG;
    }
    auto dup32 = [](const std::vector<uint32_t> &v) {
        uint32_t *p = (uint32_t *)malloc((v.size() + 1) * sizeof(uint32_t));
        if (p && !v.empty()) memcpy(p, v.data(), v.size() * sizeof(uint32_t));
        return p;
    };
    *n_query_vertices = plan.n_vertices;
    *n_paths = plan.n_paths();
    *vids = dup32(plan.vids);
    *labels = dup32(plan.labels);
    *degrees = dup32(plan.degrees);
    *pde = (double *)malloc((plan.pde.size() + 1) * sizeof(double));
    if (*pde && !plan.pde.empty()) memcpy(*pde, plan.pde.data(), plan.pde.size() * sizeof(double));
    return 0;
}

void gnnpe_host_free(void *ptr) { free(ptr); }

// ---- the online side's data load from binary sidecars ---------------------------------------------------
// `gnnpe_main --sidecars` leaves <f>gnn-pe/paths.bin (the rows of all_paths.txt as uint32 tuples) and <f>gnn-pe/vde.bin
// (x, nx, vde of every vertex).  The reference's online start re-parses the text (gen_pde, custom.h:546-572: ~95 s per
// 2e7 paths); this returns the same per-path vectors -- vids, labels, degrees, pde, pde_label -- as flat arrays.
static const char kPathsMagic[8] = {'G', 'N', 'N', 'P', 'E', 'P', 'T', 'H'};

int gnnpe_host_write_paths_header(void *file, uint32_t L, uint64_t n_paths)
{
    FILE *f = (FILE *)file;
    const uint32_t version = 1;
    if (!f || fwrite(kPathsMagic, 1, 8, f) != 8 || fwrite(&version, 4, 1, f) != 1 || fwrite(&L, 4, 1, f) != 1 ||
        fwrite(&n_paths, 8, 1, f) != 1) {
        gnnpe::set_error("paths.bin: cannot write the header");
        return GNNPE_ERR_IO;
    }
    return 0;
}

// gen_pde's arrays for all paths of paths.bin (sel == nullptr) or for the rows sel[0..n_sel) of it, in that order
static int load_path_rows(const char *paths_bin, const char *vde_bin, uint32_t n, const uint32_t *labels,
                          const uint32_t *degrees, const uint32_t *sel, uint64_t n_sel, uint64_t *n_paths, uint32_t *L_out,
                          uint32_t *e_out, uint32_t **vids, uint32_t **plabels, uint32_t **pdegrees, double **pde,
                          double **pde_label)
{
    if (!paths_bin || !vde_bin || !labels || !degrees || !n_paths || !L_out || !e_out || !vids || !plabels || !pdegrees || !pde ||
        !pde_label) {
        gnnpe::set_error("gnnpe_host_load_path_sidecar: null argument");
        return GNNPE_ERR_ARG;
    }
    *vids = *plabels = *pdegrees = nullptr;
    *pde = *pde_label = nullptr;
    // vde.bin: uint32 n, e; x[n*e], nx[n*e], vde[n*e]
    FILE *fv = fopen(vde_bin, "rb");
    if (!fv) {
        gnnpe::set_error("cannot open %s", vde_bin);
        return GNNPE_ERR_IO;
    }
    uint32_t hv[2] = {0, 0};
    if (fread(hv, 4, 2, fv) != 2 || hv[0] != n || hv[1] == 0 || hv[1] > 32) {
        fclose(fv);
        gnnpe::set_error("%s: header (n=%u, e=%u) does not match a graph of %u vertices", vde_bin, hv[0], hv[1], n);
        return GNNPE_ERR_ARG;
    }
    const uint32_t e = hv[1];
    std::vector<double> x((size_t)n * e), vde((size_t)n * e);
    bool ok = fread(x.data(), 8, x.size(), fv) == x.size() && fseek(fv, (long)((size_t)n * e * 8), SEEK_CUR) == 0 &&
              fread(vde.data(), 8, vde.size(), fv) == vde.size();
    fclose(fv);
    if (!ok) {
        gnnpe::set_error("%s: truncated", vde_bin);
        return GNNPE_ERR_IO;
    }
    // paths.bin: magic, version, L, P, then P x L uint32
    const int fd = open(paths_bin, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) {
        if (fd >= 0) close(fd);
        gnnpe::set_error("cannot open %s", paths_bin);
        return GNNPE_ERR_IO;
    }
    char *map = st.st_size ? (char *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : (char *)MAP_FAILED;
    close(fd);
    if (map == MAP_FAILED || st.st_size < 24 || memcmp(map, kPathsMagic, 8) != 0) {
        if (map != MAP_FAILED) munmap(map, (size_t)st.st_size);
        gnnpe::set_error("%s: not a paths.bin sidecar", paths_bin);
        return GNNPE_ERR_ARG;
    }
    uint32_t version, L;
    uint64_t P_file;
    memcpy(&version, map + 8, 4);
    memcpy(&L, map + 12, 4);
    memcpy(&P_file, map + 16, 8);
    // bound the count by the file size BEFORE multiplying: a crafted P_file (2^62, L = 4) would wrap the product back to
    // the 24 header bytes, pass the check and size the buffers below from the wrapped value
    const uint64_t body = (uint64_t)st.st_size - 24;
    if (version != 1 || L < 1 || L > 16 || P_file > body / (4ull * L) || P_file * L * 4 != body) {
        munmap(map, (size_t)st.st_size);
        gnnpe::set_error("%s: bad header (version %u, L %u, %llu paths, %lld bytes)", paths_bin, version, L,
                         (unsigned long long)P_file, (long long)st.st_size);
        return GNNPE_ERR_ARG;
    }
    const uint64_t P = sel ? n_sel : P_file;
    for (uint64_t i = 0; sel && i < n_sel; i++)
        if (sel[i] >= P_file) {
            munmap(map, (size_t)st.st_size);
            gnnpe::set_error("path id %u of the partition >= the %llu paths of %s", sel[i], (unsigned long long)P_file, paths_bin);
            return GNNPE_ERR_ARG;
        }
    const uint32_t *src = reinterpret_cast<const uint32_t *>(map + 24);
    const size_t D = (size_t)L * e;
    if (e < 1 || e > 4096 || P > (SIZE_MAX / 16 - 1) / D) {  // (P * D + 1) * 8 must not wrap either
        munmap(map, (size_t)st.st_size);
        gnnpe::set_error("%s: %llu paths x %zu doubles do not fit the address space", paths_bin, (unsigned long long)P, D);
        return GNNPE_ERR_ARG;
    }
    uint32_t *o_v = (uint32_t *)malloc((P * L + 1) * 4), *o_l = (uint32_t *)malloc((P * L + 1) * 4), *o_d = (uint32_t *)malloc((P * L + 1) * 4);
    double *o_p = (double *)malloc((P * D + 1) * 8), *o_x = (double *)malloc((P * D + 1) * 8);
    if (!o_v || !o_l || !o_d || !o_p || !o_x) {
        free(o_v); free(o_l); free(o_d); free(o_p); free(o_x);
        munmap(map, (size_t)st.st_size);
        gnnpe::set_error("gnnpe_host_load_path_sidecar: out of memory for %llu paths", (unsigned long long)P);
        return GNNPE_ERR_ARG;
    }
    // the loop of gen_pde (custom.h:553-568), split over the host's cores
    const unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    std::vector<uint32_t> bad(nt, 0xFFFFFFFFu);
    auto work = [&](unsigned t) {
        const uint64_t a = P * t / nt, b = P * (t + 1) / nt;
        for (uint64_t i = a; i < b; i++)
            for (uint32_t j = 0; j < L; j++) {
                const uint32_t v = src[(sel ? (uint64_t)sel[i] : i) * L + j];
                if (v >= n) {
                    bad[t] = v;
                    return;
                }
                o_v[i * L + j] = v;
                o_l[i * L + j] = labels[v];
                o_d[i * L + j] = degrees[v];
                for (uint32_t k = 0; k < e; k++) {
                    o_p[i * D + (size_t)j * e + k] = vde[(size_t)v * e + k];
                    o_x[i * D + (size_t)j * e + k] = x[(size_t)v * e + k];
                }
            }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &t : th) t.join();
    munmap(map, (size_t)st.st_size);
    for (unsigned t = 0; t < nt; t++)
        if (bad[t] != 0xFFFFFFFFu) {
            free(o_v); free(o_l); free(o_d); free(o_p); free(o_x);
            gnnpe::set_error("%s: vertex id %u >= n = %u", paths_bin, bad[t], n);
            return GNNPE_ERR_ARG;
        }
    *n_paths = P;
    *L_out = L;
    *e_out = e;
    *vids = o_v;
    *plabels = o_l;
    *pdegrees = o_d;
    *pde = o_p;
    *pde_label = o_x;
    return 0;
}

int gnnpe_host_load_path_sidecar(const char *paths_bin, const char *vde_bin, uint32_t n, const uint32_t *labels,
                                 const uint32_t *degrees, uint64_t *n_paths, uint32_t *L_out, uint32_t *e_out, uint32_t **vids,
                                 uint32_t **plabels, uint32_t **pdegrees, double **pde, double **pde_label)
{
    return load_path_rows(paths_bin, vde_bin, n, labels, degrees, nullptr, 0, n_paths, L_out, e_out, vids, plabels, pdegrees, pde,
                          pde_label);
}

// partition_paths.txt (main.cpp:98-108): "<count>\n", then one global path id per line
static int read_partition_ids(const char *path, std::vector<uint32_t> &ids)
{
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) {
        if (fd >= 0) close(fd);
        gnnpe::set_error("cannot open %s", path);
        return GNNPE_ERR_IO;
    }
    const char *map = st.st_size ? (const char *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : (const char *)MAP_FAILED;
    close(fd);
    if (map == MAP_FAILED) {
        gnnpe::set_error("%s: empty or unreadable", path);
        return GNNPE_ERR_IO;
    }
    const char *q = map, *end = map + st.st_size;
    auto next = [&](uint64_t &v) {
        while (q < end && (*q == ' ' || *q == '\n' || *q == '\r' || *q == '\t')) q++;
        if (q >= end || *q < '0' || *q > '9') return false;
        v = 0;
        while (q < end && *q >= '0' && *q <= '9') v = v * 10 + (uint64_t)(*q++ - '0');
        return true;
    };
    uint64_t cnt = 0, v = 0;
    bool ok = next(cnt) && cnt <= (uint64_t)st.st_size;  // every id takes at least a digit and a newline
    if (ok) {
        ids.resize(cnt);
        for (uint64_t i = 0; i < cnt && ok; i++) {
            ok = next(v) && v <= 0xFFFFFFFFull;
            ids[i] = (uint32_t)v;
        }
    }
    munmap((void *)map, (size_t)st.st_size);
    if (!ok) {
        gnnpe::set_error("%s: not a partition_paths.txt (count, then that many path ids)", path);
        return GNNPE_ERR_ARG;
    }
    return 0;
}

int gnnpe_host_load_partition_sidecar(const char *paths_bin, const char *vde_bin, const char *partition_paths_txt, uint32_t n,
                                      const uint32_t *labels, const uint32_t *degrees, uint64_t *n_paths, uint32_t *L_out,
                                      uint32_t *e_out, uint32_t **path_ids, uint32_t **vids, uint32_t **plabels,
                                      uint32_t **pdegrees, double **pde, double **pde_label)
{
    if (!partition_paths_txt || !path_ids) {
        gnnpe::set_error("gnnpe_host_load_partition_sidecar: null argument");
        return GNNPE_ERR_ARG;
    }
    *path_ids = nullptr;
    std::vector<uint32_t> ids;
    int rc = read_partition_ids(partition_paths_txt, ids);
    if (rc) return rc;
    static const uint32_t none = 0;
    rc = load_path_rows(paths_bin, vde_bin, n, labels, degrees, ids.empty() ? &none : ids.data(), ids.size(), n_paths, L_out, e_out,
                        vids, plabels, pdegrees, pde, pde_label);
    if (rc) return rc;
    uint32_t *o = (uint32_t *)malloc((ids.size() + 1) * 4);
    if (!o) {
        free(*vids); free(*plabels); free(*pdegrees); free(*pde); free(*pde_label);
        *vids = *plabels = *pdegrees = nullptr;
        *pde = *pde_label = nullptr;
        gnnpe::set_error("gnnpe_host_load_partition_sidecar: out of memory");
        return GNNPE_ERR_ARG;
    }
    if (!ids.empty()) memcpy(o, ids.data(), ids.size() * 4);
    *path_ids = o;
    return 0;
}

// aux_index.bin (written by gnnpe_build_aux_index): what Partition::build_auxiliary_index (custom.h:268-364) computes
int gnnpe_host_load_aux_index(const char *path, uint32_t *n_nodes, uint32_t *L_out, uint32_t *D_out, double **key,
                              uint32_t **degrees, double **label_mbr)
{
    if (!path || !n_nodes || !L_out || !D_out || !key || !degrees || !label_mbr) {
        gnnpe::set_error("gnnpe_host_load_aux_index: null argument");
        return GNNPE_ERR_ARG;
    }
    *key = *label_mbr = nullptr;
    *degrees = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) {
        gnnpe::set_error("cannot open %s", path);
        return GNNPE_ERR_IO;
    }
    char magic[8];
    uint32_t h[4] = {0, 0, 0, 0};
    uint64_t N = 0;
    struct stat st;
    const bool hdr_ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "GNNPEAUX", 8) == 0 && fread(h, 4, 4, f) == 4 &&
                        fread(&N, 8, 1, f) == 1 && fstat(fileno(f), &st) == 0;
    const uint64_t L = h[1], D = h[2];
    // N <= 2^31, L <= 16, D <= 512 keep N * (8 + 4L + 16D) far below 2^64; the count is still bounded by the file first
    const uint64_t node_bytes = 8 + 4 * L + 16 * D;
    if (!hdr_ok || h[0] != 1 || L < 1 || L > 16 || D < 1 || D > 512 || N > 0x7FFFFFFFull || st.st_size < 32 ||
        N > ((uint64_t)st.st_size - 32) / node_bytes || (uint64_t)st.st_size != 32 + N * node_bytes) {
        fclose(f);
        gnnpe::set_error("%s: not an aux_index.bin (version %u, L %u, D %u, %llu nodes)", path, h[0], h[1], h[2], (unsigned long long)N);
        return GNNPE_ERR_ARG;
    }
    double *k = (double *)malloc((N + 1) * 8), *m = (double *)malloc((N * 2 * D + 1) * 8);
    uint32_t *d = (uint32_t *)malloc((N * L + 1) * 4);
    const bool ok = k && m && d && fread(k, 8, N, f) == N && fread(d, 4, N * L, f) == N * L && fread(m, 8, N * 2 * D, f) == N * 2 * D;
    fclose(f);
    if (!ok) {
        free(k); free(m); free(d);
        gnnpe::set_error("%s: truncated or out of memory", path);
        return GNNPE_ERR_IO;
    }
    *n_nodes = (uint32_t)N;
    *L_out = (uint32_t)L;
    *D_out = (uint32_t)D;
    *key = k;
    *degrees = d;
    *label_mbr = m;
    return 0;
}

}  // extern "C"
