/*
 * gnnpe_oracle.h -- CPU restatement of the GNN-PE offline path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for the MI355X engine in gnn-pe_amd/.  It restates, in plain C,
 * the algorithm of the reference (JamesWhiteSnow/GNN-PE) for the hot path named in
 * BASELINE.json / SURVEY.md section 8.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product never links or calls it.
 *
 * Pinning: every function below is checked against the *compiled reference itself*
 * (oracle/_ref/ref_main, oracle/_ref/ref_dump; see oracle/Makefile) and against the golden
 * fixtures in tests/golden/ that were generated from those binaries
 * (tests/golden/make_golden.py).  Citations are relative to /root/reference/.
 */
#ifndef GNNPE_ORACLE_H
#define GNNPE_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- R0: graph loader (GNN-PE/libsrc/graph/graph.cpp:163-242) ------------------------- */
/* Returns 0 on success, -1 if the file cannot be opened (reference: exit(-1), graph.cpp:166-169).
 * offsets/neighbors/labels are malloc'ed; free with orc_free(). */
int orc_load_graph(const char *path, uint32_t *n_out, uint32_t *m_out,
                   uint32_t **offsets, uint32_t **neighbors, uint32_t **labels,
                   uint32_t *labels_count, uint32_t *max_degree, uint32_t *max_label_freq);
void orc_free(void *p);

/* ---- R1: membership reader (GNN-PE/src/main.cpp:77-85) -------------------------------- */
int orc_read_membership(const char *path, uint32_t n, uint32_t *sorted_nodes, uint32_t *membership);

/* ---- R2: path enumeration (GNN-PE/include/custom.h:52-92, main.cpp:87-96) -------------- */
/* Reference-faithful recursive DFS with a hash set keyed on the vertex tuple; keeps a path iff
 * neither it nor its reverse is already in the set.  L = number of vertices per path
 * (reference: always 3).  Two-phase: call with paths == NULL to get the count.
 * paths: P x L uint32 (row major) in emission order.  start_of: optional P uint32 giving the
 * start vertex' processing index.  Returns P. */
uint64_t orc_enumerate_dfs_hash(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors,
                                const uint32_t *sorted_nodes, uint32_t L,
                                uint32_t *paths, uint64_t capacity);

/* Closed form: for s in processing order, DFS over ascending neighbours,
 * simple paths only, keep iff rank[last] > rank[first].  Valid for simple graphs. */
uint64_t orc_enumerate_closed(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors,
                              const uint32_t *sorted_nodes, uint32_t L,
                              uint32_t *paths, uint64_t capacity);

/* per start vertex (indexed by processing position i) path counts, closed form */
void orc_count_per_start(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors,
                         const uint32_t *sorted_nodes, uint32_t L, uint64_t *counts);

/* l = 3 at config-5 size (parity unpinned): the per-start counts of orc_count_per_start(L = 4) without walking the
 * paths (sorted-row merges, OpenMP), and the closed-form DFS for the start vertices at positions [first, first + count) only,
 * start k's rows at paths + start_off[k] * L (start_off: count + 1 entries; returns the rows written, or >= 2^62 if a start's
 * rows are not the number it was given room for).  0 / -1 (out of memory). */
int orc_count_per_start_l3(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors, const uint32_t *sorted_nodes,
                           uint64_t *counts);
uint64_t orc_enumerate_starts(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors, const uint32_t *sorted_nodes,
                              uint32_t L, uint32_t first, uint32_t count, const uint64_t *start_off, uint32_t *paths);

/* ---- online filter, leaf test of Partition::query (custom.h:404-431) over every data path ---- */
void orc_filter_candidates(uint64_t P, uint32_t L, const uint32_t *paths, uint32_t n, const uint32_t *offsets,
                           const uint32_t *labels, const double *vde, uint32_t e, uint32_t n_qp,
                           const uint32_t *q_vids, const uint32_t *q_labels, const uint32_t *q_degrees,
                           const double *q_pde, double epsilon, uint32_t *bitmap);

/* All-core CPU port of the device-resident pass (vde + count + prefix + ids/pde fill, l=2, closed form, OpenMP):
 * bench.py's second CPU baseline.  vde: n x e; start_off: n+1; ids: P x 3, pde: P x 3e (may be NULL); returns P and
 * fills ids/pde only when P <= capacity.  threads <= 0: all cores. */
uint64_t orc_offline_parallel(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors,
                              const uint32_t *labels, const uint32_t *sorted_nodes, uint32_t e, int threads,
                              double *vde, uint64_t *start_off, uint32_t *ids, double *pde, uint64_t capacity);
int orc_max_threads(void);
/* closed-form count of 4-vertex simple paths (l = 3 rule, each path once): sum_E (du-1)(dv-1) - 3 T */
int orc_count_p4(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors, uint64_t *triangles_out, uint64_t *p4_out);

/* ---- R3: label feature (custom.h:492-511), libstdc++ mt19937 + generate_canonical ------- */
void orc_gen_vde_x(uint32_t label, uint32_t e, double *x_out);

/* ---- R4: vertex embedding (custom.h:513-544) ------------------------------------------ */
/* x, nx, vde: n x e doubles each (row major). */
void orc_gen_vde(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors,
                 const uint32_t *labels, uint32_t e, double *x, double *nx, double *vde);

/* ---- R5: path embedding (custom.h:546-572) -------------------------------------------- */
/* pde, pde_label: P x (e*L) doubles; plabels, pdegrees: P x L uint32 (any may be NULL). */
void orc_gen_pde(uint64_t P, uint32_t L, const uint32_t *paths, uint32_t e,
                 const uint32_t *offsets, const uint32_t *labels,
                 const double *x, const double *vde,
                 double *pde, double *pde_label, uint32_t *plabels, uint32_t *pdegrees);

/* ---- R7: writers (main.cpp:98-119) ----------------------------------------------------- */
int orc_write_all_paths(const char *path, uint64_t P, uint32_t L, const uint32_t *paths);
/* partition_paths.txt for partition pid: ids of paths whose start vertex is in pid, ascending */
int orc_write_partition_paths(const char *path, uint64_t P, uint32_t L, const uint32_t *paths,
                              const uint32_t *membership, uint32_t pid);
/* in-memory variants: returns byte length; buf may be NULL to size */
uint64_t orc_format_all_paths(uint64_t P, uint32_t L, const uint32_t *paths, char *buf);

/* ---- R6: index.dat reader / structural validator --------------------------------------- */
/* (rtree.cpp:318-362, rtnode.cpp:789-807,1099-1117, entry.cpp:127-136, blk_file.cpp:22-63,108-168)
 * Decodes the file image and checks every consumer constraint of SURVEY 8(a) R6.
 * On success returns 0 and fills hdr[8] = {blocklength, n_blocks, dim, num_data, dnodes,
 * inodes, root_is_data, root}; leaf_son (num_data int32) and leaf_pt (num_data x dim doubles,
 * the lo bounds) receive the leaf entries in DFS order when non-NULL.
 * Negative return = which constraint failed (see gnnpe_oracle.c). */
int orc_index_validate(const uint8_t *img, uint64_t nbytes, int32_t hdr[8],
                       int32_t *leaf_son, double *leaf_pt, uint64_t leaf_capacity,
                       int32_t *height_out);

/* ---- GNN-PGE offline ("next" row): GNN-PGE/src/main.cpp:91-195 ----------------------- */
/* Per vertex: 1-hop paths (v, nbr) (dfs to depth path_length = 2, GNN-PGE/include/custom.h:52-71); the
 * path embedding is [vde[v], vde[nbr]] (D = 2e dims); path_group = per-dimension [min, max] over the
 * vertex' paths, path_label_group the same over [x[v], x[nbr]].  Isolated vertices get
 * [vde, vde] / [x, x] in the first e dims and zeros after (main.cpp:104-121).
 * path_group, path_label_group: n x 2D doubles, laid out (lo0, hi0, lo1, hi1, ...). */
void orc_pge_groups(uint32_t n, const uint32_t *offsets, const uint32_t *neighbors, uint32_t e,
                    const double *x, const double *vde, double *path_group, double *path_label_group);
/* data_vertices.bin (main.cpp:179-194): u32 count; per vertex u32 vid, label, degree; double key
 * (uninitialised in the reference for data vertices -- written as `key_fill`); x, nx, vde (e doubles
 * each); path_group, path_label_group (2D doubles each). */
int orc_pge_write_bin(const char *path, uint32_t n, uint32_t e, const uint32_t *offsets, const uint32_t *labels,
                      const double *x, const double *nx, const double *vde, const double *path_group,
                      const double *path_label_group, double key_fill);

#ifdef __cplusplus
}
#endif
#endif
