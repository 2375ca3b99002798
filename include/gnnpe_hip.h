/*
 * gnnpe_hip.h -- C-ABI of the MI355X offline path-embedding engine (libgnnpe_hip.so).
 *
 * The reference (JamesWhiteSnow/GNN-PE) has no plugin / FFI seam: `custom.h` is a header of free
 * functions included by one translation unit.  This header is therefore the seam a
 * maintainer would add: every entry point names the reference function (file:line under
 * /root/reference/) whose work it takes over.  INTEGRATION.md shows the call sites in
 * GNN-PE/src/main.cpp.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - every function returns 0 on success, <0 on error; gnnpe_last_error() gives the message
 *     (thread local).  Unlike the reference, nothing calls exit() (functions.cpp:24-29).
 *   - "host" pointers are ordinary CPU memory owned by the caller; "dev" pointers are HIP device
 *     memory on the context's device (e.g. a torch tensor's data_ptr()).  The library owns every
 *     other device allocation behind the opaque context.
 *   - one context per GPU, driven from one host thread; work is enqueued on the context's stream
 *     (gnnpe_set_stream) and host-returning calls synchronise that stream.
 *   - ids are uint32 like the reference (`ui`, include/configuration/types.h:13-17); path counts and
 *     path ids are uint64 so inputs beyond the reference's 2^32 limit can be counted and streamed.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails.
 */
#ifndef GNNPE_HIP_H
#define GNNPE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GNNPE_ABI_VERSION 6

#define GNNPE_OK 0
#define GNNPE_ERR_ARG (-1)     /* bad argument / call order */
#define GNNPE_ERR_HIP (-2)     /* a HIP runtime call failed (message has the HIP error string) */
#define GNNPE_ERR_UNSUPPORTED (-3)
#define GNNPE_ERR_IO (-4)
#define GNNPE_ERR_RANGE (-5)   /* output does not fit the reference's 32-bit limits */

typedef struct gnnpe_ctx gnnpe_ctx;

int gnnpe_abi_version(void);
const char *gnnpe_last_error(void);

/* ---- context --------------------------------------------------------------------------------- */
/* Binds to HIP device `device_id`.  Returns NULL (and sets last_error) when no device is usable. */
gnnpe_ctx *gnnpe_create(int device_id);
void gnnpe_destroy(gnnpe_ctx *ctx);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL = the context's own. */
int gnnpe_set_stream(gnnpe_ctx *ctx, void *hip_stream);
int gnnpe_sync(gnnpe_ctx *ctx);

/* Device buffers for callers without a HIP runtime of their own (the C++ host CLI); Python callers
 * pass torch tensors instead.  gnnpe_copy_to_host synchronises the context's stream. */
int gnnpe_dev_alloc(gnnpe_ctx *ctx, uint64_t bytes, void **dev_ptr);
int gnnpe_dev_free(gnnpe_ctx *ctx, void *dev_ptr);
int gnnpe_copy_to_host(gnnpe_ctx *ctx, void *host_dst, const void *dev_src, uint64_t bytes);
int gnnpe_device_count(void);
/* The context's stream as a hipStream_t (for callers that enqueue their own work behind the engine's, e.g. the RCCL
 * send/recv of the C++ host's halo exchange), and an asynchronous device-to-device copy on it (either end may be memory
 * of another device of this process). */
int gnnpe_get_stream(gnnpe_ctx *ctx, void **hip_stream);
int gnnpe_copy_device(gnnpe_ctx *ctx, void *dev_dst, const void *dev_src, uint64_t bytes);
/* Page-locked host memory for buffers that receive gnnpe_copy_to_host / gnnpe_fill_paths output (PCIe-rate copies). */
int gnnpe_pinned_alloc(uint64_t bytes, void **host_ptr);
void gnnpe_pinned_free(void *host_ptr);

/* ---- inputs ---------------------------------------------------------------------------------- */
/* R0: the CSR that Static_Graph::loadGraphFromFile builds (graph.cpp:163-242; accessors
 * graph.h:154-176): offsets[n+1], neighbours sorted ascending per vertex, labels[n].  Host -> HBM. */
int gnnpe_load_csr(gnnpe_ctx *ctx, uint32_t n, const uint32_t *host_offsets, const uint32_t *host_nbrs,
                   const uint32_t *host_labels);

/* Vertex-partitioned variant (multi-GPU): this context holds the adjacency rows of
 * `rows[0..n_rows)` only (global vertex ids, any order; the rank's owned start vertices in
 * processing order), row k occupying row_nbrs[row_offsets[k] .. row_offsets[k+1]).  labels[n] is
 * replicated.  `nbr_capacity` reserves room (in neighbour entries) for halo rows appended later. */
int gnnpe_load_rows(gnnpe_ctx *ctx, uint32_t n, const uint32_t *host_labels, uint32_t n_rows,
                    const uint32_t *host_rows, const uint64_t *host_row_offsets, const uint32_t *host_row_nbrs,
                    uint64_t nbr_capacity);

/* Non-simple input: duplicate `e` lines.  Static_Graph::loadGraphFromFile stores what it reads (graph.cpp:211-218: no
 * de-duplication), so `degree` (graph.h:154-156) and the neighbour sum of gen_vde (custom.h:527-534) count the repeats, while
 * the hash set drops a path met again (custom.h:68-77).  The engine's enumeration takes simple rows, so such a graph is loaded
 * in two steps: gnnpe_load_csr / gnnpe_load_rows with every row WITHOUT its repeats (what the DFS + hash set amount to;
 * gnnpe_host_load_multigraph returns both forms), then this call with the same rows as the reference holds them (ascending,
 * repeats kept; row k of the loaded rows occupies row_nbrs[row_offsets[k] .. row_offsets[k+1])).  The library checks on the
 * device that the two forms belong together.  Afterwards gnnpe_vde sums over the rows given here, and with the whole graph
 * loaded the degree columns of the auxiliary index and the online filter are these rows' lengths (a slab-only context passes
 * them through gnnpe_set_degrees).  Reset by the next gnnpe_load_csr / gnnpe_load_rows.  GNN-PGE's groups and the refinement
 * refuse a context in this state.  Self entries are refused here as everywhere: for `e u u` the reference's loader writes one
 * slot twice and leaves the next one uninitialised (graph.cpp:211-218) -- it has no defined result to match. */
int gnnpe_set_multigraph_rows(gnnpe_ctx *ctx, uint32_t n_rows, const uint64_t *host_row_offsets, const uint32_t *host_row_nbrs);

/* R1: processing order and partition of every vertex, as main.cpp:77-85 reads them from
 * membership.txt (line i = "<sorted_nodes[i]> <membership[sorted_nodes[i]]>").  p = partition_num. */
int gnnpe_set_order(gnnpe_ctx *ctx, const uint32_t *host_sorted_nodes, const uint32_t *host_membership,
                    uint32_t p);

/* Slab of the processing order this context enumerates: start vertices sorted_nodes[begin..end).
 * Default (never called) = [0, n).  Multi-GPU: rank r owns one contiguous slab, so its paths are one
 * contiguous range of global path ids. */
int gnnpe_set_slab(gnnpe_ctx *ctx, uint32_t begin, uint32_t end);

/* R3: x_table[label][k] = gen_vde_x(label)[k] (custom.h:492-511), n_labels x e doubles.  The table
 * is host arithmetic (std::mt19937 re-seeded per label); host/label_table.cpp computes it. */
int gnnpe_set_label_table(gnnpe_ctx *ctx, uint32_t n_labels, uint32_t e, const double *host_x_table);

/* R3 itself: host arithmetic, exactly the reference's gen_vde_x (custom.h:492-511) for labels
 * 0..n_labels-1: std::mt19937(label), e draws of std::uniform_real_distribution<double>(0,1),
 * divided by their left-to-right sum.  out: n_labels x e doubles.  Needs no GPU. */
int gnnpe_host_label_table(uint32_t n_labels, uint32_t e, double *out);

/* R0 / R1 on the host, for callers that are not the C++ CLI (same code: host/graph_loader.cpp).
 * gnnpe_host_load_graph parses a `.graph` file like Static_Graph::loadGraphFromFile
 * (graph.cpp:163-242); the arrays are malloc'ed by the library, release them with gnnpe_host_free.
 * meta = {labels_count, max_degree, max_label_frequency} (printGraphMetaData, graph.cpp:244-247).
 * Returns 0, -1 (cannot open: the reference exits with -1) or -2 (malformed; gnnpe_last_error). */
int gnnpe_host_load_graph(const char *path, uint32_t *n, uint32_t *m, uint32_t **offsets, uint32_t **nbrs,
                          uint32_t **labels, uint32_t meta[3]);
/* The same loader without the refusal of duplicate `e` lines (gnnpe_host_load_graph answers -2 for them; both answer -2 for a
 * self-loop): offsets / nbrs are the rows as graph.cpp:211-233 leaves them; for a file with repeated lines simple_offsets /
 * simple_nbrs receive the de-duplicated rows (see gnnpe_set_multigraph_rows), else NULL. */
int gnnpe_host_load_multigraph(const char *path, uint32_t *n, uint32_t *m, uint32_t **offsets, uint32_t **nbrs,
                               uint32_t **labels, uint32_t meta[3], uint32_t **simple_offsets, uint32_t **simple_nbrs);
/* membership.txt (main.cpp:77-85): sorted_nodes[n], membership[n] caller-allocated; p = partition_num. */
int gnnpe_host_read_membership(const char *path, uint32_t n, uint32_t p, uint32_t *sorted_nodes, uint32_t *membership);
void gnnpe_host_free(void *ptr);

/* ---- halo exchange helpers (device side of the RCCL all-to-all-v) ----------------- */
/* List the vertices whose adjacency rows this context needs but does not hold -- every vertex referenced by
 * a row it holds: after gnnpe_rows_drop_halo these are the neighbours of its slab's start vertices (all that
 * l=2 needs); called again after gnnpe_rows_append they are the rows two hops out (l=3) -- grouped by owning rank: owner r holds the slab
 * [slab_bounds[r], slab_bounds[r+1]) of the processing order.  dev_ids receives the ids (capacity
 * cap entries), host_counts[r] the number per owner.  Ids are ascending inside each group. */
int gnnpe_halo_need(gnnpe_ctx *ctx, uint32_t n_ranks, const uint32_t *host_slab_bounds, void *dev_ids,
                    uint64_t cap, uint64_t *host_counts);
/* Degrees of requested rows (rows this context holds): dev_deg[k] = deg(dev_ids[k]). */
int gnnpe_rows_degree(gnnpe_ctx *ctx, uint64_t n_req, const void *dev_ids, void *dev_deg);
/* Pack the adjacency lists of the requested rows back to back into dev_out (sum of degrees entries). */
int gnnpe_rows_pack(gnnpe_ctx *ctx, uint64_t n_req, const void *dev_ids, void *dev_out, uint64_t cap);
/* Install received halo rows: ids, their degrees, and the packed adjacency (all device memory).  min_rank > 0 drops
 * the entries whose neighbour comes before processing position min_rank: a rank whose slab starts there never emits a
 * path that ends on one (kept iff rank[c] > rank[s]), so the LAST hop's halo rows can be truncated (l=2: the only hop;
 * l=3: pass 0 for the first hop).  Rows are validated (ids < n, ascending, no self-loop) and their reverse positions
 * and hub flags built here: the halo is graph structure and stays resident across steps. */
int gnnpe_rows_append(gnnpe_ctx *ctx, uint64_t n_rows, const void *dev_ids, const void *dev_deg,
                      const void *dev_nbrs, uint64_t n_nbrs, uint32_t min_rank);

/* What the context holds: adjacency rows (own + halo), their neighbour entries (halo rows count as installed, i.e.
 * after truncation), rows longer than 64 entries.  Any output may be NULL. */
int gnnpe_rows_held(gnnpe_ctx *ctx, uint64_t *n_rows, uint64_t *n_entries, uint32_t *n_hub_rows);

/* Forget every appended halo row (back to the state right after gnnpe_load_rows), so the exchange
 * can be repeated. */
int gnnpe_rows_drop_halo(gnnpe_ctx *ctx);

/* ---- R4: vertex embedding (gen_vde, custom.h:513-544) ------------------------------------------ */
/* x = table[label], nx = sum of neighbours' x in ascending-neighbour order from 0.0 (bit-identical
 * to the reference's loop :527-534), vde = x + nx.  Computed for the rows this context holds
 * adjacency for (all vertices after gnnpe_load_csr; the slab rows after gnnpe_load_rows).  Any of
 * the host outputs (n x e doubles each, indexed by vertex id) may be NULL. */
int gnnpe_vde(gnnpe_ctx *ctx, double *host_x, double *host_nx, double *host_vde);
/* Device-resident vde table (n x e doubles, vertex-id indexed) for exchange between ranks. */
int gnnpe_vde_device_ptr(gnnpe_ctx *ctx, void **dev_vde, void **dev_x);
/* Gather / scatter slab rows of the vde table to/from a packed buffer of (end-begin) x e doubles in
 * processing order -- the all-gather payload (x needs no exchange: labels are replicated). */
int gnnpe_vde_pack_slab(gnnpe_ctx *ctx, uint32_t begin, uint32_t end, void *dev_buf);
int gnnpe_vde_unpack_slab(gnnpe_ctx *ctx, uint32_t begin, uint32_t end, const void *dev_buf);
/* The same for every rank's slab at once, from the all-gathered buffer [n_ranks][stride rows][e] (stride = the longest
 * slab; bounds[n_ranks + 1] host words, the slabs' first positions): one launch per step instead of n_ranks - 1.  The
 * rows of skip_rank (this rank's own: already in the table) are left alone; pass n_ranks to scatter all. */
int gnnpe_vde_unpack_all(gnnpe_ctx *ctx, uint32_t n_ranks, const uint32_t *bounds, uint32_t stride, uint32_t skip_rank,
                         const void *dev_buf);

/* ---- R2: path enumeration (dfs + VectorHash, custom.h:52-92; driver loop main.cpp:87-96) ------- */
/* Counts the paths of the context's slab with l edges (l+1 vertices).  l=2 is the reference's path (it
 * only works for l=2); l=3 is the same rule with the DFS depth fixed (4-vertex simple paths,
 * kept iff rank[last] > rank[first]; BASELINE config 5) and needs the adjacency rows two hops from the
 * slab on the device.  host_per_start[i] (slab length entries, may be NULL) = number of paths whose
 * start is sorted_nodes[slab_begin+i]; *host_total = their sum.  Leaves the scanned offsets on the
 * device for gnnpe_fill_paths*. */
int gnnpe_count_paths(gnnpe_ctx *ctx, uint32_t l, uint64_t *host_per_start, uint64_t *host_total);
/* The same count (l=2, the default rank-sorted enumeration) ENQUEUED only: no read-back, no host synchronisation; the
 * total stays in device memory.  For a step of a multi-GPU build (main.cpp:87-96 across devices) whose outputs were
 * sized by an earlier pass: count_enqueue -> gnnpe_count_total_device (the word a collective sends) ->
 * gnnpe_fill_paths_capped_device, with nothing on the host between the launches.  gnnpe_count_total fetches the
 * number whenever the host wants it (synchronises); every entry point that needs it on the host does so itself. */
int gnnpe_count_paths_enqueue(gnnpe_ctx *ctx, uint32_t l);
int gnnpe_count_total(gnnpe_ctx *ctx, uint64_t *host_total);
/* Copies the last count's total (one uint64) to caller-owned device memory on the context's stream. */
int gnnpe_count_total_device(gnnpe_ctx *ctx, void *dev_u64);

/* ---- R2 + R5: emit paths and their embeddings (gen_pde, custom.h:546-572) ----------------------- */
/* Emits the slab-local paths [begin, end) in the reference's order (start vertices in processing
 * order, neighbours ascending, reverse-dedup): vids (end-begin) x (l+1) uint32; pde = concat of
 * vde rows, pde_label = concat of x rows, (end-begin) x e(l+1) doubles each.  Any output may be
 * NULL.  The *_device form writes device memory and only enqueues work on the stream. */
int gnnpe_fill_paths(gnnpe_ctx *ctx, uint64_t begin, uint64_t end, uint32_t *host_vids, double *host_pde,
                     double *host_pde_label);
int gnnpe_fill_paths_device(gnnpe_ctx *ctx, uint64_t begin, uint64_t end, void *dev_vids, void *dev_pde,
                            void *dev_pde_label);
/* Emits rows [0, min(total, cap_rows)) into buffers of cap_rows rows without the host knowing the total (after
 * gnnpe_count_paths_enqueue): the kernel clips against the per-start output ranges, so nothing is written beyond
 * cap_rows.  ids and pde only (either may be NULL). */
int gnnpe_fill_paths_capped_device(gnnpe_ctx *ctx, uint64_t cap_rows, void *dev_vids, void *dev_pde);

/* ---- output pool: where the emitted rows live on the device -------------------------------------------------------
 * The reference keeps `all_paths` / `pde` in host vectors (main.cpp:87-96, custom.h:546-572); here they are device
 * buffers, and on MI355X the rate a kernel streams into a multi-GiB buffer differs by up to 25 % between allocations
 * (stable for the life of a buffer, not predictable from its address: scripts/vmm_probe*.hip).
 * A pool draws `candidates` independent allocations for rows_cap rows (ids: rows_cap x L uint32; pde: rows_cap x D
 * doubles, D = 0 for none), times the consumer in each -- the emit kernel itself when the context holds an l = L-1
 * count of at most rows_cap paths (call gnnpe_count_paths first), a streaming write otherwise -- keeps the fastest and
 * frees the others before it returns (transient memory: candidates x the output size, bounded by the free memory).
 * candidates = 1 takes what comes, unprobed.  The buffers stay valid until gnnpe_output_pool_destroy.
 * With an l = 2 count on the context the pool also runs gnnpe_emit_calibrate_device on the buffer it keeps (nine launches
 * of the emit kernels): pass candidates | GNNPE_POOL_NO_CALIBRATION where the buffer is filled once or twice (a CLI that
 * fills it once per chunk), so that the launches that choose a shape do not outnumber the ones that use it. */
#define GNNPE_POOL_NO_CALIBRATION 0x80000000u
typedef struct gnnpe_pool gnnpe_pool;
int gnnpe_output_pool_create(gnnpe_ctx *ctx, uint64_t rows_cap, uint32_t L, uint32_t D, uint32_t candidates, gnnpe_pool **pool);
int gnnpe_output_pool_acquire(gnnpe_pool *pool, void **dev_ids, void **dev_pde, uint64_t *rows_cap);
/* probe_ms[0 .. min(cap, candidates drawn)) = time of the probe in every candidate, *kept = the one in use,
 * *probed_with_emit_kernel = 1 if the probe was the emit kernel (0: streaming write, or a single unprobed candidate). */
int gnnpe_output_pool_report(gnnpe_pool *pool, uint32_t cap, float *probe_ms, uint32_t *n_candidates, uint32_t *kept,
                             int *probed_with_emit_kernel);
void gnnpe_output_pool_destroy(gnnpe_pool *pool);

/* Order-sensitive 64-bit checksum of n_rows emitted rows (n_rows x L uint32 on the device) whose first
 * row has global path id first_id; checksums of consecutive chunks ADD (mod 2^64).  For outputs too large
 * to keep (config 5: count + checksum only) and for comparing rank counts. */
int gnnpe_rows_checksum_device(gnnpe_ctx *ctx, uint64_t n_rows, uint32_t L, const void *dev_ids, uint64_t first_id,
                               uint64_t *host_sum);

/* Per path, the partition of its start vertex (membership[vids[0]]): what main.cpp:98-108 groups
 * partition_paths.txt by.  dev_part: (end-begin) uint32. */
int gnnpe_path_partitions_device(gnnpe_ctx *ctx, uint64_t begin, uint64_t end, void *dev_part);

/* ---- R7: text rendering of the offline outputs (writers, main.cpp:98-119) ------------------------ */
/* all_paths.txt body: each row "<v0> <v1> <v2> \n" -- every id followed by one space, then '\n'
 * (main.cpp:114-118).  dev_vids: n_rows x L uint32.  dev_text == NULL only sizes (*nbytes). */
int gnnpe_text_paths(gnnpe_ctx *ctx, uint64_t n_rows, uint32_t L, const void *dev_vids, void *dev_text, uint64_t cap,
                     uint64_t *nbytes);
/* partition_paths.txt body: one uint64 id per line (main.cpp:102-106). */
int gnnpe_text_ids(gnnpe_ctx *ctx, uint64_t n, const void *dev_ids, void *dev_text, uint64_t cap, uint64_t *nbytes);
/* Global ids (id_base + i, ascending) of the rows i < n whose partition dev_part[i] == pid: the list
 * main.cpp:98-108 writes for partition pid.  dev_ids: room for n uint64. */
int gnnpe_select_partition(gnnpe_ctx *ctx, uint64_t n, const void *dev_part, uint32_t pid, uint64_t id_base,
                           void *dev_ids, uint64_t *count);

/* ---- R6: index.dat (Partition ctor build loop custom.h:235-257 -> RTree::insert rtree.cpp:198-284) -- */
/* Bulk-loads the R-tree over `cnt` paths given as vertex tuples (cnt x L uint32, device memory, in the
 * partition's path order: leaf `son` = row index, custom.h:243) with point MBRs lo = hi = pde row
 * (custom.h:244-248), and assembles the complete file image in the reference's block format in device
 * memory owned by the context (valid until the next call).  hdr_out = {blocklength, node blocks, dim,
 * num_data, leaf nodes, internal nodes, root_is_data, root}.  Tree shape differs from the reference's
 * insertion-built tree (it is not reproducible even by the reference); every consumer
 * constraint of the online code holds (dense block ids, internal root, enclosing MBRs). */
int gnnpe_build_index_device(gnnpe_ctx *ctx, uint64_t cnt, uint32_t L, const void *dev_vids, void **dev_image,
                             uint64_t *nbytes, int32_t hdr_out[8]);
/* The image of partition `pid` straight from the enumeration state of the context (after gnnpe_vde + gnnpe_count_paths):
 * no tuple array.  For an l = 2 count the tree is built pair-major -- the (s, b) pairs (hub pairs: their 64-entry units)
 * are sorted by [label(s) | label(b) | z-order of vde[s], vde[b]] and the leaves read their points out of the pairs' row
 * blocks; for an l = 3 count triple-major (round 6, csrc/gnnpe_index_deep.hip.h) -- the (s, b, c) triples x 64-entry pieces of
 * c's row, each with the mask of its kept fourth vertices, sorted by [label(s) | label(b) | label(c) | z-order of their vde] --
 * otherwise (embedding widths without a specialised enumeration) the partition's tuples are collected and handed to
 * gnnpe_build_index_device.  Same file format, same
 * consumer constraints; the image stays valid until the next index call on the context. */
int gnnpe_build_index_partition_device(gnnpe_ctx *ctx, uint32_t pid, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8]);
/* The same image together with the tree's auxiliary index (Partition::build_auxiliary_index, custom.h:268-364: per node
 * block key, degrees[L], label_mbr[2D]; layout as gnnpe_aux_index_device).  For the pair-major build (l = 2, whole graph
 * or gnnpe_set_degrees) the leaf kernel computes the leaves' rows while it assembles them and only the few upper levels
 * are a separate pass; otherwise the generic pass over the finished image runs.  Arrays are context-owned. */
int gnnpe_build_index_partition_aux_device(gnnpe_ctx *ctx, uint32_t pid, void **dev_image, uint64_t *nbytes, int32_t hdr_out[8],
                                           void **dev_key, void **dev_degrees, void **dev_label_mbr, uint32_t *n_nodes);
/* Whole job for partition `pid` of the context's slab: build (gnnpe_build_index_partition_device), write `path`
 * (<f>gnn-pe/partitions/partition-<pid>/index.dat).  The reference online run then skips its insert
 * loop (custom.h:222-235 only tests that the file exists). */
int gnnpe_build_index(gnnpe_ctx *ctx, uint32_t pid, const char *path);
/* index.dat of partitions 0..n_parts-1, all at once: every image is built on the device first, then the files are
 * written side by side (one writer thread per file: buffered writes to one file serialise, writes to different files do
 * not).  Same files as n_parts calls of gnnpe_build_index; aux_paths (or NULL): where to leave every partition's
 * auxiliary index (gnnpe_build_aux_index below). */
int gnnpe_build_index_files(gnnpe_ctx *ctx, uint32_t n_parts, const char *const *paths, const char *const *aux_paths);

/* Pieces for callers that assemble a partition's tuples themselves (multi-GPU host: the tuples of partition pid come
 * from every rank): rows of dev_rows (k x L uint32 selected by the global ids dev_sel[i] - sel_base) -> dev_out; and a
 * device buffer written to a file through pinned staging buffers (copy-back overlapped with the writes). */
int gnnpe_gather_rows_device(gnnpe_ctx *ctx, uint64_t k, uint32_t L, const void *dev_sel, uint64_t sel_base,
                             const void *dev_rows, void *dev_out);
int gnnpe_write_device_file(gnnpe_ctx *ctx, const void *dev_src, uint64_t nbytes, const char *path);

/* Same file format over explicit rectangles: cnt x 2*dim doubles (lo0, hi0, lo1, hi1, ...), son = row. */
int gnnpe_build_box_index_device(gnnpe_ctx *ctx, uint64_t cnt, uint32_t dim, const void *dev_boxes, void **dev_image,
                                 uint64_t *nbytes, int32_t hdr_out[8]);

/* ---- GNN-PGE offline (GNN-PGE/src/main.cpp:91-195) ------------------------- */
/* path_group / path_label_group of every vertex: per-dimension [min, max] over the embeddings
 * [vde[v], vde[u]] / [x[v], x[u]] of its 1-hop paths (v, u); n x 4e doubles each, laid out
 * (lo0, hi0, lo1, hi1, ...).  Needs gnnpe_vde.  Host outputs may be NULL. */
int gnnpe_pge_groups(gnnpe_ctx *ctx, double *host_path_group, double *host_path_label_group);
/* Device addresses of the two arrays gnnpe_pge_groups computed (n x 4e doubles each; vertex-id indexed): what a caller that
 * keeps working on the device reads instead of the host copies (GNN-PGE/src/main.cpp:141-176 fills the same values). */
int gnnpe_pge_device_ptr(gnnpe_ctx *ctx, void **dev_path_group, void **dev_path_label_group);
/* R-tree of one GNN-PGE partition (GNN-PGE/include/custom.h:141-195): entry i = path_group of
 * host_vertices[i] (the partition's vertices in membership.txt order), son = i; written to `path`
 * (<f>gnn-pge/partitions/partition-i/index.dat). */
int gnnpe_pge_build_index(gnnpe_ctx *ctx, uint64_t n_sel, const uint32_t *host_vertices, const char *path);

/* ---- the online filter ------------------------------------------------------------ */
/* Query side (host): what main.cpp:136-151 does to the query graph before it touches the index -- dfs_query
 * (custom.h:94-119), gen_vde, gen_query_pde (custom.h:574-631: paths sorted by degree weight, greedy vertex
 * cover).  Returns the plan's paths: vids / labels / degrees (n_paths x 3 uint32) and pde (n_paths x 3e doubles),
 * malloc'ed (gnnpe_host_free). */
int gnnpe_host_query_plan(const char *query_graph_path, uint32_t e, uint32_t *n_query_vertices, uint32_t *n_paths,
                          uint32_t **vids, uint32_t **labels, uint32_t **degrees, double **pde);
/* Data side (device): Partition::query (custom.h:366-489) for all partitions at once.  The R-tree traversal only
 * prunes; its result is its leaf test (custom.h:404-431) applied to every data path, which is what this does, fused
 * with the enumeration of the context's slab (gnnpe_set_order + gnnpe_vde first; no counts, no emitted paths; a
 * slab-only context also needs gnnpe_set_degrees and its halo rows).  host_bitmap:
 * n_query_vertices x ceil(n/32) uint32, bit v of row u = data vertex v is a candidate of query vertex u -- the
 * reference's candidate_set (main.cpp:165-171), ready for its refinement.  device_ms (may be NULL): time on the device. */
/* A context that holds only a slab's rows (gnnpe_load_rows) filters its own paths once it knows the degree of every
 * vertex (n uint32; the ranks' bitmaps are then OR-ed, dist.py).  Not needed after gnnpe_load_csr. */
int gnnpe_set_degrees(gnnpe_ctx *ctx, const uint32_t *host_degrees);
int gnnpe_filter_candidates(gnnpe_ctx *ctx, uint32_t n_paths, const uint32_t *q_vids, const uint32_t *q_labels,
                            const uint32_t *q_degrees, const double *q_pde, uint32_t n_query_vertices, double epsilon,
                            uint32_t *host_bitmap, double *device_ms);

/* The refinement half of the reference's online step (custom.h:634-932) is outside SURVEY section 8's scope (frozen since round 1)
 * and ships in a library of its own since round 6: include/gnnpe_online.h, libgnnpe_online.so (gnnpe_refine, gnnpe_host_refine). */

/* ---- the online side's data load ------------------------------------------------------- */
/* `gnnpe_main --sidecars` leaves <f>gnn-pe/paths.bin (magic "GNNPEPTH", uint32 version = 1, uint32 L, uint64 P, then the
 * P x L uint32 rows of all_paths.txt) and <f>gnn-pe/vde.bin (uint32 n, e; x, nx, vde: n x e doubles each).  This returns
 * what gen_pde (custom.h:546-572) builds by re-parsing the text -- per path vids / labels / degrees (P x L uint32) and
 * pde / pde_label (P x L*e doubles; labels[] and degrees[] are the graph's, `vertices[vid].label/.degree`) -- as flat
 * malloc'ed arrays (gnnpe_host_free).  Host only, no GPU.  gnnpe_host_write_paths_header writes the 24-byte header to an
 * open FILE* (the writer side, used by the CLI). */
int gnnpe_host_load_path_sidecar(const char *paths_bin, const char *vde_bin, uint32_t n, const uint32_t *labels,
                                 const uint32_t *degrees, uint64_t *n_paths, uint32_t *L, uint32_t *e, uint32_t **vids,
                                 uint32_t **path_labels, uint32_t **path_degrees, double **pde, double **pde_label);
int gnnpe_host_write_paths_header(void *file, uint32_t L, uint64_t n_paths);

/* The partition's copy of the paths -- the loop of Partition::Partition (custom.h:205-216: `paths.push_back(
 * data_paths[path_id])` for every id of partition_paths.txt): the same arrays as gnnpe_host_load_path_sidecar, but
 * only the rows of one partition, in the order of its partition_paths.txt (`path_ids`, also returned), i.e. indexed by
 * what the leaf entries of that partition's index.dat carry in `son` (custom.h:243). */
int gnnpe_host_load_partition_sidecar(const char *paths_bin, const char *vde_bin, const char *partition_paths_txt, uint32_t n,
                                      const uint32_t *labels, const uint32_t *degrees, uint64_t *n_paths, uint32_t *L,
                                      uint32_t *e, uint32_t **path_ids, uint32_t **vids, uint32_t **path_labels,
                                      uint32_t **path_degrees, double **pde, double **pde_label);

/* The auxiliary index of a partition's R-tree -- Partition::build_auxiliary_index (custom.h:268-364), which the
 * reference recomputes on every start of `-m online` by walking the tree block by block: per node block id,
 * degrees[L] (largest degree per path position below the node), label_mbr[2D] (lo0, hi0, lo1, hi1, ... of pde_label
 * below the node) and key = 0 - hi_0 - ... - hi_{D-1} of the node's entry in its parent (0 for the root).
 * gnnpe_aux_index_device: bottom-up pass over ANY index.dat image in device memory (bulk-loaded here or written by the
 * reference's insert loop); dev_tuples = the partition's paths, [cnt x L] uint32 vertex ids in partition order (what
 * the leaf entries' `son` index).  Needs the graph, the label table and gnnpe_vde in the context.  Outputs are
 * context-owned device arrays (valid until the next call): key double[n_nodes], degrees uint32[n_nodes x L], label_mbr
 * double[n_nodes x 2D]. */
int gnnpe_aux_index_device(gnnpe_ctx *ctx, const void *dev_image, uint64_t nbytes, uint64_t cnt, uint32_t L,
                           const void *dev_tuples, void **dev_key, void **dev_degrees, void **dev_label_mbr,
                           uint32_t *n_nodes, uint32_t *dim);
/* Partition pid of the current count: its index image (the one gnnpe_build_index just built, or a fresh build), its
 * auxiliary index, written to `path` as aux_index.bin: magic "GNNPEAUX", uint32 version = 1, L, D, reserved, uint64
 * n_nodes, then key, degrees, label_mbr as above.  `gnnpe_main --index --sidecars` leaves one next to every index.dat. */
int gnnpe_build_aux_index(gnnpe_ctx *ctx, uint32_t pid, const char *path);
/* Host-side reader of aux_index.bin: malloc'ed arrays (gnnpe_host_free).  No GPU. */
int gnnpe_host_load_aux_index(const char *path, uint32_t *n_nodes, uint32_t *L, uint32_t *D, double **key, uint32_t **degrees,
                              double **label_mbr);

/* ---- introspection for bench / tests ------------------------------------------------------------ */
/* Name of the kernel instantiation that dominates the fill (for matching rocprofv3 rows). */
const char *gnnpe_fill_kernel_name(void);
/* Selects the enumeration implementation (call before gnnpe_count_paths).  Both produce identical outputs:
 *   4 one wave per start vertex over rank-sorted neighbour records; rows longer than 64 are streamed in id order by
 *     the same kernel (default)
 *   1 one wave per (start, middle) pair, direct stores, run-time embedding width: the generic form used for widths
 *     without a specialised instantiation (e not in {1,2,3,4,8}); selectable as the A/B baseline */
int gnnpe_set_fill_variant(gnnpe_ctx *ctx, int variant);
/* Shape of the emit launch of variant 4 (the reference's dfs + gen_pde, custom.h:66-92, 546-572); outputs are identical.
 * Times: BASELINE config 3 into an allocation of the fast / of the slow class, profiles/r05_emit_ab.txt, r06_emit_oneshot.txt.
 *   0    whichever gnnpe_emit_calibrate_device measured fastest into the fill's output buffer; shape 1 for a buffer nobody
 *        calibrated (default)
 *   1    one wave per start vertex, ONE-SHOT since round 6 -- workgroups in launch order, exit; 128 rows staged per flush
 *        (k_fill_ranked): 2.73-2.76 / 3.23-3.3 ms (rounds 4-5: a resident grid of five workgroups per CU, start vertices from
 *        ticket counters: 2.81-2.84 / 3.3-3.4)
 *   4    the same kernel as a resident grid of three workgroups per CU, start vertices in order from ticket counters:
 *        3.05-3.25 / 3.13-3.6 ms -- the faster one in one kind of slow allocation (at widths e > 2: natural occupancy)
 *   2    one wave per output tile of 64 rows, workgroups in launch order, one store burst per wave (k_fill_tiles): 3.2-3.3 /
 *        3.3-3.55 ms; graphs with rows longer than 64 still take the start-vertex kernel, which streams such rows
 *   3    persistent waves that take output tiles in order from ticket counters, three tiles in flight per wave
 *        (k_fill_tickets, e <= 2): 3.9 / 4.1 ms -- measured and never chosen.  Since ABI 6 the kernel lives in the diagnostic
 *        build only (make DIAG=1 diag); the shipped library accepts the value and answers with shape 2
 * The environment variable GNNPE_EMIT=starts|starts_low|tiles, read when the context is created, overrides the context's setting. */
int gnnpe_set_emit_shape(gnnpe_ctx *ctx, int shape);
/* Times the emit shapes 1, 4 and 2 into the caller's output buffers (rows [0, total) of the context's current l=2 count, which
 * must fit rows_cap, the buffers' capacity in rows; three launches each: the buffers are overwritten with the paths) and
 * remembers the fastest FOR THESE BUFFERS: with emit shape 0 later fills into them take it.  The rate a kernel reaches depends
 * on the allocation it writes to and on its shape.  ms_by_shape (5 floats, indexed by shape; may be null)
 * receives the times, 0 for a shape that was not run -- all 0 when only one shape applies (hub rows, fewer than 2^24 paths).
 * The entry is dropped when the buffer is freed through gnnpe_dev_free / gnnpe_output_pool_destroy.
 * gnnpe_output_pool_create calibrates the buffer it keeps unless told not to. */
int gnnpe_emit_calibrate_device(gnnpe_ctx *ctx, uint64_t rows_cap, void *dev_vids, void *dev_pde, float *ms_by_shape, int *shape_kept);
/* Bytes of the index.dat a partition of `points` paths of dimension D becomes (header block + one 4 KiB block per node of the
 * bulk-loaded tree; rtnode.cpp:27-28, blk_file.cpp:38-52), by the builder that writes it: 0 = the pair-major / triple-major build behind
 * gnnpe_build_index / gnnpe_build_index_partition_device (nodes of min(capacity - 1, 64) entries), 1 = the tuple-array build
 * gnnpe_build_index_device (min(capacity - 2, 64): what the multi-GPU path uses).  No GPU, no context: callers check the
 * reference's 2 GiB limit (blk_file.h:32-33) BEFORE anything is built or written.  0 for a dimension no node holds. */
uint64_t gnnpe_index_file_bytes(uint64_t points, uint32_t D, int builder);
/* Name of the emit kernel the context's last fill launched ("" before the first fill): the rocprofv3 row to match. */
const char *gnnpe_emit_kernel_name(gnnpe_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
