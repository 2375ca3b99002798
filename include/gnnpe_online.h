/* gnnpe_online.h -- C-ABI of libgnnpe_online.so: the REFINEMENT half of the reference's online step (custom.h:634-932).
 *
 * Outside SURVEY section 8's scope: built in round 1 so that `gnnpe_main -m online` can print the reference's answer line, frozen
 * since, and since round 6 kept out of libgnnpe_hip.so -- the offline hot path and its (f) rows link and load
 * without it.  libgnnpe_online.so links against libgnnpe_hip.so (contexts, error text); `gnnpe_main` dlopen()s it for -m online
 * only.  Same conventions as gnnpe_hip.h: 0 = GNNPE_OK, gnnpe_last_error() for the text.
 */
#ifndef GNNPE_ONLINE_H
#define GNNPE_ONLINE_H

#include "gnnpe_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Refinement half of the reference's online step (custom.h:634-932), host side: the number of embeddings of the
 * query graph in the data graph (injective, labels equal, query degree <= data degree, query edges on data edges)
 * whose START vertex -- fewest candidates, ties to the larger degree, then the smaller id (custom.h:634-654) -- maps
 * into its candidate set, counted up to `limit` (the reference's -n).  candidate_bitmap as written by
 * gnnpe_filter_candidates. */
int gnnpe_host_refine(uint32_t n, const uint32_t *offsets, const uint32_t *nbrs, const uint32_t *labels,
                      const char *query_graph_path, const uint32_t *candidate_bitmap, uint64_t limit, uint64_t *answers);

/* The same count on the device: one thread per (start candidate, neighbour slot of its image), depth-first below that.
 * Needs the whole graph on the device (gnnpe_load_csr); query graphs of up to 32 vertices.  device_ms may be NULL. */
int gnnpe_refine(gnnpe_ctx *ctx, const char *query_graph_path, const uint32_t *candidate_bitmap, uint64_t limit,
                 uint64_t *answers, double *device_ms);

#ifdef __cplusplus
}
#endif
#endif
