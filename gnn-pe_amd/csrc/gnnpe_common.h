// gnnpe_common.h -- internal helpers shared by the HIP translation units of libgnnpe_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gnnpe_hip.h"

struct gnnpe_ctx;

namespace gnnpe {

void set_error(const char *fmt, ...);
// gnnpe_aux.hip: {degree, label} word of every vertex (c->aux_vdl)
int ensure_vertex_words(gnnpe_ctx *c);
int resolve_total(gnnpe_ctx *c);  // gnnpe_engine.hip: fetch the count's total if the last count was enqueue-only

#define GNNPE_HIP_TRY(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            gnnpe::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                             __LINE__);                                                           \
            return GNNPE_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

#define GNNPE_REQUIRE(cond, code, ...)                                                            \
    do {                                                                                          \
        if (!(cond)) {                                                                            \
            gnnpe::set_error(__VA_ARGS__);                                                        \
            return (code);                                                                        \
        }                                                                                         \
    } while (0)

// Grow-only device buffer: allocations are made outside the steady-state step (first call sizes
// them), so repeated steps enqueue kernels only.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }  // locals on error paths and the context's members free themselves
    int reserve(size_t need)
    {
        if (need <= bytes) return GNNPE_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return GNNPE_ERR_HIP;
        }
        bytes = want;
        return GNNPE_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

constexpr int kBlock = 256;            // 4 wave64 per workgroup
constexpr int kMaxGrid = 256 * 8 * 4;  // grid-stride cap: 256 CUs x 8 blocks (guide G11) x 4

inline int grid_for(uint64_t items, int per_block = kBlock)
{
    uint64_t g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > (uint64_t)kMaxGrid) g = kMaxGrid;
    return (int)g;
}

}  // namespace gnnpe

struct gnnpe_pool;
struct gnnpe_ctx;
namespace gnnpe {
// The environment switches of the shipped library, read ONCE per context (gnnpe_create) and never on a launch path:
//   GNNPE_EMIT=starts|starts_low|tiles   the emit shape, whatever was set or calibrated (gnnpe_set_emit_shape's 1 / 4 / 2)
//   GNNPE_DEEP_COUNT=merge               l = 3: the pointer-walk count kernel instead of the histogram one
//   GNNPE_DEEP_EMIT=slices|units         l = 3: force one of the two emit forms
//   GNNPE_AUX_WIDE=1                     8-byte {degree, label} words behind the aux records even where they fit the id bits
//   GNNPE_DEBUG=1                        launch shapes / pool candidates on stderr
//   GNNPE_TESTING=k=v,...                testing aids: pool_min_probe_bytes (below it the pool takes what comes: 512 MiB),
//                                        index_keep_bytes (cap on the device copies gnnpe_build_index_files keeps per wave),
//                                        index_max_units (sort units the l = 3 index build accepts before it takes the tuple-array build: 2^31)
// Everything else that rounds 2-5 switched by environment for A/B runs exists in diagnostic builds only (make DIAG=1:
// gnnpe::diag_int below): the static start-vertex walk, staged rows, LDS pads, the ticket / strip-job emit kernels, tile heights,
// rows per wave of the count kernel, the leaf kernel's XCD chunks, candidate draws of the image buffer, knock-outs, stamps.
struct Switches {
    int emit = 0;  // 0 none, else the emit shape to force
    bool deep_merge = false, aux_wide = false, debug = false;
    int deep_emit = 0;  // 1 slices, 2 units
    uint64_t pool_min_probe_bytes = 512ull << 20, index_keep_bytes = ~0ull, index_max_units = 1ull << 31;
};
Switches read_switches();  // gnnpe_engine.hip
#ifdef GNNPE_DIAG
inline long diag_int(const char *name, long dflt)
{
    const char *ev = getenv(name);
    return ev ? atol(ev) : dflt;
}
#else
constexpr long diag_int(const char *, long dflt) { return dflt; }
#endif
void pool_free(gnnpe_pool *p);  // gnnpe_pool.hip: releases a pool's memory (the context is still alive)
// gnnpe_pool.hip: the fastest of `candidates` allocations of `bytes` under a streaming write
int draw_device_buffer(gnnpe_ctx *c, uint64_t bytes, uint32_t candidates, void **out);
}

struct gnnpe_ctx {
    int device = 0;
    gnnpe::Switches sw;  // the environment as gnnpe_create found it
    std::vector<gnnpe_pool *> pools;  // output pools created on this context and not yet destroyed: freed with it
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;

    // ---- graph (R0) ----
    uint32_t n = 0;
    bool have_graph = false;
    bool rows_identity = true;  // rows == 0..n-1 (full CSR) vs. an owned-row list
    uint32_t n_rows = 0;        // rows held in storage order (owned rows; halo rows come after)
    uint64_t nbr_used = 0, nbr_owned = 0, nbr_cap = 0;
    gnnpe::DevBuf adj_start, adj_deg, present, owned, nbrs, nbr_rank, labels, rows, held, revpos, srec;
    gnnpe::DevBuf nbr_label;  // label of every OWNED adjacency entry (what gen_vde sums over): built with the rows
    // non-simple input (gnnpe_set_multigraph_rows): the owned rows with their repeated entries -- offsets by row
    // position and the entries' labels -- for gen_vde; everything else runs on the simple rows above
    bool multigraph = false;
    gnnpe::DevBuf mg_off, mg_label;
    // rows longer than 64 entries ("hub" rows of the l=2 enumeration): ids and adjacency ranges; graph-only, rebuilt
    // whenever rows are loaded / appended / dropped
    gnnpe::DevBuf hub_rows, hub_beg, hub_end;
    uint32_t n_hub = 0;
    uint64_t hub_entries = 0;
    int num_cus = 256;
    gnnpe::DevBuf text_len, text_off;  // R7 scratch
    gnnpe::DevBuf index_image, idx_keys, idx_vals, idx_mbr;
    // pair-major index build (gnnpe_index.hip): sorted pair records, prefix of their path counts, per-partition ranges;
    // valid for one count (count_gen)
    gnnpe::DevBuf px_recs, px_sorted, px_pref, px_pbase, px_first, px_tmp, px_units, px_hubs;
    std::vector<uint64_t> px_bounds, px_points;  // first sorted pair / first point of every partition (p + 1 entries)
    uint64_t count_gen = 0, px_gen = 0;
    bool px_valid = false;  // R6 scratch + the assembled index.dat image
    // the triple-major order of an l = 3 count (gnnpe_index_deep.hip.h): shares px_recs / px_sorted / px_pref / px_first / px_tmp
    gnnpe::DevBuf tx_cpre, tx_row_units, tx_toff, tx_padj;
    std::vector<uint64_t> tx_bounds, tx_points;
    uint64_t tx_gen = 0, tx_nu = 0, tx_refused_gen = ~0ull;  // tx_refused_gen: the count whose units did not fit (tuple-array build instead)
    bool tx_valid = false;
    gnnpe::DevBuf px_raux;  // {degree, label} strips beside the row blocks (k_px_raux), valid for one count like the pair order
    bool px_raux_valid = false;
    bool px_raux_compact = false;  // the word rides in the records' id bits (degree | label << px_raux_dbits)
    uint32_t px_raux_dbits = 0;
    // the pair-major leaf kernel stores used prefixes only: which buffer's block tails are known to be zero, and from which byte
    const void *img_scrub_ptr = nullptr;
    size_t img_scrub_bytes = 0;
    uint32_t img_scrub_from = 0;
    bool img_aux_valid = false;  // c->aux_key / aux_deg / aux_mbr hold the auxiliary index of the image in index_image
    uint32_t img_aux_nodes = 0;
    uint32_t n_held = 0;  // rows with adjacency on this device (owned, then appended halo rows)
    uint32_t halo_min_rank = 0;  // largest min_rank a halo row was truncated with (gnnpe_rows_append); 0 = none

    // ---- order (R1) ----
    bool have_order = false;
    uint32_t p = 1;
    uint32_t slab_begin = 0, slab_end = 0;
    bool slab_set = false;
    gnnpe::DevBuf sorted, rank, member;

    // ---- label table (R3) / vertex embeddings (R4) ----
    uint32_t n_labels = 0, e = 0;
    bool have_table = false, have_vde = false;
    bool labels_checked = false;  // every label indexes the table (checked once per label / table upload)
    gnnpe::DevBuf xtab, x, nx, vde, nbr_vde;
    // the label table by rank: xrank[label * e + k] = position of xtab[label][k] among the labels' values of dimension k
    // (equal values share a rank), xsorted[k * n_labels + r] = the value at rank r.  Min / max of label features over a set
    // of paths become min / max of 16-bit ranks (the index leaf kernel reduces them packed, two per dword)
    gnnpe::DevBuf xrank, xsorted;
    bool nbr_vde_valid = false;

    // ---- enumeration state (R2) ----
    bool counted = false;
    uint32_t l = 0;
    uint64_t n_edges = 0;  // directed (start, middle) pairs of the slab
    bool eoff_valid = false;  // the per-pair offsets of the current count exist (rank-sorted enumeration: built on demand, gnnpe_ensure_eoff)
    gnnpe::DevBuf scan_status;  // k_start_scan: one status word per tile of 256 start vertices, then its ticket counter
    uint64_t total_paths = 0;
    bool total_known = false;  // false after gnnpe_count_paths_enqueue until gnnpe::resolve_total fetches eoff[n_edges]
    gnnpe::DevBuf poffs, erow, pnbr, ecnt, eoff, cub_tmp, scratch, mark, small;
    int fill_variant = 4, counted_variant = 4;  // 4 = rank-sorted neighbour records (default), 1 = generic pair-wave
    bool slab_struct_valid = false;  // poffs / n_edges match the current graph, order and slab
    // output-tile-driven emit (gnnpe_fill_tiles.hip.h): {start, middle} of every pair (structure, valid with poffs) and the
    // tile table of one count (first pair of every 64-row output tile), built by the first fill that wants it
    gnnpe::DevBuf pst, tfirst;
    gnnpe::DevBuf tk_ctl, tk_jobs;  // emit shape 3: ticket heads + job counter; strip jobs {tile, first pair}
    bool pst_valid = false;
    uint32_t tile_rows = 0;  // rows per tile the table was built for
    uint64_t tile_gen = 0, tile_cap = 0;  // count the table belongs to; tiles it covers (+ 1 sentinel entry)
    // emit shape measured per output buffer (gnnpe_emit_calibrate_device): {buffer, faster shape}; consulted when emit_shape is 0
    // ... with the count it was measured for (paths, pairs, width) and the three times, so that a second request for the same
    // buffer and the same count is answered without the nine launches
    struct EmitPref {
        const void *key;
        int shape;
        uint64_t total, n_edges;
        uint32_t e;
        float ms[5];
    };
    std::vector<EmitPref> emit_prefs;
    const char *last_emit_kernel = "";
    int last_emit_per_cu = 0;  // workgroups per CU the last k_fill_ranked launch was held to (0: no cap)
    // 0 = as gnnpe_emit_calibrate_device measured for the fill's buffer (start-vertex waves where nothing was measured), 1 = start-vertex
    // waves, 2 = output tiles, 3 = ticket waves, 4 = start-vertex waves at three workgroups per CU (include/gnnpe_hip.h)
    int emit_shape = 0;
    bool ranked_vde_valid = false;  // the ranked records carry the current vde table
    // round 6, launches folded into their neighbours: k_vde wrote the count kernel's per-vertex records for the slab structure of
    // generation vinfo_gen (k_pack_vinfo is skipped while that is current); k_start_scan left the emit kernel's ticket heads zero;
    // the zero sentinel behind the pair records is in place for this many pairs
    bool vinfo_fused = false, heads_clean = false;
    uint64_t slab_struct_gen = 0, vinfo_gen = 0, rpairs_sentinel_at = ~0ull;
    const void *rpairs_sentinel_buf = nullptr;
    uint32_t *clear_words = nullptr;  // set by the count around build_ranked: words its row kernel zeroes for k_start_scan
    uint32_t n_clear = 0;
    bool clear_done = false;
    gnnpe::DevBuf vkey;             // R6: per-vertex sort-key parts {label, spread quantised vde}
    gnnpe::DevBuf rank_sorted, adj_end;  // l=3 count: every row's neighbour ranks in ascending order
    gnnpe::DevBuf rank_arg, rb_cnt, rb_first;  // ... the entry each sorted rank came from; row-batches (64 third vertices) per row
    gnnpe::DevBuf deep_hub_batches, deep_ubase;  // ... {row, batch} of the rows the workgroup form of the l=3 count takes; first unit of (s, b) by b's sorted position
    gnnpe::DevBuf ufirst, upair, uoff;  // l=3 work units: first unit of a pair, pair of a unit, output slot of a unit
    gnnpe::DevBuf dsl_first, dsl_kept;  // l=3 emission: first slice of every unit of the requested range, kept rows per slice (and their scan)
    uint64_t n_units = 0;
    gnnpe::DevBuf deg_all;          // online filter on a slab: degree of EVERY vertex (gnnpe_set_degrees)
    gnnpe::DevBuf q_plan, q_bitmap, q_work, q_tmp;  // online side: grow-only, so that a query allocates nothing
    bool have_deg_all = false;
    bool vkey_valid = false;
    uint32_t vkey_zb = 0, vkey_lb = 0, vkey_sbits = 32;  // sbits 32 = wide (64-bit) table only
    gnnpe::DevBuf rpairs, rrecs, vinfo;
    // row blocks of the ranked records: first 128-byte unit of every held row's block, total units; depends on the held
    // rows and the record size (e, packed ids) only, so it is laid out when one of them changes
    gnnpe::DevBuf rblock;
    uint64_t rblock_units = 0;
    bool rblock_valid = false;
    uint32_t rblock_e = 0;
    gnnpe::DevBuf slab_bounds;  // gnnpe_vde_unpack_all: the ranks' slab bounds on the device
    gnnpe::DevBuf aux_upper;  // inner node blocks listed by the leaf-level launch of the auxiliary index pass
    gnnpe::DevBuf aux_vdl;  // {degree, label} per vertex for the auxiliary index pass (+ the largest degree and label behind it)
    bool aux_vdl_valid = false;
    gnnpe::DevBuf aux_key, aux_deg, aux_mbr;  // auxiliary index of the last gnnpe_aux_index_device call (gnnpe_aux.hip)
    // which partition's image index_image holds (set by gnnpe_build_index_partition_device, cleared by the other builds)
    bool img_valid = false;
    uint32_t img_pid = 0;
    uint64_t img_gen = 0, img_bytes = 0;
    gnnpe::DevBuf pge_pg, pge_plg;  // GNN-PGE path groups (n x 4e doubles each)
    bool have_pge = false;

    // pinned host words for small read-backs
    uint64_t *h_pinned = nullptr;
};

// drops the emit shape measured for any buffer that starts in [lo, lo + bytes) (gnnpe_engine.hip; called where buffers are freed)
void gnnpe_forget_emit_pref(gnnpe_ctx *c, const void *lo, size_t bytes);
// per-pair output offsets of the current count, built on demand (gnnpe_engine.hip; internal, not part of the ABI header)
extern "C" int gnnpe_ensure_eoff(gnnpe_ctx *c);
