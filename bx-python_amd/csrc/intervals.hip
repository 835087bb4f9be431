// intervals.hip -- the interval index behind IntervalTree/Intersecter.find().
//
// Reference algorithm (lib/bx/intervals/intersection.pyx): a randomised treap
// of Python nodes, find() = pointer-chasing DFS (:180-189) that reports, in
// in-order, every interval with  end > qs  and  start < qe.
//
// MI355X design (not a treap):
//   * seal(): radix-sort the intervals into the treap's in-order, which is the
//     sort by (start, end<=start first, -i / +i) (intersection.pyx:112-116);
//     keep SoA int32 arrays in that order: s_ord, e_ord, idx, plus the prefix
//     max of e_ord (pm) and the separately sorted ends (e_sorted).
//   * every rank query goes through a static 32-ary search tree: a node is
//     32 int32 keys = one 128-byte line; 8 lanes cooperate on a node (one
//     coalesced 16-byte load each, compare 4 keys, 3 DPP adds).  The top
//     levels are staged in LDS, the lower ones come from L2 / HBM.  A 10M
//     index is 5 levels deep: 3 LDS visits + 2 line fetches per rank.
//   * count = rank_lt(starts, qe) - rank_le(ends, qs) for proper queries on
//     proper targets; anything else (zero-length / reversed query, reversed
//     target) takes the exact window scan [first pm>qs, rank_lt(starts,qe)).
//   * find = the same window, compacted with wave ballots into CSR order.
//   * batches of >= 4 Mi queries leave the trees: the queries are bucketed by coordinate (histogram
//     + LDS-ordered scatter, "part_*" kernels), every bucket is searched against its slice of the
//     sorted arrays held in LDS by direct addressing, and the counts are gathered back; a batch
//     whose starts are already sorted skips the bucketing ("ivl_local_*" kernels).
//   * clusters (ClusterTree) fall out of s_ord and pm: a boundary wherever start - d > pm[i-1].
// Integer compares and popcounts only: HBM / LDS bound, no MFMA.
//
// Map of the file: build kernels and tree search helpers; direct count kernel; partitioned count
// path (histogram, column scans, scatter, tree and cell searches, sorted-batch kernel, gather);
// partitioned / sorted find (window, permute, fill kernels); cluster kernels; host structs and the
// extern "C" entry points (bxmi_ivl_*), DESIGN.md 3 has the measurements.
#include <climits>
#include <vector>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "primitives.hpp"

#include "count_direct.hpp"
#include "count_parts.hpp"
#include "count_bitmap.hpp"
#include "count_slices.hpp"
#include "find_exchange.hpp"
#include "count_dense.hpp"
#include "find_sorted.hpp"
#include "find_direct.hpp"
#include "cluster.hpp"

namespace bxmi {

static int64_t g_opt_partition = -1;  // -1 = auto (large batches), 0 = never, 1 = always
constexpr int64_t g_opt_partition_min = 4 << 20;  // auto: partition batches of at least this many queries
constexpr int PT_MAX_SUB = 1;  // scratch regions are addressed per sub-batch; one region since sub-batch pipelining was dropped
constexpr int PT_SLOT_STRIDE = PT_SLOTS + 8;  // per sub-batch: the partial totals, then the "unsorted" flag
static int64_t g_opt_sorted_path = 1;  // 1 = batches whose starts are already sorted skip the bucketing (detected on the device)
static int64_t g_opt_bitmap = -1;      // second-generation count pass (count_bitmap.hpp): -1 = when the index qualifies, 0 = never, 1 = same as -1
static int64_t g_opt_bitmap_min = 2 << 20;  // auto: batches of at least this many queries take it (when the index qualifies)
static int64_t g_opt_bm_variant = -1;  // tile kernel shape: -1 = by batch size, 0 = 512 threads x 32 queries, 1 = 1024 x 16, 2 = 1024 x 32 (32768-query tiles)
static int64_t g_opt_fx_direct = -1;   // find() through the exchange: 1 = the fill writes straight into the CSR list (query-order prefixes from the un-permute
                                       // kernel, no copy), 0 = scratch + copy, -1 = by the size of the list the handle expects (see ivl_find_fx)
static int64_t g_opt_find_sliced = 1;  // large unsorted find() batches through the exchange (count_slices.hpp) where the slice stage fits; 0 = the bucketed find
static int64_t g_opt_slice = -1;       // search stage on staged key slices (count_slices.hpp): -1 = where the images do not pay or fit, 0 = never, 1 = wherever it fits
static int64_t g_opt_sl_f = -1;        // buckets per slice unit = 2^f: -1 = by run length and LDS, else forced (tests)
constexpr int64_t g_opt_bm_chunk = 0;     // queries per search work item (0 = BM_CHUNK, twice that for bucket pairs)
static int64_t g_opt_sl_run_cap = 160;  // a slice unit grows only while its expected (tile, unit) run stays within this many records
static int64_t g_opt_sl_flat = 1;      // 1 = count-only passes on key slices take the flat 16-byte walk of count_dense.hpp (16-bit counts, unit run table), 0 = the 16 / 64 lanes-per-run kernels of count_slices.hpp
constexpr int64_t g_opt_sl_rbits = 20;    // a slice unit's offsets take at most this many bits of the 32-bit record (the rest holds the length)
static int64_t g_opt_sl_lanes = 0;     // lanes per (tile, unit) run: 0 = by expected run length, 16 or 64, -1 (set as 1) = the flat walk for long runs
static int64_t g_opt_bm_hard_ppm = 2000;  // an index qualifies while its hard cells stay below this many per million cells
static int64_t g_opt_flat = -1;       // the flat 16-byte walk on cell images of 2^18-coordinate units (count_dense.hpp, bp_*): -1 = dense indexes that qualify, 0 = never, 1 = every index that qualifies
static int64_t g_opt_sparse = -1;     // offset-cell images for sparse indexes (offset_cells.hpp; the persistent walk): -1 = sparse indexes that qualify, batches that bring enough queries per unit; 0 = never; 1 = whatever the batch size
static int64_t g_opt_bo_cell_log2 = 0;  // their cell width: 0 = from the index's density, 6..8 = forced
constexpr int64_t g_opt_bo_min_per_unit = 4096;  // queries per unit image a batch must bring (an image is 72 KB to load whatever the batch)
static int64_t g_opt_sorted_cells = 1;  // sorted batches on indexes with cell images: 1 = answered from the images stretch by stretch (bs_*), 0 = the first-generation kernel for sorted batches
static int64_t g_opt_dense = -1;      // search stage on dense unit images (count_dense.hpp): -1 = dense indexes that qualify, 0 = never, 1 = every index that qualifies
static int64_t g_opt_bd_chunk = 0;    // queries per search work item of the dense stage (0 = 256 Ki: one item per unit on a uniform 100 M batch)
static int64_t g_opt_bd_table_from = 0;  // dense images: overflow entries from which a cell gets a rank table (read when an index is prepared); 0 = 2 where the overflow area has the room, else 6
static int64_t g_opt_bd_blocks = 0;   // 1 = dense images with block-relative ranks even where unit-relative ones fit (tests)
static int64_t g_opt_bd_w8 = -1;      // 8-bit counts out of place: -1 = by index and feedback, 0 = never, 1 = whenever the layout allows
static int64_t g_opt_order_skip = -1;  // -1 = stop launching the order check after two batches in a row were not sorted (a probe of 8192 starts rides on the parameter kernel then), 0 = always check
static int64_t g_opt_clumped = -1;     // offset cells with a rank table per hard cell for duplicate-heavy indexes: -1 / 1 = where bitmap cells do not qualify and the tables fit, 0 = never (dense unit images)
static int64_t g_opt_tot_walk = 1;     // total-only batches on cell images: 1 = the walk keeps the totals itself (no slots, no count stores, no un-permute kernel), 0 = the counts pass without its stores
static int64_t g_opt_host_chunk = 8 << 20;  // queries per chunk of the host-pointer count (upload of chunk k+1 / pass on k / download of k-1 at once); 0 = one piece
static int64_t g_opt_host_touchers = 2;  // host threads that touch the output array's pages ahead of the downloads (0 = the download faults them in)
constexpr int64_t g_opt_bd_unit_log2 = 0;   // coordinates per unit of the dense images (read when an index is prepared): 0 = 19 if the duplicated coordinates fit its 12 KiB of overflow, else 18 (64 KiB: rank tables of clumped cells); 12 .. 19 = forced

// The option table: every knob of the interval path, its variable and how a value is normalised.  bxmi_set_option writes through
// it, bxmi_get_option / bxmi_option_at read it back -- the tests take their "defaults" from the library at import instead of
// keeping a copy.  Results never depend on an option.
struct IvlOpt {
    const char *key;
    int64_t *var;
    int64_t (*norm)(int64_t);
};
static const IvlOpt IVL_OPTS[] = {
    {"ivl.partition", &g_opt_partition, nullptr},
    {"ivl.sorted_path", &g_opt_sorted_path, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.bitmap_min", &g_opt_bitmap_min, nullptr},
    {"ivl.bitmap", &g_opt_bitmap, nullptr},
    {"ivl.bm_variant", &g_opt_bm_variant, [](int64_t value) -> int64_t { return value < 0 || value > 2 ? -1 : value; }},
    {"ivl.find_sliced", &g_opt_find_sliced, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.fx_direct", &g_opt_fx_direct, [](int64_t value) -> int64_t { return value < 0 ? -1 : value != 0; }},
    {"ivl.slice", &g_opt_slice, nullptr},
    {"ivl.sl_f", &g_opt_sl_f, [](int64_t value) -> int64_t { return value > SL_MAX_F ? SL_MAX_F : value; }},
    {"ivl.sl_run_cap", &g_opt_sl_run_cap, [](int64_t value) -> int64_t { return value < 8 ? 8 : value; }},
    {"ivl.sl_flat", &g_opt_sl_flat, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.sl_lanes", &g_opt_sl_lanes, [](int64_t value) -> int64_t { return value == 16 || value == 64 ? value : (value == 1 ? -1 : 0); /* 1 = the flat walk */ }},
    {"ivl.bm_hard_ppm", &g_opt_bm_hard_ppm, nullptr},
    {"ivl.flat", &g_opt_flat, nullptr},
    {"ivl.dense", &g_opt_dense, nullptr},
    {"ivl.sparse", &g_opt_sparse, nullptr},
    {"ivl.sorted_cells", &g_opt_sorted_cells, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.bo_cell_log2", &g_opt_bo_cell_log2, [](int64_t value) -> int64_t { return value < BO_MIN_K || value > BO_MAX_K ? 0 : value; }},
    {"ivl.bd_chunk", &g_opt_bd_chunk, [](int64_t value) -> int64_t { return value < 0 ? 0 : value; }},
    {"ivl.bd_blocks", &g_opt_bd_blocks, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.bd_table_from", &g_opt_bd_table_from, [](int64_t value) -> int64_t { return value < 1 || value > 64 ? 0 : value; }},
    {"ivl.bd_w8", &g_opt_bd_w8, nullptr},
    {"ivl.order_skip", &g_opt_order_skip, nullptr},
    {"ivl.clumped", &g_opt_clumped, nullptr},
    {"ivl.tot_walk", &g_opt_tot_walk, [](int64_t value) -> int64_t { return value != 0; }},
    {"ivl.host_chunk", &g_opt_host_chunk, [](int64_t value) -> int64_t { return value <= 0 ? 0 : ((value + 4095) & ~(int64_t)4095); }},
    {"ivl.host_touchers", &g_opt_host_touchers, [](int64_t value) -> int64_t { return value < 0 ? 0 : (value > 16 ? 16 : value); }},
};
constexpr int IVL_NOPTS = (int)(sizeof(IVL_OPTS) / sizeof(IVL_OPTS[0]));

int ivl_set_option(const char *key, int64_t value)
{
    for (int i = 0; i < IVL_NOPTS; i++)
        if (!strcmp(key, IVL_OPTS[i].key)) {
            *IVL_OPTS[i].var = IVL_OPTS[i].norm ? IVL_OPTS[i].norm(value) : value;
            return 1;
        }
    return 0;
}

int ivl_option_count() { return IVL_NOPTS; }

int ivl_option_at(int i, const char **key, int64_t *value)
{
    if (i < 0 || i >= IVL_NOPTS) return 0;
    *key = IVL_OPTS[i].key;
    *value = *IVL_OPTS[i].var;
    return 1;
}

}  // namespace bxmi

using namespace bxmi;

struct bxmi_ivl {
    // host staging of appended intervals (insertion order)
    std::vector<int32_t> h_start, h_end;
    // device copies in insertion order
    DevBuf d_start, d_end;
    int64_t n_dev = 0;   // intervals resident on device (insertion order)
    int64_t n = 0;       // intervals in the sealed index
    bool sealed = false;
    int has_reversed = 0;
    // sealed index
    DevBuf keys_a, keys_b, ekeys_a, ekeys_b;
    DevBuf s_ord, e_ord, idx, pm, e_sorted, flag;
    Tree treeS, treeE, treeP;
    SortScratch sort_scratch;
    DevBuf scan_scratch;
    // query scratch
    DevBuf q_s, q_e, q_cnt, q_lo, q_hi, q_off, q_hits, q_total;
    // partitioned count path
    PartGeom geom{0, 0};
    bool images_ready = false;
    DevBuf slice_bounds, cell_images, cell_meta, p_hist, p_table, p_pairs, p_dest, p_cnt, p_plan, p_slots, p_lo, p_hi, p_boffs;
    // second-generation count pass (count_bitmap.hpp)
    int32_t cmax = 0;            // largest end of the sealed index
    DevBuf bm_recs, bm_slots, bm_tbl, bm_runT, bm_grpcnt, bm_items, bm_params, bm_tesc;
    // slice search (count_slices.hpp)
    int sl_state = 0;            // 0 = not decided yet, 1 = boundary table built and a single bucket's keys fit the LDS, -1 = they do not
    unsigned sl_need[SL_MAX_F + 1] = {0, 0, 0, 0, 0, 0, 0};  // most keys a unit of 2^f buckets stages
    DevBuf sl_meta, sl_stats, sl_unitcnt, sl_cnt, sl_loff, sl_hits, sl_eid;
    // find() through the exchange, second generation (find_exchange.hpp)
    int fx_state = 0;            // 0 = not decided yet, 1 = the half-bucket ranks are built, -1 = the grid has no half buckets (shift 0)
    std::vector<int2> fx_meta2_host;   // the ranks at the half-bucket boundaries (read back once per sealed index)
    std::vector<FxPiece> fx_pieces_host;
    int fx_pieces_f = -1;        // the unit size (2^f buckets) the piece list was cut for
    double fx_hits_per_q = -1.0; // hits per query of the handle's latest find() through the exchange (predicts the next list's size), < 0 = none yet
    DevBuf lf_state;             // sorted find() in one kernel: the chunks' look-back words and the ticket
    DevBuf fx_meta2, fx_pieces, fx_tbl2, fx_runT2, fx_hc, fx_svq, fx_parts, fx_tile_tot, fx_tile_base, fx_work;
    // dense unit images (count_dense.hpp)
    int bd_state = 0;            // 0 = not decided yet, 1 = images built and the index qualifies, -1 = it does not
    unsigned bd_worst[2] = {0, 0};  // what bd_image_kernel reported: most keys of one block, most overflow entries of one unit
    BmGeom bd_geom{0, 0, 0, 0, 0, 0, 0, BD_RSHIFT, 0};
    int bp_state = 0;            // cell images of units for the flat walk: 0 = not decided yet, 1 = built and the index qualifies, -1 = it does not
    int64_t bp_hard_cells = 0;
    BmGeom bp_geom{0, 0, 0, 0, 0, 0, 0, BP_RSHIFT, 0};
    DevBuf bp_images, bp_stats;
    int bo_state = 0;            // offset-cell images of units: 0 = not decided yet, 1 = built (standard layout: sparse indexes) and the index qualifies, 2 = built
                                 // with a rank table per hard cell (clumped layout: duplicate-heavy indexes), -1 = neither
    bool bo_tried_clumped = false;
    int64_t bo_hard_cells = 0;
    BmGeom bo_geom{0, 0, 0, 0, 0, 0, 0, BP_RSHIFT, 0};  // dshift = cell width - 5
    DevBuf bo_images;
    DevBuf bs_plan;              // sorted batches on cell images: [unit bounds][item count][items]
    bool bd_blocks = false;      // the images' ranks are relative to blocks of 1024 cells (more than 32767 keys in some unit's slice)
    DevBuf bd_images, bd_stats, bd_cnt16, bd_unitT, bd_tend;
    // 8-bit counts between the search and the un-permute kernel (bm_count_segments): the un-permute kernel keeps a running
    // total of the counts that did not fit and mirrors it into host memory
    DevBuf bd_fb;                              // the running total (device)
    unsigned long long *bd_fb_host = nullptr;  // its mirror (host memory the device can write)
    int64_t w8_queries = 0;                    // queries of the passes launched with 8-bit counts
    bool w8_off = false;                       // too many of them did not fit: this index keeps 16-bit counts
    // [1] of the same host words: what the order check of an earlier pass found (ivl_local_count_kernel writes it)
    unsigned long long order_seq = 0, order_seen = 0;  // passes launched with an order check / the last one the host has seen the answer of
    int unsorted_streak = 0;                           // consecutive answers "not sorted"
    bool order_skip = false;                           // the order check is not launched at present (the tile sort reports the order)
    bool sl_eid_ready = false;
    int32_t *one_buf = nullptr;  // host-visible result of bxmi_ivl_find_one: [n:int64][ONE_CAP hits][completion word:int64]
    unsigned long long one_seq = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream_up = nullptr, stream_down = nullptr;  // the host-pointer entry points' copies either side of the pass (ivl_count_host_chunks)
    int device = 0;
};


static IndexDev index_dev(const bxmi_ivl *h);
template <typename Kern>
static int allow_big_lds(Kern k, size_t bytes);

// Everything a bucketed pass needs to know about one (sub-)batch.
struct PartPlan {
    int64_t ntiles;
    unsigned tgrid;        // tile-kernel grid: 8 XCD ranges of ceil(ntiles/8) tiles
    unsigned *table;       // [ntiles][PT_NB] destination of every (tile, bucket) run; row 0 = bucket offsets
    int32_t *plan;         // first search workgroup of every bucket
    int2 *bq;              // the queries in bucket order, (qs, qe) pairs
    unsigned short *lpos;  // per query (original order): slot inside its tile's sorted order
};

// Bucket the batch: histogram, column scan of the tile-major table, LDS-ordered scatter.  `sub`/`q0` select the
// scratch regions of a sub-batch (regions are addressed by query offset; sub-batches start on tile boundaries).
static int part_prepare(bxmi_ivl *h, int sub, int64_t q0, const int32_t *qs, const int32_t *qe, int64_t nq, bool want_lpos, hipStream_t st,
                        PartPlan *pp, unsigned *unsorted /* zeroed flag, set by the histogram pass when the starts are not sorted */,
                        bool skip_sorted /* the passes after the histogram exit at once on a sorted batch (count path) */)
{
    const unsigned *gate = skip_sorted ? unsorted : nullptr;
    pp->ntiles = div_up(nq, PT_TILE);
    pp->tgrid = (unsigned)(((pp->ntiles + 7) >> 3) << 3);
    const int rows_per_block = (int)div_up(pp->ntiles, 64);  // ~64 row blocks: the serial middle kernel stays short
    const int nrb = (int)div_up(pp->ntiles, rows_per_block);
    pp->table = h->p_table.as<unsigned>() + (q0 / PT_TILE) * PT_NB;
    unsigned *partial = h->p_hist.as<unsigned>() + (int64_t)sub * 80 * PT_NB;
    pp->plan = h->p_plan.as<int32_t>() + (int64_t)sub * (PT_NB + 8);
    pp->bq = h->p_pairs.as<int2>() + q0;
    pp->lpos = h->p_dest.as<unsigned short>() + q0;  // written by the histogram pass, read by the scatter (and the gather)
    hipLaunchKernelGGL(part_hist_kernel, dim3(pp->tgrid), dim3(PT_THREADS), 0, st, qs, nq, h->geom, pp->table, pp->ntiles,
                       pp->lpos, unsorted);
    hipLaunchKernelGGL(part_colsum_kernel, dim3(nrb), dim3(PT_THREADS), 0, st, pp->table, pp->ntiles, rows_per_block, partial, gate);
    hipLaunchKernelGGL(part_colbase_kernel, dim3(1), dim3(PT_THREADS), 0, st, partial, nrb, nq, pp->plan, gate);
    hipLaunchKernelGGL(part_colscan_kernel, dim3(nrb), dim3(PT_THREADS), 0, st, pp->table, pp->ntiles, rows_per_block, partial, gate);
    BXMI_LAUNCH_CHECK();
    const size_t scat_lds = (size_t)PT_TILE * 8 + PT_NB * sizeof(unsigned);
    hipLaunchKernelGGL(part_scatter_kernel, dim3(pp->tgrid), dim3(PT_THREADS), scat_lds, st, qs, qe, nq, h->geom, pp->table, pp->ntiles, pp->bq,
                       pp->lpos, gate);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Scratch for bucketing a batch of nq queries (grow-only).
static int part_reserve(bxmi_ivl *h, int64_t nq, bool want_lpos)
{
    BXMI_TRY(h->p_pairs.reserve((size_t)(nq + 4) * 8));
    BXMI_TRY(h->p_plan.reserve((size_t)PT_MAX_SUB * (PT_NB + 8) * sizeof(int32_t)));
    BXMI_TRY(h->p_slots.reserve((size_t)PT_MAX_SUB * PT_SLOT_STRIDE * sizeof(unsigned long long)));
    BXMI_TRY(h->p_table.reserve((size_t)(div_up(nq, PT_TILE) + PT_MAX_SUB) * PT_NB * sizeof(unsigned)));
    BXMI_TRY(h->p_hist.reserve((size_t)PT_MAX_SUB * 80 * PT_NB * sizeof(unsigned)));
    BXMI_TRY(h->p_dest.reserve((size_t)(nq + 8) * 2));
    if (want_lpos) BXMI_TRY(h->p_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(allow_big_lds(part_scatter_kernel, (size_t)PT_TILE * 8 + PT_NB * sizeof(unsigned)));
    return BXMI_OK;
}

// One sub-batch of the partitioned count, all on stream `st`.
static int ivl_count_part_sub(bxmi_ivl *h, int sub, int64_t q0, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts,
                              int64_t *total_dev, hipStream_t st)
{
    PartPlan pp;
    // [PT_SLOTS partial totals][flag: 1 = the starts are NOT sorted], zeroed together
    unsigned long long *slots = h->p_slots.as<unsigned long long>() + (int64_t)sub * PT_SLOT_STRIDE;
    unsigned *unsorted = g_opt_sorted_path ? reinterpret_cast<unsigned *>(slots + PT_SLOTS) : nullptr;
    BXMI_HIP(hipMemsetAsync(slots, 0, PT_SLOT_STRIDE * sizeof(unsigned long long), st));
    BXMI_TRY(part_prepare(h, sub, q0, qs, qe, nq, counts != nullptr, st, &pp, unsorted, true));
    if (unsorted) {
        // sorted batch: one pass over the queries as they lie (exits at once otherwise)
        TreeDev S = h->treeS.dev, E = h->treeE.dev;
        S.lds_from = S.nlev, S.lds_ints = 0, E.lds_from = E.nlev, E.lds_ints = 0;  // walk the global levels only
        hipLaunchKernelGGL(ivl_local_count_kernel, dim3((unsigned)div_up(nq, LC_CHUNK)), dim3(LC_THREADS), 0, st, S, E, index_dev(h),
                           h->e_sorted.as<int32_t>(), qs, qe, nq, counts, total_dev ? slots : nullptr, unsorted);
    }
    const unsigned grid = (unsigned)(div_up(nq, PT_CHUNK) + PT_NB);
    unsigned short *cnt16 = counts ? h->p_cnt.as<unsigned short>() + q0 : nullptr;  // counts in bucket order, 16 bits + escape
    hipLaunchKernelGGL(part_count_cells_kernel<unsigned short>, dim3(grid), dim3(PT_THREADS), (size_t)PT_LDS_INTS * 4, st, index_dev(h),
                       h->e_sorted.as<int32_t>(), h->slice_bounds.as<SliceBound>(), h->cell_images.as<int32_t>(),
                       h->cell_meta.as<CellsMeta>(), pp.plan, pp.table, pp.bq, nq, h->geom, cnt16, total_dev ? slots : nullptr, unsorted);
    BXMI_LAUNCH_CHECK();
    if (counts) {
        hipLaunchKernelGGL(part_gather_kernel<unsigned short>, dim3(pp.tgrid), dim3(PT_THREADS), 0, st, cnt16, pp.lpos, pp.table, pp.ntiles, nq,
                           counts, unsorted, index_dev(h), h->e_sorted.as<int32_t>(), qs, qe);
        BXMI_LAUNCH_CHECK();
    }
    if (total_dev) {
        hipLaunchKernelGGL(part_fold_total_kernel, dim3(1), dim3(64), 0, st, slots, reinterpret_cast<unsigned long long *>(total_dev));
        BXMI_LAUNCH_CHECK();
    }
    return BXMI_OK;
}

// Large-batch count: bucket the queries, search each bucket against LDS-resident slices, gather back.
// (Cutting the batch into sub-batches on forked streams, so that one sub-batch's search overlaps the next one's scatter,
// measured -2 % at depth 2 and +10 % at depth 4 and was removed.)
static int ivl_count_partitioned(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts,
                                 int64_t *total_dev, hipStream_t st)
{
    if (nq >= ((int64_t)1 << 31)) return fail(BXMI_EINVAL, "bxmi_ivl_count: more than 2^31 queries in one batch");
    BXMI_TRY(part_reserve(h, nq, counts != nullptr));
    BXMI_TRY(allow_big_lds(part_count_cells_kernel<unsigned short>, (size_t)PT_LDS_INTS * 4));
    if (!h->images_ready) {
        // LDS images of every bucket for the search (159 MB, a property of the sealed index): built by the first large
        // batch, so the many small per-chromosome trees of the drop-in classes never pay for them
        BXMI_TRY(h->cell_images.reserve((size_t)PT_NB * PT_LDS_INTS * sizeof(int32_t)));
        BXMI_TRY(h->cell_meta.reserve(PT_NB * sizeof(CellsMeta)));
        BXMI_TRY(allow_big_lds(part_cells_image_kernel, (size_t)PT_LDS_INTS * 4));
        hipLaunchKernelGGL(part_cells_image_kernel, dim3(PT_NB), dim3(PT_THREADS), (size_t)PT_LDS_INTS * 4, st, index_dev(h),
                           h->e_sorted.as<int32_t>(), h->slice_bounds.as<SliceBound>(), h->geom, h->cell_images.as<int32_t>(),
                           h->cell_meta.as<CellsMeta>());
        BXMI_LAUNCH_CHECK();
        h->images_ready = true;
    }
    return ivl_count_part_sub(h, 0, 0, qs, qe, nq, counts, total_dev, st);
}


// (end, insertion index) pairs in start order (count_slices.hpp: sl_pack_eid_kernel), once per sealed index.
static int sl_ensure_eid(bxmi_ivl *h, hipStream_t st)
{
    if (h->sl_eid_ready) return BXMI_OK;
    BXMI_TRY(h->sl_eid.reserve(((size_t)h->n + SL_WALK) * sizeof(int2)));
    hipLaunchKernelGGL(sl_pack_eid_kernel, dim3((unsigned)div_up(h->n + SL_WALK, 256)), dim3(256), 0, st, h->e_ord.as<int32_t>(), h->idx.as<int32_t>(),
                       (int)h->n, h->sl_eid.as<int2>());
    BXMI_LAUNCH_CHECK();
    h->sl_eid_ready = true;
    return BXMI_OK;
}

// find() on a batch with sorted starts: windows in query order -> scan -> fill, no bucketing.
// `gate`: the word the order check -- launched just before, not waited for -- raises when the starts descend
// somewhere; every kernel of the chain stands down on it, and *stood_down tells the caller to take the exchange instead.
static int ivl_find_local(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits, int64_t cap,
                          int64_t *total_host, hipStream_t st, const unsigned *gate = nullptr, bool *stood_down = nullptr)
{
    BXMI_TRY(h->p_lo.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->p_hi.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    // (hi, count) from the sorted-batch count kernel, the offsets from one scan over the chunks' totals, then a walk down the pairs
    const bool chunk_scan = ((uintptr_t)offsets & 15) == 0;  // (lf_offsets_kernel stores 16 bytes at a time)
    {
        TreeDev S = h->treeS.dev, E = h->treeE.dev;
        S.lds_from = S.nlev, S.lds_ints = 0, E.lds_from = E.nlev, E.lds_ints = 0;  // walk the global levels only
        const int64_t nchunks = div_up(nq, LC_CHUNK);
        if (chunk_scan) BXMI_TRY(h->lf_state.reserve((size_t)(2 * nchunks + 4) * 8));
        hipLaunchKernelGGL(ivl_local_count_kernel, dim3((unsigned)nchunks), dim3(LC_THREADS), 0, st, S, E, index_dev(h),
                           h->e_sorted.as<int32_t>(), qs, qe, nq, h->q_cnt.as<int32_t>(), (unsigned long long *)nullptr, gate,
                           h->p_hi.as<int32_t>(), (unsigned long long *)nullptr, 0ull, chunk_scan ? h->lf_state.as<unsigned long long>() : nullptr);
        if (chunk_scan) {
            long long *chunk_base = h->lf_state.as<long long>() + nchunks;  // [nchunks + 1]
            hipLaunchKernelGGL(fx_tile_scan_kernel, dim3(1), dim3(1024), 0, st, h->lf_state.as<unsigned long long>(), nchunks, chunk_base,
                               reinterpret_cast<long long *>(offsets) + nq, (long long *)nullptr, gate);
            hipLaunchKernelGGL(lf_offsets_kernel, dim3((unsigned)nchunks), dim3(LC_THREADS), 0, st, h->q_cnt.as<int32_t>(), chunk_base, nq,
                               reinterpret_cast<long long *>(offsets), gate);
        }
    }
    BXMI_LAUNCH_CHECK();
    if (!chunk_scan)
        BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                               reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
    int64_t total = 0;
    // The fill is launched BEHIND the offsets without waiting for the host to learn the total (a round trip of ~25 us with the
    // device idle): it reads the total itself and leaves the list alone when it does not fit the caller's buffer.
    if (hits && cap > 0) {
        BXMI_TRY(sl_ensure_eid(h, st));
        BXMI_TRY(h->fx_work.reserve(128));
        hipLaunchKernelGGL(part_fill_pipe_kernel, dim3(device_props().cus * 8), dim3(FIND_THREADS), 0, st, h->sl_eid.as<int2>() + SL_WALK, qs, nq,
                           h->p_hi.as<int32_t>(), h->q_cnt.as<int32_t>(), reinterpret_cast<const long long *>(offsets), hits, (long long)cap,
                           h->fx_work.as<int32_t>() + 12, gate);
        BXMI_LAUNCH_CHECK();
    }
    unsigned gate_host = 0;
    if (gate) BXMI_HIP(hipMemcpyAsync(&gate_host, gate, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (gate_host) {  // (the starts descend somewhere: nothing was written that the exchange does not write again)
        if (stood_down) *stood_down = true;
        return BXMI_OK;
    }
    if (total_host) *total_host = total;
    if (total > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    return BXMI_OK;
}

// Partitioned find(): same bucketing as the count path, then window+count per query in bucket order, counts gathered
// back for the CSR offsets, offsets carried to bucket order, hits written from bucket order.
static int ivl_find_partitioned(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits,
                                int64_t cap, int64_t *total_host, hipStream_t st)
{
    if (nq >= ((int64_t)1 << 31)) return fail(BXMI_EINVAL, "bxmi_ivl_find: more than 2^31 queries in one batch");
    BXMI_TRY(part_reserve(h, nq, true));
    BXMI_TRY(h->p_lo.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->p_hi.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->p_boffs.reserve((size_t)(nq + 4) * 8));
    PartPlan pp;
    unsigned *unsorted = reinterpret_cast<unsigned *>(h->p_slots.as<unsigned long long>() + PT_SLOTS);  // hint only on this path
    BXMI_HIP(hipMemsetAsync(unsorted, 0, sizeof(unsigned), st));
    BXMI_TRY(part_prepare(h, 0, 0, qs, qe, nq, true, st, &pp, unsorted, false));
    const int64_t ntiles = pp.ntiles;
    const unsigned tgrid = pp.tgrid;
    unsigned *table = pp.table;
    unsigned short *lpos = pp.lpos;
    const size_t lds_bytes = (size_t)PT_LDS_INTS * 4;
    BXMI_TRY(allow_big_lds(part_window_kernel, lds_bytes));
    const unsigned grid = (unsigned)(div_up(nq, PT_CHUNK) + PT_NB);
    hipLaunchKernelGGL(part_window_kernel, dim3(grid), dim3(PT_THREADS), lds_bytes, st, index_dev(h), h->slice_bounds.as<SliceBound>(), pp.plan,
                       table, pp.bq, nq, h->p_lo.as<int32_t>(), h->p_hi.as<int32_t>(), h->p_cnt.as<int32_t>());
    hipLaunchKernelGGL(part_gather_kernel<int32_t>, dim3(tgrid), dim3(PT_THREADS), 0, st, h->p_cnt.as<int32_t>(), lpos, table, ntiles, nq,
                       h->q_cnt.as<int32_t>(), (const unsigned *)nullptr, index_dev(h), (const int32_t *)nullptr, (const int32_t *)nullptr,
                       (const int32_t *)nullptr);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                           reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
    int64_t total = 0;
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    if (total > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    if (total == 0) return BXMI_OK;
    const size_t perm_lds = (size_t)PT_TILE * 8 + (PT_NB + 2) * 2 + PT_NB * 4 + 64;
    BXMI_TRY(allow_big_lds(part_permute_i64_kernel, perm_lds));
    hipLaunchKernelGGL(part_permute_i64_kernel, dim3(tgrid), dim3(PT_THREADS), perm_lds, st, reinterpret_cast<const long long *>(offsets), lpos,
                       table, ntiles, nq, h->p_boffs.as<long long>());
    int fgrid = device_props().cus * 8;
    hipLaunchKernelGGL(part_fill_kernel, dim3(fgrid), dim3(FIND_THREADS), 0, st, index_dev(h), reinterpret_cast<const int32_t *>(pp.bq), 2, nq,
                       h->p_lo.as<int32_t>(),
                       h->p_hi.as<int32_t>(), h->p_cnt.as<int32_t>(), h->p_boffs.as<long long>(), hits);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// ---- the large-batch count pass (count_bitmap.hpp: tile sort, run table, plan, un-permute; its search stages in count_dense.hpp / count_slices.hpp) ----
// Cell images of 2^18-coordinate units for the flat walk (count_dense.hpp, bp_*): built once per sealed index.
static int bp_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->bp_state = -1;
    const int shift = h->geom.shift;
    // (units of at least two buckets: the 1024-thread tile sorts can then start every unit's run on a whole 16-byte slot)
    if (h->has_reversed || h->n < 4096 || shift > BP_UNIT_LOG2 - 1 || shift < BM_MIN_SHIFT) return BXMI_OK;
    BmGeom g;
    g.cmin = h->geom.cmin;
    g.cmax = h->cmax;
    g.shift = shift;
    const int f = BP_UNIT_LOG2 - shift;
    g.f = f > BD_MAX_F ? BD_MAX_F : f;
    g.rshift = BP_RSHIFT;
    g.dshift = 0;
    const BpLayout L = bp_layout(g.shift + g.f);
    g.nce = L.nce, g.ncs = L.ncs;
    g.stride = L.bytes >> 4;
    const int units = BM_NB >> g.f;
    BXMI_TRY(h->bp_images.reserve((size_t)units * L.bytes));
    BXMI_TRY(h->bp_stats.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->bp_stats.p, 0, 64, st));
    const size_t lds = (size_t)4 * L.ncs * sizeof(int32_t);
    BXMI_TRY(allow_big_lds(bp_image_kernel, lds));
    hipLaunchKernelGGL(bp_image_kernel, dim3((unsigned)units), dim3(BD_THREADS), lds, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(), (int)h->n, g,
                       h->bp_images.as<unsigned char>(), h->bp_stats.as<unsigned>());
    BXMI_LAUNCH_CHECK();
    unsigned stats[2] = {0, 0};
    BXMI_HIP(hipMemcpyAsync(stats, h->bp_stats.p, sizeof(stats), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->bp_geom = g;
    h->bp_hard_cells = stats[0];
    // cells that queries can land in: the span of the index, twice (ends and starts)
    const int64_t cells = 2 * ((((int64_t)h->cmax - (int64_t)h->geom.cmin) >> 5) + 1);
    if (stats[1] == 0 && (int64_t)stats[0] * 1000000 <= cells * g_opt_bm_hard_ppm) h->bp_state = 1;
    return BXMI_OK;
}

// Offset-cell images (offset_cells.hpp) for a sparse index: the cell width from the density, units of 4096 cells (fewer when the
// span is small: a unit is at most 2^BD_MAX_F buckets), built once per sealed index.
// Offset-cell images of an index, once per sealed index.  Two layouts:
//   standard  sparse indexes (about one key per cell): 72 KB per unit, hard cells (> 5 keys) rare and kept as lists -- bo_state = 1;
//   clumped   (round 6) duplicate-heavy indexes that bitmap cells refuse: cells of 64 coordinates, EVERY hard cell with a rank
//             table, the overflow area as large as one CU's LDS allows (a unit image of 147 KB, one 1024-thread workgroup per
//             CU) -- bo_state = 2 when every hard cell found room.
static int bo_prepare_index(bxmi_ivl *h, hipStream_t st, bool clumped = false)
{
    h->bo_state = -1;
    const int shift = h->geom.shift;
    int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin + 1;
    const int k = g_opt_bo_cell_log2 ? (int)g_opt_bo_cell_log2 : (clumped ? BO_MIN_K : bo_cell_log2_for(span, h->n));
    // (units of at least two buckets, as for bitmap cells: the tile sort can then start every unit's run on a whole slot)
    if (h->has_reversed || h->n < 4096 || k == 0 || shift > bo_rshift(k) - 1 || shift < BM_MIN_SHIFT) return BXMI_OK;
    BmGeom g;
    g.cmin = h->geom.cmin;
    g.cmax = h->cmax;
    g.shift = shift;
    const int f = bo_rshift(k) - shift;
    g.f = f > BD_MAX_F ? BD_MAX_F : f;
    g.rshift = bo_rshift(k);
    g.dshift = k - 5;
    const int big16 = BW_PF * BD_THREADS;  // what the 1024-thread walk loads: 9 x 1024 pieces of 16 bytes
    const BpLayout L = bp_layout(g.shift + g.f, k, clumped ? big16 : 0);
    if (clumped && L.ov + 4096 > L.bytes) return BXMI_OK;  // (no room for tables worth the name)
    g.nce = L.nce, g.ncs = L.ncs;
    g.stride = L.bytes >> 4;
    const int units = BM_NB >> g.f;
    BXMI_TRY(h->bo_images.reserve((size_t)units * L.bytes));
    BXMI_TRY(h->bp_stats.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->bp_stats.p, 0, 64, st));
    const size_t lds = (size_t)L.ncs * sizeof(int32_t);
    if (clumped) {
        BXMI_TRY(allow_big_lds(bo_image_kernel<true>, lds));
        hipLaunchKernelGGL(bo_image_kernel<true>, dim3((unsigned)units), dim3(BD_THREADS), lds, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(), (int)h->n, g,
                           h->bo_images.as<unsigned char>(), h->bp_stats.as<unsigned>());
    } else {
        BXMI_TRY(allow_big_lds(bo_image_kernel<false>, lds));
        hipLaunchKernelGGL(bo_image_kernel<false>, dim3((unsigned)units), dim3(BD_THREADS), lds, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(), (int)h->n, g,
                           h->bo_images.as<unsigned char>(), h->bp_stats.as<unsigned>());
    }
    BXMI_LAUNCH_CHECK();
    unsigned stats[3] = {0, 0, 0};
    BXMI_HIP(hipMemcpyAsync(stats, h->bp_stats.p, sizeof(stats), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->bo_geom = g;
    h->bo_hard_cells = stats[0];
    const int64_t cells = 2 * ((span >> k) + 1);  // cells that queries can land in: the span of the index, ends and starts
    if (clumped) {
        if (stats[1] == 0 && stats[2] == 0) h->bo_state = 2;
    } else if (stats[1] == 0 && (int64_t)stats[0] * 1000000 <= cells * g_opt_bm_hard_ppm)
        h->bo_state = 1;
    return BXMI_OK;
}

// Dense unit images (count_dense.hpp): built once per sealed index; the kernel reports whether the index fits the format.
static int bd_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->bd_state = -1;
    const int shift = h->geom.shift;
    if (h->has_reversed || h->n < 4096 || shift > BD_MAX_SHIFT || shift < BM_MIN_SHIFT) return BXMI_OK;
    // Units of 2^19 coordinates first (runs twice as long, 12 KiB of LDS for duplicated coordinates: enough for an index
    // whose duplicates are accidents), then units of 2^18 (64 KiB: rank tables for clumped cells).
    int tried = -1;  // unit width (log2) of the geometry tried last
    for (int ulog = g_opt_bd_unit_log2 ? (int)g_opt_bd_unit_log2 : BD_UNIT_LOG2; ulog >= 18 || ulog == (int)g_opt_bd_unit_log2; ulog--) {
        BmGeom g;
        g.cmin = h->geom.cmin;
        g.cmax = h->cmax;
        g.shift = shift;
        const int f = ulog - shift;
        g.f = f < 0 ? 0 : (f > BD_MAX_F ? BD_MAX_F : f);
        if (g.shift + g.f == tried) break;  // (buckets wider than the unit asked for: f clamps to 0, the same images again)
        tried = g.shift + g.f;
        g.rshift = BD_RSHIFT;
        g.dshift = 0;
        const BdLayout L = bd_layout(g.shift + g.f);
        g.nce = L.nce, g.ncs = L.ncs;
        g.stride = L.bytes >> 4;
        const int units = BM_NB >> g.f;
        BXMI_TRY(h->bd_images.reserve((size_t)units * L.bytes));
        BXMI_TRY(h->bd_stats.reserve(64));
        const size_t lds = (size_t)8 * L.ncs * sizeof(int32_t);
        BXMI_TRY(allow_big_lds(bd_image_kernel, lds));
        h->bd_geom = g;
        // ranks relative to the whole unit when every slice holds fewer than 2^15 keys (no table read per lookup), else
        // relative to blocks of 1024 cells
        // rank tables for every cell with two duplicated coordinates where the overflow area has the room, else from six (count_dense.hpp)
        const int tf_first = g_opt_bd_table_from ? (int)g_opt_bd_table_from : BD_TABLE_FROM_FIRST;
        const int tf_last = g_opt_bd_table_from ? (int)g_opt_bd_table_from : BD_TABLE_FROM_LAST;
        for (int table_from = tf_first; table_from <= tf_last; table_from += BD_TABLE_FROM_LAST - BD_TABLE_FROM_FIRST)
        for (int bshift = g_opt_bd_blocks ? 10 : 13; bshift >= 10; bshift -= 3) {
            BXMI_HIP(hipMemsetAsync(h->bd_stats.p, 0, 64, st));
            hipLaunchKernelGGL(bd_image_kernel, dim3((unsigned)units), dim3(BD_THREADS), lds, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(),
                               (int)h->n, g, bshift, h->bd_images.as<unsigned char>(), h->bd_stats.as<unsigned>(), table_from);
            BXMI_LAUNCH_CHECK();
            BXMI_HIP(hipMemcpyAsync(h->bd_worst, h->bd_stats.p, sizeof(h->bd_worst), hipMemcpyDeviceToHost, st));
            BXMI_HIP(hipStreamSynchronize(st));
            h->bd_blocks = bshift == 10;
            if (h->bd_worst[1] > (unsigned)L.ov_cap) break;  // too many duplicated coordinates: blocks do not help
            if (h->bd_worst[0] <= 32767u) {
                h->bd_state = 1;
                return BXMI_OK;
            }
        }
        if (g_opt_bd_unit_log2 || g.shift + g.f < ulog) break;  // forced, or the span is so small that the unit cannot shrink with ulog
    }
    return BXMI_OK;
}

// Slice search: the ranks at every bucket boundary, and how many keys a unit of 2^f buckets would have to stage.
static int sl_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->sl_state = -1;
    if (h->has_reversed || h->n < 1) return BXMI_OK;
    BXMI_TRY(h->sl_meta.reserve((size_t)(BM_NB + 1) * sizeof(int4)));
    BXMI_TRY(h->sl_stats.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->sl_stats.p, 0, 64, st));
    hipLaunchKernelGGL(sl_meta_kernel, dim3((BM_NB + 1 + 255) / 256), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(), (int)h->n,
                       h->geom.cmin, h->geom.shift, h->sl_meta.as<int4>());
    hipLaunchKernelGGL(sl_fit_kernel, dim3(1), dim3(1024), 0, st, h->sl_meta.as<int4>(), h->sl_stats.as<unsigned>());
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipMemcpyAsync(h->sl_need, h->sl_stats.p, sizeof(h->sl_need), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (h->sl_need[0] <= (unsigned)SL_CAP) h->sl_state = 1;
    return BXMI_OK;
}

// The slice geometry of one index for a batch with `tile` queries per tile: the unit grows while its keys fit, its
// offsets leave 12 bits for the record's length, and its runs stay short enough for one pass of a wave.
static BmGeom sl_geom(const bxmi_ivl *h, int64_t tile, size_t *lds_bytes, int64_t *run_len)
{
    BmGeom g;
    g.cmin = h->geom.cmin, g.cmax = h->cmax, g.shift = h->geom.shift;
    const int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin;
    const int64_t nb_used = ((span > 0 ? span : 0) >> g.shift) + 1;
    int f = 0;
    if (g_opt_sl_f >= 0) {
        for (f = (int)g_opt_sl_f; f > 0 && h->sl_need[f] > (unsigned)SL_CAP; f--) {}
    } else {
        for (int k = 1; k <= SL_MAX_F; k++) {
            // ... and while a (tile, unit) run of a uniform batch stays within ~2.5 waves: longer runs go through the
            // leftover passes / the workgroup's long-run list, which cost small chromosomes (narrow buckets, f = 5)
            // 15 % of the genome pass
            // (the flat walk keeps every lane busy whatever the run length, but a unit's directory has 2047 cells however
            // wide the unit is: a larger unit means more keys per cell and more halvings per lookup -- measured: the genome
            // pass 1.0 -> 2.4 ms with units as large as the LDS allows -- so the same cap serves both walks)
            if (h->sl_need[k] > (unsigned)SL_CAP || g.shift + k > g_opt_sl_rbits || (tile << k) / nb_used > g_opt_sl_run_cap) break;
            f = k;
        }
    }
    g.f = f;
    g.rshift = g.shift + f > 17 ? g.shift + f : 17;
    const int64_t Wu = (int64_t)1 << (g.shift + f);
    int d = 0;
    while (((Wu + SL_MARGIN) >> d) + 1 > SL_DIR_CELLS) d++;
    g.dshift = d;
    g.ncs = (int32_t)((Wu + SL_MARGIN) >> d) + 1;
    g.nce = (int32_t)(Wu >> d) + 1;
    g.stride = 0;
    *lds_bytes = 2 * ((size_t)h->sl_need[f] + 16) + 2 * (size_t)((g.ncs + 2 + 7) & ~7) + 2 * (size_t)(g.nce + 2 + 8);
    *run_len = (tile << f) / nb_used;
    return g;
}

// Everything the kernels of one batch need, as they are handed to every launch.
struct BmLaunch {
    const BmSeg *segs;             // device: the batch's segments
    const unsigned short *tile_seg;  // device: segment of every tile (padded numbering)
    bxmi_ivl *owner;               // whose scratch the batch uses
    int64_t ntp;                   // tiles in the padded numbering (a multiple of BM_GROUP_TILES)
    int ngroups, tile_log2;
    size_t search_lds;
    const unsigned *gate;
    bool pad = false;  // the units' runs on whole 16-byte slots (bm_tile_sort_kernel<.., PAD>): tile stride TILE + BM_PAD_ROOM
    bool w8 = false;   // 8-bit counts between the search and the un-permute kernel (padded layout, cell images)
    bool wide = false; // the cell images are offset cells (sparse indexes)
    bool big = false;  // ... in the clumped layout (a unit image beyond 80 KB: one 1024-thread workgroup per CU)
    bool tot = false;  // total-only batch on cell images: the walk keeps the totals, no slots, no counts, no un-permute (bw_search_kernel<.., TOT>)
    unsigned long long *tot_slots = nullptr;  // ... its partial totals [segments][PT_SLOTS]
    unsigned *descent = nullptr;  // no order check in this pass: bm_params_kernel's probe raises this word when it sees a descent
    unsigned *xcd_next = nullptr;  // eight item counters of the persistent search, zeroed with the partial totals
    // the parameter block written by the tile sort's first workgroup (count_bitmap.hpp: bm_write_params) instead of bm_params_kernel
    BmSegChunk par;
    int npar = 0;
    BmParOut par_out;
    int n_segs = 1;
};

template <int THREADS, int ITEMS>
static int bm_launch_tiles(const BmLaunch &L, hipStream_t st, bool sub = false)
{
    constexpr int TILE = THREADS * ITEMS;
    bxmi_ivl *h = L.owner;
    if (sub) {  // find(): ordered by half buckets, both tables (find_exchange.hpp)
        if (THREADS != 1024 || L.pad) return fail(BXMI_ESTATE, "bm_launch_tiles: half buckets need a 1024-thread shape on packed runs");
        constexpr int T2 = THREADS == 1024 ? THREADS : 1024;  // (only the 1024-thread shapes instantiate the kernel)
        constexpr int I2 = TILE / T2;
        const size_t lds = (size_t)TILE * 4 + FX_NBK * 4 + FX_NBK * 2 + 64;
        BXMI_TRY(allow_big_lds((bm_tile_sort_kernel<T2, I2, false, 2>), lds));
        hipLaunchKernelGGL((bm_tile_sort_kernel<T2, I2, false, 2>), dim3((unsigned)L.ntp), dim3(T2), lds, st, L.segs, L.tile_seg,
                           h->bm_recs.as<unsigned>(), h->bm_slots.as<unsigned short>(), h->bm_tbl.as<unsigned short>(), L.gate, (unsigned *)nullptr,
                           h->fx_tbl2.as<unsigned short>(), L.par, L.npar, L.par_out);
        BXMI_LAUNCH_CHECK();
        return BXMI_OK;
    }
    if (L.pad && L.tot) {
        const size_t lds = (size_t)(TILE + 3 * THREADS) * 4 + BM_NB * 4 + BM_NB * 2 + 64;
        BXMI_TRY(allow_big_lds((bm_tile_sort_kernel<THREADS, ITEMS, true, 1, true>), lds));
        hipLaunchKernelGGL((bm_tile_sort_kernel<THREADS, ITEMS, true, 1, true>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, L.segs, L.tile_seg,
                           h->bm_recs.as<unsigned>(), h->bm_slots.as<unsigned short>(), h->bm_tbl.as<unsigned short>(), L.gate, h->bd_tend.as<unsigned>(),
                           (unsigned short *)nullptr, L.par, L.npar, L.par_out, h->bm_tesc.as<unsigned>());
    } else if (L.pad) {
        const size_t lds = (size_t)(TILE + 3 * THREADS) * 4 + BM_NB * 4 + BM_NB * 2 + 64;
        BXMI_TRY(allow_big_lds((bm_tile_sort_kernel<THREADS, ITEMS, true>), lds));
        hipLaunchKernelGGL((bm_tile_sort_kernel<THREADS, ITEMS, true>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, L.segs, L.tile_seg,
                           h->bm_recs.as<unsigned>(), h->bm_slots.as<unsigned short>(), h->bm_tbl.as<unsigned short>(), L.gate, h->bd_tend.as<unsigned>(),
                           (unsigned short *)nullptr, L.par, L.npar, L.par_out);
    } else {
        const size_t lds = (size_t)TILE * 4 + BM_NB * 4 + BM_NB * 2 + 64;
        BXMI_TRY(allow_big_lds((bm_tile_sort_kernel<THREADS, ITEMS, false>), lds));
        hipLaunchKernelGGL((bm_tile_sort_kernel<THREADS, ITEMS, false>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, L.segs, L.tile_seg,
                           h->bm_recs.as<unsigned>(), h->bm_slots.as<unsigned short>(), h->bm_tbl.as<unsigned short>(), L.gate, (unsigned *)nullptr,
                           (unsigned short *)nullptr, L.par, L.npar, L.par_out);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

template <int THREADS, int ITEMS>
static int bm_launch_unpermute(const BmLaunch &L, unsigned long long *slots, hipStream_t st, const unsigned *cnt = nullptr, unsigned *loff = nullptr,
                               int fx = 0 /* 0, or FIND = 2 / 3 */)
{
    bxmi_ivl *h = L.owner;
    const size_t lds = (size_t)THREADS * ITEMS * sizeof(unsigned);
    if (!cnt) cnt = h->bm_recs.as<unsigned>();
    if (loff && fx == 3) {
        BXMI_TRY(allow_big_lds((bm_unpermute_kernel<THREADS, ITEMS, 3>), lds));
        hipLaunchKernelGGL((bm_unpermute_kernel<THREADS, ITEMS, 3>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, cnt,
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, loff, h->fx_svq.as<unsigned>(),
                           h->fx_parts.as<unsigned long long>(), h->fx_tile_tot.as<unsigned long long>());
    } else if (loff && fx) {
        BXMI_TRY(allow_big_lds((bm_unpermute_kernel<THREADS, ITEMS, 2>), lds));
        hipLaunchKernelGGL((bm_unpermute_kernel<THREADS, ITEMS, 2>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, cnt,
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, loff, h->fx_svq.as<unsigned>(),
                           h->fx_parts.as<unsigned long long>(), h->fx_tile_tot.as<unsigned long long>());
    } else {
        BXMI_TRY(allow_big_lds((bm_unpermute_kernel<THREADS, ITEMS>), lds));
        hipLaunchKernelGGL((bm_unpermute_kernel<THREADS, ITEMS>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, cnt,
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, (unsigned *)nullptr);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// The bitmap-cell pass over a batch of n segments (n sealed, qualifying indexes with their queries): [order check ->]
// tile sort (with the batch's parameter block) -> run table -> plan -> search -> un-permute -> totals, all on `st`, six launches
// for up to 16 segments (five when nobody asks for totals); with the order check in front or more segments the parameter kernel
// is a launch of its own.
// counts[i] may be NULL (total only: nothing is stored per query); totals_dev[i] may be.  The scratch of hs[0] serves the whole batch.
#ifndef SL_FIND_U
#define SL_FIND_U 2  // runs per lane group and round of find()'s count half
#endif
template <int LANES>
static int sl_launch_search(const BmLaunch &L, unsigned grid, hipStream_t st, unsigned *out, unsigned *hc = nullptr)
{
    bxmi_ivl *h = L.owner;
    if (out != h->bm_recs.as<unsigned>()) {
        BXMI_TRY(allow_big_lds((sl_search_pipe_kernel<LANES, SL_FIND_U, true>), L.search_lds));
        hipLaunchKernelGGL((sl_search_pipe_kernel<LANES, SL_FIND_U, true>), dim3(grid), dim3(SL_THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                           h->bm_items.as<int>(), h->bm_runT.as<unsigned>(), L.ntp, h->bm_recs.as<unsigned>(), out, L.tile_log2, L.gate, hc);
    } else {
        BXMI_TRY(allow_big_lds((sl_search_pipe_kernel<LANES, 2, false>), L.search_lds));
        hipLaunchKernelGGL((sl_search_pipe_kernel<LANES, 2, false>), dim3(grid), dim3(SL_THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                           h->bm_items.as<int>(), h->bm_runT.as<unsigned>(), L.ntp, h->bm_recs.as<unsigned>(), (unsigned *)nullptr, L.tile_log2, L.gate);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// find() through the exchange (count_slices.hpp): what the count half leaves behind for the fill half.
struct BmFindCtx {
    BmLaunch L;      // the batch as launched (items, run table and records stay in the owner's scratch)
    unsigned sgrid;
    int lanes;       // 16 or 64
    int variant;     // tile shape
    bool sub = false;  // in: the tile sort orders by half buckets and everything find_exchange.hpp needs is left behind
    bool direct = false;  // in (with sub): the offsets the un-permute kernel leaves are query-order prefixes (FIND = 3)
    int f = 0;         // out: the unit size the slice geometry picked (2^f buckets)
    int64_t ntiles = 0;
};

static int sl_launch_search_flat(const BmLaunch &L, unsigned grid, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    BXMI_TRY(allow_big_lds((sl_search_flat_kernel<4>), L.search_lds));
    hipLaunchKernelGGL((sl_search_flat_kernel<4>), dim3(grid), dim3(SL_THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                       h->bm_items.as<int>(), h->bm_runT.as<unsigned>(), L.ntp, h->bm_recs.as<unsigned>(), L.tile_log2, L.gate);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

template <int FMT, bool QB, int EXP, int DEPTH, bool PIPE, bool PAD = false, bool W8 = false>
static int bd_launch_search_t(const BmLaunch &L, unsigned grid, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    BXMI_TRY(allow_big_lds((bd_search_kernel<FMT, QB, EXP, DEPTH, PIPE, PAD, W8>), L.search_lds));
    hipLaunchKernelGGL((bd_search_kernel<FMT, QB, EXP, DEPTH, PIPE, PAD, W8>), dim3(grid), dim3(BD_THREADS), L.search_lds, st, L.segs,
                       h->bm_items.as<int4>() + 1, h->bm_items.as<int>(), h->bd_unitT.as<unsigned short>(), L.ntp, h->bm_recs.as<unsigned>(),
                       h->bd_cnt16.as<unsigned short>(), L.tile_log2, L.gate);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// the persistent walk on cell images (count_dense.hpp, bw_*): one workgroup per CU, items handed out per XCD
template <bool W8, bool WIDE, bool TOT = false, bool BIG = false>
static int bw_launch_search(const BmLaunch &L, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    constexpr int THREADS = WIDE && !BIG ? BD_THREADS / 2 : BD_THREADS;  // offset cells: two workgroups per CU (the clumped layout: one)
    constexpr int DEPTH = 3;  // (offset cells with rings of 2 / 3 / 4: genome pass 0.719 / 0.722 / 0.705 ms, an eighth of it 0.150 / 0.150 / 0.152)
    BXMI_TRY(allow_big_lds((bw_search_kernel<W8, DEPTH, WIDE, THREADS, TOT>), L.search_lds));
    hipLaunchKernelGGL((bw_search_kernel<W8, DEPTH, WIDE, THREADS, TOT>), dim3(WIDE && !BIG ? 512 : 256), dim3(THREADS), L.search_lds, st, L.segs, h->bm_items.as<int4>() + 1,
                       h->bm_items.as<int>(), h->bd_unitT.as<unsigned short>(), L.ntp, h->bm_recs.as<unsigned>(), h->bd_cnt16.as<unsigned short>(),
                       L.tile_log2, L.gate, L.xcd_next, L.tot_slots);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// One shape per stage and layout (round 3 kept every ring depth, both pipelines and the diagnostics behind knobs: 66 kernels):
//   cell images   always on padded runs (bp_prepare_index only takes geometries the tile sort can pad): the persistent walk;
//   dense images  padded runs: the ring of three hand-issued loads; packed runs (units of a single bucket): two sets of four;
//   key slices    the lean shape (two passes per round, compiler-issued loads: 64 registers) -- two workgroups share a CU when
//                 the units are small, one stages its unit while the other searches (a third of a sparse index's search time).
static int bd_launch_search(const BmLaunch &L, unsigned grid, int fmt /* 0 dense, 1 cells, 2 slices */, bool blocks, hipStream_t st)
{
    if (fmt == 1) {
        if (!L.pad) return fail(BXMI_ESTATE, "bd_launch_search: cell images on packed runs");
        if (L.tot && L.big) return bw_launch_search<false, true, true, true>(L, st);
        if (L.tot) return L.wide ? bw_launch_search<false, true, true>(L, st) : bw_launch_search<false, false, true>(L, st);
        if (L.big) return L.w8 ? bw_launch_search<true, true, false, true>(L, st) : bw_launch_search<false, true, false, true>(L, st);
        if (L.wide) return L.w8 ? bw_launch_search<true, true>(L, st) : bw_launch_search<false, true>(L, st);
        return L.w8 ? bw_launch_search<true, false>(L, st) : bw_launch_search<false, false>(L, st);
    }
    if (fmt == 2) return bd_launch_search_t<2, false, 0, 2, false>(L, grid, st);  // (never padded: see bm_count_segments)
#ifndef BD_EXP_V
#define BD_EXP_V 0  // diagnostics (compile time, wrong results): 1 = the dense walk without its look-ups
#endif
    if (L.pad) return blocks ? bd_launch_search_t<0, true, BD_EXP_V, 3, true, true>(L, grid, st) : bd_launch_search_t<0, false, BD_EXP_V, 3, true, true>(L, grid, st);
    return blocks ? bd_launch_search_t<0, true, BD_EXP_V, 4, true>(L, grid, st) : bd_launch_search_t<0, false, BD_EXP_V, 4, true>(L, grid, st);
}

template <int THREADS, int ITEMS>
static int bd_launch_unpermute(const BmLaunch &L, unsigned long long *slots, hipStream_t st)
{
    bxmi_ivl *h = L.owner;
    if (L.pad && L.w8) {
        const size_t lds = (size_t)(THREADS * ITEMS + BM_PAD_ROOM);
        BXMI_TRY(allow_big_lds((bd_unpermute_kernel<THREADS, ITEMS, true, true>), lds));
        hipLaunchKernelGGL((bd_unpermute_kernel<THREADS, ITEMS, true, true>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, h->bd_cnt16.as<unsigned short>(),
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, h->bd_tend.as<unsigned>(),
                           h->bd_fb.as<unsigned long long>(), h->bd_fb_host);
    } else if (L.pad) {
        const size_t lds = (size_t)(THREADS * ITEMS + BM_PAD_ROOM) * sizeof(unsigned short);
        BXMI_TRY(allow_big_lds((bd_unpermute_kernel<THREADS, ITEMS, true>), lds));
        hipLaunchKernelGGL((bd_unpermute_kernel<THREADS, ITEMS, true>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, h->bd_cnt16.as<unsigned short>(),
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, h->bd_tend.as<unsigned>());
    } else {
        const size_t lds = (size_t)THREADS * ITEMS * sizeof(unsigned short);
        BXMI_TRY(allow_big_lds((bd_unpermute_kernel<THREADS, ITEMS, false>), lds));
        hipLaunchKernelGGL((bd_unpermute_kernel<THREADS, ITEMS, false>), dim3((unsigned)L.ntp), dim3(THREADS), lds, st, h->bd_cnt16.as<unsigned short>(),
                           h->bm_slots.as<unsigned short>(), L.segs, L.tile_seg, slots, L.gate, (const unsigned *)nullptr);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Two 64-bit words of host memory the kernels of a pass write for the NEXT calls on this handle (never waited for):
// [0] counts that did not fit 8 bits so far, [1] what the latest order check found.
static int ensure_feedback(bxmi_ivl *h, hipStream_t st)
{
    if (h->bd_fb_host) return BXMI_OK;
    BXMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->bd_fb_host), 64, hipHostMallocDefault));
    memset(h->bd_fb_host, 0, 64);
    BXMI_TRY(h->bd_fb.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->bd_fb.p, 0, 64, st));
    return BXMI_OK;
}

// `kind`: what a search workgroup keeps in LDS -- 2 = key slices (count_slices.hpp), 3 = dense unit images, 4 = cell images of
// units (both count_dense.hpp: the flat walk, counts out of place); every index of the batch must have qualified for it.
// (kind 1 -- images of single buckets / bucket pairs, round 2's search -- is gone: an index it served qualifies for the cell
// images of units as well, so no input selected it any more.)
static int bm_count_segments(bxmi_ivl *const *hs, int n, const int32_t *const *qs, const int32_t *const *qe, const int64_t *nq,
                             int32_t *const *counts, int64_t *const *totals_dev, hipStream_t st, int kind, BmFindCtx *fx = nullptr)
{
    if (kind < 2 || kind > 5) return fail(BXMI_EINVAL, "bm_count_segments: no such search stage (%d)", kind);
    const bool wide = kind == 5;  // offset cells: the cell images of sparse indexes
    const bool slices = kind == 2, cells = kind == 4 || wide;
    const bool fxsub = fx && fx->sub;
    // (find() needs 32-bit counts apart from the records and the tile-sorted offsets: the flat walk has that form for find_exchange.hpp only)
    const bool slices_flat = slices && !fx && g_opt_sl_flat != 0;
    const bool dense = kind == 3 || cells || slices_flat /* the flat walk */;
    bxmi_ivl *h = hs[0];
    int64_t nq_all = 0;
    for (int i = 0; i < n; i++) nq_all += nq[i];
    if (nq_all >= ((int64_t)1 << 31)) return fail(BXMI_EINVAL, "bxmi_ivl_count: more than 2^31 queries in one batch");
    if (n > 4096) return fail(BXMI_EINVAL, "bxmi_ivl_count_multi: more than 4096 indexes in one batch");
    // tile shape: 32768-query tiles halve the number of (tile, bucket) runs the search has to fetch, but their sort
    // kernel runs one workgroup per CU and wants a grid of several hundred full tiles
    // (a batch over several indexes: every segment starts on a group of 64 tiles, so the big tiles only where the segments are
    // big too -- a genome of 100 M queries, not its eighth on one of eight GPUs)
    int variant = g_opt_bm_variant >= 0 ? (int)g_opt_bm_variant : (nq_all >= ((int64_t)32 << 20) * (n == 1 ? 1 : 2) ? 2 : 0);
    // cell images are searched on padded runs only: a unit of two buckets needs a tile sort whose threads own two buckets each
    // (the 1024-thread shapes), the 512-thread shape owns four
    // ... and the 1024-thread shape is the faster sort for cell images whatever the unit (a rank's share of a genome, offset cells,
    // f >= 2: 0.413 / 0.241 / 0.138 ms for 50 / 25 / 13 M queries against 0.432 / 0.259 / 0.147 with 512 threads x 32 queries)
    if (cells && variant == 0 && g_opt_bm_variant < 0) variant = 1;
    if (cells && variant == 0)
        for (int i = 0; i < n; i++)
            if ((wide ? hs[i]->bo_geom : hs[i]->bp_geom).f < 2) variant = 1;
    if (fxsub && (n != 1 || !slices)) return fail(BXMI_ESTATE, "bm_count_segments: the half-bucket order serves find() on one index's slices");
    if (fxsub && variant == 0) variant = 1;  // (the half-bucket tile sort has the 1024-thread shapes only)
    const int tile_log2 = variant == 2 ? 15 : 14;
    const int64_t tile = (int64_t)1 << tile_log2;
    // the batch's tile numbering: every segment starts on a plan-group boundary
    std::vector<BmSeg> segs((size_t)n);
    int64_t ntp = 0;
    size_t max_stride = 0, sl_lds = 0;
    int64_t sl_run = INT64_MAX;  // shortest expected (tile, unit) run of the batch
    bool any_total = false, any_blocks = false;
    for (int i = 0; i < n; i++) {
        BmSeg &sg = segs[(size_t)i];
        any_blocks |= kind == 3 && hs[i]->bd_blocks;
        if (slices) {
            size_t lds = 0;
            int64_t run_len = 0;
            sg.g = sl_geom(hs[i], tile, &lds, &run_len);
            if (lds > sl_lds) sl_lds = lds;
            if (run_len < sl_run) sl_run = run_len;
        } else if (cells) {
            sg.g = wide ? hs[i]->bo_geom : hs[i]->bp_geom;
        } else {
            sg.g = hs[i]->bd_geom;
        }
        sg.qs = qs[i], sg.qe = qe[i], sg.counts = counts[i];
        sg.nq = nq[i];
        sg.tile0 = ntp;
        sg.ntiles = div_up(nq[i], tile);
        sg.dimages = hs[i]->bd_images.as<unsigned char>();
        sg.pimages = wide ? hs[i]->bo_images.as<unsigned char>() : hs[i]->bp_images.as<unsigned char>();
        sg.smeta = slices ? hs[i]->sl_meta.as<int4>() : nullptr;
        sg.ix = index_dev(hs[i]);
        sg.e_sorted = hs[i]->e_sorted.as<int32_t>();
        ntp += div_up(sg.ntiles, BM_GROUP_TILES) * BM_GROUP_TILES;
        sg.tile_end = ntp;
        if ((size_t)sg.g.stride > max_stride) max_stride = (size_t)sg.g.stride;
        any_total |= totals_dev && totals_dev[i];
    }
    if (ntp == 0) return BXMI_OK;
    const int ngroups = (int)(ntp / BM_GROUP_TILES);
    // PAIR: a search workgroup holds the images of two neighbouring buckets (needs both in one CU's LDS)
    int chunk = dense ? (g_opt_bd_chunk ? (int)g_opt_bd_chunk : (cells || slices_flat ? 2 : 4) * BM_CHUNK) : g_opt_bm_chunk ? (int)g_opt_bm_chunk : BM_CHUNK;
    // a small batch (one rank's share of a genome on eight GPUs: 13 M queries) cut into items of 128 Ki queries is a hundred
    // workgroups on 256 CUs (measured: search 154 us of a 255 us pass); items of nq / 512, at least a tile
    // (every item stages its unit's keys again: at 25 M queries, 192 items, the smaller items already cost more than
    // the idle CUs did -- 0.33 -> 0.38 ms -- so only batches that leave a third of the chip idle are cut finer)
    if (!(dense ? g_opt_bd_chunk : g_opt_bm_chunk) && !(kind == 3 || cells) && nq_all / chunk < 160) {
        const int64_t c = nq_all / 512;
        chunk = (int)(c < 16384 ? 16384 : c);
    }
    int64_t max_items = (int64_t)n * (BM_NB + 2) + 2 * (nq_all / chunk) + 2;
    if (dense) {  // every segment has at most BM_NB >> f units; empty workgroups of 157 KB of LDS are not free
        // (the padded layout counts up to three more slots per tile and unit as "queries" of the unit)
        int64_t pad_slots = 0;
        for (int i = 0; i < n; i++) pad_slots += 3 * (int64_t)(BM_NB >> segs[(size_t)i].g.f) * (segs[(size_t)i].tile_end - segs[(size_t)i].tile0);
        max_items = 2 * ((nq_all + pad_slots) / chunk) + 2;
        for (int i = 0; i < n; i++) max_items += (BM_NB >> segs[(size_t)i].g.f) + 2;
    }
    // PAD: every unit's run of a tile on whole 16-byte slots (the search's load ring needs one store per pass); the tile
    // sort's scan keeps a unit inside one thread or a few neighbouring lanes
    bool pad = kind == 3 || cells;
    for (int i = 0; i < n && pad; i++) pad = (1 << segs[(size_t)i].g.f) >= (variant == 0 ? 4 : 2);
    const int64_t tile_stride = tile + (pad ? BM_PAD_ROOM : 0);
    // nobody wants counts, only totals, and the persistent walk serves the batch: it keeps the totals itself (bw_search_kernel<.., TOT>)
    bool tot_walk = g_opt_tot_walk != 0 && cells && pad && !fx && any_total;
    for (int i = 0; i < n && tot_walk; i++) tot_walk = counts[i] == nullptr;
    if (tot_walk) BXMI_TRY(h->bm_tesc.reserve((size_t)ntp * 4));
    BXMI_TRY(h->bm_recs.reserve((size_t)ntp * tile_stride * 4));
    if (pad) BXMI_TRY(h->bd_tend.reserve((size_t)ntp * 4));
    BXMI_TRY(h->bm_slots.reserve((size_t)ntp * tile * 2));
    BXMI_TRY(h->bm_tbl.reserve((size_t)ntp * BM_NB * 2));
    if (!dense) BXMI_TRY(h->bm_runT.reserve((size_t)ntp * BM_NB * 4));
    if (dense) BXMI_TRY(h->bd_unitT.reserve((size_t)ntp * (BM_NB + 1) * 2));  // (+ the row behind the last unit)
    BXMI_TRY(h->bm_grpcnt.reserve((size_t)ngroups * BM_NB * 4));
    BXMI_TRY(h->sl_unitcnt.reserve((size_t)ngroups * BM_NB * 4));
    if (dense && !fxsub) BXMI_TRY(h->bd_cnt16.reserve((size_t)ntp * tile_stride * 2));
    BXMI_TRY(h->bm_items.reserve((size_t)(max_items + 2) * sizeof(int4)));  // [0] = the item count, items from [1]
    if (fx) {  // find(): counts apart from the records, and the tile-sorted offsets
        BXMI_TRY(h->sl_cnt.reserve((size_t)ntp * tile * 4));
        BXMI_TRY(h->sl_loff.reserve((size_t)ntp * tile * 4));
    }
    if (fxsub) {  // ... and what find_exchange.hpp's fill and copy read
        BXMI_TRY(h->fx_tbl2.reserve((size_t)ntp * FX_NBK * 2));
        BXMI_TRY(h->fx_runT2.reserve((size_t)ntp * FX_NBK * 4));
        BXMI_TRY(h->fx_hc.reserve((size_t)ntp * tile * 4));
        BXMI_TRY(h->fx_svq.reserve((size_t)ntp * tile * 4));
        BXMI_TRY(h->fx_parts.reserve((size_t)ntp * (tile / BM_PART_Q) * 8));
        BXMI_TRY(h->fx_tile_tot.reserve((size_t)ntp * 8));
        BXMI_TRY(h->fx_tile_base.reserve((size_t)(ntp + 2) * 8));  // (+ the grand total, + the largest tile total)
        fx->f = segs[0].g.f;
        fx->ntiles = segs[0].ntiles;
    }
    unsigned *search_out = fx ? h->sl_cnt.as<unsigned>() : h->bm_recs.as<unsigned>();
    // parameter block in HBM: [segments][totals pointers][tile -> segment], written by bm_params_kernel from its arguments
    const size_t seg_bytes = (size_t)n * sizeof(BmSeg), tot_bytes = (size_t)n * sizeof(void *);
    const size_t tile_off = (seg_bytes + tot_bytes + 15) & ~(size_t)15, par_bytes = tile_off + (size_t)ntp * sizeof(unsigned short);
    BXMI_TRY(h->bm_params.reserve(par_bytes));
    // [segments][PT_SLOTS partial totals], then the flag: 1 = the starts are NOT sorted
    // ([+0] the order flag, [+4 .. +8) the search's item counters)
    BXMI_TRY(h->p_slots.reserve(((size_t)n * PT_SLOTS + 8) * sizeof(unsigned long long)));
    unsigned long long *slots = h->p_slots.as<unsigned long long>();
    // (several indexes: only the walk on cell images has a sorted-batch form over segments)
    bool multi_sorted = n > 1 && !fx && g_opt_sorted_path && g_opt_sorted_cells != 0 && cells && pad;
    for (int i = 0; i < n && multi_sorted; i++) multi_sorted = nq[i] < ((int64_t)1 << 32) - 8;
    unsigned *unsorted = g_opt_sorted_path && (n == 1 || multi_sorted) && !fx ? reinterpret_cast<unsigned *>(slots + (size_t)n * PT_SLOTS) : nullptr;
    // The order check and the stand-down of the sorted-batch kernel cost a shuffled batch 24 us (of 750).  What the order
    // checks find is mirrored into host memory (ivl_local_count_kernel, bs_walk_kernel or the workgroup that runs the probe write it, nobody waits for
    // it): after two batches in a row that were NOT sorted the check is no longer launched -- every kernel of the exchange
    // runs unconditionally -- and a PROBE rides on the parameter kernel instead: 8192 consecutive starts; a descent among
    // them says "shuffled" for certain, none brings the exact check back with the next call.  A sorted batch that arrives
    // in between goes through the exchange (0.78 instead of 0.62 ms per 100 M), exact as ever.  (Watching the order
    // exactly inside the tile sort, which has every start in registers, cost that kernel 13-19 us -- what the check costs.)
    unsigned *descent = nullptr;
    unsigned long long order_seq = 0;
    if (unsorted) {
        BXMI_TRY(ensure_feedback(h, st));
        const unsigned long long seen = reinterpret_cast<volatile unsigned long long *>(h->bd_fb_host)[1];
        if ((seen >> 1) > h->order_seen) {
            h->unsorted_streak = (seen & 1ull) ? h->unsorted_streak + 1 : 0;
            h->order_seen = seen >> 1;
            h->order_skip = h->unsorted_streak >= 2;
        }
        if (h->order_seq == 0 && g_opt_order_skip != 0) {
            // The handle's first large batch: nothing is known about the caller's order yet, and this call has waited for the
            // device already (it built the index's images) -- so the probe is asked alone and its answer read back: a descent
            // among its 8192 starts drops the exact check from this very pass (a cold pass paid 24 us of 700 for it).
            hipLaunchKernelGGL(bm_probe_kernel, dim3(1), dim3(256), 0, st, qs[0], nq[0], unsorted);
            unsigned seen_descent = 0;
            BXMI_HIP(hipMemcpyAsync(&seen_descent, unsorted, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            BXMI_HIP(hipStreamSynchronize(st));
            if (seen_descent) h->unsorted_streak = 2, h->order_skip = true;
        }
        order_seq = ++h->order_seq;
        if (h->order_skip && g_opt_order_skip != 0) descent = unsorted, unsorted = nullptr;  // (the word is zeroed with the partial totals)
    }
    BmLaunch L;
    memset(&L.par, 0, sizeof(L.par));
    // Folded: whenever the tile sort is the batch's first kernel.  With the order check in front (a handle's first batches, sorted
    // input) or more than BM_PAR_CHUNK segments (a whole genome on one GPU) the parameter kernel stays a launch of its own.
    const bool fold_params = n <= BM_PAR_CHUNK && !unsorted;
    const int n_zero = n * PT_SLOTS + 8;
    if (fold_params) {
        for (int i = 0; i < n; i++) {
            L.par.seg[i] = segs[(size_t)i];
            L.par.total[i] = totals_dev ? reinterpret_cast<unsigned long long *>(totals_dev[i]) : nullptr;
        }
        L.npar = n;
        L.par_out.segs = h->bm_params.as<BmSeg>();
        L.par_out.totals = reinterpret_cast<unsigned long long **>(h->bm_params.as<unsigned char>() + seg_bytes);
        L.par_out.tile_seg = reinterpret_cast<unsigned short *>(h->bm_params.as<unsigned char>() + tile_off);
        L.par_out.zero_u64 = slots, L.par_out.n_zero = n_zero;
        L.par_out.n_items = h->bm_items.as<int>();
        L.par_out.probe = descent;
        L.par_out.order_host = descent ? h->bd_fb_host + 1 : nullptr, L.par_out.order_seq = order_seq;
    }
    for (int first = 0; first < n && !fold_params; first += BM_PAR_CHUNK) {
        BmSegChunk c;
        memset(&c, 0, sizeof(c));
        const int cnt = n - first < BM_PAR_CHUNK ? n - first : BM_PAR_CHUNK;
        for (int i = 0; i < cnt; i++) {
            c.seg[i] = segs[(size_t)(first + i)];
            c.total[i] = totals_dev ? reinterpret_cast<unsigned long long *>(totals_dev[first + i]) : nullptr;
        }
        hipLaunchKernelGGL(bm_params_kernel, dim3((unsigned)cnt), dim3(256), 0, st, c, first, h->bm_params.as<BmSeg>(),
                           reinterpret_cast<unsigned long long **>(h->bm_params.as<unsigned char>() + seg_bytes),
                           reinterpret_cast<unsigned short *>(h->bm_params.as<unsigned char>() + tile_off), slots, n_zero,
                           h->bm_items.as<int>(), first == 0 ? descent : (unsigned *)nullptr, descent ? h->bd_fb_host + 1 : (unsigned long long *)nullptr,
                           order_seq);
    }
    BXMI_LAUNCH_CHECK();
    unsigned long long *tslots = any_total ? slots : nullptr;
    L.segs = h->bm_params.as<BmSeg>();
    L.tile_seg = reinterpret_cast<const unsigned short *>(h->bm_params.as<unsigned char>() + tile_off);
    L.owner = h;
    L.ntp = ntp, L.ngroups = ngroups, L.tile_log2 = tile_log2;
    L.search_lds = slices ? sl_lds : max_stride * 16;
    if (slices_flat && L.search_lds < 4096) L.search_lds = 4096;
    L.gate = unsorted;
    L.descent = descent;
    L.pad = pad;
    L.wide = wide;
    L.xcd_next = reinterpret_cast<unsigned *>(slots + (size_t)n * PT_SLOTS + 4);
    L.n_segs = n;
    L.tot = tot_walk, L.tot_slots = slots;
    L.big = wide && max_stride * 16 > (size_t)10 * (BD_THREADS / 2) * 16;  // (beyond what the 512-thread walk loads: 80 KB)
    // 8-bit counts (0xFF = recomputed by the un-permute kernel, exact either way): half the bytes of the second exchange
    // when the counts are small.  Cell images only serve indexes without piled-up coordinates, so the density says what to
    // expect: fewer than 128 targets per 2048 coordinates (configs[1]: 82; a count of 255 needs a query of ~6000).  What the
    // prediction misses -- long queries, targets crowded into part of the span -- the feedback catches: once more than one
    // count in 64 did not fit, the index keeps 16-bit counts (worst case before that: every count recomputed, ~2 x the pass).
    L.w8 = false;
    // (a batch over several indexes -- a genome -- keeps the feedback with its first index: every index has to be sparse enough,
    // none may have switched the narrow counts off)
    if (pad && cells && g_opt_bd_w8 != 0 && !tot_walk) {  // (a total-only walk stores no counts at all)
        BXMI_TRY(ensure_feedback(h, st));
        const unsigned long long wide_counts = *reinterpret_cast<volatile unsigned long long *>(h->bd_fb_host);
        if ((int64_t)wide_counts * 64 > h->w8_queries && wide_counts > 4096) h->w8_off = true;
        bool narrow = true;
        for (int i = 0; i < n; i++) {
            const int64_t span = (int64_t)hs[i]->cmax - (int64_t)hs[i]->geom.cmin + 1;
            narrow = narrow && !hs[i]->w8_off && (int64_t)hs[i]->n * 2048 < span * 128;
            if (wide && hs[i]->bo_state == 2) narrow = false;  // (the clumped layout: hundreds of targets around every hot spot -- its first pass on 8-bit counts recomputed all of them: 24 ms)
        }
        L.w8 = g_opt_bd_w8 > 0 || narrow;
        if (L.w8) h->w8_queries += nq_all;
    }
    if (unsorted) {
        // one index, its batch possibly sorted by start already: one pass over the queries as they lie then, and every
        // kernel below stands down (the local kernel exits at once otherwise)
        // Cell images (bitmap or offset cells): a sorted batch is answered straight from them, stretch by stretch (count_dense.hpp,
        // bs_*): the order check leaves where every unit's queries begin, a plan cuts long stretches, the walk loads a unit's
        // image and answers its queries as they lie.  Other stages keep the first-generation kernel for sorted batches below.
        const bool sorted_on_cells = cells && pad && g_opt_sorted_cells != 0 && nq[0] < ((int64_t)1 << 32) - 8;
        if (multi_sorted) {
            // a batch over several indexes: order check and plan per segment, one walk (count_dense.hpp, bs_*_multi)
            const unsigned chunk = (unsigned)(g_opt_bd_chunk ? g_opt_bd_chunk : (wide ? 1 : 2) * BM_CHUNK);
            size_t max_sorted_items = 4;
            for (int i = 0; i < n; i++) max_sorted_items += (size_t)(BM_NB >> segs[(size_t)i].g.f) + 4 + (size_t)(nq[i] / chunk);
            const size_t bounds_bytes = ((size_t)n * BS_BOUNDS_ROW * 4 + 16 + 15) & ~(size_t)15;
            BXMI_TRY(h->bs_plan.reserve(bounds_bytes + (max_sorted_items + 1) * sizeof(int4)));
            unsigned *bounds_all = h->bs_plan.as<unsigned>();
            int *n_sorted = reinterpret_cast<int *>(bounds_all + (size_t)n * BS_BOUNDS_ROW);
            int4 *sorted_items = reinterpret_cast<int4 *>(h->bs_plan.as<unsigned char>() + bounds_bytes);
            BXMI_HIP(hipMemsetAsync(n_sorted, 0, sizeof(int), st));
            hipLaunchKernelGGL(bs_check_multi_kernel, dim3((unsigned)(ntp < 2048 ? ntp : 2048)), dim3(256), 0, st, L.segs, L.tile_seg, ntp, tile_log2, unsorted, bounds_all);
            hipLaunchKernelGGL(bs_plan_multi_kernel, dim3((unsigned)n), dim3(1024), 0, st, L.segs, bounds_all, chunk, sorted_items, n_sorted, unsorted);
            if (wide && L.big) {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<true, BD_THREADS>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<true, BD_THREADS>), dim3(256), dim3(BD_THREADS), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            } else if (wide) {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<true, BD_THREADS / 2>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<true, BD_THREADS / 2>), dim3(512), dim3(BD_THREADS / 2), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            } else {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<false, BD_THREADS>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<false, BD_THREADS>), dim3(256), dim3(BD_THREADS), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            }
            BXMI_LAUNCH_CHECK();
        } else if (sorted_on_cells) {
            const BmGeom &g0 = segs[0].g;
            const int units = BM_NB >> g0.f;
            const unsigned chunk = (unsigned)(g_opt_bd_chunk ? g_opt_bd_chunk : (wide ? 1 : 2) * BM_CHUNK);
            const size_t max_sorted_items = (size_t)units + 4 + (size_t)(nq[0] / chunk);
            BXMI_TRY(h->bs_plan.reserve((size_t)(units + 2) * 4 + 16 + (max_sorted_items + 1) * sizeof(int4)));
            BmBounds B;
            B.bounds = h->bs_plan.as<unsigned>(), B.cmin = g0.cmin, B.ulog = g0.shift + g0.f, B.units = units;
            int *n_sorted = reinterpret_cast<int *>(B.bounds + units + 2);
            int4 *sorted_items = reinterpret_cast<int4 *>(h->bs_plan.as<unsigned char>() + (((size_t)(units + 2) * 4 + 16 + 15) & ~(size_t)15));
            hipLaunchKernelGGL(bm_sorted_check_kernel<true>, dim3(2048), dim3(256), 0, st, qs[0], nq[0], unsorted, B);
            hipLaunchKernelGGL(bs_plan_kernel, dim3(1), dim3(1024), 0, st, B.bounds, units, (unsigned)nq[0], chunk, sorted_items, n_sorted, unsorted);
            if (wide && L.big) {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<true, BD_THREADS>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<true, BD_THREADS>), dim3(256), dim3(BD_THREADS), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            } else if (wide) {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<true, BD_THREADS / 2>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<true, BD_THREADS / 2>), dim3(512), dim3(BD_THREADS / 2), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            } else {
                BXMI_TRY(allow_big_lds((bs_walk_kernel<false, BD_THREADS>), L.search_lds));
                hipLaunchKernelGGL((bs_walk_kernel<false, BD_THREADS>), dim3(256), dim3(BD_THREADS), L.search_lds, st, L.segs, sorted_items, n_sorted, tslots,
                                   unsorted, L.xcd_next, h->bd_fb_host + 1, order_seq);
            }
            BXMI_LAUNCH_CHECK();
        } else {
        hipLaunchKernelGGL(bm_sorted_check_kernel<false>, dim3(2048), dim3(256), 0, st, qs[0], nq[0], unsorted, BmBounds{nullptr, 0, 0, 0});
        TreeDev S = h->treeS.dev, E = h->treeE.dev;
        S.lds_from = S.nlev, S.lds_ints = 0, E.lds_from = E.nlev, E.lds_ints = 0;  // walk the global levels only
        const int64_t nchunks = div_up(nq[0], LC_CHUNK);
        hipLaunchKernelGGL(ivl_local_count_kernel, dim3((unsigned)nchunks), dim3(LC_THREADS), 0, st, S, E, index_dev(h), h->e_sorted.as<int32_t>(),
                           qs[0], qe[0], nq[0], counts[0], tslots, unsorted, (int32_t *)nullptr, h->bd_fb_host + 1, order_seq);
        BXMI_LAUNCH_CHECK();
        }
    }
    if (variant == 2)
        BXMI_TRY((bm_launch_tiles<1024, 32>(L, st, fxsub)));
    else if (variant == 1)
        BXMI_TRY((bm_launch_tiles<1024, 16>(L, st, fxsub)));
    else
        BXMI_TRY((bm_launch_tiles<512, 32>(L, st)));
    if (tot_walk) {
        // the queries behind escape records, answered from the index tile by tile (none in most batches): behind the tile sort, which
        // flags the tiles and whose first workgroup has zeroed the partial totals.  (On a stream of its own between the tile sort
        // and the fold, beside the run table, the plan and the walk: 0.499 against 0.503 ms per pass -- not worth a second stream.)
        hipLaunchKernelGGL(bm_escape_totals_kernel, dim3((unsigned)ntp), dim3(1024), 0, st, L.segs, L.tile_seg, h->bm_tesc.as<unsigned>(), ntp, tile_log2, slots,
                           unsorted);
        BXMI_LAUNCH_CHECK();
    }
    if (fxsub) {  // the half-bucket run table, half-major (nobody needs its group counts: the fill has its own plan)
        hipLaunchKernelGGL(bm_transpose_kernel<FX_NBK>, dim3((unsigned)ngroups, FX_NBK / 64), dim3(256), 0, st, h->fx_tbl2.as<unsigned short>(), L.segs,
                           L.tile_seg, tile_log2, h->fx_runT2.as<unsigned>(), ntp, (unsigned *)nullptr, unsorted);
        BXMI_LAUNCH_CHECK();
    }
    if (dense) {
        hipLaunchKernelGGL(bd_transpose_kernel, dim3((unsigned)ngroups, BM_NB / 64), dim3(256), 0, st, h->bm_tbl.as<unsigned short>(), L.segs, L.tile_seg,
                           tile_log2, h->bd_unitT.as<unsigned short>(), ntp, h->sl_unitcnt.as<unsigned>(), unsorted,
                           pad ? h->bd_tend.as<unsigned>() : (const unsigned *)nullptr);
        if (n <= BD_PLAN_SEGS)
            hipLaunchKernelGGL(bd_plan_kernel, dim3(1), dim3(1024), 0, st, h->sl_unitcnt.as<unsigned>(), n, L.segs, chunk, h->bm_items.as<int4>() + 1,
                               h->bm_items.as<int>(), unsorted);
        else
            hipLaunchKernelGGL(bm_plan_kernel<2>, dim3(BM_PLAN_BLOCKS), dim3(BM_PLAN_THREADS), 0, st, h->sl_unitcnt.as<unsigned>(), ngroups, L.segs, L.tile_seg,
                               chunk, h->bm_items.as<int4>() + 1, h->bm_items.as<int>(), unsorted);
    } else {
    hipLaunchKernelGGL(bm_transpose_kernel<BM_NB>, dim3((unsigned)ngroups, BM_NB / 64), dim3(256), 0, st, h->bm_tbl.as<unsigned short>(), L.segs, L.tile_seg,
                       tile_log2, h->bm_runT.as<unsigned>(), ntp, h->bm_grpcnt.as<unsigned>(), unsorted);
    hipLaunchKernelGGL(sl_unit_sums_kernel, dim3((unsigned)ngroups), dim3(1024), 0, st, h->bm_grpcnt.as<unsigned>(), L.segs, L.tile_seg,
                       h->sl_unitcnt.as<unsigned>(), unsorted);
    if (n <= BD_PLAN_SEGS)  // (the one-workgroup plan: 11 us where bm_plan_kernel<2> takes 37 on configs[4]'s 1526 tiles)
        hipLaunchKernelGGL(bd_plan_kernel, dim3(1), dim3(1024), 0, st, h->sl_unitcnt.as<unsigned>(), n, L.segs, chunk, h->bm_items.as<int4>() + 1,
                           h->bm_items.as<int>(), unsorted);
    else
        hipLaunchKernelGGL(bm_plan_kernel<2>, dim3(BM_PLAN_BLOCKS), dim3(BM_PLAN_THREADS), 0, st, h->sl_unitcnt.as<unsigned>(), ngroups, L.segs, L.tile_seg,
                           chunk, h->bm_items.as<int4>() + 1, h->bm_items.as<int>(), unsorted);
    }
    BXMI_LAUNCH_CHECK();
    const unsigned sgrid = (unsigned)(div_up(max_items, 8) * 8);
    if (slices_flat)
        BXMI_TRY(bd_launch_search(L, sgrid, 2, false, st));
    else if (slices) {
        // long runs (sparse index, big units): the flat walk; else L lanes per run
        int lanes = g_opt_sl_lanes < 0 ? 0 : (g_opt_sl_lanes ? (int)g_opt_sl_lanes : (sl_run >= 96 ? 0 : (sl_run >= 40 ? 64 : 16)));
        if (fx && lanes == 0) lanes = 64;  // (the fill half has no flat walk)
        if (lanes == 0)
            BXMI_TRY(sl_launch_search_flat(L, sgrid, st));
        else if (lanes == 64)
            BXMI_TRY(sl_launch_search<64>(L, sgrid, st, search_out, fxsub ? h->fx_hc.as<unsigned>() : nullptr));
        else
            BXMI_TRY(sl_launch_search<16>(L, sgrid, st, search_out, fxsub ? h->fx_hc.as<unsigned>() : nullptr));
        if (fx) fx->L = L, fx->sgrid = sgrid, fx->lanes = lanes, fx->variant = variant;
    } else
        BXMI_TRY(bd_launch_search(L, sgrid, cells ? 1 : 0, any_blocks, st));
    unsigned *loff = fx ? h->sl_loff.as<unsigned>() : nullptr;
    if (tot_walk) {
        // (the walk has added every tile's share to the partial totals: nothing to put back into query order)
    } else if (dense && !fxsub) {
        if (variant == 2)
            BXMI_TRY((bd_launch_unpermute<1024, 32>(L, tslots, st)));
        else
            BXMI_TRY((bd_launch_unpermute<1024, 16>(L, tslots, st)));
    } else if (variant == 2)
        BXMI_TRY((bm_launch_unpermute<1024, 32>(L, tslots, st, search_out, loff, fxsub ? (fx->direct ? 3 : 2) : 0)));
    else
        BXMI_TRY((bm_launch_unpermute<1024, 16>(L, tslots, st, search_out, loff, fxsub ? (fx->direct ? 3 : 2) : 0)));
    if (any_total) {
        hipLaunchKernelGGL(bm_fold_totals_kernel, dim3((unsigned)n), dim3(64), 0, st, slots,
                           reinterpret_cast<unsigned long long *const *>(h->bm_params.as<unsigned char>() + seg_bytes));
        BXMI_LAUNCH_CHECK();
    }
    return BXMI_OK;
}

// The ranks at the half-bucket boundaries (find_exchange.hpp), once per sealed index, with a copy on the host: the piece
// lists are cut there.
static int fx_prepare_index(bxmi_ivl *h, hipStream_t st)
{
    h->fx_state = -1;
    if (h->has_reversed || h->n < 1 || h->geom.shift < 1) return BXMI_OK;
    BXMI_TRY(h->fx_meta2.reserve((size_t)(FX_NBK + 1) * sizeof(int2)));
    BXMI_TRY(h->fx_work.reserve(128));  // (eight work counters, a word and sixteen bytes nobody reads)
    hipLaunchKernelGGL(fx_meta_kernel, dim3((FX_NBK + 1 + 255) / 256), dim3(256), 0, st, h->s_ord.as<int32_t>(), (int)h->n, h->geom.cmin, h->geom.shift,
                       h->fx_meta2.as<int2>());
    BXMI_LAUNCH_CHECK();
    h->fx_meta2_host.resize((size_t)FX_NBK + 1);
    BXMI_HIP(hipMemcpyAsync(h->fx_meta2_host.data(), h->fx_meta2.p, (size_t)(FX_NBK + 1) * sizeof(int2), hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->fx_pieces_f = -1;
    h->fx_state = 1;
    return BXMI_OK;
}

// The piece list for units of 2^f buckets: consecutive half buckets of one unit while the ranks their records can reach --
// [#{start < first coordinate} - FX_BACK, #{start < last coordinate + SL_MARGIN}) -- fit one LDS window.  A single half
// bucket that does not fit (a pile) is a piece of its own: the window holds the top of its range, the rest is read from HBM.
// The upload is stream-ordered and the caller synchronises `st` before the next call can change the host list.
static int fx_ensure_pieces(bxmi_ivl *h, int f, hipStream_t st)
{
    if (h->fx_pieces_f == f) return BXMI_OK;
    const std::vector<int2> &m = h->fx_meta2_host;
    std::vector<FxPiece> &out = h->fx_pieces_host;
    out.clear();
    const int per_unit = 1 << (f + 1);
    for (int u0 = 0; u0 < FX_NBK; u0 += per_unit) {
        const int u1 = u0 + per_unit < FX_NBK ? u0 + per_unit : FX_NBK;
        int sb = u0;
        while (sb < u1) {
            const int lo = m[(size_t)sb].x > FX_BACK ? m[(size_t)sb].x - FX_BACK : 0;
            int e = sb + 1;
            while (e < u1 && m[(size_t)e + 1].y - lo <= FX_CAPW) e++;
            FxPiece pc;
            pc.sb0 = sb, pc.sb1 = e;
            pc.whi = m[(size_t)e].y;
            pc.wlo = pc.whi - lo > FX_CAPW ? pc.whi - FX_CAPW : lo;
            out.push_back(pc);
            sb = e;
        }
    }
    BXMI_TRY(h->fx_pieces.reserve(out.size() * sizeof(FxPiece)));
    BXMI_HIP(hipMemcpyAsync(h->fx_pieces.p, out.data(), out.size() * sizeof(FxPiece), hipMemcpyHostToDevice, st));
    h->fx_pieces_f = f;
    return BXMI_OK;
}

// find() through the exchange, second generation (find_exchange.hpp): the count half on a tile order by half buckets, the
// CSR offsets from one scan over the tiles' totals, the fill on LDS windows, the copy that finishes the offsets.
static int ivl_find_fx(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits, int64_t cap,
                       int64_t *total_host, hipStream_t st)
{
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BmFindCtx fx;
    fx.sub = true;
    // Straight into the CSR list, a record's ~20 bytes of hits land anywhere in its tile's 0.6 MB of the list: lines that are
    // completed by other pieces much later.  While the whole list stays in the memory-side cache that is free and the copy is
    // saved (configs[4]'s index, 8 M / 16 M queries: 0.76 / 1.27 ms against 0.84 / 1.37); beyond it every partial line costs HBM
    // a read-modify-write (24 M: equal; 50 M: 3.70 ms against 3.04).  The size of the list is only known after the count half,
    // which has to know the layout -- so the handle's previous batch predicts it (hits per query; 5 before the first).
    {
        const double per_q = h->fx_hits_per_q >= 0.0 ? h->fx_hits_per_q : 5.0;
        fx.direct = g_opt_fx_direct < 0 ? per_q * (double)nq * 4.0 <= 400e6 : g_opt_fx_direct != 0;
    }
    int32_t *counts = h->q_cnt.as<int32_t>();
    int64_t *no_total = nullptr;
    BXMI_TRY(bm_count_segments(&h, 1, &qs, &qe, &nq, &counts, &no_total, st, 2, &fx));
    const int64_t ntp = fx.L.ntp;
    // (only the tiles that hold queries: the un-permute kernel leaves the padding up to the plan group alone)
    hipLaunchKernelGGL(fx_tile_scan_kernel, dim3(1), dim3(1024), 0, st, h->fx_tile_tot.as<unsigned long long>(), fx.ntiles, h->fx_tile_base.as<long long>(),
                       reinterpret_cast<long long *>(offsets) + nq, h->fx_tile_base.as<long long>() + fx.ntiles + 1);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY(fx_ensure_pieces(h, fx.f, st));
    BXMI_HIP(hipMemsetAsync(h->fx_work.p, 0, 64, st));
    if (fx.direct) {  // the offsets are final before the capacity is known to the host; the escapes' hits wait for it on the device
        hipLaunchKernelGGL(fx_offsets_kernel, dim3((unsigned)div_up(nq, 1024)), dim3(256), 0, st, fx.L.segs, h->fx_svq.as<unsigned>(),
                           h->fx_tile_base.as<long long>(), fx.ntiles, fx.L.tile_log2, reinterpret_cast<long long *>(offsets), hits, cap);
        BXMI_LAUNCH_CHECK();
    }
    int64_t total = 0;
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    h->fx_hits_per_q = (double)total / (double)nq;
    if (total >= ((int64_t)1 << 31)) {
        // The scratch offsets inside a tile are 31-bit prefixes: a tile of 16 Ki / 32 Ki queries with 2^31 hits or more (queries
        // that each cover a pile of targets) does not fit them.  Only a list this long can hold such a tile: one more word to read then.
        long long max_tile = 0;
        BXMI_HIP(hipMemcpyAsync(&max_tile, h->fx_tile_base.as<long long>() + fx.ntiles + 1, 8, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipStreamSynchronize(st));
        if (max_tile >= ((long long)1 << 31)) return ivl_find_partitioned(h, qs, qe, nq, offsets, hits, cap, total_host, st);
    }
    if (total > cap && fx.direct) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    if (total > cap) {
        // (the contract: BXMI_ERANGE comes with valid offsets.  The copy kernel, which finishes them, does not run: scan the counts.)
        BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                               reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
        BXMI_HIP(hipStreamSynchronize(st));
        return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    }
    if (total == 0) {  // (the copy kernel writes the offsets: nothing to copy, so they are zeroed here)
        if (!fx.direct) BXMI_HIP(hipMemsetAsync(offsets, 0, (size_t)(nq + 1) * 8, st));
        return BXMI_OK;
    }
    if (!fx.direct) BXMI_TRY(h->sl_hits.reserve((size_t)(total + 16) * 4));
    BXMI_TRY(sl_ensure_eid(h, st));
    const int npieces = (int)h->fx_pieces_host.size();
    // enough (piece, tile chunk) pairs to balance 256 persistent workgroups; a chunk is whole groups of 64 tiles (a wave's batch)
    int64_t nchunks = div_up(2048, npieces);
    if (nchunks > div_up(fx.ntiles, 64)) nchunks = div_up(fx.ntiles, 64);
    if (nchunks < 1) nchunks = 1;
    const int64_t tiles_per_chunk = div_up(div_up(fx.ntiles, nchunks), 64) * 64;
    nchunks = div_up(fx.ntiles, tiles_per_chunk);
    BXMI_TRY(allow_big_lds(fx_fill_kernel, FX_LDS_BYTES));
    hipLaunchKernelGGL(fx_fill_kernel, dim3((unsigned)device_props().cus), dim3(FX_THREADS), FX_LDS_BYTES, st, fx.L.segs, h->fx_pieces.as<FxPiece>(), npieces,
                       (int)nchunks, (int)tiles_per_chunk, h->fx_runT2.as<unsigned>(), ntp, h->bm_recs.as<unsigned>(), h->fx_hc.as<unsigned>(),
                       h->sl_cnt.as<unsigned>(), h->sl_loff.as<unsigned>(), h->fx_tile_base.as<long long>(), h->sl_eid.as<int2>() + SL_WALK,
                       h->fx_meta2.as<int2>(), fx.direct ? hits : h->sl_hits.as<int32_t>(), fx.L.tile_log2, h->fx_work.as<unsigned>(), h->fx_work.as<int32_t>() + 16);
    BXMI_LAUNCH_CHECK();
    if (fx.direct) return BXMI_OK;  // (every record's hits went where the CSR offsets say)
    const unsigned cgrid = (unsigned)(div_up(ntp, 8) * 8 * (((int64_t)1 << fx.L.tile_log2) / BM_PART_Q));
    {  // two consecutive queries per lane (one: 0.91 ms on configs[4], two: 0.78, four: 0.85)
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(cgrid), dim3(BM_PART_Q / 2), 0, st, fx.L.segs, h->fx_svq.as<unsigned>(), h->fx_tile_base.as<long long>(),
                               h->fx_parts.as<unsigned long long>(), h->sl_hits.as<int32_t>(), reinterpret_cast<long long *>(offsets), hits, ntp);
        };
        if (fx.variant == 2) launch(fx_hits_copy2_kernel<32768, 2>); else launch(fx_hits_copy2_kernel<16384, 2>);
    }
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

// Which search stage serves a sealed index in the large-batch pass: 0 = neither (older paths), 1 = bucket images,
// 2 = key slices, 3 = dense unit images (dense indexes try them before the bucket images).  Images cost 0.5 B per coordinate of the span and win on dense indexes; sparse ones (fewer than one
// target per 64 coordinates) and spans whose bucket image outgrows the LDS take slices.  Prepared on first use.
static int bm_choose_stage(bxmi_ivl *h, hipStream_t st, int *kind, int64_t nq)
{
    *kind = 0;
    if (h->has_reversed || h->n < 4096) return BXMI_OK;
    int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin;
    if (span < 0) span = 0;
    // Sparse indexes: offset-cell images on the persistent walk, when the batch brings enough queries per unit image (a unit's
    // image is 72 KB to load however few queries it serves; key slices stage a few KB)
    // (ivl.flat / ivl.dense / ivl.slice set to force or forbid a stage leave this one out, ivl.sparse = 1 forces it)
    if ((g_opt_sparse > 0 || (g_opt_sparse < 0 && g_opt_flat < 0 && g_opt_dense < 1 && g_opt_slice < 1)) &&
        (g_opt_bo_cell_log2 || bo_cell_log2_for(span + 1, h->n))) {
        if (h->bo_state == 0) BXMI_TRY(bo_prepare_index(h, st));
        if (h->bo_state == 1) {
            const int64_t units = (span >> (h->bo_geom.shift + h->bo_geom.f)) + 1;
            if (g_opt_sparse > 0 || nq >= units * g_opt_bo_min_per_unit) {
                *kind = 5;
                return BXMI_OK;
            }
        }
    }
    // duplicate-heavy indexes (bitmap cells refuse them): offset cells with a rank table per hard cell, if the tables fit the LDS
    // (ivl.clumped: -1 = where bitmap cells do not qualify and no knob forces or forbids another stage, 0 = never, 1 = ahead of
    // every other stage but the sparse one)
    auto try_clumped = [&]() -> int {
        if (h->bo_state == 0 || (h->bo_state == -1 && !h->bo_tried_clumped)) {
            h->bo_tried_clumped = true;
            BXMI_TRY(bo_prepare_index(h, st, true));
        }
        if (h->bo_state == 2) *kind = 5;
        return BXMI_OK;
    };
    if (g_opt_clumped > 0 && g_opt_dense != 1) {
        BXMI_TRY(try_clumped());
        if (*kind) return BXMI_OK;
    }
    const bool slices_first = g_opt_dense != 1 && g_opt_flat != 1 && (g_opt_slice == 1 || (g_opt_slice < 0 && (span / h->n >= 64 || h->geom.shift > BD_MAX_SHIFT)));
    if (slices_first) {
        if (h->sl_state == 0) BXMI_TRY(sl_prepare_index(h, st));
        if (h->sl_state == 1) {
            *kind = 2;
            return BXMI_OK;
        }
    }
    if (g_opt_flat != 0) {
        if (h->bp_state == 0) BXMI_TRY(bp_prepare_index(h, st));
        if (h->bp_state == 1) {
            *kind = 4;
            return BXMI_OK;
        }
    }
    if (g_opt_clumped < 0 && g_opt_flat != 0 && h->bp_state == -1 && g_opt_dense < 0 && g_opt_sparse != 0 && g_opt_slice < 1) BXMI_TRY(try_clumped());
    if (*kind) return BXMI_OK;
    if (g_opt_dense != 0) {
        if (h->bd_state == 0) BXMI_TRY(bd_prepare_index(h, st));
        if (h->bd_state == 1) {
            *kind = 3;
            return BXMI_OK;
        }
    }
    if (!slices_first && g_opt_slice != 0) {
        if (h->sl_state == 0) BXMI_TRY(sl_prepare_index(h, st));
        if (h->sl_state == 1) *kind = 2;
    }
    return BXMI_OK;
}

static int ivl_stream(bxmi_ivl *h)
{
    if (!h->stream) BXMI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    return BXMI_OK;
}

extern "C" int bxmi_ivl_create(bxmi_ivl_t **out)
{
    if (!out) return fail(BXMI_EINVAL, "bxmi_ivl_create: out is NULL");
    int dev = 0;
    BXMI_HIP(hipGetDevice(&dev));
    bxmi_ivl *h = new (std::nothrow) bxmi_ivl();
    if (!h) return fail(BXMI_ENOMEM, "bxmi_ivl_create: host allocation failed");
    h->device = dev;
    *out = h;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_destroy(bxmi_ivl_t *h)
{
    if (!h) return BXMI_OK;
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->stream_up) (void)hipStreamDestroy(h->stream_up);
    if (h->stream_down) (void)hipStreamDestroy(h->stream_down);
    if (h->one_buf) (void)hipHostFree(h->one_buf);
    if (h->bd_fb_host) (void)hipHostFree(h->bd_fb_host);
    delete h;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_append(bxmi_ivl_t *h, const int32_t *start, const int32_t *end, int64_t n)
{
    if (!h || n < 0 || (n > 0 && (!start || !end))) return fail(BXMI_EINVAL, "bxmi_ivl_append: bad arguments");
    if ((int64_t)h->h_start.size() + h->n_dev + n >= ((int64_t)1 << 31) - 64)
        return fail(BXMI_EINVAL, "bxmi_ivl_append: more than 2^31 intervals");
    try {
        h->h_start.insert(h->h_start.end(), start, start + n);
        h->h_end.insert(h->h_end.end(), end, end + n);
    } catch (...) {
        return fail(BXMI_ENOMEM, "bxmi_ivl_append: host allocation failed");
    }
    if (n) h->sealed = false;
    return BXMI_OK;
}

// Move host-staged intervals to the device arrays (insertion order preserved).
static int ivl_flush_host(bxmi_ivl *h, hipStream_t st)
{
    int64_t add = (int64_t)h->h_start.size();
    if (!add) return BXMI_OK;
    int64_t tot = h->n_dev + add;
    BXMI_TRY(h->d_start.reserve((size_t)tot * 4, true, st));
    BXMI_TRY(h->d_end.reserve((size_t)tot * 4, true, st));
    BXMI_HIP(hipMemcpyAsync(h->d_start.as<int32_t>() + h->n_dev, h->h_start.data(), (size_t)add * 4, hipMemcpyHostToDevice, st));
    BXMI_HIP(hipMemcpyAsync(h->d_end.as<int32_t>() + h->n_dev, h->h_end.data(), (size_t)add * 4, hipMemcpyHostToDevice, st));
    BXMI_HIP(hipStreamSynchronize(st));
    h->n_dev = tot;
    h->h_start.clear();
    h->h_end.clear();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_append_dev(bxmi_ivl_t *h, const int32_t *start, const int32_t *end, int64_t n, void *stream)
{
    if (!h || n < 0 || (n > 0 && (!start || !end))) return fail(BXMI_EINVAL, "bxmi_ivl_append_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    BXMI_TRY(ivl_flush_host(h, st));
    int64_t tot = h->n_dev + n;
    if (tot >= ((int64_t)1 << 31) - 64) return fail(BXMI_EINVAL, "bxmi_ivl_append_dev: more than 2^31 intervals");
    BXMI_TRY(h->d_start.reserve((size_t)tot * 4, true, st));
    BXMI_TRY(h->d_end.reserve((size_t)tot * 4, true, st));
    BXMI_HIP(hipMemcpyAsync(h->d_start.as<int32_t>() + h->n_dev, start, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    BXMI_HIP(hipMemcpyAsync(h->d_end.as<int32_t>() + h->n_dev, end, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    h->n_dev = tot;
    if (n) h->sealed = false;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_seal(bxmi_ivl_t *h, void *stream)
{
    if (!h) return fail(BXMI_EINVAL, "bxmi_ivl_seal: NULL handle");
    hipStream_t st = as_stream(stream);
    BXMI_TRY(ivl_flush_host(h, st));
    const int64_t n = h->n_dev;
    const int64_t n_pad = (n > 0 ? div_up(n, FAN) : 1) * FAN;
    const size_t pad_bytes = (size_t)(n_pad + FAN) * 4;

    BXMI_TRY(h->keys_a.reserve((size_t)(n + 1) * 8));
    BXMI_TRY(h->keys_b.reserve((size_t)(n + 1) * 8));
    BXMI_TRY(h->ekeys_a.reserve((size_t)(n + 1) * 4));
    BXMI_TRY(h->ekeys_b.reserve((size_t)(n + 1) * 4));
    BXMI_TRY(h->s_ord.reserve(pad_bytes));
    BXMI_TRY(h->e_ord.reserve(pad_bytes));
    BXMI_TRY(h->idx.reserve(pad_bytes));
    BXMI_TRY(h->pm.reserve(pad_bytes));
    BXMI_TRY(h->e_sorted.reserve(pad_bytes));
    BXMI_TRY(h->flag.reserve(64));

    unsigned *d_rev = h->flag.as<unsigned>();
    BXMI_HIP(hipMemsetAsync(d_rev, 0, 4, st));
    const int g = stream_grid(n_pad, 256 * 4);
    if (n > 0) {
        hipLaunchKernelGGL(ivl_make_keys_kernel, dim3(g), dim3(256), 0, st, h->d_start.as<int32_t>(), h->d_end.as<int32_t>(), n,
                           h->keys_a.as<unsigned long long>(), h->ekeys_a.as<uint32_t>(), d_rev);
        BXMI_LAUNCH_CHECK();
    }
    unsigned long long *sorted_keys = nullptr;
    uint32_t *sorted_ends = nullptr;
    BXMI_TRY(radix_sort_keys<unsigned long long>(h->keys_a.as<unsigned long long>(), h->keys_b.as<unsigned long long>(), n,
                                                 &sorted_keys, h->sort_scratch, st));
    BXMI_TRY(radix_sort_keys<uint32_t>(h->ekeys_a.as<uint32_t>(), h->ekeys_b.as<uint32_t>(), n, &sorted_ends, h->sort_scratch, st));
    hipLaunchKernelGGL(ivl_unpack_kernel, dim3(g), dim3(256), 0, st, sorted_keys, h->d_end.as<int32_t>(), n, n_pad,
                       h->s_ord.as<int32_t>(), h->e_ord.as<int32_t>(), h->idx.as<int32_t>());
    hipLaunchKernelGGL(ivl_unbias_kernel, dim3(g), dim3(256), 0, st, sorted_ends, n, n_pad, h->e_sorted.as<int32_t>());
    BXMI_LAUNCH_CHECK();
    // prefix max of ends in tree order; padding = INT_MAX so the pm tree's leaves stay sorted
    BXMI_TRY((device_scan<int32_t, int32_t, OpMax, true>(h->e_ord.as<int32_t>(), h->pm.as<int32_t>(), n, INT_MIN, nullptr,
                                                        h->scan_scratch, st)));
    if (n_pad > n) {
        hipLaunchKernelGGL(ivl_pad_kernel, dim3((unsigned)div_up(n_pad - n, 64)), dim3(64), 0, st, h->pm.as<int32_t>(), n, n_pad, INT_MAX);
        BXMI_LAUNCH_CHECK();
    }
    BXMI_TRY(h->treeS.build(h->s_ord.as<int32_t>(), n, st));
    BXMI_TRY(h->treeE.build(h->e_sorted.as<int32_t>(), n, st));
    BXMI_TRY(h->treeP.build(h->pm.as<int32_t>(), n, st));
    unsigned rev = 0;
    int32_t cmin = 0, cmax = 0;
    BXMI_HIP(hipMemcpyAsync(&rev, d_rev, 4, hipMemcpyDeviceToHost, st));
    if (n > 0) {
        BXMI_HIP(hipMemcpyAsync(&cmin, h->s_ord.as<int32_t>(), 4, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipMemcpyAsync(&cmax, h->e_sorted.as<int32_t>() + (n - 1), 4, hipMemcpyDeviceToHost, st));
    }
    BXMI_HIP(hipStreamSynchronize(st));
    h->has_reversed = rev != 0;
    h->n = n;
    // bucket grid of the partitioned count path: PT_NB buckets of width 2^shift over [min start, max end]
    {
        int64_t span = (int64_t)cmax - (int64_t)cmin;
        if (span < 0) span = 0;
        int shift = 0;
        while ((span >> shift) >= PT_NB) shift++;
        h->geom.cmin = cmin;
        h->geom.shift = shift;
        h->cmax = cmax;
        h->sl_state = 0;
        h->fx_state = 0, h->fx_pieces_f = -1;
        h->bd_state = 0;
        h->bp_state = 0;
        h->bo_state = 0, h->bo_tried_clumped = false;
        h->w8_off = false, h->w8_queries = 0;  // (the feedback of the 8-bit counts belongs to the index that was)
        if (h->bd_fb_host) {
            BXMI_HIP(hipMemsetAsync(h->bd_fb.p, 0, 64, st));
            *h->bd_fb_host = 0;
        }
        h->unsorted_streak = 0, h->order_skip = false;
        h->sl_eid_ready = false;
        BXMI_TRY(h->slice_bounds.reserve(PT_NB * sizeof(SliceBound)));
        hipLaunchKernelGGL(part_bounds_kernel, dim3(PT_NB / 256), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->e_sorted.as<int32_t>(),
                           h->pm.as<int32_t>(), (int)n, h->geom, h->slice_bounds.as<SliceBound>());
        BXMI_LAUNCH_CHECK();
        h->images_ready = false;  // built by the first large batch that needs them
        BXMI_HIP(hipStreamSynchronize(st));
    }
    h->sealed = true;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_size(const bxmi_ivl_t *h, int64_t *n)
{
    if (!h || !n) return fail(BXMI_EINVAL, "bxmi_ivl_size: bad arguments");
    *n = h->n_dev + (int64_t)h->h_start.size();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_has_reversed(const bxmi_ivl_t *h, int *flag)
{
    if (!h || !flag) return fail(BXMI_EINVAL, "bxmi_ivl_has_reversed: bad arguments");
    if (!h->sealed) return fail(BXMI_ESTATE, "bxmi_ivl_has_reversed: index not sealed");
    *flag = h->has_reversed;
    return BXMI_OK;
}

static int need_sealed(const bxmi_ivl *h, const char *who)
{
    if (!h) return fail(BXMI_EINVAL, "%s: NULL handle", who);
    if (!h->sealed) return fail(BXMI_ESTATE, "%s: index not sealed (call bxmi_ivl_seal after appending)", who);
    return BXMI_OK;
}

extern "C" int bxmi_ivl_flat_state(const bxmi_ivl_t *h, int *state, int64_t *hard_cells)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_flat_state"));
    if (state) *state = h->bp_state;
    if (hard_cells) *hard_cells = h->bp_hard_cells;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_sparse_state(const bxmi_ivl_t *h, int *state, int64_t *hard_cells, int *cell_log2)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_sparse_state"));
    if (state) *state = h->bo_state;
    if (hard_cells) *hard_cells = h->bo_hard_cells;
    if (cell_log2) *cell_log2 = h->bo_state >= 1 ? 5 + h->bo_geom.dshift : 0;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_count_width(const bxmi_ivl_t *h, int *bits, int64_t *wide_counts)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_count_width"));
    const unsigned long long wide = h->bd_fb_host ? *reinterpret_cast<volatile unsigned long long *>(h->bd_fb_host) : 0ull;
    const int64_t span = (int64_t)h->cmax - (int64_t)h->geom.cmin + 1;
    if (bits) *bits = !h->w8_off && (int64_t)h->n * 2048 < span * 128 ? 8 : 16;
    if (wide_counts) *wide_counts = (int64_t)wide;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_order_state(const bxmi_ivl_t *h, int *skipping, int64_t *answers_seen)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_order_state"));
    if (skipping) *skipping = h->order_skip && g_opt_order_skip != 0 ? 1 : 0;
    if (answers_seen) *answers_seen = (int64_t)h->order_seen;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_dense_state(const bxmi_ivl_t *h, int *state, int64_t *worst)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_dense_state"));
    if (state) *state = h->bd_state;
    if (worst) worst[0] = h->bd_worst[0], worst[1] = h->bd_worst[1];
    return BXMI_OK;
}

extern "C" int bxmi_ivl_slice_state(const bxmi_ivl_t *h, int *state, int64_t *unit_keys)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_slice_state"));
    if (state) *state = h->sl_state;
    if (unit_keys)
        for (int f = 0; f <= SL_MAX_F; f++) unit_keys[f] = h->sl_need[f];
    return BXMI_OK;
}

extern "C" int bxmi_ivl_order_dev(const bxmi_ivl_t *h, const int32_t **idx_dev, const int32_t **start_dev,
                                  const int32_t **end_dev)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_order_dev"));
    if (idx_dev) *idx_dev = h->idx.as<int32_t>();
    if (start_dev) *start_dev = h->s_ord.as<int32_t>();
    if (end_dev) *end_dev = h->e_ord.as<int32_t>();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_order(const bxmi_ivl_t *h, int32_t *idx_out)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_order"));
    if (h->n && !idx_out) return fail(BXMI_EINVAL, "bxmi_ivl_order: idx_out is NULL");
    if (h->n) BXMI_HIP(hipMemcpy(idx_out, h->idx.p, (size_t)h->n * 4, hipMemcpyDeviceToHost));
    return BXMI_OK;
}

static IndexDev index_dev(const bxmi_ivl *h)
{
    IndexDev ix;
    ix.s_ord = h->s_ord.as<int32_t>();
    ix.e_ord = h->e_ord.as<int32_t>();
    ix.idx = h->idx.as<int32_t>();
    ix.pm = h->pm.as<int32_t>();
    ix.n = (int32_t)h->n;
    ix.has_reversed = h->has_reversed;
    return ix;
}

template <typename Kern>
static int allow_big_lds(Kern k, size_t bytes)
{
    BXMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return BXMI_OK;
}

extern "C" int bxmi_ivl_count_dev(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts,
                                  int64_t *total_dev, void *stream)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_count_dev"));
    if (nq < 0 || (nq > 0 && (!qs || !qe))) return fail(BXMI_EINVAL, "bxmi_ivl_count_dev: bad arguments");
    if (nq == 0) return BXMI_OK;
    if (((uintptr_t)qs | (uintptr_t)qe | (uintptr_t)counts) & 15)
        return fail(BXMI_EINVAL, "bxmi_ivl_count_dev: query/count arrays must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const bool partition = !h->has_reversed && h->n > 0 &&
                           (g_opt_partition == 1 || (g_opt_partition < 0 && nq >= g_opt_partition_min && h->n >= 4096));
    // the bitmap-cell pass pays off earlier than the bucketed one (its fixed cost is one read of the bucket images)
    // (a caller that wants the total only takes the same pass into a scratch array of counts: 0.65 ms per 100 M where round 1's
    // bucketed pass, which can leave the counts out, takes 0.85)
    const bool bitmap = (counts || total_dev) && g_opt_bitmap != 0 && !h->has_reversed && h->n >= 4096 &&
                        (g_opt_partition == 1 || (g_opt_partition < 0 && nq >= g_opt_bitmap_min));
    if (bitmap) {
        int kind = 0;
        BXMI_TRY(bm_choose_stage(h, st, &kind, nq));
        if (kind) {
            // (counts == NULL: the same pass, its un-permute kernel sums without storing -- no scratch array of counts)
            return bm_count_segments(&h, 1, &qs, &qe, &nq, &counts, &total_dev, st, kind);
        }
    }
    if (partition) return ivl_count_partitioned(h, qs, qe, nq, counts, total_dev, st);
    TreeDev S = h->treeS.dev, E = h->treeE.dev;
    size_t lds_bytes = (size_t)(S.lds_ints + E.lds_ints) * 4 + (CNT_THREADS / 64) * sizeof(long long);
    int grid = device_props().cus;
    int64_t need = div_up(nq, (int64_t)(CNT_THREADS / 8) * CNT_Q);
    if (need < grid) grid = (int)need;
    unsigned long long *tot = reinterpret_cast<unsigned long long *>(total_dev);
    BXMI_TRY(allow_big_lds(ivl_count_kernel<true>, lds_bytes));
    hipLaunchKernelGGL(ivl_count_kernel<true>, dim3(grid), dim3(CNT_THREADS), lds_bytes, st, S, E, index_dev(h), qs, qe, nq, counts, tot);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

extern "C" int bxmi_ivl_count_multi_dev(bxmi_ivl_t *const *hs, int n, const int32_t *const *qs, const int32_t *const *qe, const int64_t *nq,
                                        int32_t *const *counts, int64_t *const *totals_dev, void *stream)
{
    if (n < 0 || (n > 0 && (!hs || !qs || !qe || !nq))) return fail(BXMI_EINVAL, "bxmi_ivl_count_multi_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    // indexes whose batch can ride the bitmap-cell pass are answered together (one pass, six launches); the others one by one
    std::vector<bxmi_ivl *> fh[5];  // by search stage (kind - 1): [1] slices, [2] dense unit images, [3] cell images of units, [4] offset cells
    std::vector<const int32_t *> fqs[5], fqe[5];
    std::vector<int64_t> fnq[5];
    std::vector<int32_t *> fc[5];
    std::vector<int64_t *> ft[5];
    std::vector<int> rest;
    int64_t nq_all = 0;
    for (int i = 0; i < n; i++) {
        BXMI_TRY(need_sealed(hs[i], "bxmi_ivl_count_multi_dev"));
        if (nq[i] < 0 || (nq[i] > 0 && (!qs[i] || !qe[i]))) return fail(BXMI_EINVAL, "bxmi_ivl_count_multi_dev: bad arguments for index %d", i);
        if (((uintptr_t)qs[i] | (uintptr_t)qe[i] | (uintptr_t)(counts ? counts[i] : nullptr)) & 15)
            return fail(BXMI_EINVAL, "bxmi_ivl_count_multi_dev: query/count arrays must be 16-byte aligned");
        nq_all += nq[i];
    }
    const bool fused = g_opt_bitmap != 0 && (counts || totals_dev) && (g_opt_partition == 1 || (g_opt_partition < 0 && nq_all >= g_opt_bitmap_min));
    for (int i = 0; i < n; i++) {
        bxmi_ivl *h = hs[i];
        if (nq[i] == 0) continue;
        int kind = 0;
        // a segment occupies whole groups of 64 tiles of scratch whatever its size: in a batch over hundreds of indexes
        // (a scaffold-level assembly) the ones with a handful of queries are answered one by one instead
        const bool tiny = n > 256 && nq[i] < 65536;
        // (an index whose caller wants neither counts nor a total has nothing to compute; total-only segments ride the same pass)
        const bool wanted = (counts && counts[i]) || (totals_dev && totals_dev[i]);
        if (fused && wanted && !tiny) BXMI_TRY(bm_choose_stage(h, st, &kind, nq[i]));
        if (kind) {
            const int k = kind - 1;
            fh[k].push_back(h), fqs[k].push_back(qs[i]), fqe[k].push_back(qe[i]), fnq[k].push_back(nq[i]), fc[k].push_back(counts ? counts[i] : nullptr);
            ft[k].push_back(totals_dev ? totals_dev[i] : nullptr);
        } else {
            rest.push_back(i);
        }
    }
    for (int k = 0; k < 5; k++)
        if (!fh[k].empty())
            BXMI_TRY(bm_count_segments(fh[k].data(), (int)fh[k].size(), fqs[k].data(), fqe[k].data(), fnq[k].data(), fc[k].data(), ft[k].data(), st,
                                       k + 1));
    for (int i : rest)
        BXMI_TRY(bxmi_ivl_count_dev(hs[i], qs[i], qe[i], nq[i], counts ? counts[i] : nullptr, totals_dev ? totals_dev[i] : nullptr, stream));
    return BXMI_OK;
}

static int upload_queries(bxmi_ivl *h, const int32_t *qs, const int32_t *qe, int64_t nq, hipStream_t st)
{
    BXMI_TRY(h->q_s.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_e.reserve((size_t)(nq + 4) * 4));
    BXMI_HIP(hipMemcpyAsync(h->q_s.p, qs, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    BXMI_HIP(hipMemcpyAsync(h->q_e.p, qe, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    return BXMI_OK;
}

#include "host_pipeline.hpp"

extern "C" int bxmi_ivl_count(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts, int64_t *total)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_count"));
    if (nq < 0 || (nq > 0 && (!qs || !qe))) return fail(BXMI_EINVAL, "bxmi_ivl_count: bad arguments");
    if (total) *total = 0;
    if (nq == 0) return BXMI_OK;
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    if (g_opt_host_chunk > 0 && nq >= 2 * g_opt_host_chunk) return ivl_count_host_chunks(h, qs, qe, nq, counts, total);
    BXMI_TRY(upload_queries(h, qs, qe, nq, st));
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_total.reserve(64));
    BXMI_HIP(hipMemsetAsync(h->q_total.p, 0, 8, st));
    BXMI_TRY(bxmi_ivl_count_dev(h, h->q_s.as<int32_t>(), h->q_e.as<int32_t>(), nq, counts ? h->q_cnt.as<int32_t>() : nullptr,
                                h->q_total.as<int64_t>(), st));
    if (counts) BXMI_HIP(hipMemcpyAsync(counts, h->q_cnt.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    int64_t t = 0;
    BXMI_HIP(hipMemcpyAsync(&t, h->q_total.p, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total) *total = t;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_find_dev(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets,
                                 int32_t *hits, int64_t cap, int64_t *total_host, void *stream)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_find_dev"));
    if (nq < 0 || !offsets || (nq > 0 && (!qs || !qe)) || cap < 0 || (cap > 0 && !hits))
        return fail(BXMI_EINVAL, "bxmi_ivl_find_dev: bad arguments");
    hipStream_t st = as_stream(stream);
    if (nq == 0) {
        BXMI_HIP(hipMemsetAsync(offsets, 0, 8, st));
        if (total_host) *total_host = 0;
        return BXMI_OK;
    }
    // The batch passes read the queries 16 bytes per lane: a query array that is not aligned for that (a slice of a device
    // array; legal per bxmi.h) is answered by the direct kernels below, which read it element by element.
    const bool q_aligned = ((((uintptr_t)qs | (uintptr_t)qe) & 15) == 0);
    if (q_aligned && !h->has_reversed && h->n > 0 &&
        (g_opt_partition == 1 || (g_opt_partition < 0 && nq >= g_opt_partition_min && h->n >= 4096)))
    {
        if (g_opt_sorted_path) {
            // one cheap look at the starts decides the path on the host (find() synchronises for the total anyway)
            BXMI_TRY(h->p_slots.reserve((size_t)PT_MAX_SUB * PT_SLOT_STRIDE * sizeof(unsigned long long)));
            unsigned *flag = reinterpret_cast<unsigned *>(h->p_slots.as<unsigned long long>() + PT_SLOTS);
            unsigned unsorted = 0;
            // (a probe of 8192 starts first: a descent among them says "shuffled" for certain and saves the full read of the
            // starts -- 55 us per 50 M -- which only a batch that passes the probe pays)
            if (nq >= (1 << 20)) {
                hipLaunchKernelGGL(bm_probe_kernel, dim3(1), dim3(256), 0, st, qs, nq, flag);
                BXMI_HIP(hipMemcpyAsync(&unsorted, flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                BXMI_HIP(hipStreamSynchronize(st));
            }
            if (!unsorted) {
                BXMI_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned), st));
                hipLaunchKernelGGL(ivl_sorted_check_kernel, dim3(stream_grid(nq, 256)), dim3(256), 0, st, qs, nq, flag);
                // the whole chain of the sorted find behind the check, every kernel gated on the check's word: one round trip to the
                // host (flag and total together) instead of three -- for batches the probe has looked at (a smaller batch that is
                // NOT sorted would pay four launches that stand down before it takes the other path)
                if (((uintptr_t)offsets & 15) == 0 && nq >= (1 << 20)) {
                    bool stood_down = false;
                    BXMI_TRY(ivl_find_local(h, qs, qe, nq, offsets, hits, cap, total_host, st, flag, &stood_down));
                    if (!stood_down) return BXMI_OK;
                    unsorted = 1;
                } else {
                    BXMI_HIP(hipMemcpyAsync(&unsorted, flag, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                    BXMI_HIP(hipStreamSynchronize(st));
                }
            }
            if (!unsorted) return ivl_find_local(h, qs, qe, nq, offsets, hits, cap, total_host, st);
        }
        if (g_opt_find_sliced && g_opt_bitmap != 0 && g_opt_slice != 0 && h->n >= 4096 && nq >= g_opt_bitmap_min &&
            !((uintptr_t)offsets & 15)) {  // (fx_offsets / fx_hits_copy2 store the offsets 16 bytes at a time)
            if (h->sl_state == 0) BXMI_TRY(sl_prepare_index(h, st));
            if (h->sl_state == 1) {
                if (h->fx_state == 0) BXMI_TRY(fx_prepare_index(h, st));
                if (h->fx_state == 1) return ivl_find_fx(h, qs, qe, nq, offsets, hits, cap, total_host, st);
            }
        }
        return ivl_find_partitioned(h, qs, qe, nq, offsets, hits, cap, total_host, st);
    }
    BXMI_TRY(h->q_cnt.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_lo.reserve((size_t)(nq + 4) * 4));
    BXMI_TRY(h->q_hi.reserve((size_t)(nq + 4) * 4));
    TreeDev S = h->treeS.dev, P = h->treeP.dev;
    size_t lds_bytes = (size_t)(S.lds_ints + P.lds_ints) * 4;
    IndexDev ix = index_dev(h);
    int grid = device_props().cus * 2;
    int64_t need = div_up(nq, (int64_t)(FIND_THREADS / 8) * FIND_Q);
    if (need < grid) grid = (int)need;
    BXMI_TRY(allow_big_lds(ivl_find_count_kernel<true>, lds_bytes));
    hipLaunchKernelGGL(ivl_find_count_kernel<true>, dim3(grid), dim3(FIND_THREADS), lds_bytes, st, S, P, ix, qs, qe, nq,
                       h->q_lo.as<int32_t>(), h->q_hi.as<int32_t>(), h->q_cnt.as<int32_t>());
    BXMI_LAUNCH_CHECK();
    // offsets[0..nq) = exclusive sum of counts, offsets[nq] = total
    BXMI_TRY((device_scan<int32_t, long long, OpSum, false>(h->q_cnt.as<int32_t>(), reinterpret_cast<long long *>(offsets), nq, 0ll,
                                                           reinterpret_cast<long long *>(offsets) + nq, h->scan_scratch, st)));
    int64_t total = 0;
    BXMI_HIP(hipMemcpyAsync(&total, offsets + nq, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    if (total_host) *total_host = total;
    if (total > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find: %lld hits need a larger buffer than cap=%lld", (long long)total, (long long)cap);
    if (total == 0) return BXMI_OK;
    int fgrid = stream_grid(nq, FIND_THREADS / 8);
    hipLaunchKernelGGL(ivl_find_fill_kernel, dim3(fgrid), dim3(FIND_THREADS), 0, st, ix, qs, nq, h->q_lo.as<int32_t>(),
                       h->q_hi.as<int32_t>(), offsets, hits);
    BXMI_LAUNCH_CHECK();
    return BXMI_OK;
}

#ifdef BXMI_DEBUG_PEEK
// (debug builds only, never in the shipped library: a look at the scratch a find() left behind)
extern "C" int bxmi_debug_peek(bxmi_ivl_t *h, const char *name, void *host, size_t bytes)
{
    const DevBuf *b = nullptr;
    const std::string n(name);
    if (n == "recs") b = &h->bm_recs;
    else if (n == "slots") b = &h->bm_slots;
    else if (n == "hc") b = &h->fx_hc;
    else if (n == "cnt") b = &h->sl_cnt;
    else if (n == "loff") b = &h->sl_loff;
    else if (n == "svq") b = &h->fx_svq;
    else if (n == "tile_base") b = &h->fx_tile_base;
    else if (n == "runT2") b = &h->fx_runT2;
    else if (n == "tbl2") b = &h->fx_tbl2;
    else if (n == "tmp_hits") b = &h->sl_hits;
    else if (n == "meta2") b = &h->fx_meta2;
    else if (n == "pieces") b = &h->fx_pieces;
    else if (n == "eid") b = &h->sl_eid;
    else if (n == "qcnt") b = &h->q_cnt;
    if (!b) return fail(BXMI_EINVAL, "peek: no such buffer");
    if (bytes > b->cap) bytes = b->cap;
    BXMI_HIP(hipDeviceSynchronize());
    BXMI_HIP(hipMemcpy(host, b->p, bytes, hipMemcpyDeviceToHost));
    return (int)0;
}
#endif

extern "C" int bxmi_ivl_find(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets, int32_t *hits,
                             int64_t cap, int64_t *total)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_find"));
    if (nq < 0 || !offsets || (nq > 0 && (!qs || !qe)) || cap < 0 || (cap > 0 && !hits))
        return fail(BXMI_EINVAL, "bxmi_ivl_find: bad arguments");
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    if (nq == 0) {
        offsets[0] = 0;
        if (total) *total = 0;
        return BXMI_OK;
    }
    // BLOCKS until offsets / hits are written.  Large batches: host threads touch the pages of `offsets` while the queries go up and
    // the device works, those of `hits` (the part the total says will be written) while the offsets come down (PageToucher).
    constexpr size_t TOUCH_CHUNK = (size_t)32 << 20;
    const size_t off_bytes = (size_t)(nq + 1) * 8;
    PageToucher t_off, t_hits;
    const bool touch = g_opt_host_touchers > 0 && off_bytes >= 2 * TOUCH_CHUNK;
    if (touch) t_off.start(offsets, off_bytes, TOUCH_CHUNK, (int)g_opt_host_touchers);
    BXMI_TRY(upload_queries(h, qs, qe, nq, st));
    BXMI_TRY(h->q_off.reserve((size_t)(nq + 2) * 8));
    BXMI_TRY(h->q_hits.reserve((size_t)(cap + 4) * 4));
    int64_t tot = 0;
    int rc = bxmi_ivl_find_dev(h, h->q_s.as<int32_t>(), h->q_e.as<int32_t>(), nq, h->q_off.as<int64_t>(), h->q_hits.as<int32_t>(), cap,
                               &tot, st);
    if (total) *total = tot;
    if (rc != BXMI_OK && rc != BXMI_ERANGE) return rc;
    const bool want_hits = rc == BXMI_OK && tot > 0;
    const bool touch_hits = want_hits && g_opt_host_touchers > 0 && (size_t)tot * 4 >= 2 * TOUCH_CHUNK;
    if (touch_hits) t_hits.start(hits, (size_t)tot * 4, TOUCH_CHUNK, (int)g_opt_host_touchers);
    BXMI_TRY(download_touched(offsets, h->q_off.p, off_bytes, st, touch ? &t_off : nullptr));
    if (want_hits) BXMI_TRY(download_touched(hits, h->q_hits.p, (size_t)tot * 4, st, touch_hits ? &t_hits : nullptr));
    BXMI_HIP(hipStreamSynchronize(st));
    return rc;
}


constexpr int ONE_CAP = 4096 - 2;  // hits that fit the 16 KiB host-visible result buffer

// IntervalTree.find(start, end) for ONE query (intersection.pyx:400-406): one kernel launch, one stream sync.
extern "C" int bxmi_ivl_find_one(bxmi_ivl_t *h, int32_t qs, int32_t qe, int32_t *hits, int64_t cap, int64_t *n_hits)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_find_one"));
    if (!n_hits || cap < 0 || (cap > 0 && !hits)) return fail(BXMI_EINVAL, "bxmi_ivl_find_one: bad arguments");
    *n_hits = 0;
    if (h->n == 0) return BXMI_OK;
    BXMI_TRY(ivl_stream(h));
    if (!h->one_buf) {
        BXMI_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->one_buf), (4096 + 2) * sizeof(int32_t), hipHostMallocDefault));
        memset(h->one_buf, 0, (4096 + 2) * sizeof(int32_t));
    }
    const unsigned long long seq = ++h->one_seq;
    Tree tS = Tree(), tP = Tree();
    tS.dev = h->treeS.dev, tP.dev = h->treeP.dev;
    tS.set_lds_budget(0), tP.set_lds_budget(0);  // every level from global memory (L2): nothing to stage for one query
    hipLaunchKernelGGL(ivl_find_one_kernel, dim3(1), dim3(ONE_THREADS), 0, h->stream, tS.dev, tP.dev, index_dev(h), qs, qe, h->one_buf, ONE_CAP, seq);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY(wait_for_host_flag(reinterpret_cast<const unsigned long long *>(h->one_buf + 2 + ONE_CAP), seq, h->stream));
    const int64_t n = *reinterpret_cast<volatile long long *>(h->one_buf);
    *n_hits = n;
    if (n > ONE_CAP) {  // a very popular region: take the batched path once
        if (n > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find_one: %lld hits need a larger buffer than cap=%lld", (long long)n, (long long)cap);
        int64_t offs[2], total = 0;
        return bxmi_ivl_find(h, &qs, &qe, 1, offs, hits, cap, &total);
    }
    if (n > cap) return fail(BXMI_ERANGE, "bxmi_ivl_find_one: %lld hits need a larger buffer than cap=%lld", (long long)n, (long long)cap);
    if (n > 0) memcpy(hits, h->one_buf + 2, (size_t)n * sizeof(int32_t));
    return BXMI_OK;
}

// Two lower-bound ranks for the single-position neighbour API:
// out[t] = #{a_t[k] < x_t}, thresholds in 64 bits so position +/- max_dist cannot overflow.
__global__ void ivl_two_ranks_kernel(const int32_t *__restrict__ a0, long long x0, const int32_t *__restrict__ a1,
                                     long long x1, int n, int *out)
{
    if (threadIdx.x < 2) {
        const int32_t *a = threadIdx.x ? a1 : a0;
        long long x = threadIdx.x ? x1 : x0;
        int lo = 0, hi = n;
        while (lo < hi) {
            int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
            if ((long long)a[mid] < x)
                lo = mid + 1;
            else
                hi = mid;
        }
        out[threadIdx.x] = lo;
    }
}

extern "C" int bxmi_ivl_clusters(bxmi_ivl_t *h, const int32_t *ids, int32_t max_dist, int64_t *n_clusters, int32_t *starts,
                                 int32_t *ends, int64_t *offsets, int32_t *members)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_clusters"));
    if (!n_clusters) return fail(BXMI_EINVAL, "bxmi_ivl_clusters: n_clusters is NULL");
    // A negative distance: the reference merges an interval into a cluster when start <= cluster end - d AND end >= cluster
    // start + d (src/cluster.c:224-232, d = -max_dist), and an interval can then be both "right of" and "left of" a cluster: the
    // tree stops being ordered and the regions depend on the insertion order and on rand().  The committed experiment on the
    // reference's own C (tests/golden/cluster_negative_distance.txt, with the script that made it: -1 ... -8, 150 interval
    // sets, 12 insertion orders x 12 seeds) finds ONE distance with an answer: max_dist = -1 on intervals of positive length --
    // "overlap by at least one base", the connected components of the overlap graph, the same sweep as below -- and none
    // for -1 with zero-length intervals or for -2 and beyond.  Those are refused.
    if (max_dist < -1)
        return fail(BXMI_EINVAL, "bxmi_ivl_clusters: max_dist=%d; with a distance below -1 the reference's result depends on the "
                                 "insertion order and is not reproduced", (int)max_dist);
    if (h->has_reversed) return fail(BXMI_ESTATE, "bxmi_ivl_clusters: the index holds intervals with start > end");
    *n_clusters = 0;
    const int64_t n = h->n;
    if (n == 0) return BXMI_OK;
    if (!starts || !ends || !offsets || !members) return fail(BXMI_EINVAL, "bxmi_ivl_clusters: output arrays must hold n (+1) entries");
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    // scratch: flags / scanned flags in the query buffers, keys in the (free after seal) sort buffers
    BXMI_TRY(h->q_lo.reserve((size_t)(n + 4) * 4));
    BXMI_TRY(h->q_hi.reserve((size_t)(n + 4) * 4));
    BXMI_TRY(h->q_cnt.reserve((size_t)(n + 4) * 4));   // cluster starts
    BXMI_TRY(h->q_hits.reserve((size_t)(n + 4) * 4));  // cluster ends, then members
    BXMI_TRY(h->q_off.reserve((size_t)(n + 4) * 8));   // cluster offsets
    BXMI_TRY(h->q_s.reserve((size_t)(n + 4) * 4));     // ids
    BXMI_TRY(h->q_e.reserve((size_t)(n + 4) * 4));     // members
    BXMI_TRY(h->keys_a.reserve((size_t)(n + 1) * 8));
    BXMI_TRY(h->keys_b.reserve((size_t)(n + 1) * 8));
    int32_t *flag = h->q_lo.as<int32_t>(), *cid = h->q_hi.as<int32_t>();
    const int g = stream_grid(n, 256);
    int32_t *empty_dev = h->q_cnt.as<int32_t>() + n;  // (a spare word of the cluster-start buffer)
    BXMI_HIP(hipMemsetAsync(empty_dev, 0, 4, st));
    hipLaunchKernelGGL(cluster_flag_kernel, dim3(g), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->e_ord.as<int32_t>(), h->pm.as<int32_t>(), (int)n,
                       (int)max_dist, flag, empty_dev);
    BXMI_LAUNCH_CHECK();
    BXMI_TRY((device_scan<int32_t, int32_t, OpSum, true>(flag, cid, n, 0, nullptr, h->scan_scratch, st)));
    int32_t nc = 0, any_empty = 0;
    BXMI_HIP(hipMemcpyAsync(&nc, cid + (n - 1), 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(&any_empty, empty_dev, 4, hipMemcpyDeviceToHost, st));
    const int32_t *d_ids = nullptr;
    if (ids) {
        BXMI_HIP(hipMemcpyAsync(h->q_s.p, ids, (size_t)n * 4, hipMemcpyHostToDevice, st));
        d_ids = h->q_s.as<int32_t>();
    }
    hipLaunchKernelGGL(cluster_keys_kernel, dim3(g), dim3(256), 0, st, cid, flag, h->s_ord.as<int32_t>(), h->idx.as<int32_t>(), d_ids, (int)n,
                       h->keys_a.as<unsigned long long>(), h->q_cnt.as<int32_t>(), h->q_off.as<long long>());
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipStreamSynchronize(st));  // nc
    if (max_dist < 0 && any_empty)
        return fail(BXMI_EINVAL, "bxmi_ivl_clusters: max_dist=-1 with zero-length intervals; the reference's result depends on the "
                                 "insertion order then and is not reproduced");
    unsigned long long *sorted = nullptr;
    BXMI_TRY(radix_sort_keys<unsigned long long>(h->keys_a.as<unsigned long long>(), h->keys_b.as<unsigned long long>(), n, &sorted,
                                                 h->sort_scratch, st));
    hipLaunchKernelGGL(cluster_finish_kernel, dim3(g), dim3(256), 0, st, sorted, h->pm.as<int32_t>(), h->q_off.as<long long>(), (int)nc, (int)n,
                       h->q_hits.as<int32_t>(), h->q_e.as<int32_t>());
    BXMI_LAUNCH_CHECK();
    BXMI_HIP(hipMemcpyAsync(starts, h->q_cnt.p, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(ends, h->q_hits.p, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(offsets, h->q_off.p, (size_t)(nc + 1) * 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipMemcpyAsync(members, h->q_e.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    *n_clusters = nc;
    return BXMI_OK;
}

extern "C" int bxmi_ivl_neighbors(bxmi_ivl_t *h, int32_t position, int32_t max_dist, int dir, int32_t *out, int64_t cap,
                                  int64_t *n_out)
{
    BXMI_TRY(need_sealed(h, "bxmi_ivl_neighbors"));
    if (!n_out || cap < 0 || (cap > 0 && !out) || dir == 0) return fail(BXMI_EINVAL, "bxmi_ivl_neighbors: bad arguments");
    *n_out = 0;
    if (h->n == 0) return BXMI_OK;
    BXMI_TRY(ivl_stream(h));
    hipStream_t st = h->stream;
    BXMI_TRY(h->q_cnt.reserve(64));
    BXMI_TRY(h->q_total.reserve(64));
    BXMI_TRY(h->q_hits.reserve((size_t)(cap + 4) * 4));
    int *d_r = h->q_cnt.as<int>();
    int r[2] = {0, 0};
    const int n = (int)h->n;
    long long vlo, vhi;
    int lo, hi;
    if (dir > 0) {
        // intersection.pyx:213-229,255: p = position + 1, keep 0 <= start - p < max_dist (in-order)
        long long p = (long long)position + 1;
        vlo = p, vhi = p + max_dist;
        hipLaunchKernelGGL(ivl_two_ranks_kernel, dim3(1), dim3(64), 0, st, h->s_ord.as<int32_t>(), vlo, h->s_ord.as<int32_t>(), vhi, n, d_r);
        BXMI_HIP(hipMemcpyAsync(r, d_r, 8, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipStreamSynchronize(st));
        lo = r[0], hi = r[1];
        hipLaunchKernelGGL(ivl_filter_window_kernel, dim3(1), dim3(256), 0, st, h->s_ord.as<int32_t>(), h->idx.as<int32_t>(), lo, hi, vlo, vhi,
                           0, h->q_hits.as<int32_t>(), cap, h->q_total.as<unsigned long long>());
    } else {
        // intersection.pyx:192-209,240: p = position - 1, keep 0 <= p - end < max_dist (reverse in-order)
        // i.e. p - max_dist < end <= p.  Candidates lie in [first k with pm[k] > p-max_dist, #{start <= p}):
        // the upper bound is the reference's own `minstart > position` prune (:196-197).
        // (With stored intervals whose start exceeds their end -- IntervalTree.insert accepts them -- the reference prunes by
        // SUBTREE (`minstart > position`, :196-197), so whether such an interval with start > position is still reported
        // there depends on the treap's random shape: it is when it shares a subtree with a start <= position.  An index
        // that holds reversed intervals therefore scans the whole window above `lo` and reports every interval whose END
        // qualifies -- everything the reference can report whatever its priorities were; Appendix A.3 of SURVEY.md pins
        // before/after for proper intervals only, for that reason.)
        long long p = (long long)position - 1;
        vlo = p - max_dist + 1, vhi = p + 1;
        hipLaunchKernelGGL(ivl_two_ranks_kernel, dim3(1), dim3(64), 0, st, h->pm.as<int32_t>(), vlo, h->s_ord.as<int32_t>(), vhi, n, d_r);
        BXMI_HIP(hipMemcpyAsync(r, d_r, 8, hipMemcpyDeviceToHost, st));
        BXMI_HIP(hipStreamSynchronize(st));
        lo = r[0], hi = h->has_reversed ? n : r[1];
        if (hi < lo) hi = lo;
        hipLaunchKernelGGL(ivl_filter_window_kernel, dim3(1), dim3(256), 0, st, h->e_ord.as<int32_t>(), h->idx.as<int32_t>(), lo, hi, vlo, vhi,
                           1, h->q_hits.as<int32_t>(), cap, h->q_total.as<unsigned long long>());
    }
    BXMI_LAUNCH_CHECK();
    unsigned long long cnt = 0;
    BXMI_HIP(hipMemcpyAsync(&cnt, h->q_total.p, 8, hipMemcpyDeviceToHost, st));
    BXMI_HIP(hipStreamSynchronize(st));
    *n_out = (int64_t)cnt;
    int64_t ncopy = (int64_t)cnt < cap ? (int64_t)cnt : cap;
    if (ncopy > 0) BXMI_HIP(hipMemcpy(out, h->q_hits.p, (size_t)ncopy * 4, hipMemcpyDeviceToHost));
    return BXMI_OK;
}
