/*
 * bxmi.h -- C ABI of libbxmi.so, the MI355X (gfx950) engine behind
 *   bx.intervals.intersection.IntervalTree / Intersecter   (insert, find)
 *   bx.bitset.BinnedBitSet                                 (set_range, iand, ior, count_range, ...)
 *
 * Plain C, opaque handles, plain pointers and sizes; no torch / Python types.
 * Every function returns a status (BXMI_OK == 0); bxmi_last_error() gives the
 * text of the last failure on the calling thread.
 *
 * Pointer convention: arguments are HOST pointers unless the function name ends
 * in `_dev`, in which case every array argument is a DEVICE pointer (HBM) and
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).  Host
 * variants stage through the library's own stream and return when the result
 * is in the caller's buffer.
 *
 * Reference interfaces replaced (bx-python 0.14.0, paths under the reference):
 *   src/binBits.h:15-26          the 12 binBits* functions  -> bxmi_bits_*
 *   lib/bx/bitset.pyx:198-241    BinnedBitSet methods        -> call bxmi_bits_*
 *   lib/bx/intervals/intersection.pyx:388-406,428-435
 *                                IntervalTree.insert/find    -> bxmi_ivl_*
 *   (intersection.pyx has no C ABI of its own: its cdef classes are the
 *    interface, so the entry points below are what a Cython/ctypes shim of
 *    those classes binds; see INTEGRATION.md.)
 */
#ifndef BXMI_H
#define BXMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BXMI_OK 0
#define BXMI_EINVAL 1 /* bad argument (NULL handle, negative count, size > 2^31-1, ...) */
#define BXMI_ENOMEM 2 /* host or device allocation failed */
#define BXMI_EHIP 3   /* a HIP runtime call or kernel launch failed */
#define BXMI_ESTATE 4 /* call not valid in the handle's state (e.g. query before seal) */
#define BXMI_ERANGE 5 /* caller's output buffer too small; the needed size is reported */

typedef struct bxmi_ivl bxmi_ivl_t;   /* one interval index == one IntervalTree */
typedef struct bxmi_bits bxmi_bits_t; /* one binned bitset  == one BinnedBitSet */

/* ---- library / device ---------------------------------------------------- */
int bxmi_version(void);
const char *bxmi_last_error(void);
int bxmi_device_count(int *n);
int bxmi_set_device(int device);
int bxmi_get_device(int *device);
int bxmi_device_info(int device, char *name, int name_len, int *compute_units, int64_t *hbm_bytes);
/* Free and total bytes of the current device's memory (hipMemGetInfo): what the host side sizes group launches against
 * (a group makes every member allocate its whole word array).  No reference counterpart. */
int bxmi_mem_info(int64_t *free_bytes, int64_t *total_bytes);
int bxmi_synchronize(void *stream);

/* Raw HBM staging for hosts that do not bring their own allocator. */
int bxmi_malloc(void **dptr, size_t bytes);
int bxmi_free(void *dptr);
int bxmi_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int bxmi_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
int bxmi_memset(void *dst_dev, int value, size_t bytes);

/* Tuning / A-B knobs (process-wide), key -> value.  Results never depend on them (the GPU tests run both sides).  The
 * authoritative list with the defaults is what bxmi_option_at enumerates (the IVL_OPTS table of csrc/intervals.hip);
 * the ones a caller may want:
 *   ivl.partition      -1 auto (batches >= 4 Mi queries take the large-batch passes), 0 never, 1 always
 *   ivl.sorted_path    1 (default): a batch whose starts are already non-decreasing is answered as it lies
 *   ivl.sorted_cells   1 (default): ... from the cell images, stretch by stretch (count_dense.hpp, bs_*); 0: first-generation kernel
 *   ivl.bitmap         -1 (default): large batches take the exchange (tile sort -> search on unit images -> un-permute)
 *                      when the index qualifies; 0 never (round 1's bucketed pass)
 *   ivl.bitmap_min     smallest batch that takes the exchange (default 2 Mi queries)
 *   ivl.flat / ivl.dense / ivl.slice / ivl.sparse   force (1), forbid (0) or leave to the index's shape (-1, default) the
 *                      search stage: bitmap-cell images / dense unit images / staged key slices / offset-cell images
 *   ivl.bo_cell_log2   offset cells: coordinates per cell (6..8, 0 = from the density)
 *   ivl.bm_variant     tile shape of the exchange (-1 auto, 0 = 512 x 32, 1 = 1024 x 16, 2 = 1024 x 32 queries per tile)
 *   ivl.bd_chunk       queries per search work item (0 = default)
 *   ivl.bd_w8          8-bit counts between the search and the un-permute kernel (-1 auto from the density + feedback)
 *   ivl.order_skip     1 (default): the exact order check is dropped after two shuffled batches (a probe stands in)
 *   ivl.find_sliced    1 (default): find() on large unsorted batches goes through the exchange (count_slices.hpp)
 *   ivl.fx_direct      the exchange's fill writes straight into the CSR list (1) or into scratch, followed by a copy (0); -1 (default):
 *                      straight while the list the handle expects (hits per query of its previous batch) stays under 400 MB
 *   ivl.sl_f, ivl.sl_lanes, ivl.sl_flat, ivl.sl_run_cap   geometry of the slice stage (tests, A/B tools)
 *   ivl.bd_table_from  dense images: duplicated coordinates from which a cell gets a rank table (0 = 2 where the LDS has the room, else 6)
 *   ivl.bm_hard_ppm, ivl.bd_blocks   thresholds / shapes of the unit images (tests)
 *   ivl.host_chunk, ivl.host_touchers   the host-pointer count: queries per chunk of its pipeline (default 8 Mi; 0 = one piece) and
 *                      the host threads that touch an output array's pages ahead of the downloads (default 2; bxmi_ivl_find too)
 *   bits.grid          grid of the per-bitset kernels
 *   core.poll          1 (default): the one-call paths (bxmi_ivl_find_one, short bxmi_bits_count_range) poll a completion
 *                      word their kernel writes to host memory; 0: they wait for the stream
 * Unknown keys return BXMI_EINVAL. */
int bxmi_set_option(const char *key, int64_t value);
/* The current value of an option, and every option in turn (i = 0, 1, ... until BXMI_EINVAL): what the tests and A/B tools
 * read the defaults from.  (No reference counterpart: bx-python has no tuning knobs.) */
int bxmi_get_option(const char *key, int64_t *value);
int bxmi_option_at(int i, const char **key, int64_t *value);

/* ---- interval index  (intersection.pyx) ---------------------------------- */
/* IntervalTree()                                   intersection.pyx:380-382 */
int bxmi_ivl_create(bxmi_ivl_t **out);
int bxmi_ivl_destroy(bxmi_ivl_t *h);
/* IntervalTree.insert(start, end, value) x n, in insertion order; the payload
 * of interval i is its insertion index (the host keeps the objects).
 * Any int32 pair is accepted, like the reference (start > end, negatives).
 *                                                  intersection.pyx:388-397 */
int bxmi_ivl_append(bxmi_ivl_t *h, const int32_t *start, const int32_t *end, int64_t n);
int bxmi_ivl_append_dev(bxmi_ivl_t *h, const int32_t *start, const int32_t *end, int64_t n, void *stream);
/* Build the device index over everything appended so far: radix sort into the
 * treap's in-order (key start, end<=start first, -i/+i), sorted ends, prefix
 * max of ends, and the 32-ary search levels.  Re-callable after more appends. */
int bxmi_ivl_seal(bxmi_ivl_t *h, void *stream);
int bxmi_ivl_size(const bxmi_ivl_t *h, int64_t *n);
/* 1 if some stored interval has end < start (forces the general count path). */
int bxmi_ivl_has_reversed(const bxmi_ivl_t *h, int *flag);
/* In-order sequence of insertion indices == IntervalTree.traverse order.
 *                                                  intersection.pyx:262-268 */
int bxmi_ivl_order(const bxmi_ivl_t *h, int32_t *idx_out);
int bxmi_ivl_order_dev(const bxmi_ivl_t *h, const int32_t **idx_dev, const int32_t **start_dev, const int32_t **end_dev);

/* len(IntervalTree.find(qs[i], qe[i])) for a batch.  counts (int32[nq]) and
 * total (sum, int64) are each optional (NULL).  Exact for ANY query/target,
 * including zero-length, reversed and negative ones.
 * Host arrays; BLOCKS until counts / total are written.  Batches of >= 2 * ivl.host_chunk (default 2 * 8 Mi) queries go up, through
 * the pass and down in chunks, PCIe busy in both directions: the call starts one host thread for the downloads and
 * ivl.host_touchers (default 2) that touch the pages of `counts` ahead of them, all joined before it returns; `counts` must
 * not be read or written by anyone else meanwhile.  100 M queries: 16 ms (0.8 GB up at 56 GB/s is 14.3) against 32-74 ms in
 * one piece.  Not thread-safe per handle, like every call that takes a bxmi_ivl_t.
 *                                                  intersection.pyx:169-189,400-406 */
int bxmi_ivl_count(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts, int64_t *total);
/* Device variant: *total_dev (device int64) is ACCUMULATED into (zero it first).  counts = NULL: the total only -- nothing is
 * stored per query.  Stream-ordered, with ONE exception: a handle's FIRST large batch (>= ivl.bitmap_min queries) builds the
 * index's unit images and answers the order probe synchronously -- it waits for `stream` once (and cannot be captured into a
 * hipGraph); every later call only enqueues. */
int bxmi_ivl_count_dev(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int32_t *counts,
                       int64_t *total_dev, void *stream);
/* A dict of per-chromosome trees queried in one go (scripts/interval_join.py:21-28 keeps {chrom: Intersecter}; a genome-wide
 * batch asks every tree with its own chromosome's queries): exactly n calls of bxmi_ivl_count_dev -- hs[i] with
 * qs[i][0..nq[i]), counts[i] and totals_dev[i] (each optional / accumulated as there) -- but the indexes that qualify for
 * the bitmap-cell pass share ONE pass: a fixed handful of launches for the whole genome instead of one set per chromosome. */
int bxmi_ivl_count_multi_dev(bxmi_ivl_t *const *hs, int n, const int32_t *const *qs, const int32_t *const *qe, const int64_t *nq,
                             int32_t *const *counts, int64_t *const *totals_dev, void *stream);
/* Which search stage of the large-batch count pass can serve this sealed index -- the slice search stage (count_slices.hpp: sorted keys staged per unit of 2^f buckets; serves sparse
 * indexes and spans whose bucket image outgrows the LDS): *state = 0 not decided yet, 1 = usable, -1 = one bucket's
 * keys alone do not fit; unit_keys[0..6] = the most keys a unit of 2^f buckets stages.  Introspection only. */
int bxmi_ivl_slice_state(const bxmi_ivl_t *h, int *state, int64_t *unit_keys);
/* The same for the dense-image search stage (count_dense.hpp: one bit per coordinate, units of 2^19 coordinates; serves
 * dense indexes, duplicated coordinates included): *state = 0 not decided yet, 1 = usable, -1 = the index does not fit
 * the format; worst[0] = most keys of one block (the unit, or 2^17 coordinates: limit 32767), worst[1] = most 16-bit
 * overflow entries of one unit (lists of duplicated coordinates and the 129-entry rank tables of clumped cells; limit
 * 32704 with units of 2^18 coordinates).  Introspection only. */
int bxmi_ivl_dense_state(const bxmi_ivl_t *h, int *state, int64_t *worst);
/* The same for the flat walk on cell images (count_dense.hpp, bp_*: the cells of the bitmap pass laid out per unit of
 * 2^18 coordinates, records walked 16 bytes at a time, 16-bit counts): *state = 0 not decided yet, 1 = usable, -1 = the
 * index does not qualify (span wider than 2^29, reversed targets, too many cells with several duplicated coordinates:
 * *hard_cells of them).  This is the stage a dense index takes first.  Introspection only. */
int bxmi_ivl_flat_state(const bxmi_ivl_t *h, int *state, int64_t *hard_cells);
/* The same for the cell images of SPARSE indexes (offset_cells.hpp: a cell of 2^k coordinates, k = 6..8 from the index's
 * density, holds the offsets of up to five keys; units of up to 2^20 coordinates on the same persistent walk, two workgroups per CU): *state = 0
 * not decided yet, 1 = usable, -1 = the index does not qualify (too dense, reversed targets, too many cells with more than
 * five keys: *hard_cells of them); *cell_log2 = k when usable.  A sparse index takes this stage when a batch brings enough
 * queries per unit image (4096), key slices otherwise.  Introspection only. */
int bxmi_ivl_sparse_state(const bxmi_ivl_t *h, int *state, int64_t *hard_cells, int *cell_log2);
/* The width of the counts a flat-walk pass over cell images hands from its search to its un-permute kernel: *bits = 8
 * while the index is sparse enough for small counts (fewer than 128 targets per 2048 coordinates) and fewer than one count
 * in 64 of the passes so far came back as "does not fit" (*wide_counts of them, as last mirrored to the host; such counts
 * are recomputed, the results are exact either way), else 16.  Introspection only. */
int bxmi_ivl_count_width(const bxmi_ivl_t *h, int *bits, int64_t *wide_counts);
/* Whether large count batches on this index currently go without the order check (bm_sorted_check_kernel + the
 * stand-down of the sorted-batch kernel): *skipping = 1 after the checks of two batches in a row found the starts NOT sorted
 * -- a probe of 8192 consecutive starts then rides on every batch, and the first one without a descent brings the check back --
 * *answers_seen = order reports the host has read so far (they arrive through host memory, a pass or more late).
 * Introspection only: a sorted batch met without the check goes through the exchange, with the same counts. */
int bxmi_ivl_order_state(const bxmi_ivl_t *h, int *skipping, int64_t *answers_seen);

/* IntervalTree.find for a batch, as CSR: offsets[nq+1] (int64) and, for query
 * i, hits[offsets[i]..offsets[i+1]) = insertion indices in the reference's
 * result order.  If the hit list needs more than `cap` entries the call
 * returns BXMI_ERANGE with offsets and *total valid and hits untouched.
 * Host arrays; BLOCKS until offsets / hits are written.  From 8 Mi queries (or 16 Mi hits) on, ivl.host_touchers (default 2)
 * host threads touch the pages of `offsets` while the queries go up and the device works, those of `hits` while the
 * offsets come down; they are joined before the call returns.  configs[4] (50 M x 50 M, 250 M hits) into fresh numpy
 * arrays: 39 ms against 67 without them (0.4 GB up + 1.4 GB down at 56 GB/s is 32). */
int bxmi_ivl_find(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets,
                  int32_t *hits, int64_t cap, int64_t *total);
/* bxmi_ivl_find_dev: device pointers of any natural alignment (4 bytes for qs / qe / hits, 8 for offsets) are legal; the
 * batch passes need qs, qe and offsets on 16-byte boundaries (what every allocator hands out) and a slice that is not is
 * answered by the direct tree kernels instead -- same results, ~8 x slower.  Blocks until the offsets and the total are
 * known (one stream synchronisation); the hits may still be in flight on `stream` when it returns. */
int bxmi_ivl_find_dev(bxmi_ivl_t *h, const int32_t *qs, const int32_t *qe, int64_t nq, int64_t *offsets,
                      int32_t *hits, int64_t cap, int64_t *total_host, void *stream);

/* IntervalTree.find for ONE query: one launch + one stream sync (the latency path of the per-call
 * drop-in API).  *n_hits = number of hits; BXMI_ERANGE if it exceeds cap. */
int bxmi_ivl_find_one(bxmi_ivl_t *h, int32_t qs, int32_t qe, int32_t *hits, int64_t cap, int64_t *n_hits);

/* IntervalNode.left / right candidate collection for before()/after():
 * dir < 0: reverse in-order, keep 0 <= (position-1) - end   < max_dist
 * dir > 0: in-order,         keep 0 <= start - (position+1) < max_dist
 * Writes up to cap insertion indices; *n_out = number of candidates.
 * An index that holds REVERSED intervals (start > end) reports, for dir < 0, every interval whose end qualifies: the
 * reference prunes by subtree (`minstart > position`), so which of those it reports depends on its treap's random shape;
 * this is the superset of every such run, found by one scan of the candidates above the window's lower end (O(n) for
 * such an index; proper indexes scan the window only).
 *                                                  intersection.pyx:192-260 */
int bxmi_ivl_neighbors(bxmi_ivl_t *h, int32_t position, int32_t max_dist, int dir, int32_t *out, int64_t cap,
                       int64_t *n_out);

/* ClusterTree (lib/bx/intervals/cluster.pyx:57-121, src/cluster.c:112-260): groups of intervals chained by gaps of at
 * most max_dist (>= 0).  All clusters in ascending start order: starts[c], ends[c], and members[offsets[c] ..
 * offsets[c+1]) = the member ids in ascending order, where an interval's id is ids[insertion index] (or the insertion
 * index itself when ids is NULL).  starts/ends/members need n entries, offsets n + 1 (n = bxmi_ivl_size).  The caller
 * applies ClusterTree's min_intervals filter.  max_dist = -1 is accepted when no interval is empty (the reference is
 * deterministic there: tests/golden/cluster_negative_distance.txt); -1 with an empty interval and every max_dist < -1
 * -> BXMI_EINVAL (the reference's result then depends on the insertion order and on unseeded rand() priorities). */
int bxmi_ivl_clusters(bxmi_ivl_t *h, const int32_t *ids, int32_t max_dist, int64_t *n_clusters, int32_t *starts, int32_t *ends,
                      int64_t *offsets, int32_t *members);

/* ---- binned bitset  (binBits.h:15-26, bitset.pyx:198-241) ----------------- */
/* binBitsAlloc(size, granularity): bin_size and nbins use the reference's
 * float32 arithmetic; size > 2^31-1 or size < 1 -> BXMI_EINVAL.
 * granularity == 0 creates a FLAT set (bitset.pyx:107-173 BitSet over
 * kent/bits.h: no bins, so no ALL_ONE arithmetic). */
int bxmi_bits_create(int64_t size, int64_t granularity, bxmi_bits_t **out);
int bxmi_bits_destroy(bxmi_bits_t *h); /* binBitsFree */
int bxmi_bits_info(const bxmi_bits_t *h, int32_t *size, int32_t *bin_size, int32_t *nbins);
/* Device view: dense LSB-first uint64 words covering [0, nbins*bin_size). */
int bxmi_bits_words_dev(bxmi_bits_t *h, uint64_t **words_dev, int64_t *nwords);
/* Per-bin state as the reference would hold it: 0 = ALL_ZERO, 1 = ALL_ONE, 2 = allocated. */
int bxmi_bits_bin_states(bxmi_bits_t *h, uint8_t *out);

int bxmi_bits_get(bxmi_bits_t *h, int32_t pos, int *bit); /* binBitsReadOne  */
int bxmi_bits_set(bxmi_bits_t *h, int32_t pos);           /* binBitsSetOne   */
int bxmi_bits_clear(bxmi_bits_t *h, int32_t pos);         /* binBitsClearOne */
/* binBitsSetRange x n.  Ranges must satisfy 0 <= start, 0 <= len,
 * start+len <= size (the wrapper raises bitset.pyx's IndexErrors first). */
int bxmi_bits_set_ranges(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n);
int bxmi_bits_set_ranges_dev(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n, void *stream);
/* binBitsCountRange x n (including the reference's ALL_ONE-bin arithmetic). */
int bxmi_bits_count_ranges(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n, int32_t *out);
int bxmi_bits_count_ranges_dev(bxmi_bits_t *h, const int32_t *start, const int32_t *len, int64_t n, int32_t *out,
                               void *stream);
/* binBitsCountRange for one (possibly chromosome-long) range, grid-wide reduction. */
int bxmi_bits_count_range(bxmi_bits_t *h, int32_t start, int32_t len, int32_t *out);
/* binBitsFindSet (val=1) / binBitsFindClear (val=0): first such bit >= start, else size. */
int bxmi_bits_next(bxmi_bits_t *h, int32_t start, int val, int32_t *out);
int bxmi_bits_and(bxmi_bits_t *h, const bxmi_bits_t *other); /* binBitsAnd */
int bxmi_bits_or(bxmi_bits_t *h, const bxmi_bits_t *other);  /* binBitsOr  */
int bxmi_bits_not(bxmi_bits_t *h);                           /* binBitsNot */
int bxmi_bits_xor(bxmi_bits_t *h, const bxmi_bits_t *other); /* bitXor (flat sets only, bits.h:56) */
/* Fused  h &= other  and  popcount(h[0,size))  in one pass over HBM. */
int bxmi_bits_and_count(bxmi_bits_t *h, const bxmi_bits_t *other, int64_t *count);
/* Stream-ordered forms used by the bench (no host sync; *count_dev accumulated). */
int bxmi_bits_and_dev(bxmi_bits_t *h, const bxmi_bits_t *other, void *stream);
int bxmi_bits_or_dev(bxmi_bits_t *h, const bxmi_bits_t *other, void *stream);
int bxmi_bits_and_count_dev(bxmi_bits_t *h, const bxmi_bits_t *other, int64_t *count_dev, void *stream);
int bxmi_bits_popcount_dev(bxmi_bits_t *h, int64_t *count_dev, void *stream);
/* Genome-scale batches: the per-chromosome loops of bed_intersect_basewise.py:25-28
 * (iand) and bed_coverage.py:27-29 (count_range(0, size)) as ONE launch over all
 * members.  counts_dev, when given, is int64[n_members] in HBM and is accumulated
 * into (zero it first): popcount of each member's result inside [0, size). */
typedef struct bxmi_bits_group bxmi_bits_group_t;
int bxmi_bits_group_create(bxmi_bits_t *const *members, int n, bxmi_bits_group_t **out);
int bxmi_bits_group_destroy(bxmi_bits_group_t *g);
int bxmi_bits_group_and_dev(bxmi_bits_group_t *g, const bxmi_bits_group_t *other, int64_t *counts_dev, void *stream);
int bxmi_bits_group_or_dev(bxmi_bits_group_t *g, const bxmi_bits_group_t *other, void *stream);
int bxmi_bits_group_popcount_dev(bxmi_bits_group_t *g, int64_t *counts_dev, void *stream);

/* Maximal runs of set bits inside [from, size), i.e. the pairs the loop
 * start=next_set(end); end=next_clear(start) of bed_intersect_basewise.py:32-38
 * produces.  Writes up to cap pairs; *n_runs = number of runs (BXMI_ERANGE if > cap). */
int bxmi_bits_runs(bxmi_bits_t *h, int32_t from, int32_t *run_start, int32_t *run_end, int64_t cap, int64_t *n_runs);

/* ---- BED text -> SoA columns on the host (the step before the hot path) ------
 * Strict single-pass parser for what lib/bx/bitset_builders.py:33-46 and
 * scripts/bed_intersect.py:46-50 do per line in Python: skip '#' and blank lines,
 * split on whitespace runs, int() the start/end columns.  It STOPS at the first line
 * that is not plain ASCII BED (stop_line/stop_off), so the caller can hand the rest to
 * the reference's own semantics; it never reinterprets or repairs input. */
typedef struct bxmi_bed bxmi_bed_t;
int bxmi_bed_parse(const char *data, int64_t len, int chrom_col, int start_col, int end_col, bxmi_bed_t **out);
int bxmi_bed_destroy(bxmi_bed_t *b);
int bxmi_bed_info(const bxmi_bed_t *b, int64_t *n_rows, int32_t *n_chroms, int64_t *stop_line, int64_t *stop_off,
                  int64_t *lines_seen);
/* Borrowed views, valid until bxmi_bed_destroy: chromosome id (first-appearance order), start, end,
 * and each row's line as (offset, length incl. newline) into the parsed buffer. */
int bxmi_bed_columns(const bxmi_bed_t *b, const int32_t **chrom_id, const int64_t **start, const int64_t **end,
                     const int64_t **line_off, const int32_t **line_len);
const char *bxmi_bed_chrom_name(const bxmi_bed_t *b, int32_t id);
/* Write the lines with mask[row] != 0, each followed by `suffix`, to file descriptor fd. */
int bxmi_bed_emit_lines(const bxmi_bed_t *b, const char *data, const uint8_t *mask, const char *suffix, int fd);

/* ---- delimited interval text -> SoA columns (the reader side of the operations layer) --------
 * What lib/bx/intervals/io.py:106-216 (GenomicIntervalReader over tabular/io.py:86-156) does per line, for the lines
 * whose outcome is certain: blank lines, comment / header lines (first line starting with one of `comment_prefixes`),
 * and rows whose TAB-separated chromosome / start / end / strand fields are already in the normal form the reader writes
 * back (stripped name, canonical integers, "+" or "-", start <= end).  It STOPS at the first other line (stop_off); the
 * caller continues from there with the reference's own per-line semantics.  strand_col < 0 or beyond the row: no strand. */
typedef struct bxmi_tab bxmi_tab_t;
int bxmi_tab_parse(const char *data, int64_t len, int chrom_col, int start_col, int end_col, int strand_col,
                   const char *const *comment_prefixes, int n_prefixes, bxmi_tab_t **out);
int bxmi_tab_destroy(bxmi_tab_t *b);
int bxmi_tab_info(const bxmi_tab_t *b, int64_t *n_lines, int32_t *n_chroms, int64_t *stop_off);
/* Borrowed per-LINE views, valid until bxmi_tab_destroy: kind (0 row, 1 blank, 2 comment, 3 header), the line's offset and
 * length (without its newline) in the parsed buffer, and for rows the chromosome id (first-appearance order), start, end
 * and strand byte ('+', '-', 0 = no strand field). */
int bxmi_tab_columns(const bxmi_tab_t *b, const uint8_t **kind, const int64_t **line_off, const int32_t **line_len, const int32_t **chrom_id,
                     const int64_t **start, const int64_t **end, const uint8_t **strand);
const char *bxmi_tab_chrom_name(const bxmi_tab_t *b, int32_t id);

/* ---- the path's only collective (multi-GPU, one process per GPU) ---------------------------
 * Intervals on different chromosomes never meet (scripts/interval_join.py:21-28 keeps one tree per chromosome,
 * lib/bx/bitset_builders.py:31-45 one bitset), so a genome is sharded by chromosome with NO data-path exchange; what
 * the ranks do exchange is the vector of per-chromosome overlap totals: an int64 sum all-reduce, RCCL over xGMI.
 * rank 0 makes the 128-byte id and hands it to the others through whatever launched them; every rank then creates
 * its communicator on its CURRENT device.  bxmi_allreduce_i64 works in place on device memory, ordered on `stream`;
 * n must be the same on every rank (n == 0 returns at once without entering the collective).
 * librccl.so is opened on first use; without it these four calls fail with BXMI_EHIP and nothing else is affected. */
typedef struct bxmi_comm bxmi_comm_t;
int bxmi_comm_unique_id(void *id128);
int bxmi_comm_create(bxmi_comm_t **out, const void *id128, int rank, int world);
int bxmi_comm_destroy(bxmi_comm_t *c);
int bxmi_allreduce_i64(bxmi_comm_t *c, int64_t *buf_dev, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BXMI_H */
