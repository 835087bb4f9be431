// count_dense.hpp -- third search stage of the large-batch count pass: DENSE UNIT IMAGES ("bd_*" kernels).
// Included by intervals.hip after count_slices.hpp: tile sort, run table, unit sums, plan and un-permute are the bm_* /
// sl_* kernels unchanged; what differs is what a search workgroup keeps in LDS and how it walks its records.
//
// Why: the bucket-pair search of count_bitmap.hpp moves 1.0 GB at 2.5 TB/s while its neighbours stream at 5 TB/s.  A
// (tile, bucket pair) run is ~32 records = 128 unaligned bytes, fetched and stored as 4-byte pieces per lane: 17.6 M +
// 22 M quarter-line requests per 100 M queries, and that request rate -- not bytes, not compute -- is its bound.
// Two changes halve the runs' number and quarter the requests:
//   1. a bitmap of 1 bit per coordinate instead of 2: one 16-byte cell per 128 coordinates holds which of them carry a
//      key, a 16-bit word per cell holds the rank of the cell's first key relative to its 2^17-coordinate block, and a
//      small overflow list describes the coordinates that carry more than one key (exact for any multiset, see below).
//      1.14 bits per coordinate: a UNIT of 2^19 coordinates (four buckets of configs[1]) fits one CU's LDS with both of
//      its arrays, so a (tile, unit) run is ~64 records = 256 bytes;
//   2. the walk is FLAT and 16 bytes wide: a wave lays the runs of 64 tiles end to end (a DPP prefix sum over their
//      lengths in 16-byte slots), every lane takes one aligned int4 of four records per pass -- found through one ballot,
//      a few readlanes and two ds_bpermutes, no LDS table, no workgroup barrier -- answers the four records and writes
//      the four counts back as one int4 (records of neighbouring units at a run's two ends are masked: their lanes
//      store single dwords).  Every lane is busy in every pass whatever the run lengths are; waves pull batches of 64
//      tiles from a workgroup counter, runs longer than 256 records (sorted / clumped input) go to the workgroup's
//      cooperative finish as in the other stages.
//
// rank(rel) = #{keys of the unit's slice below coordinate lo_u + rel}
//           = qbase[block(cell)] + base15(cell) + popcount(cell's bits below rel) + extras(cell, rel)
// A cell whose coordinates carry no duplicate keys has meta = base15 (< 2^15).  Any other cell has meta = 0x8000 | i and
// ov[i] = base15, ov[i+1..] = one 16-bit entry per duplicated coordinate: position : 7 | extra copies : 8 | more : 1
// (more than 255 extra copies take several entries; a lone entry is followed by a zero), extras(cell, rel) sums the
// copies of the entries below rel.  A cell with more than two such entries -- a clump: hundreds of keys on a few dozen
// coordinates, exon starts, peak summits -- gets a TABLE instead: ov[i] = 0x8000 | base15, then T[p] = #{keys of the
// cell below position p} for p = 0 .. 127 as bytes (ov[i + 1] = 0, 64 words) when the cell holds fewer than 256 keys,
// else as 16-bit words (ov[i + 1] = 1, 128 words): its rank is two more reads and no popcount.  Units of 2^18
// coordinates leave 64 KiB of LDS for lists and tables.  ov[0..2] are zeros.
// An index qualifies while every block holds < 2^15 keys and every unit's overflow area fits what the image leaves of the LDS;
// bd_image_kernel reports both maxima when it builds the images (once per sealed index).
//
// count(q) = (sLo + rankS(off + len)) - (eLo + rankE(off + 1))        (intersection.pyx:180-189)
#pragma once
#include "offset_cells.hpp"

namespace bxmi {

constexpr int BD_UNIT_LOG2 = 19;     // coordinates per unit, at most: the offset field of a record
constexpr int BD_RSHIFT = 19;        // record = offset : 19 | length : 13 (8191 = escape)
constexpr int BD_MARGIN = 8192;      // the starts' cells reach this far past the unit: every record's qe is covered
constexpr int BD_MAX_F = 6;          // buckets per unit = 2^f
constexpr int BD_MAX_SHIFT = 19;     // bucket width <= 2^19: spans up to 2^30
constexpr int BD_OV_MAX = 32704;     // 16-bit overflow entries per unit, at most (a cell's 15-bit index must reach them)
// list entries from which a cell gets a rank table: tried in this order per unit width (bd_prepare_index).  A table is 66 or 130
// overflow words where a short list is 3-6, but a wave whose lanes meet plain cells, lists AND tables runs all three paths:
// measured on the clustered leg of bench.py (every look-up near a hot spot) 1.19 ms with tables from 6 entries, 1.08 / 1.05 /
// 1.01 from 4 / 3 / 2; from 1 the overflow area no longer fits the LDS and the index falls to the key slices (1.43).
constexpr int BD_TABLE_FROM_FIRST = 2, BD_TABLE_FROM_LAST = 6;
constexpr int BD_THREADS = 1024;
constexpr int BD_LONG_SLOTS = 64;    // runs of more 16-byte slots than this go to the cooperative finish
constexpr int BD_LONG_CAP = 320;
constexpr int BD_HDR_QS = 6;         // header words: [0..5] qbaseE, [6..11] qbaseS, [12] overflow entries, [13] eLo, [14] sLo

// Layout of one unit's image, in bytes (the same in HBM and in LDS; every part starts 16-byte aligned).
struct BdLayout {
    int nce, ncs;                      // cells, sentinel included
    int bitsE, bitsS, metaE, metaS, hdr, ov, bytes;
    int ov_cap;                        // 16-bit entries of the overflow area: what is left of the CU's LDS, at most BD_OV_MAX
};

__host__ __device__ inline BdLayout bd_layout(int unit_log2)
{
    BdLayout L;
    const int UW = 1 << unit_log2;
    L.nce = (UW >> 7) + 1;
    L.ncs = ((UW + BD_MARGIN) >> 7) + 1;
    L.bitsE = 0;
    L.bitsS = L.nce * 16;
    L.metaE = L.bitsS + L.ncs * 16;
    L.metaS = L.metaE + ((L.nce * 2 + 15) & ~15);
    L.hdr = L.metaS + ((L.ncs * 2 + 15) & ~15);
    L.ov = L.hdr + 64;
    const int room = (160 * 1024 - 3072 /* the search kernel's static LDS */ - L.ov) / 2;
    L.ov_cap = (room < BD_OV_MAX ? room : BD_OV_MAX) & ~7;
    L.bytes = L.ov + L.ov_cap * 2;
    return L;
}

// ---------------------------------------------------------------------------
// images: built once per sealed index, one workgroup per unit
// ---------------------------------------------------------------------------
__device__ __forceinline__ int bd_lower_bound(const int32_t *__restrict__ a, int lo, int hi, int key)
{
    while (lo < hi) {
        const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// stats: [0] most keys in one block of 2^bshift cells, [1] most overflow entries of one unit
__global__ __launch_bounds__(BD_THREADS) void bd_image_kernel(const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_sorted, int n,
                                                              BmGeom g, int bshift, unsigned char *__restrict__ images, unsigned *__restrict__ stats, int table_from)
{
    const int bmask = (1 << bshift) - 1;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ int s_r[2];
    __shared__ unsigned scan_tmp[16];
    const int unit = blockIdx.x;
    const int ulog = g.shift + g.f;
    const BdLayout L = bd_layout(ulog);
    const long long UW = 1ll << ulog;
    const long long lo = (long long)g.cmin + (long long)unit * UW;
    unsigned char *__restrict__ img = images + (size_t)unit * L.bytes;
    unsigned *hdr = reinterpret_cast<unsigned *>(img + L.hdr);
    unsigned short *ov = reinterpret_cast<unsigned short *>(img + L.ov);
    // LDS: bits [ncs * 4], cnt [ncs], ecnt [ncs], ovoff [ncs], ecur [ncs]
    unsigned *bits = reinterpret_cast<unsigned *>(dyn);
    unsigned *cnt = bits + 4 * L.ncs, *ecnt = cnt + L.ncs, *ovoff = ecnt + L.ncs, *ecur = ovoff + L.ncs;
    unsigned ov_base = 3;  // overflow entries used so far (three zeros for the plain cells, then the ends' lists, then the starts')
    unsigned worst_block = 0;
    if (threadIdx.x < 2 * BD_HDR_QS) hdr[threadIdx.x] = 0u;  // (one block per unit: a kernel that reads the table finds zeros)
    if (threadIdx.x < 3) ov[threadIdx.x] = 0;
    for (int arr = 0; arr < 2; arr++) {
        const int32_t *__restrict__ A = arr == 0 ? e_sorted : s_ord;
        const int nc = arr == 0 ? L.nce : L.ncs;
        const long long cover = arr == 0 ? UW : UW + BD_MARGIN;  // keys with rel in [0, cover) belong to this image
        if (threadIdx.x < 2) s_r[threadIdx.x] = bm_rank_lt64(A, n, threadIdx.x == 0 ? lo : lo + cover);
        for (int c = threadIdx.x; c < nc; c += BD_THREADS) {
            bits[4 * c] = bits[4 * c + 1] = bits[4 * c + 2] = bits[4 * c + 3] = 0u;
            cnt[c] = ecnt[c] = ecur[c] = 0u;
        }
        __syncthreads();
        const int r0 = s_r[0], r1 = s_r[1];
        if (threadIdx.x == 0) hdr[arr == 0 ? 13 : 14] = (unsigned)r0;
        // pass 1: bits, keys per cell, overflow entries per cell
        for (int r = r0 + (int)threadIdx.x; r < r1; r += BD_THREADS) {
            const int k = A[r];
            const unsigned rel = (unsigned)((long long)k - lo);
            const unsigned c = rel >> 7, p = rel & 127u;
            atomicAdd(&cnt[c], 1u);
            const bool first = r == r0 || A[r - 1] != k, last = r + 1 == r1 || A[r + 1] != k;
            if (first) atomicOr(&bits[4 * c + (p >> 5)], 1u << (p & 31u));
            if (last && !first) {
                const unsigned extras = (unsigned)(r - bd_lower_bound(A, r0, r, k));
                atomicAdd(&ecnt[c], (extras + 254u) / 255u);
            }
        }
        __syncthreads();
        // exclusive prefixes over the cells: keys below the cell, overflow words before the cell's
        const int K = (nc + BD_THREADS - 1) / BD_THREADS;
        const int c_lo = (int)threadIdx.x * K, c_hi = c_lo + K < nc ? c_lo + K : nc;
        // overflow words of a cell: none, {base, entry, entry or zero}, or {base, 128 ranks}
        auto ov_words = [table_from](unsigned entries, unsigned keys) {
            if (entries == 0u) return 0u;
            if (entries >= (unsigned)table_from) return keys < 256u ? 66u : 130u;
            return 1u + (entries > 2u ? entries : 2u);
        };
        unsigned ksum = 0, osum = 0;
        for (int c = c_lo; c < c_hi; c++) {
            ksum += cnt[c];
            osum += ov_words(ecnt[c], cnt[c]);
        }
        unsigned ktot, otot;
        unsigned kexc = block_exclusive_scan(ksum, OpSum(), 0u, scan_tmp, &ktot);
        unsigned oexc = block_exclusive_scan(osum, OpSum(), 0u, scan_tmp, &otot);
        for (int c = c_lo; c < c_hi; c++) {
            const unsigned k = cnt[c], o = ov_words(ecnt[c], cnt[c]);
            cnt[c] = kexc;  // keys of the slice below cell c
            ovoff[c] = oexc;
            kexc += k;
            oexc += o;
        }
        __syncthreads();
        unsigned short *meta = reinterpret_cast<unsigned short *>(img + (arr == 0 ? L.metaE : L.metaS));
        for (int c = threadIdx.x; c < nc; c += BD_THREADS) {
            const int blk = c >> bshift;
            const unsigned qb = cnt[blk << bshift];
            const unsigned base = cnt[c] - qb;
            if ((c & bmask) == 0) hdr[(arr == 0 ? 0 : BD_HDR_QS) + blk] = qb;
            // keys of this block: up to the next block's first cell (or the end of the image)
            if ((c & bmask) == bmask || c == nc - 1) {
                const unsigned in_block = base + ((c == nc - 1) ? ktot - cnt[c] : cnt[c + 1] - cnt[c]);
                worst_block = in_block > worst_block ? in_block : worst_block;
            }
            unsigned m = base & 0x7FFFu;
            if (ecnt[c]) {
                const unsigned at = ov_base + ovoff[c];
                m = 0x8000u | (at & 0x7FFFu);
                const unsigned table = ecnt[c] >= (unsigned)table_from ? 0x8000u : 0u;
                if (at < (unsigned)L.ov_cap) ov[at] = (unsigned short)((base & 0x7FFFu) | table);
                if (ecnt[c] == 1u && at + 2u < (unsigned)L.ov_cap) ov[at + 2u] = 0;  // the zero after a lone entry
            }
            meta[c] = (unsigned short)m;
        }
        // rank tables: 128 threads per clumped cell, T[p] = #{keys of the cell below position p} (the cell's keys are
        // a contiguous range of the sorted array)
        for (int c = (int)(threadIdx.x >> 7); c < nc; c += BD_THREADS >> 7) {
            if (ecnt[c] < (unsigned)table_from) continue;
            const unsigned p = threadIdx.x & 127u;
            const int k0 = r0 + (int)cnt[c], k1 = r0 + (int)(c + 1 < nc ? cnt[c + 1] : ktot);
            const long long key = lo + (long long)c * 128 + (long long)p;
            const unsigned below = (unsigned)(key > INT_MAX ? k1 - k0 : bd_lower_bound(A, k0, k1, (int)key) - k0);
            const bool wide = k1 - k0 >= 256;
            const unsigned at = ov_base + ovoff[c] + 1u;
            if (p == 0 && at < (unsigned)L.ov_cap) ov[at] = wide ? 1 : 0;
            if (wide) {
                if (at + 1u + p < (unsigned)L.ov_cap) ov[at + 1u + p] = (unsigned short)below;
            } else {
                const unsigned nb = (unsigned)__shfl_down((int)below, 1, 64);  // (p and p + 1 sit in one wave: 128 threads = two waves, p even pairs with p + 1)
                if ((p & 1u) == 0u && at + 1u + (p >> 1) < (unsigned)L.ov_cap) ov[at + 1u + (p >> 1)] = (unsigned short)(below | (nb << 8));
            }
        }
        // pass 2: the overflow entries themselves
        for (int r = r0 + (int)threadIdx.x; r < r1; r += BD_THREADS) {
            const int k = A[r];
            const bool first = r == r0 || A[r - 1] != k, last = r + 1 == r1 || A[r + 1] != k;
            if (last && !first) {
                const unsigned rel = (unsigned)((long long)k - lo);
                const unsigned c = rel >> 7, p = rel & 127u;
                if (ecnt[c] >= (unsigned)table_from) continue;  // (a table cell has no list)
                unsigned extras = (unsigned)(r - bd_lower_bound(A, r0, r, k));
                const unsigned ne = (extras + 254u) / 255u;
                unsigned j = atomicAdd(&ecur[c], ne);
                const unsigned at0 = ov_base + ovoff[c] + 1u;
                for (unsigned i = 0; i < ne; i++, j++) {
                    const unsigned chunk = extras < 255u ? extras : 255u;
                    extras -= chunk;
                    const unsigned more = j + 1u < ecnt[c] ? 0x8000u : 0u;
                    if (at0 + j < (unsigned)L.ov_cap) ov[at0 + j] = (unsigned short)(p | (chunk << 7) | more);
                }
            }
        }
        // the bitmap, as it lies
        {
            int4 *dst = reinterpret_cast<int4 *>(img + (arr == 0 ? L.bitsE : L.bitsS));
            for (int c = threadIdx.x; c < nc; c += BD_THREADS) dst[c] = reinterpret_cast<const int4 *>(bits)[c];
        }
        ov_base += otot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        hdr[12] = ov_base;
        hdr[15] = 0u;
        atomicMax(&stats[1], ov_base);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned o = __shfl_down(worst_block, off, 64);
        worst_block = o > worst_block ? o : worst_block;
    }
    if (lane_id() == 0) atomicMax(&stats[0], worst_block);
}

// ---------------------------------------------------------------------------
// the same walk on CELL images ("bp_*"): count_bitmap.hpp's 8-byte cells, one unit = 2^18 coordinates
// ---------------------------------------------------------------------------
// Measured on configs[1]: the dense lookups cost ~400 vector instructions per 16-byte slot and the search is bound by
// them (65 % VALU activity, 390-425 us whatever the record pipeline's depth), while the walk alone takes 225 us.  A cell
// of count_bitmap.hpp -- {bitmap of 32 coordinates, rank of the cell's first key : 20, one duplicated coordinate's
// (position, extra copies) : 12} -- answers a rank with ONE ds_read_b64 and ~13 instructions, duplicates included, at
// 2 bits per coordinate: a unit of 2^18 coordinates (two buckets of configs[1]), (tile, unit) runs of ~32 records.
// Their walk is slower (270 us: 128-byte runs) but nothing else is in its way.  A cell with two duplicated coordinates
// (or more than 126 extra copies) is HARD: its rank is finished by a binary search in the sorted array; indexes where
// those are not rare take the dense images, whose overflow lists hold any multiset.
constexpr int BP_UNIT_LOG2 = 18;
constexpr int BP_RSHIFT = 18;     // record = offset : 18 | length : 14 (16383 = escape)
constexpr int BP_MARGIN = 16384;

// A HARD cell (two or more duplicated coordinates among its 32, or more than 126 extra copies) gets a TABLE in the image's
// overflow area: 32 16-bit words, T[p] = #{keys of the cell below position p}; its bitmap word holds the table's byte
// offset inside the image.  Round 4: until then a hard cell's rank was finished by a binary search in the sorted array
// in HBM -- rare (3 in 10 000 look-ups on configs[1]) but 17 % of the walk's passes met one, and the wait for those loads
// drains the wave's whole memory pipe, record loads and count stores alike: 38 us of a 235 us kernel.  Cells beyond the
// area's room, or with 65536 keys and more, keep the search (offset word = all ones).
constexpr int BP_TABLES = 190;       // tables per unit image: what is left of the ninth 16 KB piece the image load moves anyway
constexpr unsigned BP_NO_TABLE = 0xFFFFFFFFu;

struct BpLayout {
    int nce, ncs;              // cells, sentinel included
    int cellsE, cellsS, hdr, ov, bytes;
};

// cell_log2: 5 = bitmap cells; 6..8 = offset cells (offset_cells.hpp: same 8 bytes per cell, same header, a smaller overflow
// area; the starts' margin is the longest query the record format holds)
// bytes16 (offset cells only): the image's size in 16-byte pieces when the index keeps RANK TABLES for its hard cells in a large
// overflow area ("clumped" offset cells, bo_image_kernel<.., TABLES>: duplicate-heavy data); 0 = the standard overflow area
__host__ __device__ inline BpLayout bp_layout(int unit_log2, int cell_log2 = 5, int bytes16 = 0)
{
    BpLayout L;
    const int UW = 1 << unit_log2;
    const int margin = cell_log2 == 5 ? BP_MARGIN : bo_margin(cell_log2);
    L.nce = (UW >> cell_log2) + 2;
    L.ncs = ((UW + margin) >> cell_log2) + 1;
    L.cellsE = 0;
    L.cellsS = L.nce * 8;
    L.hdr = (L.cellsS + L.ncs * 8 + 15) & ~15;  // header: [0] eLo, [1] sLo, [2..3] first coordinate of the unit (int64)
    L.ov = L.hdr + 16;                          // the hard cells' tables, 64 bytes each
    L.bytes = L.ov + (cell_log2 == 5 ? BP_TABLES : BO_TABLES) * 64;
    if (bytes16 && cell_log2 != 5) L.bytes = bytes16 << 4;
    return L;
}

// One workgroup per unit, one array at a time (bm_image_kernel for units; the duplicate bookkeeping of a cell is a
// bitmap of its duplicated coordinates and the number of extra copies: four words of LDS per cell instead of five).
// stats: [0] hard cells, [1] units whose slice holds 2^20 keys or more
__global__ __launch_bounds__(BD_THREADS) void bp_image_kernel(const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_sorted, int n,
                                                              BmGeom g, unsigned char *__restrict__ images, unsigned *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ int s_r[2];
    __shared__ int scan_tmp[16];
    __shared__ int s_ntab;  // tables handed out so far (both arrays share the area)
    if (threadIdx.x == 0) s_ntab = 0;
    const int unit = blockIdx.x;
    const int ulog = g.shift + g.f;
    const BpLayout L = bp_layout(ulog);
    const long long UW = 1ll << ulog;
    const long long lo = (long long)g.cmin + (long long)unit * UW;
    unsigned char *__restrict__ img = images + (size_t)unit * L.bytes;
    unsigned hard = 0;
    int r0s[2];
    for (int arr = 0; arr < 2; arr++) {
        const int32_t *__restrict__ A = arr == 0 ? e_sorted : s_ord;
        const int nc = arr == 0 ? L.nce : L.ncs;
        const long long span = arr == 0 ? UW + 1 : UW + BP_MARGIN;  // keys with rel in [0, span) belong to this image
        unsigned *bm = reinterpret_cast<unsigned *>(dyn);
        int *first = dyn + nc;
        unsigned *dmask = reinterpret_cast<unsigned *>(dyn + 2 * nc), *dcnt = reinterpret_cast<unsigned *>(dyn + 3 * nc);
        if (threadIdx.x < 2) s_r[threadIdx.x] = bm_rank_lt64(A, n, threadIdx.x == 0 ? lo : lo + span);
        __syncthreads();
        const int r0 = s_r[0], r1 = s_r[1], ns = r1 - r0;
        r0s[arr] = r0;
        for (int c = threadIdx.x; c < nc; c += BD_THREADS) {
            bm[c] = 0u;
            first[c] = ns;
            dmask[c] = 0u;
            dcnt[c] = 0u;
        }
        __syncthreads();
        for (int r = r0 + (int)threadIdx.x; r < r1; r += BD_THREADS) {
            const int k = A[r];
            const unsigned rel = (unsigned)((long long)k - lo);
            const int c = (int)(rel >> 5), p = (int)(rel & 31);
            if (r == r0 || A[r - 1] != k) {
                atomicOr(&bm[c], 1u << p);
                atomicMin(&first[c], r - r0);
            } else {
                atomicAdd(&dcnt[c], 1u);
                atomicOr(&dmask[c], 1u << p);
            }
        }
        __syncthreads();
        // suffix minimum of `first`: thread t owns chunk 1023 - t, so an exclusive scan in thread order covers the higher chunks
        const int K = (nc + BD_THREADS - 1) / BD_THREADS;
        const int chunk = BD_THREADS - 1 - (int)threadIdx.x;
        const int c_lo = chunk * K < nc ? chunk * K : nc, c_hi = c_lo + K < nc ? c_lo + K : nc;
        int run = INT_MAX;
        for (int c = c_hi - 1; c >= c_lo; c--) {
            run = first[c] < run ? first[c] : run;
            first[c] = run;
        }
        int tot;
        const int above = block_exclusive_scan(run, OpMin(), INT_MAX, scan_tmp, &tot);
        uint2 *out = reinterpret_cast<uint2 *>(img + (arr == 0 ? L.cellsE : L.cellsS));
        for (int c = c_lo; c < c_hi; c++) {
            const int base = first[c] < above ? first[c] : above;
            unsigned meta = (unsigned)base & 0xFFFFFu;
            unsigned word = bm[c];
            if (dcnt[c] > 0u) {
                if (__popc(dmask[c]) == 1 && dcnt[c] < (unsigned)BM_HARD)
                    meta |= ((unsigned)(__ffs((int)dmask[c]) - 1) << 20) | (dcnt[c] << 25);
                else {
                    meta |= (unsigned)BM_HARD << 25;
                    hard++;
                    // the cell's keys are A[r0 + base .. r0 + next): its rank table, if the area has room
                    int next = c + 1 < c_hi ? (first[c + 1] < above ? first[c + 1] : above) : above;
                    next = next < ns ? next : ns;
                    word = BP_NO_TABLE;
                    if (next - base < 65536) {
                        const int slot = atomicAdd(&s_ntab, 1);
                        if (slot < BP_TABLES) {
                            word = (unsigned)(L.ov + slot * 64);
                            unsigned short *tab = reinterpret_cast<unsigned short *>(img + word);
                            const long long cell0 = lo + (long long)c * 32;
                            for (int p = 0; p < 32; p++) {
                                const long long key = cell0 + p;
                                tab[p] = (unsigned short)(key > INT_MAX ? next - base : bd_lower_bound(A, r0 + base, r0 + next, (int)key) - (r0 + base));
                            }
                        }
                    }
                }
            }
            out[c] = make_uint2(word, meta);
        }
        if (threadIdx.x == 0 && ns >= (1 << 20)) atomicAdd(&stats[1], 1u);
        __syncthreads();
    }
    if (hard) atomicAdd(&stats[0], hard);
    if (threadIdx.x == 0) {
        unsigned *hdr = reinterpret_cast<unsigned *>(img + L.hdr);
        hdr[0] = (unsigned)r0s[0], hdr[1] = (unsigned)r0s[1];
        hdr[2] = (unsigned)(unsigned long long)lo, hdr[3] = (unsigned)((unsigned long long)lo >> 32);
    }
}

// Offset cells (offset_cells.hpp; g.dshift = cell width - 5): one workgroup per unit, one array at a time -- the keys of
// every cell counted with LDS atomics, a block scan for the cells' bases, then every cell packs its (at most five) keys
// straight from the sorted array; cells with more keys get a list in the overflow area while it has room.
// stats: [0] hard cells, [1] units whose slice holds 2^20 keys or more
// TABLES ("clumped" offset cells, round 6): every hard cell gets a RANK TABLE in the image's overflow area -- T[p] = keys of the cell
// below position p, a byte each (16 bits each for a cell of 256 keys and more) -- instead of a list of <= 64 offsets: a rank in a
// hard cell is then one more LDS read whatever the cell holds, and duplicate-heavy indexes (everything around a few thousand
// hot spots: hundreds of keys in a cell) can take the persistent walk.  The area is as large as the LDS allows (g.stride carries
// the image's size); stats[2] counts the cells that found no room -- such an index does not qualify.
// low word of a hard cell: byte offset of its table | BO_TABLE (| BO_TABLE_WIDE: 16-bit entries).
constexpr unsigned BO_TABLE = 0x40000000u, BO_TABLE_WIDE = 0x80000000u;
template <bool TABLES = false>
__global__ __launch_bounds__(BD_THREADS) void bo_image_kernel(const int32_t *__restrict__ s_ord, const int32_t *__restrict__ e_sorted, int n,
                                                              BmGeom g, unsigned char *__restrict__ images, unsigned *__restrict__ stats)
{
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ int s_r[2];
    __shared__ int scan_tmp[16];
    __shared__ int s_ntab;  // lists handed out so far (both arrays share the area)
    if (threadIdx.x == 0) s_ntab = 0;
    const int unit = blockIdx.x;
    const int k = 5 + g.dshift;
    const int ulog = g.shift + g.f;
    const BpLayout L = bp_layout(ulog, k, TABLES ? g.stride : 0);
    const long long UW = 1ll << ulog;
    const long long lo = (long long)g.cmin + (long long)unit * UW;
    unsigned char *__restrict__ img = images + (size_t)unit * L.bytes;
    unsigned hard = 0, homeless = 0;
    int r0s[2];
    for (int arr = 0; arr < 2; arr++) {
        const int32_t *__restrict__ A = arr == 0 ? e_sorted : s_ord;
        const int nc = arr == 0 ? L.nce : L.ncs;
        const long long span = arr == 0 ? UW + 1 : UW + bo_margin(k);  // keys with rel in [0, span) belong to this image
        unsigned *cnt = reinterpret_cast<unsigned *>(dyn);
        if (threadIdx.x < 2) s_r[threadIdx.x] = bm_rank_lt64(A, n, threadIdx.x == 0 ? lo : lo + span);
        for (int c = threadIdx.x; c < nc; c += BD_THREADS) cnt[c] = 0u;
        __syncthreads();
        const int r0 = s_r[0], r1 = s_r[1], ns = r1 - r0;
        r0s[arr] = r0;
        for (int r = r0 + (int)threadIdx.x; r < r1; r += BD_THREADS) {
            const unsigned rel = (unsigned)((long long)A[r] - lo);
            atomicAdd(&cnt[rel >> k], 1u);
        }
        __syncthreads();
        const int K = (nc + BD_THREADS - 1) / BD_THREADS;
        const int c_lo = (int)threadIdx.x * K < nc ? (int)threadIdx.x * K : nc, c_hi = c_lo + K < nc ? c_lo + K : nc;
        int sum = 0;
        for (int c = c_lo; c < c_hi; c++) sum += (int)cnt[c];
        int tot;
        int base = block_exclusive_scan(sum, OpSum(), 0, scan_tmp, &tot);
        uint2 *out = reinterpret_cast<uint2 *>(img + (arr == 0 ? L.cellsE : L.cellsS));
        for (int c = c_lo; c < c_hi; c++) {
            const int m = (int)cnt[c];
            const long long cell0 = lo + ((long long)c << k);
            const int32_t *keys = A + r0 + base;  // the cell's keys: A is sorted, the cells partition the coordinates
            unsigned lo_w, hi_w;
            if (m <= BO_INLINE) {
                unsigned char offs[BO_INLINE];
#pragma unroll
                for (int i = 0; i < BO_INLINE; i++) offs[i] = i < m ? (unsigned char)((long long)keys[i] - cell0) : (unsigned char)0xFF;
                bo_pack(offs, m, (unsigned)base, lo_w, hi_w);
            } else {
                hard++;
                lo_w = BP_NO_TABLE, hi_w = ((unsigned)base & 0xFFFFFu) | BO_HARD;
                if (TABLES) {
                    const bool wide = m >= 256;
                    const int bytes = (wide ? 2 : 1) << k;  // (a multiple of 64: the area stays 16-byte aligned)
                    const int at = m <= 65535 ? atomicAdd(&s_ntab, bytes) : L.bytes;  // (a table's entries are 16 bits at most: a bigger pile finds no room)
                    if (L.ov + at + bytes <= L.bytes) {
                        lo_w = (unsigned)(L.ov + at) | BO_TABLE | (wide ? BO_TABLE_WIDE : 0u);
                        int i = 0;
                        for (int p = 0; p < (1 << k); p++) {
                            while (i < m && (long long)keys[i] - cell0 < (long long)p) i++;
                            if (wide)
                                reinterpret_cast<unsigned short *>(img + L.ov + at)[p] = (unsigned short)i;
                            else
                                img[L.ov + at + p] = (unsigned char)i;
                        }
                    } else {
                        homeless++;
                    }
                } else if (m <= BO_LIST) {
                    const int slot = atomicAdd(&s_ntab, 1);
                    if (slot < BO_TABLES) {
                        lo_w = (unsigned)(L.ov + slot * 64);
                        unsigned char *list = img + lo_w;
                        for (int i = 0; i < BO_LIST; i++) list[i] = i < m ? (unsigned char)((long long)keys[i] - cell0) : (unsigned char)0xFF;
                    }
                }
            }
            out[c] = make_uint2(lo_w, hi_w);
            base += m;
        }
        if (threadIdx.x == 0 && ns >= (1 << 20)) atomicAdd(&stats[1], 1u);
        __syncthreads();
    }
    if (hard) atomicAdd(&stats[0], hard);
    if (homeless) atomicAdd(&stats[2], homeless);
    if (threadIdx.x == 0) {
        unsigned *hdr = reinterpret_cast<unsigned *>(img + L.hdr);
        hdr[0] = (unsigned)r0s[0], hdr[1] = (unsigned)r0s[1];
        hdr[2] = (unsigned)(unsigned long long)lo, hdr[3] = (unsigned)((unsigned long long)lo >> 32);
    }
}

// ---------------------------------------------------------------------------
// the run table of the flat walk
// ---------------------------------------------------------------------------
// The walk needs one number per (unit, tile): the first slot of the unit's first bucket in the tile's sorted order (a
// run ends where the next unit's begins).  tbl[tile][bucket] -> unitT[unit][tile] (16 bits) and
// unitcnt[group][unit] = queries of the unit in the 64 tiles of the group, which is what the plan cuts into work items:
// 12.5 MB read and 6 MB written per 100 M queries where bm_transpose_kernel + sl_unit_sums_kernel moved 75 MB.
// One workgroup per (group of 64 tiles, 64 buckets), as there.
// unitT has one row more than there are units: where the last unit's run ends = the slots the tile uses (its queries,
// or `tend` when the tile sort left gaps between the units: PAD).
__global__ __launch_bounds__(256) void bd_transpose_kernel(const unsigned short *__restrict__ tbl, const BmSeg *__restrict__ segs,
                                                           const unsigned short *__restrict__ tile_seg, int tile_log2,
                                                           unsigned short *__restrict__ unitT /* [(BM_NB >> f) + 1][ntp] */, int64_t ntp,
                                                           unsigned *__restrict__ unitcnt /* [ngroups][BM_NB] */, const unsigned *__restrict__ gate,
                                                           const unsigned *__restrict__ tend /* PAD: slots used per tile, else NULL */)
{
    __shared__ unsigned short t[BM_GROUP_TILES][66];
    if (gate && *gate == 0) return;
    const int grp = blockIdx.x, b0 = blockIdx.y * 64;
    const int f = segs[tile_seg[(int64_t)grp * BM_GROUP_TILES]].g.f;  // (a plan group never straddles two segments)
    if (blockIdx.y == 0)  // the plan walks BM_NB columns: the ones past the last unit are empty
        for (int u = (BM_NB >> f) + (int)threadIdx.x; u < BM_NB; u += 256) unitcnt[(int64_t)grp * BM_NB + u] = 0u;
    {
        const int r = threadIdx.x >> 2, q = threadIdx.x & 3;  // 4 threads per tile row, 16 buckets each
        const int64_t tile = (int64_t)grp * BM_GROUP_TILES + r;
        const BmSeg &sg = segs[tile_seg[tile]];
        const bool live = tile - sg.tile0 < sg.ntiles;
        const int64_t left = sg.nq - ((tile - sg.tile0) << tile_log2);
        unsigned ntile = !live ? 0u : (left < ((int64_t)1 << tile_log2) ? (unsigned)left : 1u << tile_log2);
        if (tend && live) ntile = tend[tile];
        const unsigned short *row = tbl + tile * BM_NB + b0 + 16 * q;
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        if (live) {
            a = *reinterpret_cast<const uint4 *>(row);
            c = *reinterpret_cast<const uint4 *>(row + 8);
        }
        const unsigned w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            t[r][16 * q + 2 * i] = (unsigned short)(w[i] & 0xffffu);
            t[r][16 * q + 2 * i + 1] = (unsigned short)(w[i] >> 16);
        }
        // (a full tile's total is 1 << 16 when the tile has 65536 queries: lengths are taken modulo 2^16 below)
        if (q == 3) {
            t[r][64] = (unsigned short)(b0 + 64 < BM_NB ? (live ? row[16] : 0) : ntile);
            if (b0 + 64 >= BM_NB) unitT[(int64_t)(BM_NB >> f) * ntp + tile] = (unsigned short)ntile;  // the row behind the last unit
        }
    }
    __syncthreads();
    {
        // 64 >> f units in this patch, 64 tiles each: thread = (unit, 4 tiles ... ) laid out so that a unit's 64 tiles
        // are written as whole 16-byte pieces
        const int nu = 64 >> f;                      // units of the patch (>= 1: f <= 6)
        const int per_unit = 256 / nu;               // threads per unit: 4 (f = 0) .. 256 (f = 6)
        const int u = threadIdx.x / per_unit, k = threadIdx.x % per_unit;
        const int tiles_per_thread = 64 / per_unit;  // 16 .. (for per_unit > 64 some threads idle)
        const int c0 = u << f, c1 = c0 + (1 << f);   // bucket columns of the unit inside the patch
        unsigned sum = 0;
        if (tiles_per_thread >= 1) {
            unsigned short *dst = unitT + (int64_t)((b0 >> f) + u) * ntp + (int64_t)grp * BM_GROUP_TILES + k * tiles_per_thread;
            for (int i = 0; i < tiles_per_thread; i++) {
                const int r = k * tiles_per_thread + i;
                dst[i] = t[r][c0];
                sum += (unsigned)(unsigned short)(t[r][c1] - t[r][c0]);
            }
        } else if (k < 64) {  // more threads than tiles: one tile per thread
            unitT[(int64_t)((b0 >> f) + u) * ntp + (int64_t)grp * BM_GROUP_TILES + k] = t[k][c0];
            sum = (unsigned)(unsigned short)(t[k][c1] - t[k][c0]);
        }
        // per-unit sum over its threads (per_unit is a power of two, 4 .. 256)
        __shared__ unsigned s_sum[256];
        s_sum[threadIdx.x] = sum;
        __syncthreads();
        if (k == 0) {
            unsigned tot = 0;
            for (int i = 0; i < per_unit; i++) tot += s_sum[threadIdx.x + i];
            unitcnt[(int64_t)grp * BM_NB + (b0 >> f) + u] = tot;
        }
    }
}

// The plan of a batch over up to BD_PLAN_SEGS indexes (bm_plan_kernel<2> walks every column twice, three dependent batches of loads
// each time, from eight workgroups that reserve their room with an atomic: 20 us for 100 M queries, 13 us for a rank's three
// chromosomes).  One workgroup; a thread per (segment, unit), the column's group counts of ITS segment pulled 16 at a time with
// independent loads, the items of a unit kept in registers (up to six; a unit cut into more is walked a second time), one
// block scan for their places: segments and units stay in order, which is what the search's XCD-aware item mapping wants.
// items[i] = {unit | segment << 16, first tile, last tile + 1, queries}; *n_items = number of items.
// (Tried in round 6: run table and plan in ONE launch, the plan made by the workgroup that draws the last ticket, the unit
// counts written through as agent-scope atomics -- 21.5 us against 8.2 + 10.3 for one index: every workgroup waits for its
// stores and a returning atomic, and the planner reads the counts from beyond the L2.  With __threadfence() instead: 153 us --
// an agent-scope release writes the XCD's whole L2 back.)
constexpr int BD_PLAN_SEGS = 32;  // (a genome of 24 chromosomes on one GPU: 12 us where bm_plan_kernel<2> took 27)
__global__ __launch_bounds__(1024) void bd_plan_kernel(const unsigned *__restrict__ unitcnt /* [ngroups][BM_NB] */, int n_segs, const BmSeg *__restrict__ segs,
                                                       int chunk, int4 *__restrict__ items, int *__restrict__ n_items, const unsigned *__restrict__ gate)
{
    __shared__ int scan_tmp[16];
    __shared__ int s_first[BD_PLAN_SEGS + 1], s_glo[BD_PLAN_SEGS], s_ghi[BD_PLAN_SEGS], s_tlast[BD_PLAN_SEGS];
    if (gate && *gate == 0) return;
    if (threadIdx.x == 0) {
        int at = 0;
        for (int i = 0; i < n_segs; i++) {
            s_first[i] = at;
            at += segs[i].tile_end > segs[i].tile0 ? (BM_NB >> segs[i].g.f) : 0;  // (an index without queries in this batch: no units)
            s_glo[i] = (int)(segs[i].tile0 / BM_GROUP_TILES), s_ghi[i] = (int)(segs[i].tile_end / BM_GROUP_TILES);
            s_tlast[i] = (int)(segs[i].tile0 + segs[i].ntiles);
        }
        s_first[n_segs] = at;
    }
    __syncthreads();
    const int total_units = s_first[n_segs];
    int carry = 0;
    for (int j0 = 0; j0 < total_units; j0 += 1024) {
        const int j = j0 + (int)threadIdx.x;
        const bool live = j < total_units;
        int seg = 0;
        while (seg + 1 < n_segs && (live ? j : 0) >= s_first[seg + 1]) seg++;
        const int u = live ? j - s_first[seg] : 0;
        const int g_lo = s_glo[seg], ng = live ? s_ghi[seg] - g_lo : 0, t_last = s_tlast[seg];
        const unsigned *__restrict__ col = unitcnt + (int64_t)g_lo * BM_NB + u;
        int fb[6], fe[6];
        unsigned fq[6];
        int cnt = 0;
        auto walk = [&](bool emit, int at) {
            unsigned acc = 0;
            int g_first = 0, k = 0;
            auto close = [&](int g_end) {
                if (emit) {
                    const int t_end = (g_lo + g_end) * BM_GROUP_TILES;
                    items[at + k] = make_int4(u | (seg << 16), (g_lo + g_first) * BM_GROUP_TILES, t_end < t_last ? t_end : t_last, (int)acc);
                } else {
#pragma unroll
                    for (int i = 0; i < 6; i++)
                        if (k == i) fb[i] = g_first, fe[i] = g_end, fq[i] = acc;
                }
                k++;
                acc = 0;
            };
            for (int g0 = 0; g0 < ng; g0 += 16) {
                unsigned v[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int gi = g0 + i < ng ? g0 + i : ng - 1;  // a valid address: no branch around the loads
                    v[i] = col[(int64_t)gi * BM_NB];
                }
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int gi = g0 + i;
                    if (gi >= ng) break;
                    const unsigned c = v[i];
                    if (acc > 0 && acc + c > (unsigned)chunk) close(gi);
                    if (acc == 0) g_first = gi;
                    acc += c;
                }
            }
            if (acc > 0) close(ng);
            return k;
        };
        cnt = walk(false, 0);
        int tot;
        const int at = carry + block_exclusive_scan(cnt, OpSum(), 0, scan_tmp, &tot);
        if (cnt <= 6) {
#pragma unroll
            for (int i = 0; i < 6; i++)
                if (i < cnt) {
                    const int t_end = (g_lo + fe[i]) * BM_GROUP_TILES;
                    items[at + i] = make_int4(u | (seg << 16), (g_lo + fb[i]) * BM_GROUP_TILES, t_end < t_last ? t_end : t_last, (int)fq[i]);
                }
        } else {
            (void)walk(true, at);
        }
        carry += tot;
    }
    if (threadIdx.x == 0) *n_items = carry;
}

// ---------------------------------------------------------------------------
// search
// ---------------------------------------------------------------------------
typedef unsigned bd_v4u __attribute__((ext_vector_type(4)));
typedef unsigned bd_v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const bd_v4u *lds_v4u_p;
typedef __attribute__((address_space(3))) const unsigned *lds_u32_p;
typedef __attribute__((address_space(3))) const unsigned char *lds_u8_p;

struct BdImage {
    lds_v4u_p bitsE, bitsS;
    lds_u16_p metaE, metaS, ov;
    lds_u32_p qbE, qbS;
    int bias;          // sLo - eLo
    unsigned off_mask;  // offsets of a record: the unit's width - 1
    // cell images (FMT 1)
    lds_cell_p cE, cS;
    lds_u16_p img16;   // the image as 16-bit words: the hard cells' rank tables are addressed by their byte offset
    // offset cells (offset_cells.hpp): the image as bytes (the hard cells' lists), the record's length shift, the cell width
    lds_u8_p img8;
    int rshift, cell_log2;
    unsigned cell_mask;
    int eLo, sLo;
    long long lo;
    const int32_t *s_ord, *e_sorted;
    // key slices (FMT 2: count_slices.hpp's staged unit; sparse indexes, duplicated coordinates of any kind)
    SlUnit sl;
    BmGeom g;
};

// FMT 1: a rank from a unit's cell image (bm_cell_rank written for the instruction count: the walk is bound by vector
// instructions -- 274 per 16-byte slot, 58 % VALU activity at 330 us -- so the duplicate term is one bit-field extract
// and one multiply-add, and the hard-cell test is left to the caller, once per record).
__device__ __forceinline__ unsigned bp_cell_rank(lds_cell_p cells, unsigned rel, unsigned &meta_out)
{
    const unsigned long long c = cells[rel >> 5];
    const unsigned bits = (unsigned)c, meta = (unsigned)(c >> 32);
    meta_out = meta;
    const unsigned below = ~(0xFFFFFFFFu << (rel & 31u));  // the coordinates of the cell below rel
    // the duplicated coordinate counts `extra` more times when it lies below rel: bit dpos of the mask says so
    const unsigned dup = __builtin_amdgcn_ubfe(below, (meta >> 20) & 31u, 1u);
    return (meta & 0xFFFFFu) + (unsigned)__popc(bits & below) + (meta >> 25) * dup;
}

// a hard cell's rank: from its table in LDS, or (no room for a table) from the sorted array
__device__ __forceinline__ int bp_hard_rank(const BdImage &I, lds_cell_p cells, unsigned rel, const int32_t *__restrict__ a, int slice_lo)
{
    const bd_v2u c = __builtin_bit_cast(bd_v2u, cells[rel >> 5]);
    if (c.x != BP_NO_TABLE) return (int)(c.y & 0xFFFFFu) + (int)I.img16[(c.x >> 1) + (rel & 31u)];
    return bm_hard_rank(cells, rel, a, slice_lo, I.lo);
}

// the count of one record (16 bits, 0xFFFF = ask the index again; RAW: all 32 bits, an escape record counts 0)
template <bool RAW = false>
__device__ __forceinline__ unsigned bp_count_record(const BdImage &I, unsigned rec)
{
    const unsigned off = rec & I.off_mask, len = rec >> BP_RSHIFT;
    const unsigned relE = off + 1u, relS = off + len;
    unsigned mE, mS;
    const unsigned rE = bp_cell_rank(I.cE, relE, mE);
    const unsigned rS = bp_cell_rank(I.cS, relS, mS);
    unsigned c = (unsigned)I.bias + (rS - rE);
    if ((mE > mS ? mE : mS) >= ((unsigned)BM_HARD << 25)) {  // a hard cell (rare: the index qualifies only while they are)
        int hE = (int)rE, hS = (int)rS;
        if ((mE >> 25) == (unsigned)BM_HARD) hE = bp_hard_rank(I, I.cE, relE, I.e_sorted, I.eLo);
        if ((mS >> 25) == (unsigned)BM_HARD) hS = bp_hard_rank(I, I.cS, relS, I.s_ord, I.sLo);
        c = (unsigned)(I.bias + (hS - hE));
    }
    if (RAW) return rec == BM_REC_ESC ? 0u : c;
    c = c < 0xFFFFu ? c : 0xFFFFu;
    return rec == BM_REC_ESC ? 0xFFFFu : c;
}

// One lookup: read, wait, compute, and a branch for the cell with duplicated coordinates (taken by some lane of the wave
// in nearly every lookup).  (A three-step form that keeps a slot's eight lookups' LDS reads in flight together -- all
// cells, then {ov[i], ov[i+1], ov[i+2]} with i = 0 for plain cells, then the ranks -- measured no faster and needs twice
// the registers.)
// QB: the image's ranks are relative to blocks of 1024 cells (a table read per lookup); otherwise one block spans the
// unit (fewer than 2^15 keys per slice: configs[1] has 22 000) and a plain cell's 16 bits are its rank in the unit.
template <bool QB>
__device__ __forceinline__ int bd_rank(lds_v4u_p bits, lds_u16_p meta, lds_u32_p qb, lds_u16_p ov, unsigned rel)
{
    const unsigned c = rel >> 7, p = rel & 127u;
    const bd_v4u w = bits[c];
    unsigned m = meta[c];
    unsigned q = 0u;
    if (QB) q = qb[c >> 10];
    const unsigned long long lo = (unsigned long long)w.x | ((unsigned long long)w.y << 32);
    const unsigned long long hi = (unsigned long long)w.z | ((unsigned long long)w.w << 32);
    const bool up = p >= 64u;
    const unsigned long long below = (1ull << (p & 63u)) - 1ull;
    int r = __popcll((up ? hi : lo) & below) + (up ? __popcll(lo) : 0);
    if (m & 0x8000u) {
        unsigned i = m & 0x7FFFu;
        m = ov[i];
        if (m & 0x8000u) {  // a clumped cell: its ranks are tabulated, as bytes or as 16-bit words
            m &= 0x7FFFu;
            const unsigned wide = ov[i + 1u];
            const unsigned w = ov[i + 2u + (wide ? p : p >> 1)];
            r = (int)(wide ? w : (w >> ((p & 1u) << 3)) & 255u);
        } else {
            unsigned e;
            do {
                e = ov[++i];
                r += (e & 127u) < p ? (int)((e >> 7) & 255u) : 0;
            } while (e & 0x8000u);
        }
    }
    return (int)(q + m) + r;
}

// 16 bits of count: 0xFFFF = "ask the index again" (an escape record, or a count that does not fit)
__device__ __forceinline__ unsigned bd_count16(int bias, int rS, int rE, unsigned rec)
{
    unsigned c = (unsigned)(bias + (rS - rE));
    c = c < 0xFFFFu ? c : 0xFFFFu;
    return rec == BM_REC_ESC ? 0xFFFFu : c;
}

// four records of one aligned 16-byte slot -> four 16-bit counts at the same positions of the count array;
// `valid` bit j = record j belongs to this workgroup's run (the others are a neighbouring unit's: looked up all the
// same -- any 32-bit word is a safe argument, the offset is masked to the unit and the length cannot leave the margin --
// and not stored)
// EXP (diagnostics, ivl.bd_exp; wrong results): 1 = no lookups at all -- the price of the walk and of its memory traffic alone
// FMT: 0 = dense unit image, 2 = staged key slices (cell images, once FMT 1 of this kernel, have the persistent walk below: bw_search_kernel)
// W8: the counts are 8 bits wide (0xFF = "ask the index again": escape records and counts of 255 and more) -- half the bytes
// for indexes whose counts are small; the host decides per batch (bm_count_segments).
template <int FMT, bool QB, int EXP, bool W8 = false>
__device__ __forceinline__ void bd_answer_slot(const BdImage &I, unsigned short *__restrict__ out, unsigned idx4, unsigned valid, bd_v4u v)
{
    if (valid == 0u) return;
    const unsigned rec[4] = {v.x, v.y, v.z, v.w};
    unsigned c[4];
    if (EXP == 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = rec[j] & 0xffu;
    } else if (FMT == 2) {
        // (a neighbouring unit's record is a valid argument: its offset lies inside the unit's width, the directory
        // lookups stay inside the staged arrays)
        sl_count_slot(I.sl, I.g, rec, c);
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = c[j] < 0xFFFFu ? c[j] : 0xFFFFu;  // (BM_REC_ESC included)
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned off = rec[j] & I.off_mask;
            const int rE = bd_rank<QB>(I.bitsE, I.metaE, I.qbE, I.ov, off + 1u);
            const int rS = bd_rank<QB>(I.bitsS, I.metaS, I.qbS, I.ov, off + (rec[j] >> BD_RSHIFT));
            c[j] = bd_count16(I.bias, rS, rE, rec[j]);
        }
    }
    unsigned short *p = out + 4 * (size_t)idx4;
    if (EXP == 3) {  // diagnostics: the lookups alone -- nothing stored unless a count is impossible
        if ((c[0] & c[1] & c[2] & c[3]) == 0x12345u) p[0] = 1;
        return;
    }
    if (W8) {
        unsigned char *p8 = reinterpret_cast<unsigned char *>(out) + 4 * (size_t)idx4;
#pragma unroll
        for (int j = 0; j < 4; j++) c[j] = c[j] < 0xFFu ? c[j] : 0xFFu;
        if (valid == 15u) {
            *reinterpret_cast<unsigned *>(p8) = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (valid & (1u << j)) p8[j] = (unsigned char)c[j];
        }
        return;
    }
    if (valid == 15u) {
        bd_v2u o;
        o.x = c[0] | (c[1] << 16), o.y = c[2] | (c[3] << 16);
        *reinterpret_cast<bd_v2u *>(p) = o;
    } else {
        if (valid & 1u) p[0] = (unsigned short)c[0];
        if (valid & 2u) p[1] = (unsigned short)c[1];
        if (valid & 4u) p[2] = (unsigned short)c[2];
        if (valid & 8u) p[3] = (unsigned short)c[3];
    }
}

// records [ra, re) of a tile against the four records of the 16-byte slot that starts at record f0 (the slot meets the run)
__device__ __forceinline__ unsigned bd_valid_mask(unsigned f0, unsigned ra, unsigned re)
{
    const int lo = max((int)(ra - f0), 0), hi = min((int)(re - f0), 4);
    return ((1u << hi) - 1u) & ~((1u << lo) - 1u);
}

// DEPTH: passes whose records are in flight per wave.  The walk alone (no lookups) moves 0.72 GB in 225 us with two:
// neither bandwidth (3.2 TB/s, no read amplification: FETCH_SIZE = the records once) nor instructions, but 2 KB per
// wave in flight against ~2 us of loaded HBM latency.
// The record loads of the pipelined walk are issued by hand: the compiler does not know them, so it neither waits for
// them nor drains the memory pipe in front of every pass (which is what its own bookkeeping does to loads that stay in
// flight around a loop: measured, every pass waited for its own stores).  bd_wait<K> is the wait: K = the number of
// memory operations known to have been issued after the wanted load -- the counter retires in order.
// BD_LOAD_NT (diagnostics / tuning): 1 = the record loads are non-temporal -- the records stream through L2 once, the
// count lines a workgroup and its neighbours are filling should outlive them there
#ifndef BD_LOAD_NT
#define BD_LOAD_NT 0
#endif
__device__ __forceinline__ void bd_issue_load(bd_v4u &v, const unsigned *recs, unsigned idx4)
{
    const bd_v4u *p = reinterpret_cast<const bd_v4u *>(recs) + idx4;
#if BD_LOAD_NT == 1
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
#elif BD_LOAD_NT == 2
    asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
#elif BD_LOAD_NT == 3
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
#else
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
#endif
}

// (the register is named in a comment of the instruction: tools/check_ring_isa.py reads the compiled code and refuses a
// build in which the compiler moved or touched a register while a hand-issued load was on its way to it)
template <int K>
__device__ __forceinline__ void bd_wait(bd_v4u &v)
{
    asm volatile("s_waitcnt vmcnt(%1) ; ring %0" : "+v"(v) : "n"(K) : "memory");
}

// One memory operation that nobody waits for: a 4-byte store to a slot no query owns.  It stands in for "the answer of
// the pass before" where the ring starts, so that one wait count fits every pass.
__device__ __forceinline__ void bd_dummy_store(unsigned short *slot)
{
    asm volatile("global_store_dword %0, %1, off" : : "v"(slot), "v"(0u) : "memory");
}

// PIPE: two sets of DEPTH passes; while one set is answered the other's records are on their way.
// PAD: the tile sort left every unit's run on whole 16-byte slots (bm_tile_sort_kernel<.., PAD>): no slot is shared with a
// neighbouring unit, every answered pass is exactly one store, and the walk keeps a RING of DEPTH passes in flight all
// the time -- the wait in front of a pass counts the DEPTH - 1 younger loads and the DEPTH - 1 stores issued since.
template <int FMT, bool QB, int EXP = 0, int DEPTH = 2, bool PIPE = false, bool PAD = false, bool W8 = false>
__global__ __launch_bounds__(BD_THREADS) void bd_search_kernel(const BmSeg *__restrict__ segs, const int4 *__restrict__ items,
                                                               const int *__restrict__ n_items, const unsigned short *__restrict__ unitT, int64_t ntp,
                                                               const unsigned *__restrict__ recs /* tile-sorted records */,
                                                               unsigned short *__restrict__ out /* their counts, same order */, int tile_log2,
                                                               const unsigned *__restrict__ gate)
{
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ uint2 s_long[BD_LONG_CAP];  // {first record, length} of the long runs met during the walk
    __shared__ int s_nlong, s_next;
    __shared__ int s_tmp[20];  // FMT 2: sl_stage_unit's scratch
    const int nit = *n_items;
    const int per_xcd = (nit + 7) >> 3;
    const int slot = (int)(blockIdx.x >> 3);
    const int it = (int)(blockIdx.x & 7) * per_xcd + slot;  // neighbouring units on one XCD (see bm_search_kernel)
    if (slot >= per_xcd || it >= nit) return;
    const int4 item = items[it];
    const int unit = item.x & 0xffff, t0 = item.y, t1 = item.z;
    const BmSeg &sg = segs[item.x >> 16];
    const BmGeom g = sg.g;
    const BdLayout L = bd_layout(g.shift + g.f);
    const int image_bytes = L.bytes;
    const unsigned short *__restrict__ runs0 = unitT + (int64_t)unit * ntp;
    const unsigned short *__restrict__ runs1 = runs0 + ntp;  // (the next unit's first slots, or the row behind the last unit)
    const int lane = lane_id();
    const unsigned tile_slots = ((1u << tile_log2) + (PAD ? (unsigned)BM_PAD_ROOM : 0u)) >> 2;  // 16-byte slots between two tiles
    // tiles per batch of a wave: 64 when the item has plenty (configs[1]: 3052 tiles, 48 batches for 16 waves), fewer when it
    // does not -- a chromosome's item of 120 tiles in batches of 64 kept two of the sixteen waves busy (genome pass 2.3 ms
    // instead of 1.0); at least 8, so that a batch is still a few passes long
    int B = 64;
    while (B > 8 && (t1 - t0) < 2 * (BD_THREADS / 64) * B) B >>= 1;
    if (threadIdx.x == 0) s_nlong = 0, s_next = B * (BD_THREADS / 64);  // (every wave starts with the batch of its number)
    // this wave's first B tiles: their runs travel with the image
    int tb = t0 + B * (int)(threadIdx.x >> 6);
    unsigned a_nx, e_nx;
    auto load_runs = [&](int tbase) {
        const int t = tbase + lane;
        const int tc = t < t1 && lane < B ? t : t0;  // a valid address: no branch around the loads
        const unsigned a = runs0[tc];
        const unsigned e = runs1[tc];
        a_nx = t < t1 && lane < B ? a : 0u;
        e_nx = t < t1 && lane < B ? e : 0u;
    };
    load_runs(tb);
    BdImage I;
    if (FMT == 2) {
        I.sl = sl_stage_unit(sg, unit, dyn, s_tmp);
        I.g = g;
    } else {
        // the image (streams through L2 once: non-temporal loads); every load of a lane issued before its first LDS store
        const bm_v4i BX_GLOBAL *src = reinterpret_cast<const bm_v4i BX_GLOBAL *>(as_global(sg.dimages) + (size_t)unit * image_bytes);
        const int n4 = L.ov >> 4;  // (the overflow area follows, as much of it as is used)
        constexpr int SWEEPS = 5;
        for (int i0 = 0; i0 < n4; i0 += SWEEPS * BD_THREADS) {
            bm_v4i v[SWEEPS];
#pragma unroll
            for (int k = 0; k < SWEEPS; k++) {
                const int i = i0 + k * BD_THREADS + (int)threadIdx.x;
                v[k] = __builtin_nontemporal_load(src + (i < n4 ? i : n4 - 1));
            }
#pragma unroll
            for (int k = 0; k < SWEEPS; k++) {
                const int i = i0 + k * BD_THREADS + (int)threadIdx.x;
                if (i < n4) reinterpret_cast<bm_v4i *>(dyn)[i] = v[k];
            }
        }
    }
    __syncthreads();
    if (FMT == 0) {
        const bm_v4i BX_GLOBAL *src = reinterpret_cast<const bm_v4i BX_GLOBAL *>(as_global(sg.dimages) + (size_t)unit * image_bytes + L.ov);
        const int used = (int)reinterpret_cast<const unsigned *>(reinterpret_cast<unsigned char *>(dyn) + L.hdr)[12];  // overflow entries of this unit
        const int n4 = (used * 2 + 15) >> 4;
        for (int i = threadIdx.x; i < n4; i += BD_THREADS) reinterpret_cast<bm_v4i *>(reinterpret_cast<unsigned char *>(dyn) + L.ov)[i] = __builtin_nontemporal_load(src + i);
        __syncthreads();
    }
    if (FMT == 2) {
    } else {
        unsigned char *base = reinterpret_cast<unsigned char *>(dyn);
        I.bitsE = (lds_v4u_p) reinterpret_cast<bd_v4u *>(base + L.bitsE);
        I.bitsS = (lds_v4u_p) reinterpret_cast<bd_v4u *>(base + L.bitsS);
        I.metaE = (lds_u16_p) reinterpret_cast<unsigned short *>(base + L.metaE);
        I.metaS = (lds_u16_p) reinterpret_cast<unsigned short *>(base + L.metaS);
        I.ov = (lds_u16_p) reinterpret_cast<unsigned short *>(base + L.ov);
        I.qbE = (lds_u32_p) reinterpret_cast<unsigned *>(base + L.hdr);
        I.qbS = I.qbE + BD_HDR_QS;
        const unsigned *hdr = reinterpret_cast<const unsigned *>(base + L.hdr);
        I.bias = (int)hdr[14] - (int)hdr[13];
        I.off_mask = (1u << (g.shift + g.f)) - 1u;
    }
    while (tb < t1) {
        const int t = tb + lane;
        const unsigned a = a_nx, e = e_nx;
        // the batch after this one: its number from the workgroup's counter, its runs requested now
        int tn = 0;
        if (lane == 0) tn = atomicAdd(&s_next, B);
        tn = t0 + __builtin_amdgcn_readfirstlane(tn);
        load_runs(tn);
        unsigned n4 = e > a ? ((e + 3u) >> 2) - (a >> 2) : 0u;  // 16-byte slots that hold the run
        if (n4 > (unsigned)BD_LONG_SLOTS) {  // sorted / clumped input: left to the whole workgroup
            const int k = atomicAdd(&s_nlong, 1);
            if (k < BD_LONG_CAP) {
                s_long[k] = make_uint2(((unsigned)t * tile_slots << 2) + a, e - a);
                n4 = 0u;
            }  // (a full list: the wave walks the run itself, exactness never depends on it)
        }
        const unsigned incl = wave_inclusive_scan(n4, OpSum());
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
        // slot s of the flat sequence lies in run r = #{runs that end at or before s}; its int4 is at delta[r] + s
        const unsigned delta = ((unsigned)t * tile_slots + (a >> 2)) - (incl - n4);
        const unsigned ae = a | (e << 15);  // a < 2^15, e <= 2^15
        bd_v4u ring_v[DEPTH];
        unsigned ring_idx[DEPTH], ring_valid[DEPTH];
        auto prep = [&](unsigned s0, unsigned &idx4, unsigned &valid, bd_v4u &v, bool by_hand = false) {
            unsigned at = 0u, ok = 0u;
            if (!PAD || s0 < total) {  // (the ring asks for passes behind the batch's last: record 0, nothing to answer)
                const unsigned s = s0 + (unsigned)lane;
                int k = (int)__popcll(__ballot(incl <= s0));  // runs that end at or before the pass's first slot
                unsigned r = (unsigned)k;
                for (; k < 63; k++) {
                    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)incl, k);
                    if (c > s0 + 63u) break;
                    r += s >= c ? 1u : 0u;
                }
                r = r < 63u ? r : 63u;
                const unsigned d = (unsigned)__shfl((int)delta, (int)r, 64), x = (unsigned)__shfl((int)ae, (int)r, 64);
                const bool active = s < total;
                at = active ? d + s : 0u;
                if (PAD)
                    ok = active ? 15u : 0u;
                else
                    ok = active ? bd_valid_mask((at & ((1u << (tile_log2 - 2)) - 1u)) << 2, x & 0x7fffu, x >> 15) : 0u;
            }
            idx4 = at, valid = ok;
            if (EXP == 3) {  // diagnostics: no record loads -- synthetic records (offsets all over the unit, lengths < 1000)
                const unsigned h = idx4 * 2654435761u;
                v = bd_v4u{(h & 0x3ffffu) | (500u << 18), ((h >> 3) & 0x3ffffu) | (100u << 18), ((h >> 7) & 0x3ffffu) | (900u << 18),
                           ((h >> 11) & 0x3ffffu) | (300u << 18)};
            } else if (by_hand)
                bd_issue_load(v, recs, idx4);
            else
                v = reinterpret_cast<const bd_v4u *>(recs)[idx4];
        };
        // DEPTH passes at a time: their records requested together, then answered one after the other (the compiler's
        // wait counts only work out inside one iteration: with a ring carried around the loop it drains the memory
        // pipe -- loads AND the stores of the pass before -- in front of every pass)
        if (PIPE && PAD) {
            // a ring of DEPTH passes: in front of a pass, the DEPTH - 1 other slots were (re)loaded after its own load
            // and DEPTH - 1 passes stored once each.
            // The memory pipe sees  [store, load] x DEPTH  at the start (the stores are dummies) and  [answer's store, load]
            // per pass afterwards: 2 * DEPTH - 2 operations follow every load before its pass comes round again.  Passes
            // past the end of the batch load record 0 and store a dummy: the count has to hold for them too -- the
            // registers of a pass are scratch for the next load's address as soon as its wait is over (measured the hard
            // way: a late record landing in the middle of an address computation is a memory fault).
            // the end of a tile's room: past every unit's padding
            unsigned short *nobody = W8 ? reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(out) + (((size_t)tb + 1) * tile_slots << 2) - 4)
                                        : out + (((size_t)tb + 1) * tile_slots << 2) - 2;
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                if (d > 0) bd_dummy_store(nobody);
                prep(64u * d, ring_idx[d], ring_valid[d], ring_v[d], true);
            }
            for (unsigned s0 = 0; s0 < total; s0 += 64u * DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; d++) {
                    bd_wait<2 * DEPTH - 2>(ring_v[d]);
                    if (s0 + 64u * d < total)
                        bd_answer_slot<FMT, QB, EXP, W8>(I, out, ring_idx[d], ring_valid[d], ring_v[d]);
                    else
                        bd_dummy_store(nobody);
                    prep(s0 + 64u * (d + DEPTH), ring_idx[d], ring_valid[d], ring_v[d], true);
                }
            }
            // the loads of the round after the last are still on their way: nothing may reuse their registers before they land
#pragma unroll
            for (int d = 0; d < DEPTH; d++) bd_wait<0>(ring_v[d]);
        } else if (PIPE) {
            bd_v4u y_v[DEPTH];
            unsigned y_idx[DEPTH], y_valid[DEPTH];
            // When set X's pass d is wanted, at least these were issued after its load: X's passes d + 1 .. DEPTH - 1 and
            // all DEPTH of set Y (stores of the answers in between only add to that) -> wait until at most
            // 2 * DEPTH - 1 - d operations are outstanding.  Passes past the end of the batch are loaded (from record 0)
            // and not answered, so the count holds in the last round too.
#pragma unroll
            for (int d = 0; d < DEPTH; d++) prep(64u * d, ring_idx[d], ring_valid[d], ring_v[d], true);
            for (unsigned s0 = 0; s0 < total; s0 += 128u * DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; d++) prep(s0 + 64u * (DEPTH + d), y_idx[d], y_valid[d], y_v[d], true);
                if (DEPTH > 0) { bd_wait<2 * DEPTH - 1>(ring_v[0]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, ring_idx[0], ring_valid[0], ring_v[0]); }
                if (DEPTH > 1) { bd_wait<2 * DEPTH - 2>(ring_v[1 % DEPTH]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, ring_idx[1 % DEPTH], ring_valid[1 % DEPTH], ring_v[1 % DEPTH]); }
                if (DEPTH > 2) { bd_wait<2 * DEPTH - 3>(ring_v[2 % DEPTH]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, ring_idx[2 % DEPTH], ring_valid[2 % DEPTH], ring_v[2 % DEPTH]); }
                if (DEPTH > 3) { bd_wait<2 * DEPTH - 4>(ring_v[3 % DEPTH]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, ring_idx[3 % DEPTH], ring_valid[3 % DEPTH], ring_v[3 % DEPTH]); }
#pragma unroll
                for (int d = 0; d < DEPTH; d++) prep(s0 + 64u * (2 * DEPTH + d), ring_idx[d], ring_valid[d], ring_v[d], true);
                if (DEPTH > 0) { bd_wait<2 * DEPTH - 1>(y_v[0]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, y_idx[0], y_valid[0], y_v[0]); }
                if (DEPTH > 1) { bd_wait<2 * DEPTH - 2>(y_v[1 % DEPTH]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, y_idx[1 % DEPTH], y_valid[1 % DEPTH], y_v[1 % DEPTH]); }
                if (DEPTH > 2) { bd_wait<2 * DEPTH - 3>(y_v[2 % DEPTH]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, y_idx[2 % DEPTH], y_valid[2 % DEPTH], y_v[2 % DEPTH]); }
                if (DEPTH > 3) { bd_wait<2 * DEPTH - 4>(y_v[3 % DEPTH]); bd_answer_slot<FMT, QB, EXP, W8>(I, out, y_idx[3 % DEPTH], y_valid[3 % DEPTH], y_v[3 % DEPTH]); }
            }
            // the loads of the round after the last are still on their way: nothing may reuse their registers before they land
#pragma unroll
            for (int d = 0; d < DEPTH; d++) bd_wait<0>(ring_v[d]);
        } else
        for (unsigned s0 = 0; s0 < total; s0 += 64u * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) prep(s0 + 64u * d, ring_idx[d], ring_valid[d], ring_v[d]);
#pragma unroll
            for (int d = 0; d < DEPTH; d++) bd_answer_slot<FMT, QB, EXP, W8>(I, out, ring_idx[d], ring_valid[d], ring_v[d]);
        }
        tb = tn;
    }
    __syncthreads();
    {
        const int nl = s_nlong < BD_LONG_CAP ? s_nlong : BD_LONG_CAP;
        for (int k = 0; k < nl; k++) {
            const uint2 lr = s_long[k];
            const unsigned first = lr.x, end = lr.x + lr.y;
            const unsigned q0 = first >> 2, nq4 = ((end + 3u) >> 2) - q0;
            for (unsigned q = threadIdx.x; q < nq4; q += BD_THREADS) {
                const unsigned idx4 = q0 + q;
                bd_answer_slot<FMT, QB, EXP, W8>(I, out, idx4, bd_valid_mask(4u * idx4, first, end), reinterpret_cast<const bd_v4u *>(recs)[idx4]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// search, second generation ("bw_*"): persistent workgroups, a ring that never drains, run look-up by bitmap
// ---------------------------------------------------------------------------
// What round 4's experiments said about bd_search_kernel on configs[1] (profiles/r04_search_*.txt): its memory operations
// alone take 200 us, its look-ups alone 118 us, together 258-270 us.  The memory side is bound by how many cache LINES a CU
// keeps in flight (a 137-byte run touches 2.07 lines, a CU gets ~0.27 lines per ns whatever the ring's depth), the
// compute side paid ~100 scalar instructions and a dozen branches per pass to find which run a slot belongs to (the
// readlane walk) and looked its four records up one after the other.  This kernel keeps the data layout (padded runs,
// cell images, 8-bit counts) and changes what the compute side costs, so that it hides under the memory side:
//   1. a slot's run comes from a BITMAP: every non-empty run of the batch sets the bit of its last slot (one LDS atomic
//      per lane and batch), lane p keeps the 64-bit window of pass p and the number of runs that end before it, and a
//      pass is three readlanes, two mbcnt and one ds_bpermute -- no loop, no branch (look-ups alone: 118 -> 70 us);
//   2. the ring is fed across batches (a wave sets the next batch up while its loads are in flight: nothing drains
//      before the item's last pass);
//   3. workgroups are persistent: one per CU, items handed out per XCD in unit order by a counter (no 157 KB workgroups
//      to launch and retire per item);
//   4. the four records of a slot are looked up together (eight LDS reads in flight, one hard-cell test per slot).
// Measured: 265 -> 226 us.  Tried on top and dropped: the next item's image requested into registers when a wave runs
// out of batches (the compiler keeps 36 more registers only by spilling them: 261 us), the same as a touch of the
// image's lines (L2 prefetch: 0.5 % and one memory fault under the profiler), non-temporal record loads (+4 %), batches
// of 16 / 32 tiles (+10 / +4 %).
constexpr int BW_MASK_WORDS = 128;  // run-end bitmap of a batch: 64 runs of at most 64 slots each
constexpr int BW_PF = 9;            // 16-byte pieces per thread of an image (9 x 1024 x 16 = 147 KB)
constexpr unsigned BW_IDLE = 0xFFFFFFFFu;
// diagnostics (compile time, wrong results): bit 0 = no look-ups, bit 1 = no count stores, bit 2 = no record loads
#ifndef BW_EXP
#define BW_EXP 0
#endif
#ifndef BW_B
#define BW_B 64  // tiles per batch of a wave, at most
#endif
#ifndef BS_UNROLL
#define BS_UNROLL 1  // sorted batches: groups of four queries per lane whose loads are issued together (measured on configs[1]: 1: 0.393 ms per pass, 2: 0.426, 4: 0.524 -- the walk streams at 4.7 TB/s as it is)
#endif

__device__ __forceinline__ unsigned bw_wave_inclusive_sum(unsigned v)
{
    // Hillis-Steele inside the rows of 16 lanes, then lane 15 -> next row, lane 31 -> upper half (gfx9 DPP)
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// the four counts of a slot from a cell image (16 bits each, 0xFFFF = ask the index again); all eight cells are read
// before the first is used, hard cells are noticed once per slot
// RAW (total-only batches: the walk sums what it finds): the counts' 32 bits as they are, escape records (the runs' padding, and
// the queries bm_escape_totals_kernel answers) count 0
template <bool RAW = false>
__device__ __forceinline__ void bp_count_slot(const BdImage &I, bd_v4u v, unsigned (&c)[4])
{
    const unsigned rec[4] = {v.x, v.y, v.z, v.w};
    unsigned relE[4], relS[4];
    unsigned long long cellE[4], cellS[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const unsigned off = rec[j] & I.off_mask;
        relE[j] = off + 1u, relS[j] = off + (rec[j] >> BP_RSHIFT);
        cellE[j] = I.cE[relE[j] >> 5];
        cellS[j] = I.cS[relS[j] >> 5];
    }
    unsigned worst = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bd_v2u ce = __builtin_bit_cast(bd_v2u, cellE[j]), cs = __builtin_bit_cast(bd_v2u, cellS[j]);
        const unsigned bE = ce.x, mE = ce.y, bS = cs.x, mS = cs.y;
        const unsigned geE = 0xFFFFFFFFu << (relE[j] & 31u), geS = 0xFFFFFFFFu << (relS[j] & 31u);  // the coordinates of the cell from rel on
        // the duplicated coordinate counts `extra` more times when it lies below rel
        const unsigned rE = (mE & 0xFFFFFu) + (unsigned)__popc(bE & ~geE) + (mE >> 25) * __builtin_amdgcn_ubfe(~geE, (mE >> 20) & 31u, 1u);
        const unsigned rS = (mS & 0xFFFFFu) + (unsigned)__popc(bS & ~geS) + (mS >> 25) * __builtin_amdgcn_ubfe(~geS, (mS >> 20) & 31u, 1u);
        c[j] = (unsigned)I.bias + (rS - rE);
        const unsigned m = mE > mS ? mE : mS;
        worst = worst > m ? worst : m;
    }
    if (worst >= ((unsigned)BM_HARD << 25)) {  // some hard cell among the eight (rare): those records again, one by one
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned mE = __builtin_bit_cast(bd_v2u, cellE[j]).y, mS = __builtin_bit_cast(bd_v2u, cellS[j]).y;
            if ((mE > mS ? mE : mS) >= ((unsigned)BM_HARD << 25)) c[j] = bp_count_record<RAW>(I, rec[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (RAW) {
            c[j] = rec[j] == BM_REC_ESC ? 0u : c[j];
        } else {
            c[j] = c[j] < 0xFFFFu ? c[j] : 0xFFFFu;
            c[j] = rec[j] == BM_REC_ESC ? 0xFFFFu : c[j];
        }
    }
}

// ---- offset cells (offset_cells.hpp) ----
// a hard cell's rank: from its list in LDS, or (no room for a list, more than 64 keys) from the sorted array
__device__ __forceinline__ int bo_hard_rank(const BdImage &I, lds_cell_p cells, unsigned rel, const int32_t *__restrict__ a, int slice_lo)
{
    const unsigned ci = rel >> I.cell_log2, p = rel & I.cell_mask;
    const bd_v2u c = __builtin_bit_cast(bd_v2u, cells[ci]);
    const unsigned base = c.y & 0xFFFFFu, next = (unsigned)(cells[ci + 1u] >> 32) & 0xFFFFFu;  // (every cell carries its base, hard or not)
    if (c.x != BP_NO_TABLE && (c.x & BO_TABLE)) {  // a rank table (clumped offset cells): one read
        const unsigned at = c.x & 0x3FFFFFFFu;
        return (int)(base + ((c.x & BO_TABLE_WIDE) ? (unsigned)I.img16[(at >> 1) + p] : (unsigned)I.img8[at + p]));
    }
    if (c.x != BP_NO_TABLE) {
        unsigned r = base;
        for (unsigned i = 0; i < next - base; i++) r += (unsigned)I.img8[c.x + i] < p ? 1u : 0u;
        return (int)r;
    }
    const long long key = I.lo + (long long)rel;
    if (key > INT_MAX) return (int)next;  // every int32 key is below it
    return global_rank_lt(a, slice_lo + (int)base, slice_lo + (int)next, (int)key) - slice_lo;
}

// the count of one record whose cells may be hard (16 bits, 0xFFFF = ask the index again; RAW: see bp_count_slot)
template <bool RAW = false>
__device__ __forceinline__ unsigned bo_count_record(const BdImage &I, unsigned rec)
{
    const unsigned off = rec & I.off_mask, len = rec >> I.rshift;
    const unsigned relE = off + 1u, relS = off + len;
    const bd_v2u ce = __builtin_bit_cast(bd_v2u, I.cE[relE >> I.cell_log2]), cs = __builtin_bit_cast(bd_v2u, I.cS[relS >> I.cell_log2]);
    const int rE = ce.y >= BO_HARD ? bo_hard_rank(I, I.cE, relE, I.e_sorted, I.eLo) : (int)bo_rank(ce.x, ce.y, relE & I.cell_mask);
    const int rS = cs.y >= BO_HARD ? bo_hard_rank(I, I.cS, relS, I.s_ord, I.sLo) : (int)bo_rank(cs.x, cs.y, relS & I.cell_mask);
    unsigned c = (unsigned)(I.bias + (rS - rE));
    if (RAW) return rec == BM_REC_ESC ? 0u : c;
    c = c < 0xFFFFu ? c : 0xFFFFu;
    return rec == BM_REC_ESC ? 0xFFFFu : c;
}

// the four counts of a slot: all eight cells read before the first is used, hard cells noticed once per slot (bp_count_slot's shape)
template <bool RAW = false>
__device__ __forceinline__ void bo_count_slot_lists(const BdImage &I, bd_v4u v, unsigned (&c)[4])
{
    const unsigned rec[4] = {v.x, v.y, v.z, v.w};
    unsigned relE[4], relS[4];
    unsigned long long cellE[4], cellS[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const unsigned off = rec[j] & I.off_mask;
        relE[j] = off + 1u, relS[j] = off + (rec[j] >> I.rshift);
        cellE[j] = I.cE[relE[j] >> I.cell_log2];
        cellS[j] = I.cS[relS[j] >> I.cell_log2];
    }
    unsigned worst = 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bd_v2u ce = __builtin_bit_cast(bd_v2u, cellE[j]), cs = __builtin_bit_cast(bd_v2u, cellS[j]);
        const unsigned rE = bo_rank(ce.x, ce.y, relE[j] & I.cell_mask), rS = bo_rank(cs.x, cs.y, relS[j] & I.cell_mask);
        c[j] = (unsigned)I.bias + (rS - rE);
        const unsigned m = ce.y > cs.y ? ce.y : cs.y;
        worst = worst > m ? worst : m;
    }
    if (worst >= BO_HARD) {  // some hard cell among the eight (rare): those records again, one by one
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned mE = __builtin_bit_cast(bd_v2u, cellE[j]).y, mS = __builtin_bit_cast(bd_v2u, cellS[j]).y;
            if ((mE > mS ? mE : mS) >= BO_HARD) c[j] = bo_count_record<RAW>(I, rec[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (RAW) {
            c[j] = rec[j] == BM_REC_ESC ? 0u : c[j];
        } else {
            c[j] = c[j] < 0xFFFFu ? c[j] : 0xFFFFu;
            c[j] = rec[j] == BM_REC_ESC ? 0xFFFFu : c[j];
        }
    }
}

// TAB: the image is in the clumped layout (hard cells carry rank tables); the standard layout keeps round 5's shape (bo_count_slot_lists)
template <bool RAW = false, bool TAB = false>
__device__ __forceinline__ void bo_count_slot(const BdImage &I, bd_v4u v, unsigned (&c)[4])
{
    if (!TAB) {
        bo_count_slot_lists<RAW>(I, v, c);
        return;
    }
    const unsigned rec[4] = {v.x, v.y, v.z, v.w};
    unsigned relE[4], relS[4];
    unsigned long long cellE[4], cellS[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const unsigned off = rec[j] & I.off_mask;
        relE[j] = off + 1u, relS[j] = off + (rec[j] >> I.rshift);
        cellE[j] = I.cE[relE[j] >> I.cell_log2];
        cellS[j] = I.cS[relS[j] >> I.cell_log2];
    }
    // a hard cell with a RANK TABLE (clumped layout: the rule around a hot spot) is answered in line -- one more LDS read, no second
    // pass over the record; what is left for the slow path are lists and cells without room (standard layout: rare)
    auto rank_of = [&](bd_v2u cell, unsigned p, bool &slow) -> unsigned {
        unsigned r = bo_rank(cell.x, cell.y, p);
        const bool hard = cell.y >= BO_HARD;
        const bool tab = hard && cell.x != BP_NO_TABLE && (cell.x & BO_TABLE) != 0u;
        if (tab) {
            const unsigned at = cell.x & 0x3FFFFFFFu;
            r = (cell.y & 0xFFFFFu) + ((cell.x & BO_TABLE_WIDE) ? (unsigned)I.img16[(at >> 1) + p] : (unsigned)I.img8[at + p]);
        }
        slow |= hard && !tab;
        return r;
    };
    bool any_slow = false;
    bool slow[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bd_v2u ce = __builtin_bit_cast(bd_v2u, cellE[j]), cs = __builtin_bit_cast(bd_v2u, cellS[j]);
        slow[j] = false;
        const unsigned rE = rank_of(ce, relE[j] & I.cell_mask, slow[j]), rS = rank_of(cs, relS[j] & I.cell_mask, slow[j]);
        c[j] = (unsigned)I.bias + (rS - rE);
        any_slow |= slow[j];
    }
    if (any_slow) {  // a list or a cell without room among the eight: those records again, one by one
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (slow[j]) c[j] = bo_count_record<true>(I, rec[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (RAW) {
            c[j] = rec[j] == BM_REC_ESC ? 0u : c[j];
        } else {
            c[j] = c[j] < 0xFFFFu ? c[j] : 0xFFFFu;
            c[j] = rec[j] == BM_REC_ESC ? 0xFFFFu : c[j];
        }
    }
}

template <bool W8>
__device__ __forceinline__ void bw_store_slot(unsigned short *__restrict__ out, unsigned idx4, const unsigned (&c)[4])
{
    if (W8) {
        unsigned b[4];
#pragma unroll
        for (int j = 0; j < 4; j++) b[j] = c[j] < 0xFFu ? c[j] : 0xFFu;
        // (an ordinary store: the 34 bytes of a run meet their neighbours' in L2 -- as non-temporal stores the pass takes 0.80 instead of 0.71 ms)
        reinterpret_cast<unsigned *>(out)[idx4] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    } else {
        bd_v2u o;
        o.x = c[0] | (c[1] << 16), o.y = c[2] | (c[3] << 16);
        reinterpret_cast<bd_v2u *>(out)[idx4] = o;
    }
}

// WIDE: the unit images are offset cells (sparse indexes; the cell width travels in the segment's geometry, g.dshift) --
// units of up to 2^20 coordinates, so a (tile, unit) run can be hundreds of slots long: a wave's batch is then fewer tiles
// (the run-end bitmap holds 4096 slots whatever their number), and a run is "long" from 4096 / B slots on.
// THREADS: 1024 = one workgroup per CU (bitmap cells: a unit's image is 147 KB); 512 = two per CU (offset cells: 72 KB) -- one
// loads its next image while the other looks records up.  (Measured on configs[3] with one workgroup per CU and units of 2^21:
// 13.7 us of every 41 us item were image load, ring start and drain, with the vector units idle.)
// TOT: the batch wants its overlap TOTAL only (every segment's counts == NULL).  The walk then sums what it finds -- the counts'
// full 32 bits, escape records 0 (bp_count_slot<RAW>) -- and adds the sum to its segment's partial totals when the segment
// changes and at the end; no count is stored, no un-permute kernel follows (16 bytes of HBM traffic per query instead of 26).
// The ring keeps its shape: where a pass stored its counts it stores to the slot nobody owns.  `out` is still the count array
// (that slot lies in it).  The queries behind real escape records are bm_escape_totals_kernel's.
template <bool W8, int DEPTH, bool WIDE = false, int THREADS = BD_THREADS, bool TOT = false>
__global__ __launch_bounds__(THREADS) void bw_search_kernel(const BmSeg *__restrict__ segs, const int4 *__restrict__ items,
                                                               const int *__restrict__ n_items, const unsigned short *__restrict__ unitT, int64_t ntp,
                                                               const unsigned *__restrict__ recs /* tile-sorted records, padded runs */,
                                                               unsigned short *__restrict__ out /* their counts, same order */, int tile_log2,
                                                               const unsigned *__restrict__ gate, unsigned *__restrict__ xcd_next /* [8], zero */,
                                                               unsigned long long *__restrict__ total_slots = nullptr /* TOT: [segments][PT_SLOTS] */)
{
    if (gate && *gate == 0) return;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ long long s_red[TOT ? THREADS / 64 : 1];
    unsigned long long acc = 0ull;  // TOT: this lane's share of the current segment's total
    int acc_seg = -1;
    constexpr int LONG_CAP = THREADS == BD_THREADS ? BD_LONG_CAP : BD_LONG_CAP / 2;
    constexpr int PF = THREADS == BD_THREADS ? BW_PF : 10;  // 16-byte pieces per thread of an image (10 x 512 x 16 = 80 KB)
    __shared__ uint2 s_long[LONG_CAP];  // {first record, length} of the long runs met during the walk
    __shared__ int s_nlong, s_next, s_item_next;
    __shared__ __attribute__((aligned(8))) unsigned s_mask[THREADS / 64][BW_MASK_WORDS];
    const int nit = *n_items;
    const int per_xcd = (nit + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7);
    const int it_lo = xcd * per_xcd, it_hi = it_lo + per_xcd < nit ? it_lo + per_xcd : nit;  // neighbouring units on one XCD, in order
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (said out loud: a wave's number is uniform)
    unsigned *const wmask = s_mask[wave];
    wmask[lane] = 0u, wmask[64 + lane] = 0u;
    if (threadIdx.x == 0) s_item_next = it_lo + (int)atomicAdd(&xcd_next[xcd], 1u);
    __syncthreads();
    int it = __builtin_amdgcn_readfirstlane(s_item_next);
    __syncthreads();  // (thread 0 writes the word again at the top of the loop)
    const unsigned tile_slots = ((1u << tile_log2) + (unsigned)BM_PAD_ROOM) >> 2;  // 16-byte slots between two tiles
    while (it < it_hi) {
        const int4 item = items[it];
        const int unit = item.x & 0xffff, t0 = item.y, t1 = item.z;
        const BmSeg &sg = segs[item.x >> 16];
        if (TOT && (item.x >> 16) != acc_seg) {  // (item-uniform) the totals are per segment
            if (acc_seg >= 0) {
                block_accumulate_i64((long long)acc, s_red, total_slots + (int64_t)acc_seg * PT_SLOTS + (blockIdx.x & (PT_SLOTS - 1)));
                __syncthreads();
            }
            acc = 0ull, acc_seg = item.x >> 16;
        }
        const BmGeom g = sg.g;
        const int cell_log2 = WIDE ? 5 + g.dshift : 5;
        const BpLayout LP = bp_layout(g.shift + g.f, cell_log2, WIDE ? g.stride : 0);
        const unsigned short *__restrict__ runs0 = unitT + (int64_t)unit * ntp;
        const unsigned short *__restrict__ runs1 = runs0 + ntp;  // (the next unit's first slots, or the row behind the last unit)
        int B = BW_B;
        while (B > 8 && (t1 - t0) < 2 * (THREADS / 64) * B) B >>= 1;
        if (WIDE) {  // one and a half times the item's mean run, B times, has to fit the bitmap
            const int mean4 = item.w / (t1 - t0) / 4 + 2;
            while (B > 8 && B * (mean4 + (mean4 >> 1)) > BW_MASK_WORDS * 32) B >>= 1;
        }
        const unsigned long_slots = WIDE ? (unsigned)(BW_MASK_WORDS * 32 / B) : (unsigned)BD_LONG_SLOTS;
        if (threadIdx.x == 0) {
            s_nlong = 0, s_next = B * (THREADS / 64);  // (every wave starts with the batch of its number)
            s_item_next = it_lo + (int)atomicAdd(&xcd_next[xcd], 1u);
        }
        // The run table of the NEXT batch is requested by hand as well (two 16-bit loads per lane): left to the compiler they
        // were issued where they are used, behind a full drain of the memory pipe -- record loads and count stores -- once per
        // batch.  They travel under the batch's passes; bw_wait_runs says when they are certain to have landed.
        unsigned a_nx = 0u, e_nx = 0u;
        bool nx_in = false;
        auto load_runs = [&](int tbase) {
            const int t = tbase + lane;
            nx_in = t < t1 && lane < B;
            const int tc = nx_in ? t : t0;  // a valid address: no branch around the loads
            asm volatile("global_load_ushort %0, %1, off" : "=v"(a_nx) : "v"(runs0 + tc) : "memory");
            asm volatile("global_load_ushort %0, %1, off" : "=v"(e_nx) : "v"(runs1 + tc) : "memory");
        };
        int tb_next = t0 + B * wave;
        load_runs(tb_next);
        {   // the image (streams through L2 once: non-temporal loads); every load of a lane issued before its first LDS store
            // (Requested one item ahead instead -- into registers, in front of the barrier at the end of the item before, so that
            // the loads travel while the slower waves finish: the 36 registers are live across the loop's back edge and the
            // allocator spills five of the nine pieces beside the 1024-thread walk's 127 -- search 204 -> 245 us on configs[1];
            // the 512-thread walk keeps them (126 registers) and gains 0.5-2 %: its CU's other workgroup hides the load already.)
            const bm_v4i BX_GLOBAL *src = reinterpret_cast<const bm_v4i BX_GLOBAL *>(as_global(sg.pimages) + (size_t)unit * LP.bytes);
            const int n4 = LP.bytes >> 4;
            bm_v4i v[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int i = k * THREADS + (int)threadIdx.x;
                v[k] = __builtin_nontemporal_load(src + (i < n4 ? i : n4 - 1));
            }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int i = k * THREADS + (int)threadIdx.x;
                if (i < n4) reinterpret_cast<bm_v4i *>(dyn)[i] = v[k];
            }
        }
        __syncthreads();
        const int it_nx = __builtin_amdgcn_readfirstlane(s_item_next);
        BdImage I;
        {
            unsigned char *base = reinterpret_cast<unsigned char *>(dyn);
            I.img16 = (lds_u16_p) reinterpret_cast<unsigned short *>(base);
            I.cE = (lds_cell_p) reinterpret_cast<unsigned long long *>(base + LP.cellsE);
            I.cS = (lds_cell_p) reinterpret_cast<unsigned long long *>(base + LP.cellsS);
            const unsigned *hdr = reinterpret_cast<const unsigned *>(base + LP.hdr);
            I.eLo = (int)hdr[0], I.sLo = (int)hdr[1];
            I.bias = I.sLo - I.eLo;
            I.lo = (long long)((unsigned long long)hdr[2] | ((unsigned long long)hdr[3] << 32));
            I.s_ord = sg.ix.s_ord, I.e_sorted = sg.e_sorted;
            I.off_mask = (1u << (g.shift + g.f)) - 1u;
            I.img8 = (lds_u8_p)base;
            I.rshift = g.rshift, I.cell_log2 = cell_log2, I.cell_mask = (1u << cell_log2) - 1u;
        }
        // a slot no query owns (the end of a tile's room, past every unit's padding): where the ring's idle passes store
        unsigned *const nobody = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(out) +
                                                              (((size_t)(t0 + (unit * 7 + wave * 64) % (t1 - t0)) + 1) * tile_slots - 1) * (W8 ? 4 : 8));
        // --- the batch a wave is issuing passes of ---
        unsigned total = 0u, s0 = 0u;      // slots of the batch, first slot of the next pass
        unsigned delta_c = 0u;             // lane k: where run k's slots lie (non-empty runs, in order): slot s of the batch is at delta_c[k] + s
        unsigned win_lo = 0u, win_hi = 0u, win_before = 0u;  // lane p: run-end bits of pass p, runs that end before pass p
        bool live = true;
        auto advance = [&]() -> bool {
            while (tb_next < t1) {
                const int t = tb_next + lane;
                // the runs of this batch were requested one batch ago: 2 x (its passes) memory operations followed them, so
                // once the batch had DEPTH - 1 passes the ring's own wait count has seen them land; a shorter batch (or none:
                // the item's first, or an empty one skipped just now) waits for everything
                // (one statement names the registers, on every path: two of them made the compiler copy the registers in front of
                // the waits -- reading them while the loads were in flight; tools/check_ring_isa.py follows these loads too)
                if (((total + 63u) >> 6) < (unsigned)(DEPTH - 1)) asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
                asm volatile("s_waitcnt vmcnt(%2) ; runs %0 %1" : "+v"(a_nx), "+v"(e_nx) : "n"(2 * DEPTH - 2) : "memory");
                const unsigned a = nx_in ? a_nx : 0u, e = nx_in ? e_nx : 0u;
                total = 0u;  // (nothing of this batch has been issued yet)
                int tn = 0;
                if (lane == 0) tn = atomicAdd(&s_next, B);
                tb_next = t0 + __builtin_amdgcn_readfirstlane(tn);
                load_runs(tb_next);
                unsigned n4 = e > a ? ((e + 3u) >> 2) - (a >> 2) : 0u;  // 16-byte slots that hold the run
                if (n4 > long_slots) {  // sorted / clumped input: left to the whole workgroup
                    const int k = atomicAdd(&s_nlong, 1);
                    const unsigned first = ((unsigned)t * tile_slots << 2) + a;
                    if (k < LONG_CAP) {
                        s_long[k] = make_uint2(first, e - a);
                    } else if (TOT) {  // (a full list, and nobody to ask again: the lane walks its run itself)
                        for (unsigned q = first >> 2; q < ((first + (e - a) + 3u) >> 2); q++) {
                            unsigned c[4];
                            if (WIDE)
                                bo_count_slot<true, THREADS == BD_THREADS>(I, reinterpret_cast<const bd_v4u *>(recs)[q], c);
                            else
                                bp_count_slot<true>(I, reinterpret_cast<const bd_v4u *>(recs)[q], c);
                            acc += (unsigned long long)c[0] + c[1] + c[2] + c[3];
                        }
                    } else {  // (a full list: the run's counts say "ask the index again" -- exactness never depends on the list)
                        const unsigned esc[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
                        for (unsigned q = first >> 2; q < ((first + (e - a) + 3u) >> 2); q++) bw_store_slot<W8>(out, q, esc);
                    }
                    n4 = 0u;
                }
                const unsigned incl = bw_wave_inclusive_sum(n4);
                total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
                if (total == 0u) continue;
                const unsigned delta = ((unsigned)t * tile_slots + (a >> 2)) - (incl - n4);
                // the non-empty runs move to the low lanes (the empty ones fill the lanes from the top)
                const bool ne = n4 != 0u;
                const unsigned long long bal = __ballot(ne);
                const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                const unsigned dest = ne ? below : 63u - ((unsigned)lane - below);
                delta_c = (unsigned)__builtin_amdgcn_ds_permute((int)(dest << 2), (int)delta);
                if (ne) atomicOr(&wmask[(incl - 1u) >> 5], 1u << ((incl - 1u) & 31u));
                const uint2 w = reinterpret_cast<uint2 *>(wmask)[lane];  // (this wave's own LDS operations stay in order)
                reinterpret_cast<uint2 *>(wmask)[lane] = make_uint2(0u, 0u);
                win_lo = w.x, win_hi = w.y;
                const unsigned ends = (unsigned)__popc(w.x) + (unsigned)__popc(w.y);
                win_before = bw_wave_inclusive_sum(ends) - ends;
                s0 = 0u;
                return true;
            }
            return false;
        };
        // The ring: DEPTH passes of records in flight per wave, each slot of it a fixed set of registers (the loop is
        // unrolled over the slots).  The memory pipe sees [store, load] per pass from the first to the last: 2 * DEPTH - 2
        // operations follow every load before its slot comes round again (run-table and image loads in between only make
        // a wait stronger: the counter retires in order).  Idle passes -- before the first batch, behind the last -- load
        // record 0 and store to the slot nobody owns.
        bd_v4u ring_v[DEPTH];
        unsigned ring_idx[DEPTH];
        int inflight = 0;  // passes of the ring that hold records of a batch
        auto issue = [&](unsigned &idx4, bd_v4u &v) {
            if (live && s0 >= total) live = advance();
            unsigned at = BW_IDLE;
            if (live) {
                const int p = (int)(s0 >> 6);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)win_lo, p), hi = (unsigned)__builtin_amdgcn_readlane((int)win_hi, p);
                const unsigned before = (unsigned)__builtin_amdgcn_readlane((int)win_before, p);
                // slot s lies in run r = #{runs whose last slot is below s}
                const unsigned r = before + __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
                const unsigned dl = (unsigned)__builtin_amdgcn_ds_bpermute((int)(r << 2), (int)delta_c);
                const unsigned s = s0 + (unsigned)lane;
                at = s < total ? dl + s : BW_IDLE;
                s0 += 64u;
            }
            idx4 = at;
            if (live) inflight++;
            bd_issue_load(v, recs, at == BW_IDLE || (BW_EXP & 4) ? 0u : at);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            if (d > 0) bd_dummy_store(reinterpret_cast<unsigned short *>(nobody));
            ring_idx[d] = BW_IDLE;
            bd_issue_load(ring_v[d], recs, 0u);
        }
        for (;;) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                bd_wait<2 * DEPTH - 2>(ring_v[d]);
                if (__any(ring_idx[d] != BW_IDLE)) {
                    inflight--;
                    if (ring_idx[d] != BW_IDLE) {
                        unsigned c[4];
                        bd_v4u rv = ring_v[d];
                        if (BW_EXP & 4) {  // synthetic records: offsets all over the unit, lengths < 1000
                            const unsigned h = ring_idx[d] * 2654435761u;
                            rv = bd_v4u{(h & 0x3ffffu) | (500u << 18), ((h >> 3) & 0x3ffffu) | (100u << 18), ((h >> 7) & 0x3ffffu) | (900u << 18),
                                        ((h >> 11) & 0x3ffffu) | (300u << 18)};
                        }
                        if (BW_EXP & 1)
                            c[0] = rv.x & 0xffu, c[1] = rv.y & 0xffu, c[2] = rv.z & 0xffu, c[3] = rv.w & 0xffu;
                        else if (WIDE)
                            bo_count_slot<TOT, THREADS == BD_THREADS>(I, rv, c);
                        else
                            bp_count_slot<TOT>(I, rv, c);
                        if (TOT) acc += (unsigned long long)c[0] + c[1] + c[2] + c[3];
                        else if (!(BW_EXP & 2) || (c[0] & c[1] & c[2] & c[3]) == 0x12345u) bw_store_slot<W8>(out, ring_idx[d], c);
                        else bd_dummy_store(reinterpret_cast<unsigned short *>(nobody));
                    }
                    if (TOT) bd_dummy_store(reinterpret_cast<unsigned short *>(nobody));  // (every lane of the pass: one memory operation, as a count store would be)
                } else {
                    bd_dummy_store(reinterpret_cast<unsigned short *>(nobody));
                }
                issue(ring_idx[d], ring_v[d]);
            }
            if (!live && inflight == 0) break;
        }
        // the loads of the round after the last are still on their way: nothing may reuse their registers before they land
#pragma unroll
        for (int d = 0; d < DEPTH; d++) bd_wait<0>(ring_v[d]);
        __syncthreads();
        {
            const int nl = s_nlong < LONG_CAP ? s_nlong : LONG_CAP;
            for (int k = 0; k < nl; k++) {
                const uint2 lr = s_long[k];
                const unsigned q0 = lr.x >> 2, nq4 = ((lr.x + lr.y + 3u) >> 2) - q0;  // (padded runs: whole slots)
                for (unsigned q = threadIdx.x; q < nq4; q += THREADS) {
                    unsigned c[4];
                    if (WIDE)
                        bo_count_slot<TOT, THREADS == BD_THREADS>(I, reinterpret_cast<const bd_v4u *>(recs)[q0 + q], c);
                    else
                        bp_count_slot<TOT>(I, reinterpret_cast<const bd_v4u *>(recs)[q0 + q], c);
                    if (TOT) acc += (unsigned long long)c[0] + c[1] + c[2] + c[3];
                    else bw_store_slot<W8>(out, q0 + q, c);
                }
            }
        }
        __syncthreads();
        it = it_nx;
    }
    if (TOT && acc_seg >= 0) block_accumulate_i64((long long)acc, s_red, total_slots + (int64_t)acc_seg * PT_SLOTS + (blockIdx.x & (PT_SLOTS - 1)));
}

// ---------------------------------------------------------------------------
// sorted batches on cell images ("bs_*"): no exchange at all
// ---------------------------------------------------------------------------
// A batch sorted by start (the usual BED file) has every unit's queries in ONE stretch of the query arrays, and the order
// check that detects it has just read every start: bm_sorted_check_kernel<true> leaves the stretches' bounds.  A workgroup
// then takes a unit's image into LDS and answers the stretch as it lies -- 16 bytes of starts, 16 of ends in, 16 bytes of
// counts out per lane, the same look-ups as the walk above -- 12 bytes of HBM traffic per query and no scratch, where the
// first-generation kernel for sorted batches (ivl_local_count_kernel: per 4096 queries a walk down the index's global trees,
// two slices staged as LDS search trees, two tree searches per query) is bound by the latency chain of its setup.
// bs_plan_kernel: stretches longer than `chunk` queries are cut (a batch crowded into a few units), the queries left of the grid
// and right of it become items of unit -1 (answered from the sealed index, like every escape record).
// items[i] = {unit or -1, first query, last query + 1, 0}
__global__ __launch_bounds__(1024) void bs_plan_kernel(const unsigned *__restrict__ bounds /* [units + 1] */, int units, unsigned nq, unsigned chunk,
                                                       int4 *__restrict__ items, int *__restrict__ n_items, const unsigned *__restrict__ gate)
{
    __shared__ int scan_tmp[16];
    if (gate && *gate != 0) return;  // not sorted: the exchange answers the batch
    int carry = 0;
    // stretch s = -1: queries in front of unit 0; s = units: behind the last unit
    for (int s0 = -1; s0 <= units; s0 += 1024) {
        const int s = s0 + (int)threadIdx.x;
        unsigned lo = 0u, hi = 0u;
        if (s <= units) {
            lo = s < 0 ? 0u : bounds[s];
            hi = s == units ? nq : bounds[s + 1];
        }
        const unsigned n = hi > lo ? hi - lo : 0u;
        const int cnt = (int)((n + chunk - 1u) / chunk);
        int tot;
        const int at = carry + block_exclusive_scan(cnt, OpSum(), 0, scan_tmp, &tot);
        for (int k = 0; k < cnt; k++) {
            const unsigned a = lo + (unsigned)k * chunk, b = a + chunk < hi ? a + chunk : hi;
            items[at + k] = make_int4(s >= 0 && s < units ? s : -1, (int)a, (int)b, 0);
        }
        carry += tot;
    }
    if (threadIdx.x == 0) *n_items = carry;
}

// A batch over SEVERAL indexes (a sorted BED file against a genome: one index per chromosome, bxmi_ivl_count_multi_dev;
// scripts/interval_join.py:21-28 loops over the chromosomes, lib/bx/bitset_builders.py:31-45 keeps one set per chromosome): the
// order check and the plan per SEGMENT, one walk over all segments' items.  A workgroup checks one tile of the batch's tile
// numbering (every segment starts on a group of 64 tiles) and leaves the bounds of its segment's units in the segment's row of
// `bounds_all`; ONE descent anywhere sends the whole batch through the exchange (a sorted file is sorted in every chromosome).
constexpr int BS_BOUNDS_ROW = BM_NB + 2;
__global__ __launch_bounds__(256) void bs_check_multi_kernel(const BmSeg *__restrict__ segs, const unsigned short *__restrict__ tile_seg, int64_t ntp, int tile_log2,
                                                             unsigned *__restrict__ unsorted, unsigned *__restrict__ bounds_all)
{
    for (int64_t tile = blockIdx.x; tile < ntp; tile += gridDim.x) {
        if (__hip_atomic_load(unsorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
        const int seg = tile_seg[tile];
        const BmSeg &sg = segs[seg];
        const int64_t ltile = tile - sg.tile0;
        if (ltile >= sg.ntiles) continue;  // padding up to the next plan group
        BmBounds B;
        B.bounds = bounds_all + (int64_t)seg * BS_BOUNDS_ROW, B.cmin = sg.g.cmin, B.ulog = sg.g.shift + sg.g.f, B.units = BM_NB >> sg.g.f;
        if (ltile == 0 && threadIdx.x == 0) bm_bounds_outer(sg.qs, sg.nq, B);
        const int64_t c0 = (ltile << tile_log2) / BM_CHECK_CH, c1 = ((ltile + 1) << tile_log2) / BM_CHECK_CH;
        bool descent = false;
        for (int64_t c = c0; c < c1 && c * BM_CHECK_CH < sg.nq; c++) descent |= bm_check_chunk<true>(sg.qs, sg.nq, c, B);
        if (__syncthreads_or(descent)) {
            if (threadIdx.x == 0) __hip_atomic_store(unsorted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
    }
}

// bs_plan_kernel per segment (a workgroup each): items[i] = {unit or -1, first query, last query + 1, segment}; room is
// reserved in the shared list 1024 stretches at a time (n_items zeroed by the host).
__global__ __launch_bounds__(1024) void bs_plan_multi_kernel(const BmSeg *__restrict__ segs, const unsigned *__restrict__ bounds_all, unsigned chunk,
                                                             int4 *__restrict__ items, int *__restrict__ n_items, const unsigned *__restrict__ gate)
{
    __shared__ int scan_tmp[16];
    __shared__ int s_at;
    if (gate && *gate != 0) return;  // not sorted: the exchange answers the batch
    const int seg = blockIdx.x;
    const BmSeg &sg = segs[seg];
    const int units = BM_NB >> sg.g.f;
    const unsigned nq = (unsigned)sg.nq;
    if (nq == 0u) return;
    const unsigned *bounds = bounds_all + (int64_t)seg * BS_BOUNDS_ROW;
    for (int s0 = -1; s0 <= units; s0 += 1024) {
        const int s = s0 + (int)threadIdx.x;
        unsigned lo = 0u, hi = 0u;
        if (s <= units) {
            lo = s < 0 ? 0u : bounds[s];
            hi = s == units ? nq : bounds[s + 1];
        }
        const unsigned n = hi > lo ? hi - lo : 0u;
        const int cnt = (int)((n + chunk - 1u) / chunk);
        int tot;
        const int rel = block_exclusive_scan(cnt, OpSum(), 0, scan_tmp, &tot);
        if (threadIdx.x == 0) s_at = tot ? atomicAdd(n_items, tot) : 0;
        __syncthreads();
        const int at = s_at + rel;
        for (int k = 0; k < cnt; k++) {
            const unsigned a = lo + (unsigned)k * chunk, b = a + chunk < hi ? a + chunk : hi;
            items[at + k] = make_int4(s >= 0 && s < units ? s : -1, (int)a, (int)b, seg);
        }
        __syncthreads();
    }
}

// (items carry their segment in .w: one walk serves a batch over several indexes; a plain call is segment 0 of its parameter block)
template <bool WIDE, int THREADS>
__global__ __launch_bounds__(THREADS) void bs_walk_kernel(const BmSeg *__restrict__ segs, const int4 *__restrict__ items, const int *__restrict__ n_items,
                                                          unsigned long long *__restrict__ total_slots /* [segments][PT_SLOTS] */, const unsigned *__restrict__ gate,
                                                          unsigned *__restrict__ xcd_next /* [8], zero */,
                                                          unsigned long long *__restrict__ order_host, unsigned long long seq)
{
    // what the order check found, into host memory (as ivl_local_count_kernel reports it: pass number << 1 | 1 = not sorted)
    if (order_host && blockIdx.x == 0 && threadIdx.x == 0) *order_host = (seq << 1) | (*gate != 0 ? 1ull : 0ull);
    if (*gate != 0) return;
    constexpr int PF = THREADS == BD_THREADS ? BW_PF : 10;
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    __shared__ int s_item_next;
    __shared__ long long red[THREADS / 64];
    const int nit = *n_items;
    const int per_xcd = (nit + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7);
    const int it_lo = xcd * per_xcd, it_hi = it_lo + per_xcd < nit ? it_lo + per_xcd : nit;  // neighbouring stretches on one XCD
    long long acc = 0;
    int acc_seg = -1;  // whose total `acc` belongs to
    int loaded = -1;   // segment << 16 | unit of the image the LDS holds
    if (threadIdx.x == 0) s_item_next = it_lo + (int)atomicAdd(&xcd_next[xcd], 1u);
    __syncthreads();
    int it = s_item_next;
    __syncthreads();
    while (it < it_hi) {
        const int4 item = items[it];
        const int unit = item.x;
        const int seg = item.w;
        if (seg != acc_seg) {  // (item-uniform) the totals are per segment
            if (acc_seg >= 0 && total_slots) {
                block_accumulate_i64(acc, red, total_slots + (int64_t)acc_seg * PT_SLOTS + (blockIdx.x & (PT_SLOTS - 1)));
                __syncthreads();
            }
            acc = 0, acc_seg = seg;
        }
        const BmSeg &sg = segs[seg];
        const BmGeom g = sg.g;
        const int cell_log2 = WIDE ? 5 + g.dshift : 5;
        const BpLayout LP = bp_layout(g.shift + g.f, cell_log2, WIDE ? g.stride : 0);
        if (threadIdx.x == 0) s_item_next = it_lo + (int)atomicAdd(&xcd_next[xcd], 1u);
        if (unit >= 0 && ((seg << 16) | unit) != loaded) {  // (the barrier at the end of the item before: nobody reads the old image any more)
            const bm_v4i BX_GLOBAL *src = reinterpret_cast<const bm_v4i BX_GLOBAL *>(as_global(sg.pimages) + (size_t)unit * LP.bytes);
            const int n4 = LP.bytes >> 4;
            bm_v4i v[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int i = k * THREADS + (int)threadIdx.x;
                v[k] = __builtin_nontemporal_load(src + (i < n4 ? i : n4 - 1));
            }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                const int i = k * THREADS + (int)threadIdx.x;
                if (i < n4) reinterpret_cast<bm_v4i *>(dyn)[i] = v[k];
            }
            loaded = (seg << 16) | unit;
        }
        __syncthreads();
        const int it_nx = s_item_next;
        BdImage I;
        {
            unsigned char *base = reinterpret_cast<unsigned char *>(dyn);
            I.img16 = (lds_u16_p) reinterpret_cast<unsigned short *>(base);
            I.cE = (lds_cell_p) reinterpret_cast<unsigned long long *>(base + LP.cellsE);
            I.cS = (lds_cell_p) reinterpret_cast<unsigned long long *>(base + LP.cellsS);
            const unsigned *hdr = reinterpret_cast<const unsigned *>(base + LP.hdr);
            I.eLo = (int)hdr[0], I.sLo = (int)hdr[1];
            I.bias = I.sLo - I.eLo;
            I.lo = (long long)((unsigned long long)hdr[2] | ((unsigned long long)hdr[3] << 32));
            I.s_ord = sg.ix.s_ord, I.e_sorted = sg.e_sorted;
            I.off_mask = (1u << (g.shift + g.f)) - 1u;
            I.img8 = (lds_u8_p)base;
            I.rshift = g.rshift, I.cell_log2 = cell_log2, I.cell_mask = (1u << cell_log2) - 1u;
        }
        // groups of four consecutive queries (the arrays are 16-byte aligned): [4 q, 4 q + 4) for q in [q0, q1)
        const int64_t lo = (unsigned)item.y, hi = (unsigned)item.z;
        const int64_t q0 = lo >> 2, q1 = (hi + 3) >> 2;
        // BS_UNROLL groups per lane and round: their loads are issued together
        for (int64_t qb = q0 + threadIdx.x; qb < q1; qb += (int64_t)BS_UNROLL * THREADS) {
            int s[BS_UNROLL][4], e[BS_UNROLL][4];
            bool whole[BS_UNROLL];
#pragma unroll
            for (int u = 0; u < BS_UNROLL; u++) {
                const int64_t q = qb + (int64_t)u * THREADS, k0 = 4 * q;
                whole[u] = q < q1 && k0 >= lo && k0 + 4 <= hi;  // (hi <= nq: a whole group never reads past the arrays)
                if (whole[u]) {
                    const int4 vs = load_int4(as_global(sg.qs) + 4 * q), ve = load_int4(as_global(sg.qe) + 4 * q);
                    s[u][0] = vs.x, s[u][1] = vs.y, s[u][2] = vs.z, s[u][3] = vs.w;
                    e[u][0] = ve.x, e[u][1] = ve.y, e[u][2] = ve.z, e[u][3] = ve.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool in = q < q1 && k0 + j >= lo && k0 + j < hi;
                        s[u][j] = in ? as_global(sg.qs)[k0 + j] : 0, e[u][j] = in ? as_global(sg.qe)[k0 + j] : 0;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BS_UNROLL; u++) {
                const int64_t q = qb + (int64_t)u * THREADS, k0 = 4 * q;
                if (q >= q1) break;
                unsigned c[4];
                if (unit >= 0) {
                    bd_v4u rec;
                    rec.x = bm_record_of(s[u][0], e[u][0], g), rec.y = bm_record_of(s[u][1], e[u][1], g);
                    rec.z = bm_record_of(s[u][2], e[u][2], g), rec.w = bm_record_of(s[u][3], e[u][3], g);
                    if (WIDE)
                        bo_count_slot<false, THREADS == BD_THREADS>(I, rec, c);
                    else
                        bp_count_slot(I, rec, c);
                } else {
                    c[0] = c[1] = c[2] = c[3] = 0xFFFFu;
                }
                // (0xFFFF: an escape record -- improper, off the grid, longer than a record holds -- or a count that does not fit 16 bits)
                if (c[0] == 0xFFFFu || c[1] == 0xFFFFu || c[2] == 0xFFFFu || c[3] == 0xFFFFu || !whole[u]) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const bool in = k0 + j >= lo && k0 + j < hi;
                        if (in && c[j] == 0xFFFFu) c[j] = (unsigned)bm_escape_count(sg.ix, sg.e_sorted, g, s[u][j], e[u][j]);
                        if (!in) c[j] = 0u;
                    }
                }
                if (sg.counts) {
                    if (whole[u]) {
                        store_int4(as_global(sg.counts) + 4 * q, (int)c[0], (int)c[1], (int)c[2], (int)c[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (k0 + j >= lo && k0 + j < hi) as_global(sg.counts)[k0 + j] = (int)c[j];
                    }
                }
                acc += (long long)c[0] + c[1] + c[2] + c[3];
            }
        }
        __syncthreads();  // (thread 0 writes the next item's number, the next image may replace this one)
        it = it_nx;
    }
    if (total_slots && acc_seg >= 0) block_accumulate_i64(acc, red, total_slots + (int64_t)acc_seg * PT_SLOTS + (blockIdx.x & (PT_SLOTS - 1)));
}

// ---------------------------------------------------------------------------
// counts back into query order
// ---------------------------------------------------------------------------
// bm_unpermute_kernel for 16-bit counts: half the bytes to read, half the LDS (two workgroups share a CU).
// 0xFFFF = recompute from the sealed index (escape records, counts of 65535 and more).
template <int THREADS, int ITEMS, bool PAD, bool W8>
__device__ __forceinline__ void bd_unpermute_tile(const unsigned short *__restrict__ cnt /* tile-sorted; W8: bytes */,
                                                  const unsigned short *__restrict__ slots, const BmSeg *__restrict__ segs,
                                                  const unsigned short *__restrict__ tile_seg,
                                                  unsigned long long *__restrict__ total_slots /* [segments][PT_SLOTS], may be NULL */,
                                                  const unsigned *__restrict__ gate, const unsigned *__restrict__ tend,
                                                  unsigned long long *__restrict__ fb_dev /* W8: counts that did not fit, ever */,
                                                  unsigned long long *__restrict__ fb_host /* its copy in host memory */)
{
    constexpr int TILE = THREADS * ITEMS;
    constexpr int STRIDE = PAD ? TILE + BM_PAD_ROOM : TILE;  // slots between two tiles of the count array (PAD: gaps between the units)
    static_assert(ITEMS % 8 == 0, "whole 16-byte vectors of 16-bit counts per thread");
    extern __shared__ __attribute__((aligned(16))) int32_t dyn[];
    if (gate && *gate == 0) return;
    unsigned short *vals = reinterpret_cast<unsigned short *>(dyn);  // [TILE]
    __shared__ long long red[THREADS / 64];
    const int64_t tile = blockIdx.x;
    const int seg_id = tile_seg[tile];
    const BmSeg &sg = segs[seg_id];
    const int64_t ltile = tile - sg.tile0;
    if (ltile >= sg.ntiles) return;  // padding up to the next plan group
    // (one of the segment's 64 partial totals; bm_fold_totals_kernel adds them up.  Straight to the caller's word instead -- no fold
    // launch -- was tried in round 6: 3052 atomics on ONE address per pass stretch this kernel from 124 to 136 us, the fold costs 4.5;
    // folding by the workgroup that draws the last ticket: 147 us, every workgroup waits for a returning atomic while it holds half a CU.)
    unsigned long long *const acc_to = total_slots ? total_slots + (int64_t)seg_id * PT_SLOTS + (blockIdx.x & (PT_SLOTS - 1)) : nullptr;
    const IndexDev ix = sg.ix;
    const BmGeom g = sg.g;
    const int32_t *__restrict__ e_sorted = sg.e_sorted;
    const int32_t BX_GLOBAL *__restrict__ qs_arr = as_global(sg.qs) + ltile * TILE, *__restrict__ qe_arr = as_global(sg.qe) + ltile * TILE;  // escapes only
    int32_t BX_GLOBAL *__restrict__ out = as_global(sg.counts) + ltile * TILE;  // (as_global: common.hpp)
    static_assert(!W8 || PAD, "8-bit counts come with the padded layout");
    const unsigned char *vals8 = reinterpret_cast<const unsigned char *>(dyn);
    cnt = W8 ? reinterpret_cast<const unsigned short *>(reinterpret_cast<const unsigned char *>(cnt) + tile * STRIDE) : cnt + tile * STRIDE;
    slots += tile * TILE;  // scratch is laid out by the batch's tile numbering
    const int64_t nq = sg.nq - ltile * TILE;
    const int n = (int)(nq < TILE ? nq : TILE);
    constexpr unsigned ESC = W8 ? 0xFFu : 0xFFFFu;
    if (PAD) {
        const int n8 = W8 ? ((int)tend[tile] + 15) >> 4 : ((int)tend[tile] + 7) >> 3;  // the slots the tile's sorted order uses
        const int4 *src = reinterpret_cast<const int4 *>(cnt);
        for (int i = threadIdx.x; i < n8; i += THREADS) reinterpret_cast<int4 *>(vals)[i] = src[i];
    } else {
        const int4 *src = reinterpret_cast<const int4 *>(cnt);
        if (n == TILE) {
            int4 v[ITEMS / 8];
#pragma unroll
            for (int j = 0; j < ITEMS / 8; j++) v[j] = src[j * THREADS + threadIdx.x];
#pragma unroll
            for (int j = 0; j < ITEMS / 8; j++) reinterpret_cast<int4 *>(vals)[j * THREADS + threadIdx.x] = v[j];
        } else {
            const int n8 = (n + 7) >> 3;  // (the scratch is padded to whole tiles)
            for (int i = threadIdx.x; i < n8; i += THREADS) reinterpret_cast<int4 *>(vals)[i] = src[i];
        }
    }
    __syncthreads();
    long long acc = 0;
    unsigned wide = 0;  // W8: counts of this thread that came back as "ask again"
    if (PAD && W8 && !sg.counts) {
        // TOTAL ONLY (the caller passed no counts array): the tile's total is the sum of its count bytes -- no slot is needed -- unless
        // a REAL query's count came back as "ask again".  The tile sort's padding comes back as 0xFF too, but how many pad slots
        // the tile has is known (tend - n): if exactly that many bytes are 0xFF, none belongs to a query and the 0.2 GB of slots
        // stay unread; otherwise the tile takes the ordinary path below.
        const int n8 = ((int)tend[tile] + 15) >> 4;
        unsigned sum = 0, nff = 0;
        for (int i = threadIdx.x; i < n8; i += THREADS) {
            const int4 v = reinterpret_cast<const int4 *>(vals)[i];
            const unsigned w[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                // bytes beyond the tile's used slots (the last vector's tail) are whatever the search left there: mask them out
                const int first = 16 * i + 4 * k, used = (int)tend[tile] - first;
                const unsigned keep = used >= 4 ? 0xFFFFFFFFu : (used <= 0 ? 0u : (1u << (8 * used)) - 1u);
                const unsigned x = w[k] & keep;
                const unsigned ff = ((x & 0x7F7F7F7Fu) + 0x01010101u) & x & 0x80808080u;  // bit 7 of every byte that is 0xFF
                const unsigned c = (unsigned)__popc(ff);
                nff += c;
                sum += __builtin_amdgcn_sad_u8(x, 0u, 0u) - 255u * c;
            }
        }
        __shared__ unsigned s_tot[2];
        if (threadIdx.x < 2) s_tot[threadIdx.x] = 0u;
        __syncthreads();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum += (unsigned)__shfl_down((int)sum, off, 64);
            nff += (unsigned)__shfl_down((int)nff, off, 64);
        }
        if (lane_id() == 0) {
            atomicAdd(&s_tot[0], sum);
            atomicAdd(&s_tot[1], nff);
        }
        __syncthreads();
        if (s_tot[1] == tend[tile] - (unsigned)n) {  // (block-uniform) every 0xFF is padding
            if (acc_to && threadIdx.x == 0 && s_tot[0]) atomicAdd(acc_to, (unsigned long long)s_tot[0]);
            return;
        }
    }
    if (n == TILE) {
        const uint2 *l4 = reinterpret_cast<const uint2 *>(slots);
        int32_t BX_GLOBAL *o4 = out;
        uint2 sl[ITEMS / 4];
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) sl[j] = l4[j * THREADS + threadIdx.x];
#pragma unroll
        for (int j = 0; j < ITEMS / 4; j++) {
            unsigned c[4];
            if (W8) {
                c[0] = vals8[sl[j].x & 0xffffu], c[1] = vals8[sl[j].x >> 16], c[2] = vals8[sl[j].y & 0xffffu], c[3] = vals8[sl[j].y >> 16];
            } else {
                c[0] = vals[sl[j].x & 0xffffu], c[1] = vals[sl[j].x >> 16], c[2] = vals[sl[j].y & 0xffffu], c[3] = vals[sl[j].y >> 16];
            }
            if (c[0] == ESC || c[1] == ESC || c[2] == ESC || c[3] == ESC) {
                const int64_t k0 = 4 * (int64_t)(j * THREADS + threadIdx.x);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (c[u] == ESC) {
                        c[u] = (unsigned)bm_escape_count(ix, e_sorted, g, qs_arr[k0 + u], qe_arr[k0 + u]);
                        wide++;
                    }
            }
            // (no counts array: the caller asked for the total only -- the 4 bytes per query of this store are 0.4 of the kernel's
            // 0.71 GB per 100 M queries)
            if (sg.counts) store_int4(o4 + 4 * (j * THREADS + (int)threadIdx.x), (int)c[0], (int)c[1], (int)c[2], (int)c[3]);
            acc += (long long)c[0] + c[1] + c[2] + c[3];
        }
    } else {
        for (int k = threadIdx.x; k < n; k += THREADS) {
            unsigned c = W8 ? (unsigned)vals8[slots[k]] : (unsigned)vals[slots[k]];
            if (c == ESC) {
                c = (unsigned)bm_escape_count(ix, e_sorted, g, qs_arr[k], qe_arr[k]);
                wide++;
            }
            if (sg.counts) out[k] = (int)c;
            acc += c;
        }
    }
    if (acc_to) block_accumulate_i64(acc, red, acc_to);
    if (W8 && fb_dev) {
        // feedback for the host's choice of the count width (bm_count_segments): a running total of the counts that did
        // not fit 8 bits, and -- from whichever workgroup comes first -- its value so far into host memory, where the
        // next call finds it without a synchronisation (a pass or two late: the choice only has to converge)
        if (__any(wide != 0u)) {
            unsigned w = wide;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) w += (unsigned)__shfl_down((int)w, off, 64);
            if (lane_id() == 0 && w) atomicAdd(fb_dev, (unsigned long long)w);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0 && fb_host) *fb_host = __hip_atomic_load(fb_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// (66-68 registers = seven waves per SIMD = one 1024-thread workgroup per CU; held to eight waves, and with the staging loads below
// requested together, the pass is no faster: 0.662-0.672 against 0.668-0.669 ms, round 6)
template <int THREADS, int ITEMS, bool PAD = false, bool W8 = false>
__global__ __launch_bounds__(THREADS) void bd_unpermute_kernel(const unsigned short *__restrict__ cnt, const unsigned short *__restrict__ slots,
                                                               const BmSeg *__restrict__ segs, const unsigned short *__restrict__ tile_seg,
                                                               unsigned long long *__restrict__ total_slots, const unsigned *__restrict__ gate,
                                                               const unsigned *__restrict__ tend = nullptr, unsigned long long *__restrict__ fb_dev = nullptr,
                                                               unsigned long long *__restrict__ fb_host = nullptr)
{
    bd_unpermute_tile<THREADS, ITEMS, PAD, W8>(cnt, slots, segs, tile_seg, total_slots, gate, tend, fb_dev, fb_host);
}

}  // namespace bxmi
