// Offset cells: the unit images of SPARSE indexes for the persistent walk (count_dense.hpp, bw_search_kernel<.., true>).
//
// A bitmap cell (count_bitmap.hpp) spends one bit per coordinate: 131 KB of LDS hold a unit of 2^18 coordinates, and an
// index with one target per 300 coordinates (a chromosome of configs[3]) would stream 1.5 GB of images per pass for its
// 100 M queries -- which is why such indexes were left to the key slices, whose look-up is a search (5 x the time per
// query of a cell look-up).  An OFFSET cell covers 2^k coordinates (k = 6..8, chosen from the index's density so that a cell
// holds about one key) in the same 8 bytes:
//     low word   offsets of the cell's keys 0..3 inside the cell, one byte each, ascending; 0xFF = no key
//     high word  [19:0] rank of the cell's first key in the unit's slice, [27:20] offset of key 4 (0xFF = none),
//                [31:28] 0 = plain, 0xF = HARD (more than five keys)
// so a unit of 4096 cells per side is 2^(12+k) coordinates in 72 KB -- 2^20 at k = 8: four times fewer images than bitmap
// cells for the same span, each half the size, so that TWO search workgroups share a CU and one loads its image while the
// other looks records up -- and a rank is base + #{offsets below the position}: one ds_read_b64 and five compares.
// Duplicated coordinates are simply repeated offsets.  (0xFF as "no key" needs no count field: a position is at most 255,
// so neither an absent key nor a real key at offset 255 is ever below it -- and a key at offset 255 only matters to the
// next cell's base.)
// A HARD cell keeps its keys' offsets as a list of up to 64 bytes in the image's overflow area (low word = byte offset of
// the list in the image; its length is the next cell's base minus its own); longer ones are finished by a search in
// the sorted array (low word = all ones).
//
// This header is plain C++ as well: tests/test_host_logic.py compiles it with g++ and checks pack / rank / layout
// against brute force on the CPU.
#pragma once
#if defined(__HIPCC__)
#define BO_HD __host__ __device__ __forceinline__
#else
#define BO_HD inline
#endif

namespace bxmi {

constexpr int BO_INLINE = 5;               // keys a cell holds itself
constexpr unsigned BO_HARD = 0xF0000000u;  // high word of a hard cell (or'ed to its base)
constexpr int BO_LIST = 64;                // bytes of a hard cell's list in the overflow area
constexpr int BO_CELLS_LOG2 = 12;          // cells per side of a full unit
constexpr int BO_TABLES = 96;              // lists per unit image (its overflow area)
constexpr int BO_MIN_K = 6, BO_MAX_K = 8;  // cell widths (log2); 5 is the bitmap cell's

// n <= BO_INLINE ascending offsets -> the cell's two words
BO_HD void bo_pack(const unsigned char *offs, int n, unsigned base, unsigned &lo_w, unsigned &hi_w)
{
    unsigned b[BO_INLINE];
    for (int i = 0; i < BO_INLINE; i++) b[i] = i < n ? (unsigned)offs[i] : 0xFFu;
    lo_w = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    hi_w = (base & 0xFFFFFu) | (b[4] << 20);
}

// #{keys of the unit's slice below position p of this (plain) cell}
BO_HD unsigned bo_rank(unsigned lo_w, unsigned hi_w, unsigned p)
{
    unsigned r = hi_w & 0xFFFFFu;
    r += (lo_w & 0xFFu) < p ? 1u : 0u;
    r += ((lo_w >> 8) & 0xFFu) < p ? 1u : 0u;
    r += ((lo_w >> 16) & 0xFFu) < p ? 1u : 0u;
    r += (lo_w >> 24) < p ? 1u : 0u;
    r += ((hi_w >> 20) & 0xFFu) < p ? 1u : 0u;
    return r;
}

// the record format that goes with cells of 2^k coordinates: offset : 12 + k | length : 20 - k (all ones = escape)
BO_HD int bo_rshift(int k) { return BO_CELLS_LOG2 + k; }
BO_HD int bo_margin(int k) { return 1 << (32 - bo_rshift(k)); }  // the starts' cells reach this far past the unit

// The cell width for an index of n keys over `span` coordinates: the widest k <= 8 whose cells hold 1.2 keys or fewer on
// average (Poisson: 2 in 10 000 cells of such an index are hard at the limit); below 6 the index is dense enough for bitmap cells.
BO_HD int bo_cell_log2_for(long long span, long long n)
{
    int k = 0;
    for (int c = BO_MIN_K; c <= BO_MAX_K; c++)
        if (n * ((long long)1 << c) * 10 <= span * 12) k = c;
    return k;  // 0 = none
}

}  // namespace bxmi
