// CPU check of bx-python_amd/csrc/offset_cells.hpp (compiled by tests/test_host_logic.py with g++): the cell encoding
// against brute force, and a scalar model of one unit's two images answering overlap counts the way the search kernel
// does -- count = (sLo + rankS(off + len)) - (eLo + rankE(off + 1)) -- against the definition of an overlap.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "offset_cells.hpp"

using namespace bxmi;

struct Image {
    std::vector<unsigned> lo_w, hi_w;              // the cells
    std::vector<std::vector<unsigned char>> list;  // hard cells' lists (index in lo_w), empty = "search the sorted array"
};

// the builder's logic, serially: keys (sorted) with rel = key - lo in [0, span) -> cells of 2^k coordinates
static Image build(const std::vector<long long> &keys, long long lo, long long span, int k, int ncells)
{
    Image im;
    im.lo_w.assign(ncells, 0), im.hi_w.assign(ncells, 0);
    std::vector<std::vector<unsigned char>> per(ncells);
    for (long long key : keys)
        if (key >= lo && key - lo < span) per[(key - lo) >> k].push_back((unsigned char)((key - lo) & ((1 << k) - 1)));
    unsigned base = 0;
    for (int c = 0; c < ncells; c++) {
        const int m = (int)per[c].size();
        if (m <= BO_INLINE) {
            bo_pack(per[c].data(), m, base, im.lo_w[c], im.hi_w[c]);
        } else {
            im.hi_w[c] = (base & 0xFFFFFu) | BO_HARD;
            im.lo_w[c] = (unsigned)im.list.size();
            im.list.push_back(per[c]);
        }
        base += (unsigned)m;
    }
    return im;
}

static unsigned rank_of(const Image &im, unsigned rel, int k)
{
    const unsigned c = rel >> k, p = rel & ((1u << k) - 1u);
    if (im.hi_w[c] >= BO_HARD) {
        unsigned r = im.hi_w[c] & 0xFFFFFu;
        const unsigned n = (im.hi_w[c + 1] & 0xFFFFFu) - r;
        if (n != im.list[im.lo_w[c]].size()) { printf("list length of a hard cell\n"); exit(1); }
        for (unsigned char o : im.list[im.lo_w[c]]) r += o < p;
        return r;
    }
    return bo_rank(im.lo_w[c], im.hi_w[c], p);
}

int main()
{
    std::mt19937_64 rng(12345);
    // 1. pack / rank of single cells, every position
    for (int it = 0; it < 20000; it++) {
        const int n = (int)(rng() % (BO_INLINE + 1));
        unsigned char offs[BO_INLINE];
        for (int i = 0; i < n; i++) offs[i] = (unsigned char)(rng() % 256);
        std::sort(offs, offs + n);
        const unsigned base = (unsigned)(rng() % (1u << 20));
        unsigned lo_w, hi_w;
        bo_pack(offs, n, base, lo_w, hi_w);
        if (hi_w >= BO_HARD) { printf("a plain cell looks hard\n"); return 1; }
        for (unsigned p = 0; p < 256; p++) {
            unsigned want = base;
            for (int i = 0; i < n; i++) want += offs[i] < p;
            if (bo_rank(lo_w, hi_w, p) != want) { printf("rank: n %d p %u\n", n, p); return 1; }
        }
    }
    // 2. the record format and the density rule
    for (int k = BO_MIN_K; k <= BO_MAX_K; k++)
        if (bo_rshift(k) != 12 + k || bo_margin(k) != (1 << (20 - k))) { printf("format k %d\n", k); return 1; }
    if (bo_cell_log2_for(249000000, 805000) != 8 || bo_cell_log2_for(250000000, 10000000) != 0 || bo_cell_log2_for(64 * 1000, 1000) != 6 ||
        bo_cell_log2_for(1000 * 128 * 10 / 12 + 1, 1000) != 7) { printf("density rule\n"); return 1; }
    // 3. one unit: counts from the two images against the definition
    for (int it = 0; it < 60; it++) {
        const int k = BO_MIN_K + (int)(rng() % 3), ulog = k + 4 + (int)(rng() % 4);  // 16 .. 128 cells per unit
        const long long UW = 1ll << ulog, lo = (long long)(rng() % 100000) - 50000;
        const int margin = bo_margin(k), rshift = bo_rshift(k);
        const int n = 20 + (int)(rng() % 400);
        std::vector<long long> S(n), E(n);
        for (int i = 0; i < n; i++) {
            S[i] = lo - 3000 + (long long)(rng() % (UW + margin + 6000));
            if (rng() % 4 == 0 && i) S[i] = S[i - 1];  // repeated coordinates, piles
            E[i] = S[i] + (long long)(rng() % 3000);
        }
        std::vector<long long> Ss(S), Es(E);
        std::sort(Ss.begin(), Ss.end()), std::sort(Es.begin(), Es.end());
        const int nce = (int)(UW >> k) + 2, ncs = (int)((UW + margin) >> k) + 1;
        const Image IE = build(Es, lo, UW + 1, k, nce), IS = build(Ss, lo, UW + margin, k, ncs);
        const long long eLo = std::lower_bound(Es.begin(), Es.end(), lo) - Es.begin(), sLo = std::lower_bound(Ss.begin(), Ss.end(), lo) - Ss.begin();
        for (int q = 0; q < 3000; q++) {
            const unsigned off = (unsigned)(rng() % UW), len = 1 + (unsigned)(rng() % (margin - 2));  // (margin - 1 = the escape length)
            const unsigned rec = off | (len << rshift);
            const unsigned o2 = rec & ((1u << ulog) - 1u), l2 = rec >> rshift;
            const long long qs = lo + off, qe = qs + len;
            long long want = 0;
            for (int i = 0; i < n; i++) want += S[i] < qe && E[i] > qs;
            const long long got = (sLo + rank_of(IS, o2 + l2, k)) - (eLo + rank_of(IE, o2 + 1, k));
            if (got != want) { printf("count: k %d ulog %d off %u len %u got %lld want %lld\n", k, ulog, off, len, got, want); return 1; }
        }
    }
    printf("offset cells ok\n");
    return 0;
}
