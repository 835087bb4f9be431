#!/usr/bin/env python3
"""Stage-by-stage check of find() through the exchange against the oracle: counts, query-order and tile-sorted scratch offsets, `hc`,
the half-bucket run table and every record's hits in the scratch list, read back through bxmi_debug_peek -- a symbol only DEBUG builds
export: tools/build_variant.sh peek "-DBXMI_DEBUG_PEEK=1", then copy build_variants/libbxmi_peek.so over bx-python_amd/bxmi/libbxmi.so
for the run.  (How round 5 found the shuffle that ran inside a divergent select: DESIGN.md 3.2.)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bx-python_amd"))
import numpy as np
from bxmi import _ffi
from bxmi.intervals import IntervalIndex
from oracle import oracle as O
def set_opt(k, v): _ffi.call("bxmi_set_option", k.encode(), int(v))
L = _ffi.load()
def peek(ix, name, dtype, count):
    a = np.empty(count, dtype=dtype)
    rc = L.bxmi_debug_peek(ix._h, name.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes))
    assert rc == 0, (name, rc)
    return a
rng = np.random.default_rng(70)
n, span = 100_000, 30_000_000
s = rng.integers(1000, span, size=n); e = s + rng.integers(0, 1200, size=n)
NQ = int(os.environ.get("NQ", 50_000))
qs = rng.integers(0, span + 2000, size=50_000)[:NQ]; qe = (qs + rng.integers(1, 2500, size=50_000)[:NQ])
s, e, qs, qe = (a.astype(np.int32) for a in (s, e, qs, qe))
t = O.OracleIntervalTree(); t.insert_many_arrays(s, e)
w_off, w_hits = t.find_batch(qs, qe)
ix = IntervalIndex(); ix.append(s, e); ix.seal()
set_opt("ivl.partition", 1); set_opt("ivl.bitmap_min", 1); set_opt("ivl.sorted_path", 0)
for rep in range(int(os.environ.get("REPS", 2))):
    off, hits = ix.find(qs, qe)
    TILE, ntp = 16384, 64
    ntiles = (NQ + TILE - 1) // TILE
    slots = peek(ix, "slots", np.uint16, ntp * TILE); recs = peek(ix, "recs", np.uint32, ntp * TILE)
    hc = peek(ix, "hc", np.uint32, ntp * TILE); cnt = peek(ix, "cnt", np.uint32, ntp * TILE); loff = peek(ix, "loff", np.uint32, ntp * TILE)
    svq = peek(ix, "svq", np.uint32, ntp * TILE); tile_base = peek(ix, "tile_base", np.int64, ntiles + 1)
    runT2 = peek(ix, "runT2", np.uint32, 4096 * ntp).reshape(4096, ntp); tmp = peek(ix, "tmp_hits", np.int32, int(w_off[-1]))
    meta2 = peek(ix, "meta2", np.int32, 4097 * 2).reshape(4097, 2); qcnt = peek(ix, "qcnt", np.int32, NQ)
    pcs = peek(ix, "pieces", np.int32, 4096 * 4).reshape(4096, 4)
    f = int(np.log2(pcs[0, 1] - pcs[0, 0])) - 1
    npieces = int(np.argmax(pcs[:, 1] >= 4096)) + 1
    cmin = int(s.min()); sp = int(e.max()) - cmin; shift = 0
    while (sp >> shift) >= 2048: shift += 1
    q = np.arange(NQ); tile = q // TILE; pos = tile * TILE + slots[q].astype(np.int64)
    want_c = np.diff(w_off).astype(np.int64)
    esc = (loff[pos] >> 31) != 0
    ok = ~esc
    print("rep", rep, "f", f, "npieces", npieces, "escapes", int(esc.sum()), "final bad hits", int((hits != w_hits).sum()), "offsets ok", np.array_equal(off, w_off))
    print(" A cnt != want:", int((cnt[pos][ok] != want_c[ok]).sum()), " qcnt != want:", int((qcnt != want_c).sum()))
    print(" B svq != loff[pos]:", int((svq[q] != loff[pos]).sum()))
    ssort = np.sort(s.astype(np.int64))
    sub = np.clip((qs.astype(np.int64) - cmin) >> (shift - 1), 0, 4095); sub[qs < cmin] = 0
    unit = sub >> (f + 1)
    hi_want = np.searchsorted(ssort, qe.astype(np.int64), side="left")
    hi_got = meta2[unit << (f + 1), 0].astype(np.int64) + (hc[pos] >> 16).astype(np.int64)
    print(" C hc count16 mismatch:", int(((hc[pos] & 0xffff)[ok] != np.minimum(want_c[ok], 0xffff)).sum()), " hi mismatch:", int((hi_got[ok] != hi_want[ok]).sum()))
    base = tile_base[tile] + (loff[pos] & 0x7fffffff).astype(np.int64)
    badD = []
    for i in np.nonzero(ok & (want_c > 0))[0]:
        if not np.array_equal(tmp[base[i]:base[i] + want_c[i]], w_hits[w_off[i]:w_off[i + 1]]): badD.append(int(i))
    print(" D tmp_hits region != oracle list:", len(badD), badD[:12])
    # E: the runs of every piece cover exactly the slots of its half buckets
    inv = np.full(ntp * TILE, -1, dtype=np.int64); inv[pos] = q
    badE = 0
    for t_ in range(ntiles):
        for p in range(npieces):
            sb0, sb1 = int(pcs[p, 0]), int(pcs[p, 1])
            a = int(runT2[sb0, t_] & 0xffff); r1 = int(runT2[sb1 - 1, t_]); b = (r1 & 0xffff) + (r1 >> 16)
            qq = inv[t_ * TILE + a: t_ * TILE + b]
            if (qq < 0).any() or ((sub[qq] < sb0) | (sub[qq] >= sb1)).any(): badE += 1
            # and nothing of the piece outside the run
    cover = sum(int(((int(runT2[int(pcs[p, 1]) - 1, t_]) & 0xffff) + (int(runT2[int(pcs[p, 1]) - 1, t_]) >> 16)) - (int(runT2[int(pcs[p, 0]), t_]) & 0xffff)) for t_ in range(ntiles) for p in range(npieces))
    print(" E runs with foreign records:", badE, " records covered:", cover, "of", NQ)
    for i in badD[:8]:
        t_, u = int(tile[i]), int(unit[i])
        p = int(np.nonzero((pcs[:npieces, 0] <= sub[i]) & (pcs[:npieces, 1] > sub[i]))[0][0]); sb0, sb1 = int(pcs[p, 0]), int(pcs[p, 1])
        lens = []
        for tt in range(ntiles):
            a = int(runT2[sb0, tt] & 0xffff); r1 = int(runT2[sb1 - 1, tt]); lens.append((r1 & 0xffff) + (r1 >> 16) - a)
        a3 = int(runT2[sb0, t_] & 0xffff)
        fl = sum(lens[:t_]) + int(slots[i]) - a3
        g = tmp[base[i]:base[i] + want_c[i]]
        print("  q", i, "tile", t_, "slot", int(slots[i]), "piece", p, (sb0, sb1), "run lens", lens, "flat pos", fl, "pass", fl // 64, "lane", fl % 64, "cnt", int(want_c[i]),
              "loff", int(loff[pos[i]]), "tmp got", g.tolist()[:8], "want", w_hits[w_off[i]:w_off[i + 1]].tolist()[:8], "final ok", bool(np.array_equal(hits[w_off[i]:w_off[i+1]], w_hits[w_off[i]:w_off[i+1]])))
