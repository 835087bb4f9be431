"""Replays a golden BinnedBitSet op sequence against any implementation of the
bx.bitset.BinnedBitSet interface and reports the first divergence."""
import numpy as np


def run_call(fn, *a):
    try:
        return ["ok", fn(*a)]
    except (IndexError, ValueError, OverflowError, TypeError) as ex:
        return [type(ex).__name__, str(ex)]


def norm(r):
    if r[0] == "ok" and r[1] is not None:
        return ["ok", int(r[1])]
    return list(r)


def replay(case, factory, check_final=True):
    size, gran = case["size"], case["granularity"]
    sets = {"A": factory(size, gran), "B": factory(size, gran)}
    assert sets["A"].bin_size == case["bin_size"], ("bin_size", sets["A"].bin_size, case["bin_size"])
    assert sets["A"].size == size
    for i, (which, op, x, y, expect) in enumerate(case["ops"]):
        t = sets[which]
        if op in ("set_range", "count_range"):
            got = run_call(getattr(t, op), x, y)
        elif op in ("next_set", "next_clear", "set", "clear"):
            got = run_call(getattr(t, op), x)
        elif op == "get":
            got = run_call(t.__getitem__, x)
        elif op == "invert":
            got = run_call(t.invert)
        elif op in ("iand", "ior"):
            got = run_call(getattr(t, op), sets[x])
        else:
            raise AssertionError(op)
        assert norm(got) == norm(expect), "size=%d gran=%d op#%d %s.%s(%r,%r): got %r want %r" % (
            size, gran, i, which, op, x, y, got, expect)
    if check_final and "final" in case:
        for w in ("A", "B"):
            bits = np.array([sets[w][p] for p in range(size)], dtype=np.uint8)
            got = np.packbits(bits, bitorder="little").tobytes().hex()
            assert got == case["final"][w], "final bits of %s differ (size=%d gran=%d)" % (w, size, gran)
            assert norm(run_call(sets[w].count_range, 0, size)) == norm(case["full_count"][w])
    return sets
