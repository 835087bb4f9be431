#!/usr/bin/env python
"""
Two interval files with chromosome columns: counts how many entries of the second overlap at least one entry of the
first.

usage: %prog bed1 bed2 > out
"""
# Counterpart of the reference's scripts/interval_count_intersections.py:19-47: per chromosome one batched
# count; an entry of the second set is counted when at least one interval of the first overlaps it.
import bz2
import gzip
import sys

import numpy as np

from bxmi.intervals import IntervalIndex


def open_compressed(filename):  # bx/misc/__init__.py:9-15
    if filename.endswith(".bz2"):
        return bz2.open(filename, "rt")
    if filename.endswith(".gz"):
        return gzip.open(filename, "rt")
    return open(filename)


def read_intervals(inp):
    for line in inp:
        fields = line.split()
        yield fields[0], int(fields[1]), int(fields[2])


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    targets, queries = {}, {}
    for chrom, start, end in read_intervals(open_compressed(args[0])):
        assert start <= end, "start must be less than end"  # Interval() in the reference
        s, e = targets.setdefault(chrom, ([], []))
        s.append(start), e.append(end)
    for chrom, start, end in read_intervals(open_compressed(args[1])):
        if chrom in targets:
            s, e = queries.setdefault(chrom, ([], []))
            s.append(start), e.append(end)
    total = 0
    for chrom, (qs, qe) in queries.items():
        ix = IntervalIndex()
        ix.append(np.array(targets[chrom][0], dtype=np.int64), np.array(targets[chrom][1], dtype=np.int64))
        counts, _ = ix.count(np.array(qs, dtype=np.int64), np.array(qe, dtype=np.int64))
        ix.close()
        total += int((counts > 0).sum())
    out.write("%d\n" % total)
    out.flush()


if __name__ == "__main__":
    main()
