#!/usr/bin/env python
"""
One number per line of `bed1`: how many regions of `bed2` it overlaps.

usage: %prog bed1 bed2
"""
# Counterpart of the reference's scripts/bed_count_by_interval.py:17-35 (identical to
# bed_count_overlapping.py in the reference): per chromosome ONE batched count instead of a find() per line.
import sys

import numpy as np

from bxmi.intervals import IntervalIndex


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    bed1, bed2 = args[0:2]
    ranges = {}
    for line in open(bed2):
        fields = line.strip().split()
        start, end = int(fields[1]), int(fields[2])
        assert start <= end, "start must be less than end"  # Interval() in the reference (intersection.pyx:291)
        s, e = ranges.setdefault(fields[0], ([], []))
        s.append(start), e.append(end)
    rows, per, error = [], {}, None
    for line in open(bed1):
        try:
            fields = line.strip().split()
            chrom, start, end = fields[0], int(fields[1]), int(fields[2])
            if not (-2147483648 <= start <= 2147483647 and -2147483648 <= end <= 2147483647) and chrom in ranges:
                raise OverflowError("value too large to convert to int")
            rows.append(" ".join(fields[:3] + [" ".join(fields[3:])]))
            if chrom in ranges:
                s, e, idx = per.setdefault(chrom, ([], [], []))
                s.append(start), e.append(end), idx.append(len(rows) - 1)
        except (ValueError, IndexError, OverflowError) as ex:
            error = ex
            break
    counts = np.zeros(len(rows), dtype=np.int64)
    for chrom, (s, e, idx) in per.items():
        ix = IntervalIndex()
        ix.append(np.array(ranges[chrom][0], dtype=np.int64), np.array(ranges[chrom][1], dtype=np.int64))
        counts[np.array(idx, dtype=np.int64)] = ix.count(np.array(s, dtype=np.int32), np.array(e, dtype=np.int32))[0]
        ix.close()
    w = out.write
    for row, c in zip(rows, counts.tolist()):
        w("%s %d\n" % (row, c))
    out.flush()
    if error is not None:
        raise error


if __name__ == "__main__":
    main()
