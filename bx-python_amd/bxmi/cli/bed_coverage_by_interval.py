#!/usr/bin/env python
"""
For every line of `bed1`, the covered fraction: bases of that interval that `bed2` touches, divided by its length.

usage: %prog bed1 bed2 [mask]
"""
# Counterpart of the reference's scripts/bed_coverage_by_interval.py:9-53: all count_range calls of a
# chromosome go out as one batched launch (two with a mask).
import sys

import numpy as np

from bx.bitset import BinnedBitSet
from bxmi.builders import _check_range, binned_bitsets_from_file


def clone(bits):
    b = BinnedBitSet(bits.size)
    b.ior(bits)
    return b


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    bed1_fname, bed2_fname = args[0:2]
    bitsets = binned_bitsets_from_file(open(bed2_fname))
    mask = None
    if len(args) > 2:
        mask = binned_bitsets_from_file(open(args[2]))
        new_bitsets = {}
        for key in bitsets:
            if key in mask:
                b = clone(mask[key])
                b.invert()
                b.iand(bitsets[key])
                new_bitsets[key] = b
        bitsets = new_bitsets
    rows, error = [], None
    per = {}  # chrom -> (starts, counts, row indices)
    for line in open(bed1_fname):
        try:
            fields = line.split()
            chrom, start, end = fields[0], int(fields[1]), int(fields[2])
            for sets in (bitsets, mask):
                if sets and chrom in sets and error is None:
                    error = _check_range(sets[chrom].size, start, end - start)
            if error is not None:
                break
            s, c, idx = per.setdefault(chrom, ([], [], []))
            s.append(start), c.append(end - start), idx.append(len(rows))
            rows.append((chrom, end - start))
        except (ValueError, IndexError) as ex:
            error = ex
            break
    covered = np.zeros(len(rows), dtype=np.int64)
    masked = np.zeros(len(rows), dtype=np.int64)
    for chrom, (s, c, idx) in per.items():
        s, c, idx = np.array(s, dtype=np.int32), np.array(c, dtype=np.int32), np.array(idx, dtype=np.int64)
        if chrom in bitsets:
            covered[idx] = bitsets[chrom].count_ranges(s, c)
        if mask and chrom in mask:
            masked[idx] = mask[chrom].count_ranges(s, c)
    w = out.write
    for (chrom, length), bases_covered, bases_masked in zip(rows, covered.tolist(), masked.tolist()):
        length -= bases_masked
        assert bases_covered <= length, f"{bases_covered!r}, {bases_masked!r}, {length!r}"
        w("0.0\n" if length == 0 else "%r\n" % (bases_covered / length))
    out.flush()
    if error is not None:
        raise error


if __name__ == "__main__":
    main()
