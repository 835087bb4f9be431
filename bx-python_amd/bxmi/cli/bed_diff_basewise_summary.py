#!/usr/bin/env python
"""
Three base counts for a pair of BED files: covered by both, by the first only, by the second only.

usage: %prog bed_file_1 bed_file_2
"""
# Counterpart of the reference's scripts/bed_diff_basewise_summary.py:13-44: the three coverage sums are
# group popcounts (one launch per genome) and the intersection is the fused group AND + count.
import sys

from bxmi.bitset import BitSetGroup
from bxmi.builders import binned_bitsets_from_file


def coverage(bitsets):
    if not bitsets:
        return 0
    for b in bitsets.values():
        b._flush()
    return int(BitSetGroup([b._d for b in bitsets.values()]).popcounts().sum())


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    try:
        in_fname, in2_fname = args
    except ValueError:
        raise SystemExit(__doc__.replace("%prog", sys.argv[0]))
    bits1 = binned_bitsets_from_file(open(in_fname))
    bits2 = binned_bitsets_from_file(open(in2_fname))
    bits1_covered = coverage(bits1)
    bits2_covered = coverage(bits2)
    shared = [k for k in bits1 if k in bits2]
    both_covered = 0
    if shared:
        for k in shared:
            bits1[k]._flush(), bits2[k]._flush()
            bits1[k]._touch()
        ga, gb = BitSetGroup([bits1[k]._d for k in shared]), BitSetGroup([bits2[k]._d for k in shared])
        both_covered = int(ga.iand(gb, want_counts=True).sum())
    out.write("in both:  \t%d\n" % both_covered)
    out.write("only in %s:\t%d\n" % (in_fname, bits1_covered - both_covered))
    out.write("only in %s:\t%d\n" % (in2_fname, bits2_covered - both_covered))
    out.flush()


if __name__ == "__main__":
    main()
