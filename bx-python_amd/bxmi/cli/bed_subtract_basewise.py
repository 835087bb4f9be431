#!/usr/bin/env python
"""
Base-level difference: the maximal stretches `bed_file_1` covers and `bed_file_2` does not.

usage: %prog bed_file_1 bed_file_2
"""
# Counterpart of the reference's scripts/bed_subtract_basewise.py:22-43: invert + iand on the device,
# then one run extraction per chromosome instead of the next_set/next_clear walk.
import sys

from bxmi.builders import binned_bitsets_from_file, group_iand, write_runs


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    try:
        in_fname, in2_fname = args
    except ValueError:
        raise SystemExit(__doc__.replace("%prog", sys.argv[0]))
    bitsets1 = binned_bitsets_from_file(open(in_fname))
    bitsets2 = binned_bitsets_from_file(open(in2_fname))
    shared = [chrom for chrom in bitsets1 if chrom in bitsets2]
    for chrom in shared:
        bitsets2[chrom].invert()
    group_iand([bitsets1[c] for c in shared], [bitsets2[c] for c in shared])  # one launch for the genome's and-not
    for chrom, bits1 in bitsets1.items():
        write_runs(out, chrom, bits1)
    out.flush()


if __name__ == "__main__":
    main()
