#!/usr/bin/env python
"""
Total bases covered by the BED input (files named on the command line, else stdin); a base under several
intervals counts once.

usage: %prog bed files ...
"""
# Counterpart of the reference's scripts/bed_coverage.py:19-31: one set_ranges launch and one
# grid-wide popcount per chromosome.
import fileinput
import sys

from bxmi.builders import binned_bitsets_from_file, binned_bitsets_from_paths


def main(argv=None, out=None, stdin=None):
    out = out or sys.stdout
    bed_filenames = sys.argv[1:] if argv is None else argv
    if bed_filenames and "-" not in bed_filenames:
        bitsets = binned_bitsets_from_paths(bed_filenames)  # what fileinput would chain, each file ingested in bulk
    else:
        bitsets = binned_bitsets_from_file(fileinput.input(bed_filenames) if bed_filenames else (stdin or sys.stdin))
    total = 0
    for chrom in bitsets:
        total += bitsets[chrom].count_range(0, bitsets[chrom].size)
    out.write("%d\n" % total)
    out.flush()


if __name__ == "__main__":
    main()
