#!/usr/bin/env python
"""
Total bases covered by the BED input (files named on the command line, else stdin); a base under several
intervals counts once.

usage: %prog bed files ...
"""
# Counterpart of the reference's scripts/bed_coverage.py:19-31: one set_ranges launch per chromosome and ONE
# grid-wide popcount for the genome (bxmi_bits_group_popcount_dev) where the reference loops count_range over the chromosomes.
import fileinput
import sys

from bxmi.builders import binned_bitsets_from_file, binned_bitsets_from_paths, group_coverage


def main(argv=None, out=None, stdin=None):
    out = out or sys.stdout
    bed_filenames = sys.argv[1:] if argv is None else argv
    if bed_filenames and "-" not in bed_filenames:
        bitsets = binned_bitsets_from_paths(bed_filenames)  # what fileinput would chain, each file ingested in bulk
    else:
        bitsets = binned_bitsets_from_file(fileinput.input(bed_filenames) if bed_filenames else (stdin or sys.stdin))
    total = group_coverage(bitsets.values())  # (fresh sets, never inverted: count_range(0, size) is the popcount)
    out.write("%d\n" % total)
    out.flush()


if __name__ == "__main__":
    main()
