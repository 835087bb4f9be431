#!/usr/bin/env python
"""
Merge any overlapping regions of bed files. Bed files can be provided on the
command line or on stdin. Merged regions are always reported on the '+'
strand, and any fields beyond chrom/start/stop are lost.

usage: %prog bed files ...
"""
# Counterpart of the reference's scripts/bed_merge_overlapping.py:17-35.
import fileinput
import sys

from bxmi.builders import binned_bitsets_from_bed_file, write_runs


def main(argv=None, out=None, stdin=None):
    out = out or sys.stdout
    bed_filenames = sys.argv[1:] if argv is None else argv
    inp = fileinput.input(bed_filenames) if bed_filenames else (stdin or sys.stdin)
    bitsets = binned_bitsets_from_bed_file(inp)
    for chrom, bits in bitsets.items():
        write_runs(out, chrom, bits)
    out.flush()


if __name__ == "__main__":
    main()
