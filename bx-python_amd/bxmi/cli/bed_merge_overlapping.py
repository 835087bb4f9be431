#!/usr/bin/env python
"""
Union of BED input (command line or stdin): overlapping or touching regions come out as one, on the '+' strand,
with only chrom / start / end kept.

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
