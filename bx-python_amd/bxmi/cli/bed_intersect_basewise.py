#!/usr/bin/env python
"""
Base-level AND of two BED files: prints each maximal stretch of bases that both inputs cover.

usage: %prog bed_file_1 bed_file_2
"""
# Counterpart of the reference's scripts/bed_intersect_basewise.py:14-38: the per-chromosome iand loop is ONE group launch
# (bxmi_bits_group_and_dev), the next_set/next_clear walk one run-extraction pass on the device per chromosome.
import sys

from bxmi.builders import binned_bitsets_from_file, group_iand, write_runs


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    try:
        in_fname, in2_fname = args
    except ValueError:
        raise SystemExit(__doc__.replace("%prog", sys.argv[0]))
    bits1 = binned_bitsets_from_file(open(in_fname))
    bits2 = binned_bitsets_from_file(open(in2_fname))
    bitsets = {key: bits1[key] for key in bits1 if key in bits2}  # first-appearance order of file 1
    group_iand(bitsets.values(), [bits2[key] for key in bitsets])
    for chrom, bits in bitsets.items():
        write_runs(out, chrom, bits)
    out.flush()


if __name__ == "__main__":
    main()
