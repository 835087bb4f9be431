#!/usr/bin/env python
"""
Print what a BED file leaves uncovered: for every chromosome named in the length table (whitespace-separated
`name size` lines), the gaps between the file's regions.

usage: %prog bed_file chrom_length_file
"""
# Counterpart of the reference's scripts/bed_complement.py:13-51 (same clipping of the last run to the
# chromosome length, same output order = LEN file order).
import sys

from bxmi.builders import binned_bitsets_from_file, write_runs


def read_len(f):
    """Length table -> {chromosome: length}, in file order (what scripts/bed_complement.py:13-19 reads; a line with fewer
    than two fields raises IndexError there and here)."""
    table = {}
    for fields in map(str.split, f):
        table[fields[0]] = int(fields[1])
    return table


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    try:
        in_fname, len_fname = args
    except Exception:
        raise SystemExit(__doc__.replace("%prog", sys.argv[0]))
    bitsets = binned_bitsets_from_file(open(in_fname))
    lens = read_len(open(len_fname))
    for chrom, length in lens.items():
        if chrom in bitsets:
            bits = bitsets[chrom]
            bits.invert()
            write_runs(out, chrom, bits, clip=length)
        else:
            out.write("%s\t0\t%d\n" % (chrom, length))
    out.flush()


if __name__ == "__main__":
    main()
