#!/usr/bin/env python
"""
Echo the lines of the first BED file that overlap something in the second one, every column kept as it was.

The padding options -u / -d are parsed and ignored, as in the script this mirrors.

usage: %prog bed_file_1 bed_file_2
    -m, --mincols=N: Require this much overlap (default 1bp)
    -u, --upstream_pad=N: upstream interval padding (default 0bp)
    -d, --downstream_pad=N: downstream interval padding (default 0bp)
    -v, --reverse: Print regions that DO NOT overlap
    -b, --booleans: Just print '1' if interval overlaps or '0' otherwise
"""
# Counterpart of the reference's scripts/bed_intersect.py:26-68 -- same flags, same stdout
# (including the "line\n" + " " quirk of print(line, end=" ")), but every query line of a
# chromosome is answered by one batched count_ranges launch instead of one call per line.
import optparse
import sys
from warnings import warn

import numpy as np

from bxmi.builders import _check_range, binned_bitsets_from_file


def parse_args(argv):
    p = optparse.OptionParser("%prog bed_file_1 bed_file_2", conflict_handler="resolve")
    p.add_option("-m", "--mincols", action="store", help="Require this much overlap (default 1bp)")
    p.add_option("-u", "--upstream_pad", action="store", help="upstream interval padding (default 0bp)")
    p.add_option("-d", "--downstream_pad", action="store", help="downstream interval padding (default 0bp)")
    p.add_option("-v", "--reverse", action="store_true", help="Print regions that DO NOT overlap")
    p.add_option("-b", "--booleans", action="store_true", help="Just print '1' if interval overlaps or '0' otherwise")
    return p.parse_args(argv)


def main(argv=None, out=None):
    out = out or sys.stdout
    options, args = parse_args(sys.argv[1:] if argv is None else argv)
    mincols = 1
    try:
        if options.mincols:
            mincols = int(options.mincols)
        if options.upstream_pad:
            int(options.upstream_pad)  # parsed, unused: "not functional" in the reference too
        if options.downstream_pad:
            int(options.downstream_pad)
        reverse = bool(options.reverse)
        booleans = bool(options.booleans)
        in_fname, in2_fname = args
    except Exception:
        raise SystemExit(__doc__.replace("%prog", sys.argv[0]))

    bitsets = binned_bitsets_from_file(open(in2_fname))
    query = open(in_fname)
    rest = _bulk_prefix(query, bitsets, mincols, reverse, booleans, out)
    _per_line(rest, bitsets, mincols, reverse, booleans, out)


def _bulk_prefix(query, bitsets, mincols, reverse, booleans, out):
    """Fast path: the query file is parsed in C++ (bxmi.bedio), every chromosome's rows are answered by one
    count_ranges launch and the selected lines are written straight from the file buffer.  Stops at the first
    row the reference would reject (or the parser is unsure about) and returns the remaining lines."""
    from bxmi import bedio

    data = bedio.file_bytes(query) if bedio.enabled() else None
    if data is None:
        return query
    bed = bedio.ParsedBed(data)
    try:
        known = np.array([name in bitsets for name in bed.names] or [False])
        size_of = np.array([bitsets[name].size if name in bitsets else 0 for name in bed.names] or [0], dtype=np.int64)
        has = known[bed.chrom] if bed.n else np.empty(0, bool)
        sz = size_of[bed.chrom] if bed.n else np.empty(0, np.int64)
        s, e = bed.start, bed.end
        bad = has & ((s < 0) | (s >= sz) | (e < s) | (e > sz))
        k = int(np.argmax(bad)) if bad.any() else bed.n
        if (s[:k] > e[:k]).any():
            warn("Bed interval start after end!")
        hit = np.zeros(bed.n, dtype=bool)
        ids = bed.chrom[:k]
        order = np.argsort(ids, kind="stable")
        bounds = np.searchsorted(ids[order], np.arange(len(bed.names) + 1))
        for c, name in enumerate(bed.names):
            rows = order[bounds[c]:bounds[c + 1]]
            if known[c] and len(rows):
                counts = bitsets[name].count_ranges(s[rows].astype(np.int32), (e[rows] - s[rows]).astype(np.int32))
                hit[rows] = counts >= mincols
        sel = hit != reverse
        sel[k:] = False
        out.flush()
        if booleans:
            text = np.where(sel[:k], ord("1"), ord("0")).astype(np.uint8)
            both = np.empty((k, 2), dtype=np.uint8)
            both[:, 0], both[:, 1] = text, ord("\n")
            out.write(both.tobytes().decode("ascii"))
        else:
            try:
                fd = out.fileno()
            except (AttributeError, OSError, ValueError):
                fd = None
            if fd is not None:
                bed.emit(sel, b" ", fd)
            else:
                for i in np.nonzero(sel)[0].tolist():
                    o = int(bed.line_off[i])
                    out.write(data[o:o + int(bed.line_len[i])].decode("utf-8") + " ")
        out.flush()
        return bed.rest_lines(k if k < bed.n else None, query)
    finally:
        bed.close()


def _per_line(lines_in, bitsets, mincols, reverse, booleans, out):
    # Pass 1 (host): parse the query lines in order; stop at the first line the reference would die on.
    lines, per_chrom = [], {}
    error = None
    for line in lines_in:
        if line.startswith("#") or line.isspace():
            continue
        try:
            fields = line.split()
            start, end = int(fields[1]), int(fields[2])
            if start > end:
                warn("Bed interval start after end!")
            chrom = fields[0]
            if chrom in bitsets:
                error = _check_range(bitsets[chrom].size, start, end - start)
                if error is not None:
                    break
                s, c, idx = per_chrom.setdefault(chrom, ([], [], []))
                s.append(start)
                c.append(end - start)
                idx.append(len(lines))
            lines.append(line)
        except (ValueError, IndexError) as ex:
            error = ex
            break
    # Pass 2 (device): one count_ranges launch per chromosome.
    hit = np.zeros(len(lines), dtype=bool)
    for chrom, (s, c, idx) in per_chrom.items():
        counts = bitsets[chrom].count_ranges(np.array(s, dtype=np.int32), np.array(c, dtype=np.int32))
        hit[np.array(idx, dtype=np.int64)] = counts >= mincols
    # Pass 3 (host): emit in file order, exactly as the reference prints.
    w = out.write
    for line, h in zip(lines, hit.tolist()):
        if booleans:
            w("1\n" if h != reverse else "0\n")
        elif h != reverse:
            w(line)
            w(" ")
    out.flush()
    if error is not None:
        raise error


if __name__ == "__main__":
    main()
