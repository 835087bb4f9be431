#!/usr/bin/env python
"""
Side-by-side listing of every overlapping pair between two interval files (a full join on base overlap).

usage: %prog bed1 bed2
"""
# Counterpart of the reference's scripts/interval_join.py:16-30: per chromosome, one device index
# over file 2 and ONE batched find() (CSR hit list in HBM) for all rows of file 1.
import sys

import numpy as np

from bxmi.intervals import IntervalIndex
from bxmi.rows import read_rows


def main(argv=None, out=None):
    out = out or sys.stdout
    args = sys.argv[1:] if argv is None else argv
    # Read second set into per-chromosome indexes (insertion order = file order)
    targets = {}
    for row in read_rows(open(args[1])):
        targets.setdefault(row.chrom, []).append(row)
    # Read the first set; rows on chromosomes absent from file 2 print nothing
    queries, error = [], None
    try:
        for row in read_rows(open(args[0])):
            queries.append(row)
    except Exception as ex:  # the reference prints everything before the bad line, then dies
        error = ex
    by_chrom = {}
    for qi, row in enumerate(queries):
        if row.chrom in targets:
            by_chrom.setdefault(row.chrom, []).append(qi)
    hits_of = {}
    for chrom, qis in by_chrom.items():
        trows = targets[chrom]
        ix = IntervalIndex()
        ix.append(np.array([r.start for r in trows], dtype=np.int64), np.array([r.end for r in trows], dtype=np.int64))
        offs, hits = ix.find(np.array([queries[q].start for q in qis], dtype=np.int64),
                             np.array([queries[q].end for q in qis], dtype=np.int64))
        ix.close()
        offs, hits = offs.tolist(), hits.tolist()
        for k, q in enumerate(qis):
            hits_of[q] = (trows, hits[offs[k]:offs[k + 1]])
    w = out.write
    for qi, row in enumerate(queries):
        if qi in hits_of:
            trows, hs = hits_of[qi]
            left = str(row)
            for h in hs:
                w(left + "\t" + str(trows[h]) + "\n")
    out.flush()
    if error is not None:
        raise error


if __name__ == "__main__":
    main()
