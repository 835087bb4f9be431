#!/usr/bin/env python3
"""Cut a rocprofv3 kernel trace of tools/count_variants.py (or any run that pauses between phases) into segments at
idle gaps and print, per segment, every kernel's calls and average duration.  A segment with few dispatches (warm-up,
index build) is printed in one line.   usage: trace_segments.py <dir with *kernel_trace.csv> [gap_ms=20] [min_calls=3]"""
import csv
import glob
import os
import sys
from collections import OrderedDict

d = sys.argv[1]
gap_ns = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 20e6
min_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hits = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not hits:
    sys.exit("no kernel trace under " + d)


def short(name):
    n = name.split("(")[0]
    for pre in ("void ", "bxmi::"):
        n = n.replace(pre, "")
    return n[:70]


rows = []
for r in csv.DictReader(open(hits[0])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
segs, cur, last_end = [], [], None
for s, e, n in rows:
    if last_end is not None and s - last_end > gap_ns and cur:
        segs.append(cur)
        cur = []
    cur.append((s, e, n))
    last_end = max(e, last_end or 0)
if cur:
    segs.append(cur)
for i, seg in enumerate(segs):
    acc = OrderedDict()
    for s, e, n in seg:
        a = acc.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    span = (seg[-1][1] - seg[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in seg) / 1e3
    top = max(a[0] for a in acc.values())
    if top < min_calls:
        print("segment %d: %d dispatches, span %.1f us, busy %.1f us (%s)" % (i, len(seg), span, busy, ", ".join(list(acc)[:6])))
        continue
    print("segment %d: %d dispatches, span %.1f us, busy %.1f us, per pass: span %.1f busy %.1f" % (i, len(seg), span, busy, span / top, busy / top))
    for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("    %-72s calls=%-4d avg=%9.1f us" % (n, c, t / c / 1e3))
