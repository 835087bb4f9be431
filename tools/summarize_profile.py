#!/usr/bin/env python3
"""Condense rocprofv3 output (kernel stats + FETCH_SIZE / WRITE_SIZE passes) into small files:
   <dir>/summary_kernel_stats.csv, <dir>/summary_pmc.json.   usage: summarize_profile.py gpurun_out"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def find(sub, pat):
    hits = glob.glob(os.path.join(d, sub, "**", pat), recursive=True)
    return hits[0] if hits else None


def short(name):
    n = name.split("(")[0]
    for pre in ("void ", "bxmi::"):
        n = n.replace(pre, "")
    return n[:90]


out = {}
stats = find("prof_stats", "*kernel_stats.csv")
if stats:
    rows = list(csv.DictReader(open(stats)))
    with open(os.path.join(d, "summary_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"])
        for r in rows[:40]:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]])
            print("%-70s calls=%-6s avg=%10.1f us  %5s%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
trace = find("prof_stats", "*kernel_trace.csv")
dur = defaultdict(list)
if trace:
    for r in csv.DictReader(open(trace)):
        dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for tag, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    cc = find(tag, "*counter_collection.csv")
    if not cc:
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") == counter:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[counter] = dict(dispatches=len(v), mean=sum(v) / len(v), min=min(v), max=max(v))
for k, v in dur.items():
    if k in out:
        out[k]["avg_duration_us"] = sum(v) / len(v) / 1e3
        out[k]["dispatches_traced"] = len(v)
# HBM bytes of one count pass = its seven kernels.
# FETCH_SIZE is in KiB and reads HALF of a streaming read on gfx950 (checked below on the bitset popcount, whose
# byte count is known); WRITE_SIZE is in KiB and exact.  Correction as MI355X_MICROARCH.md prescribes.
# the bitmap-cell pass (count_bitmap.hpp); whatever template arguments the run used
pass_prefixes = ["bm_params_kernel", "bm_probe_kernel", "bm_sorted_check_kernel", "ivl_local_count_kernel", "bm_tile_sort_kernel", "bm_transpose_kernel",
                 "bm_plan_kernel", "bm_unpermute_kernel", "bm_fold_totals_kernel", "bd_transpose_kernel", "bd_plan_kernel", "bd_search_kernel",
                 "bw_search_kernel", "bd_unpermute_kernel"]
pass_kernels = [k for k in out if any(k.startswith(pre) for pre in pass_prefixes)]
tot = 0.0
detail = {}
for k in pass_kernels:
    v = out.get(k)
    if v and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        b = (2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024.0
        detail[k] = dict(hbm_bytes=b, fetch_kib_raw=v["FETCH_SIZE"]["mean"], write_kib=v["WRITE_SIZE"]["mean"], avg_us=v.get("avg_duration_us"))
        tot += b
if detail:
    out["count_pass"] = dict(hbm_bytes_per_launch=round(tot), kernels=detail,
                             note="sum over the pass's kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024")
v = out.get("ivl_count_kernel<true>")
if v and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    out["ivl_count_kernel"] = dict(hbm_bytes_per_launch=round((2.0 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024.0))
# what code this was measured on (bench.py quotes count_pass.hbm_bytes_per_launch only when these match its own)
try:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import datetime

    import bench

    out["stamps"] = dict(bench.source_stamps(), date=datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%MZ"),
                         command=os.environ.get("PROFILE_CMD", "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sorted --no-find --no-bitset --no-genome"),
                         head=os.environ.get("PROFILE_HEAD", "(stamped when copied into profiles/)"))
except Exception as ex:
    out["stamps"] = {"error": repr(ex)}
json.dump(out, open(os.path.join(d, "summary_pmc.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k in ("count_pass", "bits_popcount_kernel", "stamps") or k.startswith("bits_group")}, indent=1))
