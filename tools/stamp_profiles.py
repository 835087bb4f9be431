#!/usr/bin/env python3
"""Copy the summaries of the last tools/profile.sh + bench.py run (gpurun_out/) into profiles/ under a round prefix,
stamped with the commit they were taken on; refuses when bench.py or the count-pass sources changed since the run.
usage: python tools/stamp_profiles.py r02"""
import csv
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

prefix = sys.argv[1] if len(sys.argv) > 1 else "rXX"
out, prof = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
d = json.load(open(os.path.join(out, "summary_pmc.json")))
now = bench.source_stamps()
assert all(now[k] == d["stamps"].get(k) for k in now), ("profile taken on other code", now, d["stamps"])
d["stamps"]["head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
for name in (prefix + "_pmc.json", "pmc_latest.json"):
    json.dump(d, open(os.path.join(prof, name), "w"), indent=1)
shutil.copy(os.path.join(out, "summary_kernel_stats.csv"), os.path.join(prof, prefix + "_kernel_stats.csv"))
for src, dst in (("bench.json", "_bench_line.json"), ("prof_stats.json", "_bench_line_under_rocprof.json")):
    line = open(os.path.join(out, src)).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(prof, prefix + dst), "w").write(line + "\n")
rows = {r["kernel"]: float(r["avg_ns"]) / 1e6 for r in csv.DictReader(open(os.path.join(prof, prefix + "_kernel_stats.csv")))}
keys = [k for k in rows if k.startswith(("bm_params", "bm_sorted_check", "ivl_local_count", "bm_tile_sort", "bm_transpose", "bm_plan", "bm_unpermute",
                                         "bm_fold_totals", "bd_transpose", "bd_plan", "bd_search", "bw_search", "bd_unpermute"))]
readme = os.path.join(prof, "README.md")
text = open(readme).read()
text = re.sub(r"their averages \(.*?\) are the pass time", "their averages (%s = %.3f ms) are the pass time"
              % (" + ".join("%.4f" % rows[k] for k in keys), sum(rows[k] for k in keys)), text, count=1)
open(readme, "w").write(text)
b = json.loads(open(os.path.join(prof, prefix + "_bench_line.json")).read())
print("stamped", d["stamps"], "| bench", b["ms_per_step"], "ms, frac", b["roofline"]["frac"], "| traffic", d["count_pass"]["hbm_bytes_per_launch"])
