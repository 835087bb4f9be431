#!/usr/bin/env python3
"""Per kernel of a translation unit compiled to assembly: instructions, FLAT accesses, `s_waitcnt vmcnt(0)`, global / LDS
instructions, registers -- what HISTORY.md section 11 reads.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/iv.s bx-python_amd/csrc/intervals.hip
       python tools/scan_isa.py /tmp/iv.s [name filter]"""
import re
import shutil
import subprocess
import sys

lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur, stats = None, {}
for l in lines:
    m = re.match(r"^(_ZN4bxmi\w+):", l)
    if m:
        cur = m.group(1)
        stats[cur] = dict(n=0, flat=0, vm0=0, vmem=0, lds=0, vgpr=0)
        continue
    if l.strip().startswith(".amdhsa_kernel"):
        cur = None
    if cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
        s = stats[cur]
        s["n"] += 1
        s["flat"] += "flat_" in l
        s["vm0"] += bool(re.search(r"s_waitcnt vmcnt\(0\)", l))
        s["vmem"] += "global_" in l or "buffer_" in l
        s["lds"] += l.strip().startswith("ds_")
    m = re.match(r"\s*\.set (_ZN4bxmi\w+)\.num_vgpr, (\d+)", l)
    if m and m.group(1) in stats:
        stats[m.group(1)]["vgpr"] = int(m.group(2))
names = subprocess.run([shutil.which("c++filt")] + list(stats), capture_output=True, text=True).stdout.splitlines()
for k, nm in zip(stats, names):
    s = stats[k]
    if pat in nm:
        print("%-92s n=%5d flat=%3d vmcnt0=%3d vmem=%3d lds=%3d vgpr=%3d" % (re.sub(r"\(.*", "", nm)[-92:], s["n"], s["flat"], s["vm0"], s["vmem"], s["lds"], s["vgpr"]))
