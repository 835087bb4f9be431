#!/usr/bin/env python3
"""Per kernel of a translation unit compiled to assembly: instructions, FLAT accesses, `s_waitcnt vmcnt(0)`, global / LDS
instructions, registers, and every vmcnt the compiler waits for -- what HISTORY.md section 11 reads, and what
tests/test_isa_waits.py pins.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/iv.s bx-python_amd/csrc/intervals.hip
       python tools/scan_isa.py /tmp/iv.s [name filter]"""
import re
import shutil
import subprocess
import sys


def scan(path):
    """{demangled kernel name (without its argument list): dict(n, flat, vm0, vmem, lds, vgpr, waits = {N: times vmcnt(N) is waited for})}"""
    lines = open(path).read().splitlines()
    cur, stats = None, {}
    for l in lines:
        m = re.match(r"^(_ZN4bxmi\w+):", l)
        if m:
            cur = m.group(1)
            stats[cur] = dict(n=0, flat=0, vm0=0, vmem=0, lds=0, vgpr=0, waits={})
            continue
        if l.strip().startswith(".amdhsa_kernel"):
            cur = None
        if cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
            s = stats[cur]
            s["n"] += 1
            s["flat"] += "flat_" in l
            s["vmem"] += "global_" in l or "buffer_" in l
            s["lds"] += l.strip().startswith("ds_")
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if w:
                k = int(w.group(1))
                s["waits"][k] = s["waits"].get(k, 0) + 1
                s["vm0"] += k == 0
        m = re.match(r"\s*\.set (_ZN4bxmi\w+)\.num_vgpr, (\d+)", l)
        if m and m.group(1) in stats:
            stats[m.group(1)]["vgpr"] = int(m.group(2))
    names = subprocess.run([shutil.which("c++filt")] + list(stats), capture_output=True, text=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", nm).replace("void ", ""): stats[k] for k, nm in zip(stats, names)}


if __name__ == "__main__":
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    for nm, s in scan(sys.argv[1]).items():
        if pat in nm:
            print("%-92s n=%5d flat=%3d vmcnt0=%3d vmem=%3d lds=%3d vgpr=%3d" % (nm[-92:], s["n"], s["flat"], s["vm0"], s["vmem"], s["lds"], s["vgpr"]))
