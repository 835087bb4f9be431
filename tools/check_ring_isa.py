#!/usr/bin/env python3
"""Read the compiled code of the flat walk's ring kernels (bd_search_kernel<.., PAD = true>, count_dense.hpp) and check what
the compiler cannot know: the record loads of the ring are issued by hand (inline asm), so nothing tells the register
allocator that a destination register is still being written.  For every such kernel, in program order:

  * a hand-issued `global_load_dwordx4 v[a:b]` (or `global_load_ushort vN`: the run table of the persistent walk) puts its
    destination "in flight";
  * a hand-issued `s_waitcnt vmcnt(K) ; ring v[a:b]` (`; runs vA vB`) lands them -- and the registers it names must BE the destination of
    a load in flight (if the compiler copied the value somewhere else in between, the copy read a register before its
    data arrived);
  * any other instruction that names a register in flight is an error.

usage: python tools/check_ring_isa.py [file.s]   (without a file: compiles intervals.hip to assembly first, ~40 s)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bx-python_amd", "csrc")


def vregs(text):
    """every VGPR number an operand string names: v7, v[14:17]"""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


def parse_blocks(lines):
    """basic blocks of one kernel: [(label or None, [(line number, kind, text, raw)], successors)]; kind = 'asm' for
    instructions of an inline-asm statement"""
    blocks, cur, in_asm = [], {"label": None, "ins": [], "succ": [], "fall": True}, False

    def close(fall=True):
        nonlocal cur
        cur["fall"] = fall
        blocks.append(cur)
        cur = {"label": None, "ins": [], "succ": [], "fall": True}

    for no, line in lines:
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        text = line.strip() if in_asm else line.split(";")[0].strip()
        if not text or text.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", text)
            if not m:
                continue
            if cur["ins"] or cur["label"]:
                close()
            cur["label"] = m.group(1)
            continue
        cur["ins"].append((no, "asm" if in_asm else "", text, line))
        op = text.split()[0]
        if op == "s_branch":
            cur["succ"].append(text.split()[1])
            close(fall=False)
        elif op.startswith("s_cbranch"):
            cur["succ"].append(text.split()[1])
            close()
        elif op == "s_endpgm":
            close(fall=False)
    if cur["ins"] or cur["label"]:
        close(fall=False)
    return blocks


def transfer(block, state, errors=None):
    """state: {register: line of the load that is writing it}"""
    st = dict(state)
    loads = waits = 0
    for no, kind, code, raw in block["ins"]:
        if kind == "asm" and (code.startswith("global_load_dwordx4") or code.startswith("global_load_ushort")):
            dest = vregs(code.split(",")[0])
            addr = vregs(",".join(code.split(",")[1:]).split(";")[0])
            bad = ((dest | addr) - (dest & addr)) & set(st) | (addr & set(st))
            if bad and errors is not None:
                errors.append((no, "load names registers in flight %s: %s" % (sorted(bad), code)))
            for r in dest:
                st[r] = no
            loads += 1
        elif kind == "asm" and code.startswith("s_waitcnt") and ("ring" in code or "runs" in code):
            named = vregs(code.split("ring")[1] if "ring" in code else code.split("runs")[1])
            waits += 1
            if "vmcnt(0)" in code:  # (the drain behind the loop: everything lands)
                st.clear()
                continue
            missing = named - set(st)
            if missing and errors is not None:
                errors.append((no, "wait names %s, which no load in flight writes: %s" % (sorted(missing), code)))
            for r in named:
                st.pop(r, None)
        elif code.startswith("s_waitcnt") and "vmcnt(0)" in code:
            st.clear()  # the compiler's own drain (the rare global look-ups of a hard cell)
        else:
            bad = vregs(code.split(";")[0]) & set(st)
            if bad and errors is not None:
                errors.append((no, "touches registers in flight %s (loaded at line %d): %s" % (sorted(bad), st[min(bad)], code)))
    return st, loads, waits


def check_kernel(name, lines):
    blocks = parse_blocks(lines)
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    succ = []
    for i, b in enumerate(blocks):
        out = [index[t] for t in b["succ"] if t in index]
        if b["fall"] and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    state_in = [None] * len(blocks)  # may-analysis: a register is in flight if it is on any path
    state_in[0] = {}
    work = [0]
    while work:
        i = work.pop()
        out, _, _ = transfer(blocks[i], state_in[i])
        for j in succ[i]:
            merged = dict(state_in[j] or {})
            before = len(merged) if state_in[j] is not None else -1
            for r, no in out.items():
                merged.setdefault(r, no)
            if len(merged) != before:
                state_in[j] = merged
                work.append(j)
    errors, loads, waits = [], 0, 0
    for i, b in enumerate(blocks):
        if state_in[i] is None:
            continue
        _, l, w = transfer(b, state_in[i], errors)
        loads, waits = loads + l, waits + w
    return loads, waits, sorted(set(errors))


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.gettempdir(), "bxmi_intervals.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                               "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, "intervals.hip"), "-o", path], stderr=subprocess.DEVNULL)
    kernels, cur = {}, None
    for no, line in enumerate(open(path), 1):
        m = re.match(r"^(_ZN4bxmi16b[dw]_search_kernel\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
        elif cur and line.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            kernels[cur].append((no, line.rstrip("\n")))
    failed = 0
    for name, lines in kernels.items():
        if "bw_search_kernel" in name:  # the persistent walk: bw_search_kernel<W8, DEPTH>
            t = re.search(r"ILb(\d)ELi(\d)ELb(\d)E", name)
            tag = "persistent walk W8 %s DEPTH %s%s" % (t.group(1), t.group(2), " offset cells" if t.group(3) == "1" else "")
        else:
            t = re.search(r"ILi(\d)ELb(\d)ELi(\d)ELi(\d)ELb(\d)ELb(\d)E", name)
            fmt, qb, exp, depth, pipe, pad = (int(x) for x in t.groups())
            if not pipe:
                continue
            if exp == 3:
                continue  # (diagnostics: synthetic records, no loads to wait for)
            tag = "FMT %d QB %d EXP %d DEPTH %d %s" % (fmt, qb, exp, depth, "ring" if pad else "two sets")
        loads, waits, errors = check_kernel(name, lines)
        if loads == 0 or waits == 0:
            errors.append((0, "no hand-issued loads / waits found (%d / %d)" % (loads, waits)))
        print("%-40s loads %2d waits %2d  %s" % (tag, loads, waits, "ok" if not errors else "%d PROBLEMS" % len(errors)))
        for no, msg in errors[:6]:
            print("    line %d: %s" % (no, msg))
        failed += bool(errors)
    print("%d kernels checked, %d with problems" % (sum(1 for k in kernels if "ELb1ELb" in k or "bw_search" in k), failed))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
