#!/usr/bin/env python3
"""Instruction mix of the hottest loops of every kernel in a hipcc -S listing (CPU-side proxy for the
issue-bound kernels here): for each kernel, every loop (by header label, from the "in Loop: Header=" block
annotations) with its VALU / SALU / LDS / VMEM / branch / waitcnt instruction counts.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -S --cuda-device-only -o k.s file.hip; isa_loop_mix.py k.s [min_instrs]"""
import collections
import re
import sys


def kind(op):
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    min_ins = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    kern = None
    loops = collections.OrderedDict()
    cur = None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
            cur = None
            continue
        if kern is None:
            continue
        if "s_endpgm" in l:
            kern = None
            continue
        if l.startswith(".L") or l.startswith("; %bb"):
            m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
            if m:
                cur = (kern, m.group(1), m.group(2))
            else:
                m = re.match(r"^\.L(BB\d+_\d+):.*Loop Header: Depth=(\d+)", l)
                if m is None and l.startswith(".L"):
                    # header annotation can sit on the following comment line
                    cur = ("pending", l.split(":")[0][2:])
                    continue
                cur = (kern, m.group(1), m.group(2)) if m else None
            continue
        if cur and cur[0] == "pending":
            m = re.search(r"Loop Header: Depth=(\d+)", l)
            if m and not l.startswith("\t"):
                cur = (kern, cur[1], m.group(1))
                continue
            if l.startswith("\t"):
                cur = None
        if cur and cur[0] != "pending" and l.startswith("\t"):
            t = l.strip()
            if t.startswith((";", ".")):
                continue
            loops.setdefault(cur, collections.Counter())[kind(t.split()[0])] += 1
    for (k, h, d), c in loops.items():
        n = sum(c.values())
        if n >= min_ins:
            short = re.sub(r"^_ZN11lz4flex_dev", "", k)[:90]
            print("%-90s %-10s depth %s  total %4d  %s" % (short, h, d, n, dict(c)))


if __name__ == "__main__":
    main()
