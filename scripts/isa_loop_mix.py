#!/usr/bin/env python3
"""Instruction mix of the hot loops of a gfx950 assembly listing (hipcc -S --cuda-device-only): per kernel, every loop
(a backward branch to a label) with its VALU / SALU / LDS / VMEM instruction counts — the numbers DESIGN.md quotes as
"VALU per event" are the innermost tile loop's VALU count divided by the events a lane walks per tile."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    return "other"


def main(path, only=None):
    kernel, lines = None, []
    kernels = {}
    for raw in open(path):
        line = raw.split(";")[0].rstrip()
        m = re.match(r"^(\w+):\s*$", line)
        if m and not line.startswith(".L"):
            kernel = m.group(1)
            kernels[kernel] = []
            continue
        if kernel is not None:
            kernels[kernel].append(line)
    for name, body in kernels.items():
        if only and only not in name:
            continue
        labels, instrs = {}, []
        for line in body:
            s = line.strip()
            if not s or s.startswith("."):
                m = re.match(r"^(\.LBB\w+):", s)
                if m:
                    labels[m.group(1)] = len(instrs)
                continue
            m = re.match(r"^(\.LBB\w+):", s)
            if m:
                labels[m.group(1)] = len(instrs)
                continue
            instrs.append(s)
        if not instrs:
            continue
        total = Counter(classify(i.split()[0]) for i in instrs)
        print(f"{name}: total {dict(total)}")
        loops = []
        for idx, ins in enumerate(instrs):
            parts = ins.split()
            if parts[0].startswith("s_cbranch") or parts[0] == "s_branch":
                tgt = parts[-1]
                if tgt in labels and labels[tgt] <= idx:
                    loops.append((labels[tgt], idx))
        for a, b in sorted(loops, key=lambda t: t[1] - t[0]):
            c = Counter(classify(i.split()[0]) for i in instrs[a:b + 1])
            if b - a > 100:
                print(f"   loop [{a}, {b}] len {b - a + 1}: {dict(c)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
