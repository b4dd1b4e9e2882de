#!/usr/bin/env python3
"""Basic blocks of one kernel in the gfx950 assembly of ksw2_shim_hip.hip, with VALU / SALU / memory instruction counts:
the static side of "how many instructions does one step issue" (the dynamic side is SQ_INSTS_VALU, pmc_summary.py).

usage: python tools/scripts/asm_blocks.py <mangled-name-prefix> [file.s]
       (file.s from: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o file.s ksw2_amd/csrc/ksw2_shim_hip.hip)
"""
import re
import sys


def main():
    prefix = sys.argv[1]
    path = sys.argv[2] if len(sys.argv) > 2 else "/tmp/shim.s"
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith(prefix) and l.rstrip().endswith(":") or (l.startswith(prefix) and ": " in l)][0]
    end = start
    while not lines[end].startswith(".Lfunc_end"):
        end += 1
    name, cnt, out = "entry", [0, 0, 0], []
    notes = []

    def flush():
        if sum(cnt) or notes:
            out.append("%-14s valu %4d  salu %4d  mem %3d  %s" % (name, cnt[0], cnt[1], cnt[2], " ".join(notes)))

    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %(bb\.\d+):", l)
        if m:
            flush()
            name, cnt, notes = m.group(1), [0, 0, 0], []
            if "Loop Header" in l:
                notes.append("<loop header>")
            continue
        s = l.strip()
        if not s or s[0] in ";.":
            continue
        op = s.split()[0]
        if op.startswith("v_"):
            cnt[0] += 1
        elif op.startswith(("global_", "ds_", "scratch_", "buffer_", "flat_", "s_load", "s_buffer")):
            cnt[2] += 1
        else:
            cnt[1] += 1
        if op.startswith(("s_cbranch", "s_branch")):
            notes.append(s.replace("\t", " "))
    flush()
    print("\n".join(out))


if __name__ == "__main__":
    main()
