#!/usr/bin/env python3
"""Audit of the k2a_load_async registers (ksw2_lane_pk.h) in the gfx950 assembly of ksw2_shim_hip.hip.

k2a_load_async is an inline-asm global_load_dword the compiler does not know as a load: nothing waits for it until the
inline-asm `s_waitcnt vmcnt(0)` of k2a_load_wait.  That is only correct if, on EVERY path from the load to a wait, no
instruction reads or overwrites the destination register (a compiler-inserted copy, spill or re-use of a register whose
load is still in flight moves stale data or loses the loaded value), and if no path reaches s_endpgm with a load in flight.
This script checks exactly that on the control-flow graph of every kernel: forward data flow of the set of in-flight
registers over basic blocks (union at joins), cleared by any `s_waitcnt vmcnt(0)`.

usage: python tools/scripts/async_load_audit.py [file.s]      (without a file: compiles ksw2_shim_hip.hip to /tmp first)
exit status 1 if anything is reported.  tests/test_build_and_abi.py runs it in the CPU tier.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "ksw2_amd", "csrc", "ksw2_shim_hip.hip")


def regs_of(tok):
    """VGPR numbers an operand token names: v12, v[4:7]"""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def audit_kernel(name, body):
    # ---- split into basic blocks
    blocks, cur, label_of = [], {"label": "entry", "ins": []}, {}
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            if cur["ins"] or cur["label"] != "entry":
                blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            cur["ins"].append(("asm+", None))
            continue
        if s.startswith(";;#ASMEND"):
            cur["ins"].append(("asm-", None))
            continue
        if not s or s[0] in ";.":
            continue
        cur["ins"].append(("i", s.split(";")[0].strip()))
        op = s.split()[0]
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"]:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        last = b["ins"][-1][1] if b["ins"] and b["ins"][-1][0] == "i" else ""
        op = last.split()[0] if last else ""
        if op.startswith("s_endpgm"):
            succ.append([])
        elif op.startswith("s_branch"):
            succ.append([label_of[last.split()[1]]])
        elif op.startswith("s_cbranch"):
            succ.append([label_of[last.split()[1]]] + ([i + 1] if i + 1 < len(blocks) else []))
        else:
            succ.append([i + 1] if i + 1 < len(blocks) else [])

    # ---- transfer function of a block: returns (out set, findings)
    def run(b, inset, report):
        pend, inasm = set(inset), False
        for kind, s in b["ins"]:
            if kind == "asm+":
                inasm = True
                continue
            if kind == "asm-":
                inasm = False
                continue
            ops = s.split(None, 1)
            op, rest = ops[0], (ops[1] if len(ops) > 1 else "")
            if op == "s_waitcnt" and "vmcnt(0)" in rest:
                pend = set()
                continue
            if op == "s_endpgm" and pend:
                report.append("%s: loads into %s still in flight at s_endpgm" % (name, sorted("v%d" % r for r in pend)))
                continue
            if inasm and op == "global_load_dword":
                dst = regs_of(rest.split(",")[0])
                if dst & pend:
                    report.append("%s: second async load into v%d before a wait" % (name, min(dst & pend)))
                pend |= dst
                continue
            if pend:
                touched = regs_of(rest) & pend
                if touched:
                    report.append("%s: `%s` touches %s while its load is in flight" % (name, s[:70], sorted("v%d" % r for r in touched)))
                    pend -= touched          # report once
        return pend

    n = len(blocks)
    ins = [set() for _ in range(n)]
    work = [0]
    outs = [None] * n
    while work:
        i = work.pop()
        o = run(blocks[i], ins[i], [])
        if outs[i] is not None and o == outs[i]:
            continue
        outs[i] = o
        for j in succ[i]:
            if not o <= ins[j]:
                ins[j] |= o
                work.append(j)
            elif outs[j] is None:
                work.append(j)
    report = []
    for i in range(n):
        run(blocks[i], ins[i], report)
    nload = sum(1 for b in blocks for k, (kind, s) in enumerate(b["ins"]) if kind == "i" and s.startswith("global_load_dword") and k and b["ins"][k - 1][0] == "asm+")
    return nload, sorted(set(report))


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = "/tmp/ksw2_shim_audit.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", path, SRC],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(path).read().split("\n")
    total, findings, kernels, i = 0, [], 0, 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        nload, rep = audit_kernel(m.group(1), lines[i + 1:j])
        if nload:
            kernels += 1
        total += nload
        findings += rep
        i = j
    for f in findings:
        print(f)
    print("kernels with async loads: %d, async loads: %d, findings: %d" % (kernels, total, len(findings)))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
