#!/usr/bin/env python3
"""Distance (in instructions of the wavefront's stream) from every VALU instruction to the closest earlier instruction that wrote one of
its source registers, per basic block of one kernel in a `hipcc -S` listing: a lone wavefront issues a VALU instruction that depends on
its predecessor every ~8.7 cycles and one that does not every ~5.3 (profiles/r5_issue_probe.txt), so blocks with many distance-1
instructions are where a single wavefront per SIMD loses time.

usage: dep_distance.py listing.s kernel-substring [min-block-length]"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or "s_endpgm" in lines[i])
def regs(tok):
    out = []
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", tok):
        if m.group(1): out.append(int(m.group(1)))
        else: out += list(range(int(m.group(2)), int(m.group(3)) + 1))
    return out
blocks, cur = [], []
for l in lines[start + 1:end]:
    t = l.split(";")[0].strip()
    if not t: continue
    if re.match(r"^\.?\w+:$", t):
        blocks.append(cur); cur = []; continue
    if t.startswith("."): continue
    cur.append(t)
blocks.append(cur)
tot = {}
for b in blocks:
    last = {}
    hist = {}
    nv = 0
    for k, t in enumerate(b):
        op, _, rest = t.partition(" ")
        ops = [x.strip() for x in rest.split(",")]
        if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readfirstlane") and not op.startswith("v_readlane"):
            d = min([k - last[r] for o in ops[1:] for r in regs(o) if r in last] + [99])
            if "+v" in t: pass
            nv += 1
            key2 = d if d <= 3 else 4
            hist[key2] = hist.get(key2, 0) + 1
            for r in regs(ops[0]): last[r] = k
        elif op.startswith("v_cmp"):
            d = min([k - last[r] for o in ops for r in regs(o) if r in last] + [99])
            nv += 1
            key2 = d if d <= 3 else 4
            hist[key2] = hist.get(key2, 0) + 1
        elif op.startswith(("ds_", "global_", "buffer_", "flat_", "scratch_")) and ops and "load" in op or op.startswith("ds_read") or op.startswith("ds_bpermute"):
            for r in regs(ops[0]): last[r] = k - 50         # results wait on counters, not on issue distance
    if nv >= minlen:
        print("block of %4d instructions, %4d VALU: distance 1: %4d  2: %4d  3: %4d  more: %4d" % (len(b), nv, hist.get(1, 0), hist.get(2, 0), hist.get(3, 0), hist.get(4, 0)))
    for k2, v in hist.items(): tot[k2] = tot.get(k2, 0) + v
print("kernel:", {("d%d" % k if k < 4 else "more"): v for k, v in sorted(tot.items())})
