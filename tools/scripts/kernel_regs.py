#!/usr/bin/env python3
"""Register / occupancy table of every kernel in ksw2_shim_hip.hip (hipcc -Rpass-analysis=kernel-resource-usage).

usage: python tools/scripts/kernel_regs.py [filter-substring]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "ksw2_amd", "csrc", "ksw2_shim_hip.hip")


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", SRC, "-o", "/dev/null",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "")}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>6s} {'occ':>4s}")
    for r in rows:
        if flt in r["name"]:
            print(f"{r['name']:70s} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('scratch', 0):8d} {r.get('lds', 0):6d} {r.get('occ', 0):4d}")


if __name__ == "__main__":
    main()
