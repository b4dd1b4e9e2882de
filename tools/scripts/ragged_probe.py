"""Unique shapes (no two alignments of a batch can be paired): packed kernels with every task paired with itself, the solo
kernel (KSW2AMD_SOLO=1: one alignment on both register halves) and the int32 kernels (KSW2AMD_NO_PK=1).  GPU box:  python tools/scripts/ragged_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402

lib = ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
rng = np.random.Generator(np.random.PCG64(1))
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # batch size multiplier: more wavefronts per SIMD
for lo, hi, n, w, zd, flag, name in ((400, 600, 32768, 64, -1, 1, "short score-only"), (400, 600, 32768, 64, 400, 0, "short cigar"),
                                     (8000, 12000, 2048, 500, 400, 1, "long score-only"), (8000, 12000, 2048, 500, 400, 0, "long cigar")):
    n *= scale
    pairs = synth.ragged_pairs(rng, n, lo, hi, sub=0.05, ind=0.06)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    for nopk, solo in ((0, "0"), (0, "1"), (1, "0")):
        os.environ["KSW2AMD_SOLO"] = solo
        if nopk:
            os.environ["KSW2AMD_NO_PK"] = "1"
        else:
            os.environ.pop("KSW2AMD_NO_PK", None)
        p = lib.make_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag).plan(False)
        p.run(); p.timing()
        ms = []
        for _ in range(3):
            p.run(); ms.append(p.timing()[1])
        print("%-18s %-7s packed pairs %6d  %8.2f ms  %7.1f GCUPS" % (name, "int32" if nopk else "solo" if solo == "1" else "packed", p.packed_pairs(), np.mean(ms), p.cells() / np.mean(ms) / 1e6))
        p.close()
