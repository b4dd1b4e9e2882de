"""Gap-linear X-drop kernel forms: register window (auto) against the LDS-state kernel (KSW2AMD_EXTF_LDS=1) by band width.
GPU box:  python tools/scripts/extf_classes.py"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402

lib = ka.library()
for n, ln, w in ((16384, 1000, 30), (16384, 1000, 100), (16384, 1000, 146), (8192, 2000, 200), (4096, 3000, 300), (4096, 3000, 402), (4096, 3000, 600)):
    q, t = synth.fixed_batch(8, n, ln, ln, sub=0.05, ind=0.01)
    for lds in (0, 1):
        if lds:
            os.environ["KSW2AMD_EXTF_LDS"] = "1"
        else:
            os.environ.pop("KSW2AMD_EXTF_LDS", None)
        p = lib.make_linear_batch(list(q), list(t), 2, -4, 2, w=w, xdrop=-1).plan()
        p.run(); p.timing()
        ms = []
        for _ in range(3):
            p.run(); ms.append(p.timing()[1])
        print("%5d x %d^2 w=%3d %-8s %8.2f ms  %7.1f GCUPS" % (n, ln, w, "lds" if lds else "auto", np.mean(ms), p.cells() / np.mean(ms) / 1e6))
        p.close()
