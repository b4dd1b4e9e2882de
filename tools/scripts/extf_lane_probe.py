"""Gap-linear X-drop extension: one extension per wavefront (register window / LDS forms) against one per lane (KSW2AMD_EXTF_LANE=1)
by batch size.  GPU box:  python tools/scripts/extf_lane_probe.py"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402

lib = ka.library()
for n, ln, w in ((16384, 1000, 100), (65536, 1000, 100), (262144, 1000, 100), (65536, 1000, 30), (65536, 2000, 300), (262144, 300, 50)):
    q, t = synth.fast_fixed(8, n, ln, ln, sub=0.05, ind=0.01)
    for lane, ring in ((0, "1"), (1, "0"), (1, "1")):          # one per wavefront | one per lane, state in HBM scratch | in LDS rings
        os.environ["KSW2AMD_EXTF_LANE"] = str(lane)
        os.environ["KSW2AMD_EXTF_RING"] = ring
        p = lib.make_linear_batch(list(q), list(t), 2, -4, 2, w=w, xdrop=-1).plan()
        d = p.describe()
        p.run(); p.timing()
        ms = []
        for _ in range(3):
            p.run(); ms.append(p.timing()[1])
        print("%6d x %d^2 w=%3d %-22s %8.2f ms  %7.1f GCUPS  %s" % (n, ln, w, "one per lane" if lane else "one per wavefront", np.mean(ms), p.cells() / np.mean(ms) / 1e6,
                                                                  ["%s %s %s" % (x["kernel"], x["form"], x["ring"]) for x in d]), flush=True)
        p.close()
