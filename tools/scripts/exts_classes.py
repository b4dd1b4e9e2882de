"""Kernel time of the splice-aware extension per window class (GPU box): register windows of 8 / 16 / 24 slots versus the
scratch-array kernel on the same batches.   python tools/scripts/exts_classes.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402

lib = ka.library()
mat = synth.simple_mat(5, 1, 2, 0)
rng = np.random.Generator(np.random.PCG64(3))
for ql, tl, n in ((400, 1500, 8192), (900, 2500, 4096), (1400, 3000, 2048), (2400, 4000, 1024)):
    t = rng.integers(0, 4, size=(n, tl), dtype=np.uint8)
    ex1 = ql // 2
    a, b = 50, tl - 50 - (ql - ex1)
    t[:, a + ex1], t[:, a + ex1 + 1], t[:, b - 2], t[:, b - 1] = 2, 3, 0, 2
    q = np.concatenate([t[:, a:a + ex1], t[:, b:b + ql - ex1]], axis=1).copy()
    mm = rng.random(q.shape) < 0.03
    q[mm] = (q[mm] + 1) & 3
    for big in (0, 1, 2):                      # 0: the host's choice, 1: HBM state for everything, 2: 16 register slots also with traceback
        os.environ.pop("KSW2AMD_EXTS_BIG", None); os.environ.pop("KSW2AMD_EXTS_REG", None)
        if big == 1:
            os.environ["KSW2AMD_EXTS_BIG"] = "1"
        if big == 2:
            os.environ["KSW2AMD_EXTS_REG"] = "1"
        for flag, name in ((ka.KSW_EZ_SCORE_ONLY | 0x100, "score"), (0x100, "cigar")):
            p = lib.make_splice_batch(list(q), list(t), mat, 2, 1, 32, 4, zdrop=-1, flag=flag).plan()
            p.run(); p.timing()
            ms = []
            for _ in range(3):
                p.run(); ms.append(p.timing()[1])
            cells = p.cells()
            p.close()
            print("q=%d t=%d n=%d %-5s %-8s %8.2f ms  %7.1f GCUPS" % (ql, tl, n, name, ["auto", "scratch", "reg16"][big], np.mean(ms), cells / np.mean(ms) / 1e6))
