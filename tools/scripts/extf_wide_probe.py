"""GPU box: ksw_extf2_sse on bands wider than the four-per-wavefront form takes: the register forms with 32 / 64 lanes per extension
(KSW2AMD_EXTF_GRP unset / 2) against the register windows and LDS forms (KSW2AMD_EXTF_GRP=1), kernel time of one resident plan."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ksw2_amd as ka
from oracle.gen_golden_extf import noisy_pair
L = ka.Library()
rng = np.random.Generator(np.random.PCG64(5))
for length, w, n in ((1000, 100, 16384), (2000, 200, 8192), (2000, 300, 8192), (3000, 400, 4096), (4000, 600, 4096), (4000, 900, 4096)):
    base = [noisy_pair(rng, length, k % 3) for k in range(32)]
    qs = [base[k % 32][0] for k in range(n)]; ts = [base[k % 32][1] for k in range(n)]
    for env in ("1", ""):
        os.environ["KSW2AMD_EXTF_GRP"] = env
        b = L.make_linear_batch(qs, ts, 2, -4, 2, w=w, xdrop=-1)
        p = b.plan()
        kinds = [(d["kernel"], d["tasks"]) for d in p.describe()]
        best = 1e9
        for _ in range(4):
            p.run(); raw = p.fetch_raw(); f, t = p.timing(); best = min(best, t)
        cells = p.cells()
        p.close()
        print("len %5d w %4d n %6d  EXTF_GRP=%-2s %-28s %8.3f ms  %8.1f GCUPS" % (length, w, n, env or "-", kinds, best, cells / best / 1e6))
