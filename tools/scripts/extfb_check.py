"""Soak of the four-extensions-per-wavefront form of ksw_extf2_sse (k2a_extf_grp_kernel) against the golden vectors and the pinned oracle.
  usage: [LIBP=<lib.so>] extfb_check.py [seed [rounds]]   (LIBP = tests/sim/libksw2_amd_sim.so: the simulator build)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ksw2_amd as ka
from oracle import pyoracle as po
from oracle.gen_golden_extf import noisy_pair
from tests import golden_util as gu
L = ka.Library(os.environ["LIBP"]) if os.environ.get("LIBP") else ka.Library()
t0 = time.time()
fc = gu.ExtfCases()
cases = [fc.case(k) for k in range(fc.n)]
ngrp = nbad = 0
for sc in sorted({(c["mch"], c["mis"], c["e"]) for c in cases}):
    sub = [c for c in cases if (c["mch"], c["mis"], c["e"]) == sc]
    b = L.make_linear_batch([c["q"] for c in sub], [c["t"] for c in sub], *sc, w=[c["w"] for c in sub], xdrop=[c["xdrop"] for c in sub]) if hasattr(L, "make_linear_batch") else None
    if b is not None:
        p = b.plan(); ngrp += sum(d["tasks"] for d in p.describe() if d["kernel"] == "extf-grp"); p.close()
    res = L.extf_batch([c["q"] for c in sub], [c["t"] for c in sub], *sc, w=[c["w"] for c in sub], xdrop=[c["xdrop"] for c in sub])
    for r, c in zip(res, sub):
        bad = [f for f in gu.FIELDS if r[f] != c["expect"][f]]
        if bad:
            nbad += 1
            if nbad < 10: print("BAD golden", sc, len(c["q"]), len(c["t"]), c["w"], c["xdrop"], {f: (c["expect"][f], r[f]) for f in bad})
print("golden", len(cases), "through extf-grp:", ngrp, "bad:", nbad, round(time.time() - t0, 1))
rng = np.random.Generator(np.random.PCG64(int(sys.argv[1]) if len(sys.argv) > 1 else 3))
nb2 = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    n = 9
    qs, ts = zip(*[noisy_pair(rng, int(rng.integers(20, 3000)), (it + k) % 3) for k in range(n)])
    w = [int(x) for x in rng.choice([0, 1, 5, 15, 16, 40, 100, 158, 159, 160, 300], size=n)]
    xd = [int(x) for x in rng.choice([-1, 30, 200], size=n)]
    mch, mis, e = [(2, -4, 2), (1, -3, 1), (3, -2, 4)][it % 3]
    res = L.extf_batch(list(qs), list(ts), mch, mis, e, w=w, xdrop=xd)
    for k in range(n):
        exp = po.extf2("oracle", qs[k], ts[k], mch, mis, e, w[k], xd[k])
        bad = [f for f in gu.FIELDS if res[k][f] != exp[f]]
        if bad:
            nb2 += 1
            print("BAD", it, k, len(qs[k]), len(ts[k]), w[k], xd[k], {f: (exp[f], res[k][f]) for f in bad})
print("random done, bad =", nb2, round(time.time() - t0, 1))
