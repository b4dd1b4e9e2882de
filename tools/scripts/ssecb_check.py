"""Soak of the SSE-compatible mode (score-only tasks: k2a_ssec_blk_kernel) against the golden vectors and the pinned oracle.
  usage: [LIBP=<lib.so>] ssecb_check.py [seed [rounds]]   (LIBP = tests/sim/libksw2_amd_sim.so: the simulator build)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ksw2_amd as ka
from oracle import pyoracle as po
from tests import golden_util as gu, sse_compat_util as su
from ksw2_amd import synth
L = ka.Library(os.environ["LIBP"]) if os.environ.get("LIBP") else ka.Library()
COMPAT = ka.KSW2AMD_EZ_SSE_COMPAT
ALL = gu.FIELDS + ["cigar"]
t0 = time.time()
print("golden", su.check_golden(L), time.time() - t0)
su.check_routing(L); print("routing ok")
rng = np.random.Generator(np.random.PCG64(int(sys.argv[1]) if len(sys.argv) > 1 else 5))
nbad = 0
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = 6
    lo, hi = [(20, 200), (200, 1200), (900, 2600), (30, 90)][rnd % 4]
    pairs = synth.ragged_pairs(rng, n, lo, hi, sub=float(rng.choice([0.03, 0.1, 0.25])), ind=float(rng.choice([0.02, 0.1, 0.2])), n_rate=float(rng.choice([0, 0.01])))
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    w = rng.choice([0, 1, 2, 5, 15, 16, 17, 40, 100, 300, 700, 959, 960, 1100, -1], size=n)
    zd = rng.choice([-1, 20, 100, 400], size=n)
    base = rng.choice([po.SCORE_ONLY, po.SCORE_ONLY | po.EXTZ_ONLY, po.SCORE_ONLY | po.APPROX_MAX | po.EXTZ_ONLY, po.SCORE_ONLY | po.APPROX_MAX | po.APPROX_DROP | po.EXTZ_ONLY, po.SCORE_ONLY | po.APPROX_MAX,
                       0, po.RIGHT, po.EXTZ_ONLY, po.RIGHT | po.REV_CIGAR | po.EXTZ_ONLY, po.APPROX_MAX, po.APPROX_MAX | po.APPROX_DROP | po.RIGHT, po.REV_CIGAR], size=n)   # (with a traceback too)
    a, b, scn = [(2, 4, -1), (1, 3, 0), (5, 4, 1), (1, 9, -3)][rnd % 4]
    mat = po.simple_mat(5, a, b, scn)
    gq, ge, gq2, ge2 = [(4, 2, 24, 1), (6, 1, 13, 0), (2, 3, 20, 2), (5, 2, 5, 2)][(rnd // 4) % 4]
    for dual in (False, True):
        flag = base | COMPAT
        res = L.extd_batch(qs, ts, mat, gq, ge, gq2, ge2, w=w, zdrop=zd, end_bonus=int(rng.integers(0, 8)) * 0 + 3, flag=flag) if dual else L.extz_batch(qs, ts, mat, gq, ge, w=w, zdrop=zd, end_bonus=3, flag=flag)
        for i in range(n):
            exp = po.align("oracle", "extd2_sse" if dual else "extz2_sse", qs[i], ts[i], mat, gq, ge, gq2, ge2, w=int(w[i]), zdrop=int(zd[i]), end_bonus=3, flag=int(base[i]))
            bad = [f for f in ALL if exp[f] != res[i][f]]
            if bad:
                nbad += 1
                print("BAD", rnd, dual, i, len(qs[i]), len(ts[i]), int(w[i]), int(zd[i]), hex(int(base[i])), bad, {f: (exp[f], res[i][f]) for f in bad if f != "cigar"})
print("random done, bad =", nbad, time.time() - t0)
# unequal lengths and the narrowest bands: the band ends against the target's or the query's end, one position wide for several anti-diagonals
nbad = 0
for rnd in range(60):
    n = 8
    qs = [rng.integers(0, 4, size=int(rng.integers(1, 120)), dtype=np.uint8) for _ in range(n)]
    ts = [rng.integers(0, 4, size=int(rng.integers(1, 120)), dtype=np.uint8) for _ in range(n)]
    w = rng.choice([0, 1, 2, 3, 4, 7, 33], size=n)
    zd = rng.choice([-1, 10, 50], size=n)
    base = rng.choice([po.SCORE_ONLY, po.SCORE_ONLY | po.EXTZ_ONLY, po.SCORE_ONLY | po.APPROX_MAX | po.APPROX_DROP | po.EXTZ_ONLY], size=n)
    mat = po.simple_mat(5, 2, 4, -1)
    for dual in (False, True):
        res = L.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=3, flag=base | COMPAT) if dual else L.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, end_bonus=3, flag=base | COMPAT)
        for i in range(n):
            exp = po.align("oracle", "extd2_sse" if dual else "extz2_sse", qs[i], ts[i], mat, 4, 2, 24, 1, w=int(w[i]), zdrop=int(zd[i]), end_bonus=3, flag=int(base[i]))
            bad = [f for f in ALL if exp[f] != res[i][f]]
            if bad:
                nbad += 1
                print("BAD2", rnd, dual, i, len(qs[i]), len(ts[i]), int(w[i]), int(zd[i]), hex(int(base[i])), bad, {f: (exp[f], res[i][f]) for f in bad if f != "cigar"})
print("unequal done, bad =", nbad)
