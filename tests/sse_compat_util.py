"""Shared checks of the SSE-compatible mode (ksw2_lane_ssec.h) for the simulator tier and the GPU tier: the library against the
golden vectors of the unmodified ksw_extz2_sse / ksw_extd2_sse (tests/golden/sse_cases.npz) and against the pinned oracle
restatement (oracle/ksw2_oracle_sse.c) -- every ksw_extz_t field and the CIGAR."""
import numpy as np

import ksw2_amd as ka
from oracle import pyoracle as po
from tests import golden_util as gu

ALL = gu.FIELDS + ["cigar"]
COMPAT = ka.KSW2AMD_EZ_SSE_COMPAT


def check_golden(lib, step=1):
    """Golden cases in batches that share a scoring, per-pair opt-in flag; returns the number of cases checked."""
    sc = gu.SseCases()
    groups = {}
    for k in range(0, sc.n, step):
        c = sc.case(k)
        groups.setdefault((c["dual"], c["mat"].tobytes(), c["gq"], c["ge"], c["gq2"], c["ge2"]), []).append(c)
    n = 0
    for (dual, _, gq, ge, gq2, ge2), cs in groups.items():
        kw = dict(w=np.array([c["w"] for c in cs]), zdrop=np.array([c["zdrop"] for c in cs]), end_bonus=np.array([c["end_bonus"] for c in cs]),
                  flag=np.array([c["flag"] | COMPAT for c in cs]))
        qs, ts = [c["q"] for c in cs], [c["t"] for c in cs]
        res = lib.extd_batch(qs, ts, cs[0]["mat"], gq, ge, gq2, ge2, **kw) if dual else lib.extz_batch(qs, ts, cs[0]["mat"], gq, ge, **kw)
        for c, r in zip(cs, res):
            bad = [f for f in ALL if r[f] != c["expect"][f]]
            assert not bad, (bad, dual, c["w"], c["zdrop"], hex(c["flag"]), len(c["q"]), len(c["t"]))
            n += 1
    return n


def check_routing(lib):
    """A mixed batch (exact-contract pairs, opted-in pairs, APPROX_DROP pairs) comes back pair by pair as the respective
    definition says; the process-wide switch does the same for an unchanged caller; EQX works in the mode."""
    from ksw2_amd import synth
    rng = np.random.Generator(np.random.PCG64(31))
    mat = po.simple_mat(5, 2, 4, -1)
    pairs = synth.ragged_pairs(rng, 24, 30, 300, sub=0.1, ind=0.2)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    w = rng.choice([3, 8, 20, 64], size=24)
    zd = rng.choice([-1, 30], size=24)
    base = rng.choice([0, po.SCORE_ONLY, po.RIGHT, po.EXTZ_ONLY], size=24)
    kind = np.arange(24) % 3                    # 0 exact contract, 1 opted in, 2 APPROX_MAX | APPROX_DROP (routed by itself)
    flag = np.where(kind == 1, base | COMPAT, np.where(kind == 2, base | po.APPROX_MAX | po.APPROX_DROP, base))
    for dual in (False, True):
        res = lib.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=3, flag=flag) if dual else \
            lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, end_bonus=3, flag=flag)
        for i in range(24):
            func = ("extd2" if dual else "extz2") + ("_sse" if kind[i] else "")
            exp = po.align("oracle", func, qs[i], ts[i], mat, 4, 2, 24, 1, w=int(w[i]), zdrop=int(zd[i]), end_bonus=3, flag=int(flag[i]) & ~COMPAT)
            assert all(exp[f] == res[i][f] for f in ALL), (dual, i, int(kind[i]), [f for f in ALL if exp[f] != res[i][f]])
    # process-wide: single calls of an unchanged caller, with a reused ksw_extz_t
    lib.set_sse_compat(True)
    try:
        ez = ka.KswExtz()
        for i in range(6):
            r = lib.extd2(qs[i], ts[i], mat, 4, 2, 24, 1, w=int(w[i]), zdrop=int(zd[i]), end_bonus=3, flag=int(base[i]) | po.EQX, ez=ez)
            exp = po.align("oracle", "extd2_sse", qs[i], ts[i], mat, 4, 2, 24, 1, w=int(w[i]), zdrop=int(zd[i]), end_bonus=3, flag=int(base[i]) | po.EQX)
            assert all(exp[f] == r[f] for f in ALL), (i, [f for f in ALL if exp[f] != r[f]])
            r = lib.extz2(qs[i], ts[i], mat, 4, 2, w=int(w[i]), zdrop=int(zd[i]), flag=int(base[i]))
            exp = po.align("oracle", "extz2_sse", qs[i], ts[i], mat, 4, 2, w=int(w[i]), zdrop=int(zd[i]), flag=int(base[i]))
            assert all(exp[f] == r[f] for f in ALL), (i, [f for f in ALL if exp[f] != r[f]])
        if ez.cigar:
            ka._libc.free(ez.cigar)
    finally:
        lib.set_sse_compat(False)
    # and off again: the exact contract
    r = lib.extz2(qs[0], ts[0], mat, 4, 2, w=3, zdrop=-1, flag=0)
    exp = po.align("oracle", "extz2", qs[0], ts[0], mat, 4, 2, w=3, zdrop=-1, flag=0)
    assert all(exp[f] == r[f] for f in ALL)


def check_long(lib, n=4, length=3000, w=100):
    """Longer reads with a band, wildcards and Z-drop: every block boundary of the 64-position passes is crossed many times."""
    from ksw2_amd import synth
    rng = np.random.Generator(np.random.PCG64(77))
    mat = po.simple_mat(5, 2, 4, -1)
    pairs = synth.ragged_pairs(rng, n, length // 2, length, sub=0.06, ind=0.1, n_rate=0.002)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    for dual, flag in ((False, 0), (True, po.RIGHT), (True, po.APPROX_MAX | po.APPROX_DROP), (False, po.SCORE_ONLY)):
        fl = np.full(n, flag | COMPAT)
        res = lib.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=200, flag=fl) if dual else lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=200, flag=fl)
        for i in range(n):
            exp = po.align("oracle", "extd2_sse" if dual else "extz2_sse", qs[i], ts[i], mat, 4, 2, 24, 1, w=w, zdrop=200, flag=flag)
            assert all(exp[f] == res[i][f] for f in ALL), (dual, hex(flag), i, [f for f in ALL if exp[f] != res[i][f]])


def check_register_form(lib, setenv, rounds=6, long_len=2600):
    """Tasks with simple scoring take the kernel with the state in registers (k2a_ssec_blk_kernel, ksw2_lane_ssecb.h; with a traceback it
    writes the direction bytes the reference's walk reads): bands from one
    position to the 960 the ring holds (wider ones keep the position-per-lane kernel), targets longer than the ring so that every
    lane takes several blocks, both gap models, exact maximum and the approximate modes, wildcards, Z-drop, unequal lengths whose
    band ends against a sequence end -- against the oracle, and the same batch with the form switched off (KSW2AMD_SSEC_BLK=0)."""
    from ksw2_amd import synth
    rng = np.random.Generator(np.random.PCG64(4242))
    mats = [(po.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (po.simple_mat(5, 1, 3, 0), 6, 1, 13, 0), (po.simple_mat(5, 5, 4, 1), 2, 3, 20, 2)]
    nblk = 0
    for rnd in range(rounds):
        n = 8
        lo, hi = [(20, 200), (long_len // 3, long_len), (30, 90)][rnd % 3]
        pairs = synth.ragged_pairs(rng, n - 2, lo, hi, sub=float(rng.choice([0.03, 0.1, 0.25])), ind=float(rng.choice([0.02, 0.1, 0.2])), n_rate=float(rng.choice([0, 0.01])))
        qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
        qs += [rng.integers(0, 4, size=int(rng.integers(1, 150)), dtype=np.uint8) for _ in range(2)]       # unrelated, unequal lengths
        ts += [rng.integers(0, 4, size=int(rng.integers(1, 150)), dtype=np.uint8) for _ in range(2)]
        w = rng.choice([0, 1, 2, 5, 16, 17, 100, 700, 959, 960, 1100, -1], size=n)
        zd = rng.choice([-1, 20, 100, 400], size=n)
        so = po.SCORE_ONLY
        base = rng.choice([so, so | po.EXTZ_ONLY, so | po.APPROX_MAX | po.EXTZ_ONLY, so | po.APPROX_MAX | po.APPROX_DROP | po.EXTZ_ONLY, so | po.APPROX_MAX,
                           0, po.RIGHT, po.EXTZ_ONLY | po.REV_CIGAR, po.RIGHT | po.EXTZ_ONLY, po.APPROX_MAX, po.APPROX_MAX | po.APPROX_DROP | po.RIGHT], size=n)      # (half of them with a traceback)
        mat, gq, ge, gq2, ge2 = mats[rnd % 3]
        for dual in (False, True):
            func = "extd2_sse" if dual else "extz2_sse"
            exp = [po.align("oracle", func, qs[i], ts[i], mat, gq, ge, gq2, ge2, w=int(w[i]), zdrop=int(zd[i]), end_bonus=3, flag=int(base[i])) for i in range(n)]
            for off in (False, True):
                setenv("KSW2AMD_SSEC_BLK", "0" if off else "")
                b = lib.make_batch(qs, ts, mat, gq, ge, gq2, ge2, w=w, zdrop=zd, end_bonus=3, flag=base | COMPAT)
                p = b.sse_plan(dual)
                forms = {}
                for d in p.describe():
                    forms[d["form"]] = forms.get(d["form"], 0) + d["tasks"]
                p.close()
                span = [min(len(qs[i]), len(ts[i]), (int(w[i]) if 0 <= w[i] <= max(len(qs[i]), len(ts[i])) else max(len(qs[i]), len(ts[i]))) + 1) for i in range(n)]
                assert forms.get("blk", 0) == (0 if off else sum(1 for x in span if x <= 960)), (forms, span)
                nblk += forms.get("blk", 0)
                res = lib.extd_batch(qs, ts, mat, gq, ge, gq2, ge2, w=w, zdrop=zd, end_bonus=3, flag=base | COMPAT) if dual else \
                    lib.extz_batch(qs, ts, mat, gq, ge, w=w, zdrop=zd, end_bonus=3, flag=base | COMPAT)
                for i in range(n):
                    bad = [f for f in ALL if exp[i][f] != res[i][f]]
                    assert not bad, (rnd, dual, off, i, len(qs[i]), len(ts[i]), int(w[i]), int(zd[i]), hex(int(base[i])), bad)
    setenv("KSW2AMD_SSEC_BLK", "")
    return nblk
