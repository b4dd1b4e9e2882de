"""Shared comparison helpers: product library (HIP or simulator build) vs the oracle."""
import numpy as np

from oracle import pyoracle as po

CMP_FIELDS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar", "cigar"]


def diff(a, b, fields=CMP_FIELDS):
    return [f for f in fields if a[f] != b[f]]


def oracle_batch(dual, qs, ts, mat, q, e, q2, e2, w, zdrop, end_bonus, flag):
    n = len(qs)
    bc = lambda v: np.full(n, v) if np.ndim(v) == 0 else np.asarray(v)
    w, zdrop, end_bonus, flag = bc(w), bc(zdrop), bc(end_bonus), bc(flag)
    return [po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, q, e, q2, e2, w=int(w[i]), zdrop=int(zdrop[i]),
                     end_bonus=int(end_bonus[i]), flag=int(flag[i])) for i in range(n)]


def check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=-1, zdrop=-1, end_bonus=0, flag=0, sample=None, fields=CMP_FIELDS, m=None):
    """Run the batch through lib, compare (all pairs, or the index list `sample`) with the oracle; returns #checked."""
    if dual:
        res = lib.extd_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zdrop, end_bonus=end_bonus, flag=flag, m=m)
    else:
        res = lib.extz_batch(qs, ts, mat, q, e, w=w, zdrop=zdrop, end_bonus=end_bonus, flag=flag, m=m)
    n = len(qs)
    idx = range(n) if sample is None else sample
    bc = lambda v: np.full(n, v) if np.ndim(v) == 0 else np.asarray(v)
    w, zdrop, end_bonus, flag = bc(w), bc(zdrop), bc(end_bonus), bc(flag)
    for i in idx:
        exp = po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, q, e, q2, e2, w=int(w[i]), zdrop=int(zdrop[i]),
                       end_bonus=int(end_bonus[i]), flag=int(flag[i]), m=m)
        d = diff(exp, res[i], fields)
        assert not d, ("pair %d dual=%s w=%d zdrop=%d flag=%d qlen=%d tlen=%d" % (i, dual, w[i], zdrop[i], flag[i], len(qs[i]), len(ts[i])),
                       {k: (exp[k], res[i][k]) for k in d if k != "cigar"})
    return len(list(idx)), res


def cigar_score(cigar, q, t, mat, m, gq, ge, gq2=None, ge2=None):
    """Score of a CIGAR (M/I/D runs, start of both sequences) under the (two-piece) affine model; returns (score, qlen used, tlen used)."""
    x = y = 0
    sc = 0
    mat = np.asarray(mat, dtype=np.int64).reshape(m, m)
    for c in cigar:
        op, ln = c & 0xf, c >> 4
        if op == 0:
            sc += int(mat[t[x:x + ln], q[y:y + ln]].sum())
            x += ln
            y += ln
        else:
            cost = gq + ln * ge if gq2 is None else min(gq + ln * ge, gq2 + ln * ge2)
            sc -= cost
            if op == 2:
                x += ln
            else:
                y += ln
    return sc, y, x


def sprinkle_target_wildcards(rng, ts, every=2, runs=(1, 50)):
    """Runs of the wildcard code (4) in every `every`-th target: somewhere inside, and now and then at the very first / last rows."""
    out = [np.array(t, dtype=np.uint8) for t in ts]
    for i in range(0, len(out), every):
        t = out[i]
        for _ in range(int(rng.integers(1, 3))):
            ln = int(min(len(t), rng.integers(runs[0], runs[1] + 1)))
            at = int(rng.integers(0, len(t) - ln + 1))
            t[at:at + ln] = 4
        if rng.random() < 0.3:
            t[0] = 4
        if rng.random() < 0.3:
            t[-1] = 4
    return out


def check_target_wildcards(lib, setenv, delenv, scale=1):
    """Round 6: a TARGET wildcard (code 4) is a row of the packed kernels like any other wherever its scores do not depend on the query
    code (K2aScoring.pk_tn1): selector 0x0c takes penalty 0 out of the column profile and the constant comes off the candidate in a
    branch only wavefronts that hold such a row enter.  Every packed family -- (8,18), (16,8), (64,8), (64,16), plain and re-based, the
    deferred arg-max and its two re-run passes, LDS selectors, solo, generation-serial -- score only and both traceback modes, both
    gap models, Z-drops, against the oracle on every pair, with the plan's own description saying that nothing was demoted to the int32
    kernels and nothing handed back; KSW2AMD_TN=0 and a matrix whose wildcard row varies keep the old rule."""
    rng = np.random.Generator(np.random.PCG64(6061))
    from ksw2_amd import synth
    mat = synth.simple_mat(5, 2, 4, -1)
    seen = set()
    # (pairs, qlen, tlen, w, zdrop, flag, dual, env)
    cases = [(24, 120, 128, 16, -1, po.SCORE_ONLY, False, {}),                                  # (8, 18)
             (24, 128, 120, 16, 60, 0, False, {}),                                             # (16, 8) with CIGAR
             (12, 400, 420, 100, 100, po.RIGHT, True, {}),                                     # (64, 8) two-piece, right-aligned
             (12, 900 * scale, 880 * scale, 300, 200, po.SCORE_ONLY, False, {"KSW2AMD_DEFER": "0", "KSW2AMD_LDSCODES": "0"}),      # (64, 16) registers
             (12, 900 * scale, 880 * scale, 300, 200, po.SCORE_ONLY, False, {"KSW2AMD_DEFER": "0", "KSW2AMD_LDSCODES": "1"}),      # ... selectors in LDS
             (12, 900 * scale, 880 * scale, 300, 60, po.SCORE_ONLY, False, {"KSW2AMD_DEFER": "1"}),                               # deferred arg-max + frozen books
             (8, 1200, 1250, 300, 400, 0, True, {}),                                           # (64, 16) two-piece with CIGAR (row state in LDS or registers)
             (9, 700, 690, 64, 100, 0, False, {"KSW2AMD_SOLO": "all"}),                        # solo
             (9, 700, 690, 64, 100, po.SCORE_ONLY, True, {"KSW2AMD_SOLO": "all"}),
             (12, 300, 310, 40, -1, po.SCORE_ONLY | po.APPROX_MAX, False, {}),                 # KSW_EZ_APPROX_MAX alone: the kernels without maximum tracking
             (12, 900, 880, 300, -1, po.APPROX_MAX, True, {}),                                 # ... two-piece, with the corner CIGAR
             (4, 2300, 2337, -1, -1, 0, False, {}),                                            # generation-serial
             (4, 2300, 2337, -1, 300, po.SCORE_ONLY, True, {})]
    keys = ("KSW2AMD_DEFER", "KSW2AMD_LDSCODES", "KSW2AMD_SOLO", "KSW2AMD_TN", "KSW2AMD_SIMDS")
    for ci, (n, ql, tl, w, zd, flag, dual, env) in enumerate(cases):
        for k in keys:
            delenv(k, raising=False)
        setenv("KSW2AMD_SIMDS", "0")
        for k, v in env.items():
            setenv(k, v)
        qs, ts = synth.fixed_batch(7000 + ci, n, ql, tl, sub=0.05, ind=0.06, tail_random_frac=0.3 if zd >= 0 else 0.0, tail_pairs=0.5 if zd >= 0 else 0.0)
        qs = [np.array(x, dtype=np.uint8) for x in qs]
        ts = sprinkle_target_wildcards(rng, ts, every=2 if n > 4 else 1)
        for i in range(1, n, 5):                                  # query wildcards too: N against N is a cell like any other
            qs[i][len(qs[i]) // 3] = 4
        if n >= 8:                                                # the same position in query and target of one pair
            qs[2][10] = 4; ts[2][10] = 4
        p = lib.make_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag).plan(dual)
        d = p.describe()
        assert d and all(k["kernel"] in ("pk", "solo", "pkmp") and k["tn"] == 1 for k in d), (ci, d)
        assert p.packed_pairs() == n, (ci, p.packed_pairs())
        p.close()
        seen |= {(k["kernel"], k["G"], k["C"], k["form"]) for k in d}
        r0 = lib.rerun_count()
        check_batch(lib, dual, qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag)
        # the flat entry: the arena goes up unscanned -- nothing may come back for a re-run either
        fres = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=0, flag=flag).run_oneshot(dual)
        exp = oracle_batch(dual, qs, ts, mat, 4, 2, 24, 1, w, zd, 0, flag)
        bad = [(i, diff(exp[i], fres[i])) for i in range(n) if diff(exp[i], fres[i])]
        assert not bad, (ci, "flat", bad[:3])
        assert lib.rerun_count() == r0, (ci, lib.rerun_count() - r0)
        if ci in (0, 5):                                          # the old rule, forced: int32 for the scanned batch, hand-back for the flat one; same records
            setenv("KSW2AMD_TN", "0")
            p = lib.make_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag).plan(dual)
            d0 = p.describe()
            assert any(k["kernel"] == "int32" for k in d0) and all(k["tn"] == 0 for k in d0), (ci, d0)
            p.close()
            check_batch(lib, dual, qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag)
            fres = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=0, flag=flag).run_oneshot(dual)
            assert lib.rerun_count() > r0 and not [i for i in range(n) if diff(exp[i], fres[i])], ci
            delenv("KSW2AMD_TN", raising=False)
    for k in keys:
        delenv(k, raising=False)
    assert {"pk", "solo", "pkmp"} <= {s[0] for s in seen} and {(8, 18), (16, 8), (64, 8), (64, 16)} <= {(s[1], s[2]) for s in seen if s[0] == "pk"}, seen
    assert {"registers", "ldscodes", "defer"} <= {s[3] for s in seen}, seen
    # a generic matrix: a constant wildcard row keeps the rule, a wildcard row that depends on the query code does not
    gm = np.array([[3, -4, -2, -4, -1], [-4, 3, -4, -2, -1], [-2, -4, 3, -4, -1], [-4, -2, -4, 3, -1], [-1, -1, -1, -1, -1]], dtype=np.int8)
    qs, ts = synth.fixed_batch(7100, 12, 300, 310, sub=0.05, ind=0.06)
    qs = [np.array(x, dtype=np.uint8) for x in qs]
    ts = sprinkle_target_wildcards(rng, ts, every=2)
    for name, m_, tn in (("constant", gm, 1), ("varying", np.where(np.arange(25).reshape(5, 5) == 21, -3, gm).astype(np.int8), 0)):
        fl = po.SCORE_ONLY | po.GENERIC_SC
        p = lib.make_batch(qs, ts, m_.reshape(-1), 4, 2, 0, 0, w=40, zdrop=-1, flag=fl).plan(False)
        d = p.describe()
        assert all(k["tn"] == tn for k in d if k["kernel"] != "int32") and (any(k["kernel"] == "int32" for k in d) == (tn == 0)), (name, d)
        p.close()
        check_batch(lib, False, qs, ts, m_.reshape(-1), 4, 2, 0, 0, w=40, zdrop=-1, flag=fl)
