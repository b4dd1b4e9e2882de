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
