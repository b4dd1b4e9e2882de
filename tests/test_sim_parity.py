"""CPU: the product's host code (ksw2_host_*.c) and per-lane kernel code (ksw2_lane.h) linked against the
host-memory lock-step wave simulator (tests/sim), checked against the oracle through the same C-ABI.

This does not replace the GPU parity tests (tests/test_gpu_parity.py): it pins the packing, geometry choice,
strip schedule, band masks, row bookkeeping and traceback layout on every commit without a GPU.
"""
import os
import subprocess

import numpy as np
import pytest

import ksw2_amd as ka
from ksw2_amd import synth
from oracle import pyoracle as po
from tests import golden_util as gu
from tests.parity_util import check_batch, diff, CMP_FIELDS as CMP

SIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sim")


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", SIM_DIR], check=True, capture_output=True)
    L = ka.Library(os.path.join(SIM_DIR, "libksw2_amd_sim.so"))
    assert L.backend() == "sim"
    return L


@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("mode", [po.SCORE_ONLY, 0, po.RIGHT])
def test_sim_ragged(sim, dual, mode):
    rng = np.random.Generator(np.random.PCG64(4321 + mode + 10 * dual))
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    for rnd in range(3):
        n = 40
        pairs = synth.ragged_pairs(rng, n, 1, [120, 700, 2200][rnd], sub=0.05, ind=0.12, n_rate=0.01 if rnd % 2 else 0.0)
        qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
        w = rng.choice([-1, 0, 1, 5, 20, 64, 68, 69, 100, 284, 285, 500, 536, 537, 1040, 1041], size=n)
        zd = rng.choice([-1, 50, 200, 400], size=n)
        eb = rng.choice([0, 10, 50], size=n)
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) |
                       (po.GENERIC_SC if rng.random() < 0.3 else 0) for _ in range(n)])
        check_batch(sim, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)


def test_sim_generation_serial(sim):
    rng = np.random.Generator(np.random.PCG64(99))
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    pairs = synth.ragged_pairs(rng, 6, 2100, 4200, sub=0.05, ind=0.12, indel_mean=4.0)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    for dual, flag in ((False, 0), (True, po.RIGHT), (False, po.SCORE_ONLY)):
        check_batch(sim, dual, qs, ts, mat, q, e, q2, e2, w=np.array([-1, 1041, 2000, -1, 3000, 1100]),
                    zdrop=np.array([-1, 400, -1, 200, 2000, -1]), flag=flag)


def test_sim_golden_subset(sim):
    """Every 6th committed random case (reference outputs) through the simulator build."""
    rc = gu.RandomCases()
    n = 0
    for k in range(0, rc.n, 6):
        c = rc.case(k)
        dual = "extd" in c["func"]
        if c["func"].endswith("2_sse"):
            r = (sim.extd2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"],
                           flag=c["flag"]) if dual else
                 sim.extz2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"]))
            assert not diff(c["expect"], r, gu.SSE_LOOSE_FIELDS)
        else:
            r = (sim.extd(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=c["zdrop"], flag=c["flag"]) if dual else
                 sim.extz(c["q"], c["t"], c["mat"], c["gq"], c["ge"], w=c["w"], zdrop=c["zdrop"], flag=c["flag"]))
            assert not diff(c["expect"], r, gu.FIELDS + ["cigar"]), (c["func"], c["w"], c["zdrop"], c["flag"])
        n += 1
    assert n == 600


def test_sim_t1q1_and_gg(sim):
    ka_ = gu.known_answers()
    _, ts = gu.read_fasta("t1.fa")
    _, qs = gu.read_fasta("q1.fa")
    mat = gu.simple_mat(5, 2, 4, 0)
    for k, rec in enumerate(ka_["t1q1"]):
        res = sim.extz(qs[k], ts[k], mat, 4, 2)
        assert not diff(rec["ksw_extz/flag=0"], res, gu.FIELDS)
        assert gu.cigar_string(res["cigar"]) == rec["ksw_extz/flag=0"]["cigar"]
        for g in ("gg", "gg2", "gg2_sse"):
            s, c = sim.gg(g, qs[k], ts[k], mat, 4, 2, w=-1)
            assert s == rec["ksw_gg"]["score"] and gu.cigar_string(c) == rec["ksw_gg"]["cigar"]
        s, _ = sim.gg("gg2", qs[k], ts[k], mat, 4, 2, w=-1, with_cigar=False)
        assert s == rec["ksw_gg"]["score"]


def test_sim_edge_cases(sim):
    mat = synth.simple_mat(5, 2, 4, -1)
    one = np.array([1], dtype=np.uint8)
    r = sim.extz2(np.zeros(0, np.uint8), one, mat, 4, 2)
    assert (r["score"], r["max"], r["max_t"], r["n_cigar"], r["zdropped"]) == (ka.KSW_NEG_INF, 0, -1, 0, 0)
    r = sim.extz2(one, one, synth.simple_mat(5, 1, 20, -1), 4, 2)
    assert r["score"] == ka.KSW_NEG_INF and r["n_cigar"] == 0
    rng = np.random.Generator(np.random.PCG64(5))
    t = rng.integers(0, 4, 300, dtype=np.uint8)
    qv = t[:100].copy()
    for qq, tt in ((qv, t), (t, qv)):
        exp = po.align("oracle", "extz2", qq, tt, mat, 4, 2, w=10)
        res = sim.extz2(qq, tt, mat, 4, 2, w=10)
        assert not diff(exp, res) and res["zdropped"] == 1 and res["score"] == ka.KSW_NEG_INF
    # more residue types than int8_t m can hold are refused loudly
    with pytest.raises(ka.Ksw2Error):
        sim.extz_batch([one], [one], np.zeros(128 * 128, np.int8), 4, 2, m=128)


def _wide_alphabet_cases(rng, rnd):
    m = int(rng.choice([6, 21, 24, 64, 127]))
    mat = rng.integers(-6, 3, size=(m, m)).astype(np.int8)
    mat = np.minimum(mat, mat.T)
    np.fill_diagonal(mat, rng.integers(2, 9, size=m))
    mat[m - 1, :] = -1
    mat[:, m - 1] = -1
    n = int(rng.integers(2, 12))
    qs, ts = [], []
    for _ in range(n):
        tl = int(rng.integers(1, 500))
        t = rng.integers(0, m, tl, dtype=np.uint8)
        qq = t.copy()
        mask = rng.random(tl) < 0.15
        qq[mask] = rng.integers(0, m, int(mask.sum()), dtype=np.uint8)
        if rng.random() < 0.5 and tl > 20:
            qq = np.delete(qq, slice(5, 5 + int(rng.integers(1, 10))))
        qs.append(qq)
        ts.append(t)
    w = rng.choice([-1, 5, 30, 100], size=n)
    zd = rng.choice([-1, 50, 200], size=n)
    mode = [0, po.RIGHT, po.SCORE_ONLY][rnd % 3]
    fl = np.array([mode | (po.GENERIC_SC if rng.random() < 0.7 else 0) | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) for _ in range(n)])
    return m, mat.reshape(-1).copy(), qs, ts, w, zd, fl


def test_sim_wide_alphabets(sim):
    """m > 5 residue types (protein-sized matrices): scores come from the LDS copy of the matrix (K2aLane::step)."""
    rng = np.random.Generator(np.random.PCG64(8))
    for rnd in range(9):
        m, mat, qs, ts, w, zd, fl = _wide_alphabet_cases(rng, rnd)
        for dual in (False, True):
            check_batch(sim, dual, qs, ts, mat, 6, 2, 20, 1, w=w, zdrop=zd, flag=fl, m=m)


def test_sim_eqx(sim):
    rng = np.random.Generator(np.random.PCG64(3))
    mat = synth.simple_mat(5, 2, 4, -1)
    for _ in range(20):
        (qq, tt), = synth.ragged_pairs(rng, 1, 20, 200, sub=0.1, ind=0.1)
        exp = po.align("oracle", "extd2", qq, tt, mat, 4, 2, 24, 1, flag=po.EQX)
        res = sim.extd2(qq, tt, mat, 4, 2, 24, 1, flag=po.EQX)
        assert res["cigar"] == exp["cigar"] and all((c & 0xf) != 0 for c in res["cigar"])


def _fixed_shape_cases(rng, rnd):
    mat, q, e, q2, e2 = [(synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (synth.simple_mat(5, 1, 3, 0), 5, 1, 20, 1),
                         (synth.simple_mat(5, 2, 4, -3), 4, 2, 13, 1)][rnd % 3]
    n = int(rng.integers(3, 30))
    ql = int(rng.integers(50, 700))
    tl = max(1, ql + int(rng.integers(-30, 30)))
    w = int(rng.choice([20, 64, 68, 100, 284, 400, -1]))
    qs, ts = synth.fixed_batch(100 + rnd, n, ql, tl, sub=0.05, ind=0.08, tail_random_frac=0.3, tail_pairs=0.3)
    if rnd % 2:
        qs, ts = qs.copy(), ts.copy()
        qs[rng.random(qs.shape) < 0.01] = 4
        ts[rng.random(ts.shape) < 0.01] = 4
    zd = rng.choice([-1, 30, 100, 400], size=n)
    eb = rng.choice([0, 10, 50], size=n)
    fl = np.array([po.SCORE_ONLY | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.GENERIC_SC if rnd % 3 == 0 else 0) |
                   (po.RIGHT if rng.random() < 0.2 else 0) for _ in range(n)])
    return mat, q, e, q2, e2, qs, ts, w, zd, eb, fl


def test_sim_packed_int16_class(sim):
    """Same-shape score-only batches go through the packed-int16 kernels (two alignments per lane): wildcards,
    per-pair Z-drop, odd leftovers, every resident geometry class."""
    rng = np.random.Generator(np.random.PCG64(5))
    npk = 0
    for rnd in range(24):
        mat, q, e, q2, e2, qs, ts, w, zd, eb, fl = _fixed_shape_cases(rng, rnd)
        for dual in (False, True):
            b = sim.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
            p = b.plan(dual)
            npk += p.packed_pairs()
            p.close()
            check_batch(sim, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
    assert npk > 300          # batches with wildcards stay on the int32 kernels


def test_sim_packed_rebased(sim):
    """Reads too long for absolute 16-bit scores: per-strip bases (K2aLanePk RB = true).  Score-only and both traceback
    modes, Z-drop on and off, all-match / all-mismatch rows (fastest drift of the base), bands up to the window limit."""
    rng = np.random.Generator(np.random.PCG64(77))
    scs = [(synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (synth.simple_mat(5, 10, 12, 0), 12, 4, 40, 2),
           (synth.simple_mat(5, 1, 3, 0), 5, 1, 20, 1), (synth.simple_mat(5, 6, 9, -3), 9, 3, 30, 1)]
    wide = [[400, 536, 560], [140, 170, 180], [500, 600, 700], [200, 250, 270]]
    npk = ntot = 0
    for rnd in range(12):
        mat, q, e, q2, e2 = scs[rnd % 4]
        n = int(rng.integers(2, 6))
        ql = int(rng.integers(1500, 4000))
        tl = ql + int(rng.integers(-60, 60))
        w = int(rng.choice([10, 20, 64, 68, 100, 150])) if rnd < 8 else int(rng.choice(wide[rnd % 4]))
        qs, ts = synth.fixed_batch(900 + rnd, n, ql, tl, sub=0.05, ind=0.1, tail_random_frac=0.3, tail_pairs=0.3)
        qs, ts = qs.copy(), ts.copy()
        if rnd % 5 == 1:
            qs[0, :] = 0; ts[0, :] = 0
        if rnd % 5 == 2:
            qs[0, :] = 1; ts[0, :] = 2
        zd = rng.choice([-1, 100, 400, 2000], size=n)
        eb = rng.choice([0, 10, 50], size=n)
        mode = [po.SCORE_ONLY, 0, po.RIGHT][rnd % 3]
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) |
                       (po.GENERIC_SC if rnd % 4 == 0 else 0) for _ in range(n)])
        for dual in (False, True):
            p = sim.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl).plan(dual)
            npk += p.packed_pairs(); ntot += n
            p.close()
            check_batch(sim, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
    assert npk > ntot // 2


def test_sim_packed_range_guard(sim):
    """Shapes whose scores could leave the int16 window must stay on the int32 kernels."""
    mat = synth.simple_mat(5, 2, 4, -1)
    q, t = synth.fixed_batch(6, 2, 9000, 9000)
    p = sim.make_batch(q, t, mat, 4, 2, 24, 1, w=100, zdrop=-1, flag=po.SCORE_ONLY).plan(False)
    assert p.packed_pairs() == 2       # long reads, narrow band: re-based packed kernels
    p.close()
    p = sim.make_batch(q, t, mat, 4, 2, 24, 1, w=700, zdrop=-1, flag=po.SCORE_ONLY).plan(False)
    assert p.packed_pairs() == 0       # band window too wide for 16 bits
    p.close()
    os.environ["KSW2AMD_NO_RB"] = "1"
    try:
        p = sim.make_batch(q, t, mat, 4, 2, 24, 1, w=100, zdrop=-1, flag=po.SCORE_ONLY).plan(False)
        assert p.packed_pairs() == 0
        p.close()
    finally:
        del os.environ["KSW2AMD_NO_RB"]
    q, t = synth.fixed_batch(2, 4, 512, 512)
    p = sim.make_batch(q, t, mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=po.SCORE_ONLY).plan(False)
    assert p.packed_pairs() == 4
    p.close()
    # a generic matrix without match/mismatch structure IS packed (round 5: column profiles, any matrix over <= 5 codes)
    m2 = mat.copy(); m2[1] = -3
    p = sim.make_batch(q, t, m2, 4, 2, 24, 1, w=64, zdrop=-1, flag=po.SCORE_ONLY | po.GENERIC_SC).plan(False)
    assert p.packed_pairs() == 4
    p.close()
    check_batch(sim, False, q, t, m2, 4, 2, 0, 0, w=64, flag=po.SCORE_ONLY | po.GENERIC_SC)


def _generic_matrices(rng):
    """Scoring matrices WITHOUT match / mismatch structure (KSW_EZ_GENERIC_SC; ksw2_extz2_sse.c:142-143): transitions cheaper than
    transversions, an asymmetric random one, an all-different one, and alphabets of four and three codes (no wildcard row / column)."""
    tt = np.array([[2, -4, -2, -4, -1], [-4, 2, -4, -2, -1], [-2, -4, 2, -4, -1], [-4, -2, -4, 2, -1], [-1, -1, -1, -1, -1]], dtype=np.int8)
    yield 5, tt.reshape(-1), 4, 2, 24, 1
    r = rng.integers(-6, 1, size=(5, 5)).astype(np.int8)
    r[np.arange(5), np.arange(5)] = rng.integers(1, 6, size=5)
    yield 5, r.reshape(-1), 5, 2, 20, 1
    yield 5, (np.arange(25, dtype=np.int8) % 11 - 7).reshape(-1), 6, 1, 18, 1
    r4 = rng.integers(-5, 0, size=(4, 4)).astype(np.int8)
    r4[np.arange(4), np.arange(4)] = [3, 2, 4, 1]
    yield 4, r4.reshape(-1), 4, 2, 24, 1
    yield 3, np.array([2, -3, -1, -2, 3, -4, -1, -3, 1], dtype=np.int8), 3, 1, 15, 1


def _check_generic_packed(lib, scale=1.0):
    """Every packed kernel family on generic matrices (round 5: column profiles): one-shape batches through the resident geometries
    (plain and re-based), unique shapes through the solo kernel, an unbanded long pair through the generation-serial class, the
    deferred arg-max by the plan's own rules; score-only and both traceback modes, both gap models, Z-drops, wildcards in the QUERY
    (entry 4 of the table) -- and what the launch really took is asserted from the plan's description.  Against the oracle, every field."""
    rng = np.random.Generator(np.random.PCG64(2605))
    kinds, npk, ntot = set(), 0, 0
    shapes = [(24, 150, 160, 20), (26, 420, 400, 64), (12, 700, 690, 100), (10, 900, 930, 300), (6, int(3000 * scale), int(3000 * scale), 100), (4, 2300, 2250, -1)]
    for mi, (m, mat, q, e, q2, e2) in enumerate(_generic_matrices(rng)):
        for si, (n, ql, tl, w) in enumerate(shapes):
            if (mi + si) % 2 and si >= 3:
                continue                                        # (the long shapes with every other matrix: CPU seconds)
            qs = [rng.integers(0, min(m, 4), ql, dtype=np.uint8) for _ in range(n)]
            ts = []
            for x in qs:                                        # targets: the query through a substitution / indel channel, trimmed to tl
                y = x.copy()
                y[rng.random(ql) < 0.08] = rng.integers(0, min(m, 4))
                cut = int(rng.integers(0, max(1, ql - 40)))
                y = np.concatenate([y[:cut], y[cut + int(rng.integers(0, 12)):], rng.integers(0, min(m, 4), 64, dtype=np.uint8)])[:tl]
                ts.append(np.ascontiguousarray(np.concatenate([y, rng.integers(0, min(m, 4), max(0, tl - len(y)), dtype=np.uint8)])))
            if m == 5:
                for x in qs[::3]:
                    x[rng.random(ql) < 0.01] = 4                # wildcards in the query: scored in place by the packed kernels
            for dual in (False, True):
                mode = [po.SCORE_ONLY, 0, po.RIGHT][(mi + si + dual) % 3]
                zd = rng.choice([-1, 60, 400], size=n)
                fl = np.full(n, mode | po.GENERIC_SC)
                b = lib.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=0, flag=fl, m=m)
                p = b.plan(dual)
                kinds.update(d["kernel"] + ("-defer" if d.get("form") == "defer" else "") for d in p.describe())
                npk += p.packed_pairs(); ntot += n
                p.close()
                check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=0, flag=fl, m=m)
        # unique shapes: the solo kernel
        pairs = synth.ragged_pairs(rng, 12, 200, 900, sub=0.06, ind=0.10)
        qs, ts = [np.minimum(p_[0], min(m, 4) - 1) for p_ in pairs], [np.minimum(p_[1], min(m, 4) - 1) for p_ in pairs]
        if m == 5:
            qs[0][7] = 4
        for dual in (False, True):
            fl = np.full(12, [0, po.SCORE_ONLY][dual] | po.GENERIC_SC)
            b = lib.make_batch(qs, ts, mat, q, e, q2, e2, w=80, zdrop=200, end_bonus=0, flag=fl, m=m)
            p = b.plan(dual)
            kinds.update(d["kernel"] for d in p.describe())
            npk += p.packed_pairs(); ntot += 12
            p.close()
            check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=80, zdrop=200, end_bonus=0, flag=fl, m=m)
    assert npk == ntot, (npk, ntot)                              # no pair left the packed kernels for its matrix (no target holds a wildcard)
    assert {"pk", "solo", "pkmp"} <= kinds, kinds
    return kinds


def test_sim_packed_generic_matrices(sim, monkeypatch):
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")
    _check_generic_packed(sim, scale=0.5)
    monkeypatch.setenv("KSW2AMD_DEFER", "1")                     # ... and with the deferred arg-max forced on for the score-only single-gap classes
    kinds = _check_generic_packed(sim, scale=0.5)
    assert "pk-defer" in kinds, kinds


def _exts_cases(rng, rnd, big_limit):
    from oracle.gen_golden_exts import spliced_pair
    n = int(rng.integers(1, 8))
    qs, ts, js = [], [], []
    for _ in range(n):
        q, t = spliced_pair(rng, int(rng.integers(1, big_limit if rnd % 7 == 0 else 400)), low_complexity=(rnd % 5 == 2))
        qs.append(q)
        ts.append(t)
        js.append((rng.integers(0, 16, len(t), dtype=np.uint8) * (rng.random(len(t)) < 0.05)).astype(np.uint8) if rnd % 3 == 0 else None)
    a, b, scn, q_, e_, q2, nc = [(1, 2, 0, 2, 1, 32, 4), (2, 4, -1, 4, 2, 24, 5), (1, 3, 0, 2, 1, 20, 9)][rnd % 3]
    flag = np.array([int(rng.choice([0, po.SCORE_ONLY, po.RIGHT, po.EXTZ_ONLY, po.REV_CIGAR, po.GENERIC_SC, po.RIGHT | po.REV_CIGAR])) |
                     int(rng.choice([0, po.SPLICE_FOR, po.SPLICE_REV, po.SPLICE_FOR | po.SPLICE_FLANK, po.SPLICE_FOR | po.SPLICE_REV])) for _ in range(n)])
    zd = rng.choice([-1, 20, 100, 400], size=n)
    return qs, ts, js, synth.simple_mat(5, a, b, scn), q_, e_, q2, nc, (3 if rnd % 3 == 0 else 0), flag, zd


def check_exts_batch(lib, qs, ts, js, mat, q, e, q2, nc, jb, flag, zd):
    res = lib.exts_batch(qs, ts, mat, q, e, q2, nc, zdrop=zd, junc_bonus=jb, flag=flag, juncs=js)
    for i in range(len(qs)):
        exp = po.exts2("oracle", qs[i], ts[i], mat, q, e, q2, nc, zdrop=int(zd[i]), junc_bonus=jb, flag=int(flag[i]), junc=js[i])
        d = diff(exp, res[i], gu.FIELDS + ["cigar"])
        assert not d, (i, len(qs[i]), len(ts[i]), hex(int(flag[i])), int(zd[i]), {k: (exp[k], res[i][k]) for k in d if k != "cigar"})


def _intron_pair(rng, tl):
    """A long target with one GT..AG intron and a query made of the two exons around it."""
    t = rng.integers(0, 4, tl, dtype=np.uint8)
    a = int(rng.integers(50, tl // 3))
    ex1, intr, ex2 = int(rng.integers(40, 500)), int(rng.integers(100, tl // 2)), int(rng.integers(40, 700))
    b = min(tl - 5, a + ex1 + intr)
    t[a + ex1], t[a + ex1 + 1], t[b - 2], t[b - 1] = 2, 3, 0, 2
    q = np.concatenate([t[a:a + ex1], t[b:min(tl, b + ex2)]]).copy()
    mm = rng.random(len(q)) < 0.04
    q[mm] = rng.integers(0, 4, int(mm.sum()), dtype=np.uint8)
    return q, t


def test_sim_splice_aware_16_slots_with_traceback(sim, monkeypatch):
    """The 16-slot register window is only chosen for score-only launches; KSW2AMD_EXTS_REG forces it with traceback."""
    monkeypatch.setenv("KSW2AMD_EXTS_REG", "1")
    rng = np.random.Generator(np.random.PCG64(31))
    for rnd in (0, 7, 14):
        check_exts_batch(sim, *_exts_cases(rng, rnd, 950))


def test_sim_splice_aware(sim):
    """ksw_exts2_sse semantics through the diagonal-major kernel (ksw2_lane_dm.h): random spliced pairs with junction
    annotation, every flag combination, tie-heavy two-letter sequences, and diagonals longer than one register slot."""
    rng = np.random.Generator(np.random.PCG64(21))
    for rnd in range(22):
        check_exts_batch(sim, *_exts_cases(rng, rnd, 1400))


def test_sim_splice_aware_sliding_window(sim):
    rng = np.random.Generator(np.random.PCG64(4))
    mat = synth.simple_mat(5, 1, 2, 0)
    for rnd in range(8):
        q, t = _intron_pair(rng, int(rng.integers(1500, 6000)))
        if rnd % 2:
            q, t = t, q                              # long query, short target
        flag = int(rng.choice([0, po.RIGHT, po.SCORE_ONLY, po.EXTZ_ONLY])) | po.SPLICE_FOR
        zd = int(rng.choice([-1, 200, 1000]))
        exp = po.exts2("oracle", q, t, mat, 2, 1, 32, 4, zdrop=zd, flag=flag)
        res = sim.exts2(q, t, mat, 2, 1, 32, 4, zdrop=zd, flag=flag)
        assert not diff(exp, res, gu.FIELDS + ["cigar"]), (rnd, len(q), len(t))
        if flag == po.SPLICE_FOR and zd == -1 and not rnd % 2:
            assert any((c & 0xf) == 3 and (c >> 4) > 50 for c in res["cigar"])       # the intron comes back as N
    # a diagonal that does not fit the largest register window: state in the scratch array (k2a_exts_big_kernel)
    q, t = _intron_pair(rng, 2600)
    q = np.concatenate([q, t[-1500:]])[:1700]
    for flag in (po.SPLICE_FOR, po.SPLICE_FOR | po.SCORE_ONLY):
        exp = po.exts2("oracle", q, t, mat, 2, 1, 32, 4, zdrop=500, flag=flag)
        assert not diff(exp, sim.exts2(q, t, mat, 2, 1, 32, 4, zdrop=500, flag=flag), gu.FIELDS + ["cigar"])


@pytest.mark.parametrize("big", [False, True])
def test_sim_splice_aware_golden_subset(sim, big, monkeypatch):
    if big:
        monkeypatch.setenv("KSW2AMD_EXTS_BIG", "1")      # every case through the scratch-array kernel
    ec = gu.ExtsCases()
    for k in range(1 if big else 0, ec.n, 5):
        c = ec.case(k)
        res = sim.exts2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["noncan"], zdrop=c["zdrop"], junc_bonus=c["junc_bonus"],
                        flag=c["flag"], junc=c["junc"])
        assert not diff(c["expect"], res, gu.FIELDS + ["cigar"]), (k, hex(c["flag"]))


def test_sim_approx_max_mode(sim):
    ac = gu.ApproxCases()
    for k in range(0, ac.n, 2):
        c = ac.case(k)
        assert not diff(c["expect"], gu.ApproxCases.run(sim, c), gu.FIELDS + ["cigar"]), (k, c["func"], hex(c["flag"]))


def _approx_batches(rng, rnd):
    mat, q, e, q2, e2 = [(synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (synth.simple_mat(5, 1, 3, 0), 5, 1, 20, 1)][rnd % 2]
    n = int(rng.integers(3, 14))
    long_reads = rnd % 3 == 2
    ql = int(rng.integers(2500, 6000)) if long_reads else int(rng.integers(40, 800))
    tl = max(1, ql + int(rng.integers(-30, 30)))
    w = int(rng.choice([20, 64, 100, 150])) if long_reads else int(rng.choice([20, 64, 68, 100, 284, 400, -1]))
    qs, ts = synth.fixed_batch(4000 + rnd, n, ql, tl, sub=0.05, ind=0.08, tail_random_frac=0.3, tail_pairs=0.3)
    fl = np.array([0x08 | int(rng.choice([0, po.RIGHT, po.SCORE_ONLY, po.EXTZ_ONLY, po.REV_CIGAR])) for _ in range(n)])
    return mat, q, e, q2, e2, qs, ts, w, rng.choice([-1, 100, 400], size=n), rng.choice([0, 50], size=n), fl


def test_sim_approx_max_packed_classes(sim):
    """KSW_EZ_APPROX_MAX batches run the packed kernels without row-maximum tracking (NOMAX), plain and re-based."""
    rng = np.random.Generator(np.random.PCG64(66))
    npk = ntot = 0
    for rnd in range(15):
        mat, q, e, q2, e2, qs, ts, w, zd, eb, fl = _approx_batches(rng, rnd)
        for dual in (False, True):
            p = sim.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl).plan(dual)
            npk += p.packed_pairs(); ntot += len(qs)
            p.close()
            check_batch(sim, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
    assert npk > ntot // 2


def _solo_cases(rng, rnd, n):
    scs = [(synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (synth.simple_mat(5, 1, 3, 0), 4, 1, 24, 1),
           (synth.simple_mat(5, 2, 4, -3), 4, 2, 13, 1), (synth.simple_mat(5, 2, 5, -1), 5, 3, 20, 2)]
    mat, q, e, q2, e2 = scs[rnd % 4]
    pairs = synth.ragged_pairs(rng, n, 1, [120, 700, 2200, 6000][rnd % 4], sub=0.05, ind=0.12)
    for i in range(0, n, 5):              # very uneven shapes: long leading / trailing gaps through the long piece
        pairs[i] = (rng.integers(0, 4, int(rng.integers(1, 300))).astype(np.uint8), rng.integers(0, 4, int(rng.integers(1, 300))).astype(np.uint8))
    if rnd % 3 == 0:
        pairs[1] = (np.full(39, 1, np.uint8), np.full(12, 2, np.uint8))
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    w = rng.choice([-1, 0, 1, 5, 8, 9, 15, 16, 17, 20, 30, 64, 68, 100, 284, 500], size=n)
    zd = rng.choice([-1, 50, 200, 400], size=n)
    eb = rng.choice([0, 10, 50], size=n)
    return mat, q, e, q2, e2, qs, ts, w, zd, eb


@pytest.mark.parametrize("dual", [False, True])
def test_sim_solo_kernel(sim, dual, monkeypatch):
    """One alignment per wavefront on both register halves (ksw2_lane_solo.h): KSW2AMD_SOLO=all sends every eligible
    alignment there.  Ragged shapes, every band regime of the double strips, Z-drop, all three modes."""
    monkeypatch.setenv("KSW2AMD_SOLO", "all")
    rng = np.random.Generator(np.random.PCG64(808 + dual))
    nsolo = ntot = 0
    for rnd in range(9):
        n = 20
        mat, q, e, q2, e2, qs, ts, w, zd, eb = _solo_cases(rng, rnd, n)
        mode = [po.SCORE_ONLY, 0, po.RIGHT][rnd % 3]
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) for _ in range(n)])
        p = sim.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl).plan(dual)
        nsolo += p.packed_pairs(); ntot += n
        p.close()
        check_batch(sim, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
    assert nsolo > ntot * 3 // 4


def test_sim_solo_takes_unpaired_leftovers(sim, monkeypatch):
    """KSW2AMD_SOLO=1: same-shape pairs stay on the two-per-lane kernels, alignments without a partner go solo."""
    monkeypatch.setenv("KSW2AMD_SOLO", "1")
    mat = synth.simple_mat(5, 2, 4, -1)
    qs, ts = synth.fixed_batch(31, 6, 900, 880, sub=0.05, ind=0.1)
    rng = np.random.Generator(np.random.PCG64(12))
    extra = synth.ragged_pairs(rng, 5, 500, 1500, sub=0.05, ind=0.1)
    qs = [x for x in qs] + [p[0] for p in extra]; ts = [x for x in ts] + [p[1] for p in extra]
    for flag in (0, po.SCORE_ONLY):
        p = sim.make_batch(qs, ts, mat, 4, 2, 24, 1, w=64, zdrop=400, flag=flag).plan(True)
        assert p.packed_pairs() == 11
        p.close()
        check_batch(sim, True, qs, ts, mat, 4, 2, 24, 1, w=64, zdrop=400, flag=flag)


def test_sim_linear_xdrop_golden_and_batches(sim, monkeypatch):
    """ksw_extf2_sse through the simulator build: every 4th reference case one by one, the rest as one batch (LDS-state
    classes), the same batch again with the state arrays in HBM scratch, empty sequences."""
    fc = gu.ExtfCases()
    cases = [fc.case(k) for k in range(fc.n)]
    for c in cases[::4]:
        r = sim.extf2(c["q"], c["t"], c["mch"], c["mis"], c["e"], c["w"], c["xdrop"])
        assert not diff(r, c["expect"], gu.FIELDS), (len(c["q"]), len(c["t"]), c["w"], c["xdrop"])
    for c in cases[1::4]:                             # LDS-state kernel for every band width
        monkeypatch.setenv("KSW2AMD_EXTF_LDS", "1")
        r = sim.extf2(c["q"], c["t"], c["mch"], c["mis"], c["e"], c["w"], c["xdrop"])
        assert not diff(r, c["expect"], gu.FIELDS), (len(c["q"]), len(c["t"]), c["w"], c["xdrop"])
    monkeypatch.delenv("KSW2AMD_EXTF_LDS")
    monkeypatch.setenv("KSW2AMD_EXTF_WIN", "1")        # the register window wherever the band fits it
    for c in cases[2::4]:
        r = sim.extf2(c["q"], c["t"], c["mch"], c["mis"], c["e"], c["w"], c["xdrop"])
        assert not diff(r, c["expect"], gu.FIELDS), (len(c["q"]), len(c["t"]), c["w"], c["xdrop"])
    monkeypatch.delenv("KSW2AMD_EXTF_WIN")
    rng = np.random.Generator(np.random.PCG64(99))     # both register windows at their limits, windows sliding over long targets
    from oracle.gen_golden_extf import noisy_pair
    for it in range(60):
        q, t = noisy_pair(rng, int(rng.integers(300, 3000)), it % 3)
        w = int(rng.choice([100, 145, 146, 147, 300, 401, 402, 403]))
        xd = int(rng.choice([-1, 200]))
        assert not diff(sim.extf2(q, t, 2, -4, 2, w, xd), po.extf2("oracle", q, t, 2, -4, 2, w, xd), gu.FIELDS), (len(q), len(t), w, xd)
    for hbm in (False, True):
        if hbm:
            monkeypatch.setenv("KSW2AMD_EXTF_HBM", "1")
        for sc in sorted({(c["mch"], c["mis"], c["e"]) for c in cases}):
            sub = [c for c in cases if (c["mch"], c["mis"], c["e"]) == sc][:: 3 if hbm else 1]
            res = sim.extf_batch([c["q"] for c in sub], [c["t"] for c in sub], *sc, w=[c["w"] for c in sub], xdrop=[c["xdrop"] for c in sub])
            for r, c in zip(res, sub):
                assert not diff(r, c["expect"], gu.FIELDS), (sc, len(c["q"]), len(c["t"]), c["w"], c["xdrop"])
    e = np.zeros(0, np.uint8); one = np.array([2], np.uint8); five = np.arange(5, dtype=np.uint8) % 4
    for q, t in ((e, e), (one, e), (e, one), (five, e), (e, five), (one, one)):
        assert not diff(sim.extf2(q, t, 2, -4, 2, -1, 50), po.extf2("oracle", q, t, 2, -4, 2, -1, 50), gu.FIELDS), (len(q), len(t))


def _check_extf_group_form(lib, monkeypatch, rounds, maxlen):
    """ksw_extf2_sse for narrow bands, four extensions per wavefront (k2a_extf_grp_kernel, ksw2_lane_extfb.h): the golden cases whose
    band fits (asserted from the plan's description), random extensions with bands of 1 to 160 positions over targets several rings
    long, X-drop, groups of unequal lengths and task counts that leave a wavefront's last groups empty -- against the reference's
    outputs and the oracle; the same batches with the form off take the other kernels."""
    from oracle.gen_golden_extf import noisy_pair
    fc = gu.ExtfCases()
    cases = [fc.case(k) for k in range(fc.n)]
    ngrp = 0
    for sc in sorted({(c["mch"], c["mis"], c["e"]) for c in cases}):
        sub = [c for c in cases if (c["mch"], c["mis"], c["e"]) == sc]
        kw = dict(w=[c["w"] for c in sub], xdrop=[c["xdrop"] for c in sub])
        p = lib.make_linear_batch([c["q"] for c in sub], [c["t"] for c in sub], *sc, **kw).plan()
        span = [min(len(c["q"]), len(c["t"]), (c["w"] if c["w"] >= 0 else max(len(c["q"]), len(c["t"]))) + 1) for c in sub if len(c["q"]) and len(c["t"])]
        got = sum(d["tasks"] for d in p.describe() if d["kernel"] == "extf-grp")
        p.close()
        assert got == sum(1 for x in span if x <= 160), (got, sc)
        ngrp += got
        res = lib.extf_batch([c["q"] for c in sub], [c["t"] for c in sub], *sc, **kw)
        for r, c in zip(res, sub):
            assert not diff(r, c["expect"], gu.FIELDS), (sc, len(c["q"]), len(c["t"]), c["w"], c["xdrop"])
    assert ngrp > 1500
    rng = np.random.Generator(np.random.PCG64(77))
    for it in range(rounds):
        n = int(rng.choice([1, 3, 4, 5, 9]))
        qs, ts = zip(*[noisy_pair(rng, int(rng.integers(20, maxlen)), (it + k) % 3) for k in range(n)])
        w = [int(x) for x in rng.choice([0, 1, 5, 15, 16, 40, 100, 158, 159, 160, 300], size=n)]
        xd = [int(x) for x in rng.choice([-1, 30, 200], size=n)]
        mch, mis, e = [(2, -4, 2), (1, -3, 1), (3, -2, 4)][it % 3]
        exp = [po.extf2("oracle", qs[k], ts[k], mch, mis, e, w[k], xd[k]) for k in range(n)]
        for off in ("", "0"):
            monkeypatch.setenv("KSW2AMD_EXTF_GRP", off)
            res = lib.extf_batch(list(qs), list(ts), mch, mis, e, w=w, xdrop=xd)
            for k in range(n):
                assert not diff(res[k], exp[k], gu.FIELDS), (it, off, k, len(qs[k]), len(ts[k]), w[k], xd[k])
    monkeypatch.delenv("KSW2AMD_EXTF_GRP")


def _check_extf_wide_group_forms(lib, monkeypatch, rounds, maxlen):
    """The same kernel with 32 / 64 lanes per extension (two / one per wavefront; bands of 161..416 / 417..928 positions): bands on
    both sides of each limit, targets several rings long (a lane takes its next block every 512 / 1 024 positions), X-drop, task
    counts that leave the second group of a wavefront empty -- against the oracle, with the plan's description asserted for the
    switch's four settings (unset / 2: every band that fits; 1: the four-per-wavefront form only; 0: none)."""
    from oracle.gen_golden_extf import noisy_pair
    rng = np.random.Generator(np.random.PCG64(78))
    seen = {"extf-grp": 0, "extf-grp32": 0, "extf-grp64": 0}
    for it in range(rounds):
        n = int(rng.choice([1, 2, 3, 5]))
        qs, ts = zip(*[noisy_pair(rng, int(rng.integers(maxlen // 3, maxlen)), (it + k) % 3) for k in range(n)])
        w = [int(x) for x in rng.choice([100, 159, 160, 161, 250, 414, 415, 416, 500, 700, 926, 927, 928, 1200, -1], size=n)]
        xd = [int(x) for x in rng.choice([-1, 100, 600], size=n)]
        mch, mis, e = [(2, -4, 2), (1, -3, 1), (3, -2, 4)][it % 3]
        exp = [po.extf2("oracle", qs[k], ts[k], mch, mis, e, w[k], xd[k]) for k in range(n)]
        span = [min(len(qs[k]), len(ts[k]), (w[k] if w[k] >= 0 else max(len(qs[k]), len(ts[k]))) + 1) for k in range(n)]
        for env in ("", "2", "1", "0"):
            monkeypatch.setenv("KSW2AMD_EXTF_GRP", env)
            p = lib.make_linear_batch(list(qs), list(ts), mch, mis, e, w=w, xdrop=xd).plan()
            got = {}
            for d in p.describe():
                got[d["kernel"]] = got.get(d["kernel"], 0) + d["tasks"]
            p.close()
            want = {"extf-grp": sum(1 for x in span if x <= 160) if env != "0" else 0,
                    "extf-grp32": sum(1 for x in span if 160 < x <= 416) if env in ("", "2") else 0,
                    "extf-grp64": sum(1 for x in span if 416 < x <= 928) if env in ("", "2") else 0}
            assert {k: got.get(k, 0) for k in want} == want, (env, got, span)
            if env == "":
                for k in want:
                    seen[k] += want[k]
            res = lib.extf_batch(list(qs), list(ts), mch, mis, e, w=w, xdrop=xd)
            for k in range(n):
                assert not diff(res[k], exp[k], gu.FIELDS), (it, env, k, len(qs[k]), len(ts[k]), w[k], xd[k])
    monkeypatch.delenv("KSW2AMD_EXTF_GRP")
    assert seen["extf-grp32"] >= rounds // 3 and seen["extf-grp64"] >= rounds // 3, seen


def test_sim_linear_xdrop_group_form(sim, monkeypatch):
    _check_extf_group_form(sim, monkeypatch, rounds=24, maxlen=2500)


def test_sim_linear_xdrop_wide_group_forms(sim, monkeypatch):
    _check_extf_wide_group_forms(sim, monkeypatch, rounds=10, maxlen=3200)


@pytest.mark.parametrize("lds", ["0", "1"])
def test_sim_row_state_in_registers_and_in_lds(sim, lds, monkeypatch):
    """The classes that can keep row maxima / arg-max columns (and, packed, target codes) in LDS -- packed (64, 16) two-piece
    with traceback, generation-serial single-gap with traceback -- in both forms (KSW2AMD_LDSROWS)."""
    monkeypatch.setenv("KSW2AMD_LDSROWS", lds)
    rng = np.random.Generator(np.random.PCG64(2024))
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    npk = 0
    for rnd, (ql, w) in enumerate(((2500, 400), (5200, 500), (1800, 330))):      # w > 284: the 16-row geometry
        qs, ts = synth.fixed_batch(700 + rnd, 4, ql, ql - 20, sub=0.05, ind=0.1, tail_random_frac=0.3, tail_pairs=0.3)
        zd = rng.choice([-1, 400, 2000], size=4)
        for mode in (0, po.RIGHT):
            fl = np.array([mode | (po.REV_CIGAR if rng.random() < 0.3 else 0) for _ in range(4)])
            p = sim.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, flag=fl).plan(True)
            npk += p.packed_pairs()
            p.close()
            check_batch(sim, True, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, flag=fl)
    assert npk >= 16
    pairs = synth.ragged_pairs(rng, 3, 2100, 3000, sub=0.05, ind=0.12, indel_mean=4.0)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    for flag in (0, po.RIGHT):
        check_batch(sim, False, qs, ts, mat, q, e, q2, e2, w=np.array([-1, 1500, 2000]), zdrop=np.array([-1, 400, -1]), flag=flag)


def _code_plane_batches(rng, scale=1.0):
    """Score-only batches for the two packed geometries whose launcher may keep the target-code planes in LDS -- (64, 16): band
    285..536 on targets of more than 512 rows; (8, 18): band up to 67; (16, 8): band 68 -- in the plain (short reads) and the re-based (long reads)
    number format, exact and KSW_EZ_APPROX_MAX launches, Z-drop on / off, extension flags; shapes in twos and threes (tasks pair up,
    a leftover is paired with itself).  Yields (qs, ts, w, zdrop, end_bonus, flag, (G, C), rebased, nomax)."""
    for geom in ((64, 16), (8, 18), (16, 8)):
        for rebased in (False, True):
            for nomax in (False, True):
                qs, ts, ws = [], [], []
                for k in range(5):
                    ql = int(rng.integers(5000, 9000) * scale) if rebased else int(rng.integers(600, 2600) if geom[0] == 64 else rng.integers(30, 1500))
                    tl = max(520 if geom[0] == 64 else 1, ql + int(rng.integers(-40, 40)))
                    w = int(rng.choice([285, 300, 400, 500, 536] if geom[0] == 64 else [0, 1, 7, 20, 40, 64, 67] if geom[0] == 8 else [68]))
                    q, t = synth.fixed_batch(5100 + 10 * rebased + k, 2 + k % 2, ql, tl, sub=0.05, ind=0.08 if geom[0] == 64 else 0.02, tail_random_frac=0.3, tail_pairs=0.4)
                    qs += list(q); ts += list(t); ws += [w] * len(q)
                n = len(qs)
                zd = rng.choice([-1, 100, 400, 2000], size=n)
                eb = rng.choice([0, 50], size=n)
                fl = np.array([po.SCORE_ONLY | (0x08 if nomax else 0) | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) for _ in range(n)])
                yield qs, ts, np.array(ws), zd, eb, fl, geom, rebased, nomax


def _check_code_plane_forms(lib, form, scale=1.0, seed=31):
    """All twelve instantiations behind g_fill_pk_ldscodes ((64, 16) / (8, 18) / (16, 8) x plain / re-based x exact / no-maximum) or their
    register twins, whichever `form` the caller forced, against the oracle; the plan must report exactly that form."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mat, q, e = synth.simple_mat(5, 2, 4, -1), 4, 2
    seen = set()
    for qs, ts, w, zd, eb, fl, geom, rebased, nomax in _code_plane_batches(rng, scale):
        p = lib.make_batch(qs, ts, mat, q, e, 0, 0, w=w, zdrop=zd, end_bonus=eb, flag=fl).plan(False)
        for c in p.describe():
            if c["kernel"] == "pk" and (c["G"], c["C"]) == geom and c["mode"] == "score":
                assert c["form"] == form, c
                seen.add((c["G"], c["rebased"], c["nomax"]))
        p.close()
        check_batch(lib, False, qs, ts, mat, q, e, 0, 0, w=w, zdrop=zd, end_bonus=eb, flag=fl)
    assert seen == {(g, r, m) for g in (64, 8, 16) for r in (0, 1) for m in (0, 1)}, seen


@pytest.mark.parametrize("ldc", ["0", "1"])
def test_sim_code_planes_in_registers_and_in_lds(sim, ldc, monkeypatch):
    """KSW2AMD_LDSCODES=0 / 1: the exact and no-maximum score-only kernels of the (64, 16) and (8, 18) geometries with the target-code
    planes in registers / in LDS (the forms the headline benchmark and config 2 run), plain and re-based."""
    monkeypatch.setenv("KSW2AMD_LDSCODES", ldc)
    monkeypatch.setenv("KSW2AMD_DEFER", "0")              # (the deferred arg-max kernels have their own forms: test below)
    _check_code_plane_forms(sim, "ldscodes" if ldc == "1" else "registers", scale=0.6)


def _check_deferred_argmax(lib, on, scale=1.0, seed=41):
    """KSW2AMD_DEFER=1 / 0 (set by the caller): exact score-only single-gap batches of all four packed geometries, plain and re-based,
    through the kernels that track row maxima without columns + the arg-max recovery pass, or through the ordinary ones; Z-drop
    thresholds low enough that some alignments are handed back as inexact and re-run.  Every pair against the oracle."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mat, q, e = synth.simple_mat(5, 2, 4, -1), 4, 2
    seen, r0, ndrop = set(), lib.rerun_count(), 0
    for geom, wset, lens in (((8, 18), [0, 1, 7, 40, 64, 67], (30, 1500)), ((16, 8), [68], (100, 1500)), ((64, 8), [69, 100, 284], (520, 2500)),
                             ((64, 16), [285, 400, 536], (600, 2600))):
        for rebased in (False, True):
            qs, ts, ws = [], [], []
            for k in range(5):
                ql = int(rng.integers(5000, 8000) * scale) if rebased else int(rng.integers(*lens))
                tl = max(520 if geom[0] == 64 else 1, ql + int(rng.integers(-40, 40)))
                qq, tt = synth.fixed_batch(7100 + 10 * rebased + k, 2 + k % 2, ql, tl, sub=0.05, ind=0.05, tail_random_frac=0.3, tail_pairs=0.4)
                qs += list(qq); ts += list(tt); ws += [int(rng.choice(wset))] * len(qq)
            n = len(qs)
            zd = rng.choice([-1, 60, 400, 2000], size=n)
            eb = rng.choice([0, 50], size=n)
            fl = np.array([po.SCORE_ONLY | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) for _ in range(n)])
            p = lib.make_batch(qs, ts, mat, q, e, 0, 0, w=np.array(ws), zdrop=zd, end_bonus=eb, flag=fl).plan(False)
            for c in p.describe():
                if c["kernel"] == "pk" and (c["G"], c["C"]) == geom:
                    assert (c["form"] == "defer") == on, c
                    seen.add((c["G"], c["rebased"]))
            p.run(); raw = p.fetch_raw().copy(); p.close()
            _, res = check_batch(lib, False, qs, ts, mat, q, e, 0, 0, w=np.array(ws), zdrop=zd, end_bonus=eb, flag=fl)
            assert all(raw[i, 2] == res[i]["max_q"] and raw[i, 7] == res[i]["mte_q"] and raw[i, 8] == res[i]["score"] for i in range(n))   # resident plan == batch entry
            ndrop += sum(1 for r in res if r["zdropped"])
    assert seen == {(g, r) for g in (8, 16, 64) for r in (0, 1)}, seen
    # some alignments dropped.  Rounds 3-4 handed every alignment whose book the deferred fill froze back to the host; the third pass
    # (k2a_zscan_kernel, round 5) settles them all on the device: nothing is run again
    assert ndrop >= 20 and lib.rerun_count() == r0, (ndrop, lib.rerun_count() - r0)


def _check_frozen_books_with_wildcards(lib, monkeypatch):
    """Round 5's fuzz find: an unscanned (streamed / flat) plan whose deferred fill froze a book AND met a wildcard code in the same
    pair's target -- the third pass rewrote the record and with it the fill's wildcard report, and the pair came back mis-scored
    instead of being re-run.  Diverging tails (every pair freezes at Z-drop 60 / 200), wildcards in every third target and in
    some queries, through the streamed pointer entry and the flat entry, against the oracle."""
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")
    monkeypatch.setenv("KSW2AMD_DEFER", "1")
    monkeypatch.setenv("KSW2AMD_STREAM", "1")
    monkeypatch.setenv("KSW2AMD_STREAM_PIECE_KB", "64")
    mat = synth.simple_mat(5, 2, 4, -3)
    for ql, tl, w, zd in ((900, 880, 20, 200), (1500, 1500, 300, 60)):
        qs, ts = synth.fixed_batch(4242 + w, 24, ql, tl, sub=0.05, ind=0.05, tail_random_frac=0.4, tail_pairs=1.0)
        qs, ts = [np.array(x) for x in qs], [np.array(x) for x in ts]
        for i in range(0, 24, 3):
            ts[i][tl // 4] = 4
        for i in range(1, 24, 5):
            qs[i][ql // 3] = 4
        for tn in ("0", None):                                  # KSW2AMD_TN=0: target wildcards are handed back (before round 6, and still for matrices whose wildcard row varies);
            if tn is None:                                          # default: they are rows of the packed kernels like any other (K2aScoring.pk_tn1) -- nothing is re-run
                monkeypatch.delenv("KSW2AMD_TN", raising=False)
            else:
                monkeypatch.setenv("KSW2AMD_TN", tn)
            r0 = lib.rerun_count()
            res = lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, flag=po.SCORE_ONLY)
            fres = lib.make_flat_batch(qs, ts, mat, 4, 2, 0, 0, w=w, zdrop=zd, end_bonus=0, flag=po.SCORE_ONLY).run_oneshot(False)
            if tn == "0":
                assert lib.rerun_count() >= r0 + 16                 # the eight target wildcards, twice; nothing else
            else:
                assert lib.rerun_count() == r0
            ndrop = 0
            for i in range(24):
                exp = po.align("oracle", "extz2", qs[i], ts[i], mat, 4, 2, w=w, zdrop=zd, flag=po.SCORE_ONLY)
                assert not diff(exp, res[i], CMP) and not diff(exp, fres[i], CMP), (w, tn, i, diff(exp, res[i], CMP), diff(exp, fres[i], CMP))
                ndrop += exp["zdropped"]
            assert ndrop >= 12, ndrop


def test_sim_frozen_books_with_wildcards(sim, monkeypatch):
    _check_frozen_books_with_wildcards(sim, monkeypatch)


@pytest.mark.parametrize("defer", ["0", "1"])
def test_sim_deferred_argmax(sim, defer, monkeypatch):
    """K2aLanePk DEFER + k2a_argmax twin on the simulator."""
    monkeypatch.setenv("KSW2AMD_DEFER", defer)
    _check_deferred_argmax(sim, defer == "1", scale=0.6)


def _flat_cases(rng):
    """Batches for the flat entry points: ragged and one-shape, score-only / CIGAR / EQX, both gap models, every band class, and in
    each batch a few pairs with a wildcard code somewhere -- in the query, in the target, in the last byte -- which the packed
    kernels must report so that the host re-runs them (ksw2_host_plan.c::pair_rerun)."""
    mat = synth.simple_mat(5, 2, 4, -1)
    for rnd in range(6):
        n = 36
        if rnd % 2:
            q, t = synth.fixed_batch(6100 + rnd, n, [300, 700, 2500][rnd // 2], [310, 690, 2480][rnd // 2], sub=0.05, ind=0.08, tail_random_frac=0.3, tail_pairs=0.3)
            qs, ts = [x.copy() for x in q], [x.copy() for x in t]
        else:
            pairs = synth.ragged_pairs(rng, n, 1, [150, 900, 3000][rnd // 2], sub=0.05, ind=0.12)
            qs, ts = [p[0].copy() for p in pairs], [p[1].copy() for p in pairs]
        for i in range(2, n, 9):
            (qs if i % 2 else ts)[i][int(rng.integers(len((qs if i % 2 else ts)[i])))] = 4
        qs[5][-1] = 4
        ts[6][-1] = 4
        w = rng.choice([-1, 5, 20, 64, 68, 100, 284, 285, 500, 537, 1041], size=n)
        zd = rng.choice([-1, 100, 400], size=n)
        eb = rng.choice([0, 30], size=n)
        for dual in (False, True):
            mode = [po.SCORE_ONLY, 0, po.RIGHT][(rnd + dual) % 3]
            fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.EQX if dual and mode != po.SCORE_ONLY and rng.random() < 0.3 else 0) |
                           (0x08 if rng.random() < 0.15 else 0) for _ in range(n)])
            yield dual, qs, ts, mat, w, zd, eb, fl


def _check_flat(lib, device_copy=None):
    """ksw2amd_ext?_batch_flat and ksw2amd_plan_create_flat against the ordinary entry points on the same pairs (every field, CIGAR
    included) and against the oracle; device_copy(arena) -> device address runs the same through a device-resident arena."""
    rng = np.random.Generator(np.random.PCG64(2027))
    tot = 0
    kinds = set()
    for dual, qs, ts, mat, w, zd, eb, fl in _flat_cases(rng):
        ref = (lib.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=eb, flag=fl) if dual else
               lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, end_bonus=eb, flag=fl))
        fb = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=eb, flag=fl)
        got = [fb.run_oneshot(dual)]
        p = fb.plan(dual)
        kinds.update(d["kernel"] for d in p.describe())
        p.run(); got.append(p.fetch()); p.close()
        if device_copy is not None:
            keep = device_copy(fb.arena)
            fd = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=eb, flag=fl, device_base=keep[0])
            got.append(fd.run_oneshot(dual))
            p = fd.plan(dual); p.run(); got.append(p.fetch()); p.close()
        for i in range(len(qs)):
            for g in got:
                assert not diff(ref[i], g[i]), (i, dual, int(w[i]), hex(int(fl[i])), diff(ref[i], g[i]))
            if i % 5 == 0 and not (fl[i] & po.EQX):
                exp = po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, 4, 2, 24, 1, w=int(w[i]), zdrop=int(zd[i]), end_bonus=int(eb[i]), flag=int(fl[i]))
                assert not diff(exp, ref[i]), (i, dual)
            tot += 1
    assert {"pk", "solo", "int32"} <= kinds, kinds         # unscanned arenas reach the packed AND the solo kernels (both report wildcard codes)
    # wide alphabets through the flat entry: the effective matrices travel in the arena's tail block (page-locked staging since round 4)
    for rnd in range(3):
        m, wmat, qs, ts, w, zd, fl = _wide_alphabet_cases(rng, rnd)
        for dual in (False, True):
            ref = (lib.extd_batch(qs, ts, wmat, 6, 2, 20, 1, w=w, zdrop=zd, flag=fl, m=m) if dual else lib.extz_batch(qs, ts, wmat, 6, 2, w=w, zdrop=zd, flag=fl, m=m))
            fb = lib.make_flat_batch(qs, ts, wmat, 6, 2, 20, 1, w=w, zdrop=zd, flag=fl, m=m)
            got = fb.run_oneshot(dual)
            for i in range(len(qs)):
                assert not diff(ref[i], got[i]), ("flat, wide alphabet", m, i, dual, hex(int(fl[i])), diff(ref[i], got[i]))
                if i % 3 == 0:
                    exp = po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], wmat, 6, 2, 20, 1, w=int(w[i]), zdrop=int(zd[i]), flag=int(fl[i]), m=m)
                    assert not diff(exp, got[i]), ("flat vs oracle, wide alphabet", m, i, dual)
    if device_copy is not None:
        # a DEVICE arena whose pairs ask for the SSE kernels' own results (what a sharded run with those flags hands every receiving
        # rank, ksw2_amd/parallel.py): the span comes back to the host and takes the SSE-compatible plans -- same results as the pointer entry
        pairs = synth.ragged_pairs(rng, 24, 30, 300, sub=0.08, ind=0.15)
        qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
        mat = synth.simple_mat(5, 2, 4, -1)
        w = rng.choice([-1, 3, 9, 33, 100], size=24)
        fl = np.array([int(rng.choice([ka.KSW2AMD_EZ_SSE_COMPAT, po.APPROX_MAX | po.APPROX_DROP, 0])) | int(rng.choice([0, po.SCORE_ONLY])) for _ in range(24)])
        for dual in (False, True):
            ref = (lib.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=60, end_bonus=5, flag=fl) if dual else lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=60, end_bonus=5, flag=fl))
            fb = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=60, end_bonus=5, flag=fl)
            keep = device_copy(fb.arena)
            fd = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=60, end_bonus=5, flag=fl, device_base=keep[0])
            got = fd.run_oneshot(dual)
            for i in range(24):
                assert not diff(ref[i], got[i]), ("device arena + SSE-compatible flags", i, dual, hex(int(fl[i])), diff(ref[i], got[i]))
    return tot


def test_sim_flat_batches(sim):
    """One arena + offsets instead of 2 n pointers (include/ksw2_amd.h, "Flat batches"): same results as the pointer entry points,
    wildcard pairs included -- the flat path uploads the arena unscanned, the packed kernels report codes >= 4, the host re-runs."""
    held = []

    def device_copy(arena):                      # the simulator's "device memory" is host memory: exercises the on_device code path
        held.append(sim.device_copy(arena))
        return held[-1], None

    try:
        assert _check_flat(sim, device_copy) == 6 * 2 * 36
    finally:
        for d in held:
            sim.device_free(d)


def test_sim_pairs_share_the_true_target_length(sim):
    """Found by tools/scripts/fuzz_gpu.py: two alignments with the same query length, band and rows inside the band but
    different true target lengths (one target cut off by the band, so it has no last row: mte / score stay unset) must not
    share a packed task."""
    rng = np.random.Generator(np.random.PCG64(77))
    mat = synth.simple_mat(5, 2, 4, -1)
    q = rng.integers(0, 4, 519).astype(np.uint8)
    t = np.concatenate([q, rng.integers(0, 4, 10).astype(np.uint8)])
    qs, ts = [q, q, q, q], [t[:524], t[:529], t[:524], t[:531]]
    for dual in (False, True):
        for flag in (po.SCORE_ONLY, 0):
            check_batch(sim, dual, qs, ts, mat, 4, 2, 13, 1, w=5, zdrop=-1, end_bonus=10, flag=flag)


def test_sim_eqx_golden_subset(sim):
    """Every 4th committed KSW_EZ_EQX case of the reference through the simulator build: single calls and one batch."""
    ec = gu.EqxCases()
    cs = [ec.case(k) for k in range(0, ec.n, 4)]
    for c in cs:
        r = sim.extd2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=-1, end_bonus=c["end_bonus"], flag=c["flag"])
        bad, _ = gu.EqxCases.check(c, r)
        assert not bad, (bad, c["flag"])
    g = [c for c in cs if (c["gq"], c["ge2"]) == (4, 1) and c["mat"][0] == 2 and c["mat"][24] == -1]
    assert len(g) >= 20
    res = sim.extd_batch([c["q"] for c in g], [c["t"] for c in g], g[0]["mat"], 4, 2, 24, 1, w=np.array([c["w"] for c in g]), zdrop=-1,
                         end_bonus=np.array([c["end_bonus"] for c in g]), flag=np.array([c["flag"] for c in g]))
    for c, r in zip(g, res):
        bad, _ = gu.EqxCases.check(c, r)
        assert not bad, (bad, c["flag"])


def test_sim_fuzz_script_runs(sim):
    """The soak script itself (tools/scripts/fuzz_gpu.py, a -m gpu test on the box) for a few seconds against the simulator build."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("KSW2AMD_")}
    env["KSW2AMD_FUZZ_LIB"] = os.path.join(SIM_DIR, "libksw2_amd_sim.so")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scripts", "fuzz_gpu.py"), "8", "7"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_sim_sse_compatible_mode(sim):
    """ksw_extz2_sse / ksw_extd2_sse as the reference's SSE kernels return them (opt-in; ksw2_lane_ssec.h): a third of the golden
    cases of the unmodified reference, the routing of mixed batches and of the process-wide switch, longer banded reads."""
    from tests import sse_compat_util as su
    assert su.check_golden(sim, step=3) >= 500
    su.check_routing(sim)
    su.check_long(sim, n=3, length=1500, w=60)


def test_sim_sse_compatible_register_form(sim, monkeypatch):
    """Score-only SSE-compatible tasks through the simulator twin of k2a_ssec_blk_kernel (one 16-position block of the reference's
    arrays per lane, tests/sse_compat_util.check_register_form), and the full golden set once more with the form off."""
    from tests import sse_compat_util as su
    assert su.check_register_form(sim, monkeypatch.setenv, rounds=6, long_len=2200) > 40
    monkeypatch.setenv("KSW2AMD_SSEC_BLK", "0")
    assert su.check_golden(sim, step=2) >= 750


def test_sim_packed_generation_serial(sim, monkeypatch):
    """Same-shape pairs whose band no resident geometry holds take the packed generation-serial class (ksw2_lane_pkmp.h: sliding
    score base, row maxima merged as keys, boundary entries between generations): unbanded and wide bands, one to six
    generations, both gap models, Z-drop; and the int32 generation-serial kernels still serve when it is switched off."""
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")
    mat = synth.simple_mat(5, 2, 4, -1)
    cases = [(900, -1, False, 0, -1), (2100, -1, False, po.RIGHT, -1), (2600, -1, False, po.SCORE_ONLY, -1), (2200, 1100, True, 0, -1),
             (2300, -1, True, po.RIGHT, 300), (3300, -1, False, po.EXTZ_ONLY, 100), (5200, 1200, True, 0, 400)]
    for L, w, dual, flag, zd in cases:
        q, t = synth.fixed_batch(9, 2, L, L + 37, sub=0.05, ind=0.08, tail_random_frac=0.3 if zd >= 0 else 0.0, tail_pairs=0.5 if zd >= 0 else 0.0)
        qs, ts = [q[0], q[0], q[1], q[1]], [t[0], t[0], t[1], t[1]]
        p = sim.make_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag).plan(dual)
        assert p.packed_pairs() == 4          # 900 rows: all strips resident in the packed (64, 16) array; beyond 2 048 rows: this class
        p.close()
        check_batch(sim, dual, qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag)
    # rows of 9 000 columns: a row's maximum drifts more than 12 000 units above the sliding base and is merged into its key mid-row
    q, t = synth.fixed_batch(12, 1, 9000, 8800, sub=0.05, ind=0.08)
    check_batch(sim, False, [q[0], q[0]], [t[0], t[0]], mat, 4, 2, 24, 1, w=-1, zdrop=-1, flag=po.SCORE_ONLY)
    monkeypatch.setenv("KSW2AMD_NO_PKMP", "1")
    q, t = synth.fixed_batch(9, 2, 2300, 2337, sub=0.05, ind=0.08)
    p = sim.make_batch(q, t, mat, 4, 2, 24, 1, w=-1, zdrop=-1, flag=0).plan(False)
    assert p.packed_pairs() == 0
    p.close()


def _extf_ring_rows(c):
    """K2A_EXTF_RING_ROWS (ksw2_types.h) of a case: rows of four positions around the band"""
    w = c["w"] if c["w"] >= 0 else max(len(c["q"]), len(c["t"]))
    return (min(min(len(c["q"]), len(c["t"])) - 1, w) + 30) // 4 + 3


def _check_extf_lane_forms(lib, cases, monkeypatch):
    """Every reference case through the lane form, per scoring: the narrow-band cases as batches that fit LDS rings of up to 20, 32,
    48 and 64 rows (plan diagnostics say which form ran and with how many rows), everything again with the state arrays in HBM
    scratch (KSW2AMD_EXTF_RING=0)."""
    seen = set()
    for sc in sorted({(c["mch"], c["mis"], c["e"]) for c in cases}):
        sub = [c for c in cases if (c["mch"], c["mis"], c["e"]) == sc and len(c["q"]) and len(c["t"])]
        def rows_of(part):
            return max(16, (max(_extf_ring_rows(c) for c in part) + 3) & ~3)
        parts = [[c for c in sub if _extf_ring_rows(c) <= 20], [c for c in sub if 20 < _extf_ring_rows(c) <= 32], [c for c in sub if 32 < _extf_ring_rows(c) <= 48],
                 [c for c in sub if 48 < _extf_ring_rows(c) <= 64]]
        parts = [("ldsring", rows_of(x), x) for x in parts if x] + [("hbm", 0, [c for c in sub if _extf_ring_rows(c) > 64]), ("hbm", 0, sub)]
        for form, ring, part in parts:
            if not part:
                continue
            monkeypatch.setenv("KSW2AMD_EXTF_RING", "0" if form == "hbm" and part is sub else "1")
            b = lib.make_linear_batch([c["q"] for c in part], [c["t"] for c in part], *sc, w=[c["w"] for c in part], xdrop=[c["xdrop"] for c in part])
            p = b.plan()
            d = p.describe()
            assert len(d) == 1 and d[0]["kernel"] == "extf-lane" and d[0]["form"] == form and d[0]["ring"] == ring, (d, form, ring)
            p.run()
            res = p.fetch()
            p.close()
            monkeypatch.delenv("KSW2AMD_EXTF_RING", raising=False)
            seen.add((form, ring))
            for r, c in zip(res, part):
                assert not diff(r, c["expect"], gu.FIELDS), (sc, form, ring, len(c["q"]), len(c["t"]), c["w"], c["xdrop"])
    assert ("hbm", 0) in seen and len({r for f, r in seen if f == "ldsring"}) >= 3, seen


def test_sim_linear_xdrop_one_extension_per_lane(sim, monkeypatch):
    """The lane-per-extension form of ksw_extf2_sse (k2a_extf_lane_kernel: interleaved sequences and state, four positions per dword):
    every reference case in batches of mixed shapes, so groups of 64 hold very different lengths and bands."""
    monkeypatch.setenv("KSW2AMD_EXTF_LANE", "1")
    fc = gu.ExtfCases()
    cases = [fc.case(k) for k in range(fc.n)]
    _check_extf_lane_forms(sim, cases, monkeypatch)
    e = np.zeros(0, np.uint8); one = np.array([2], np.uint8)
    res = sim.extf_batch([e, one, one, e], [e, e, one, one], 2, -4, 2, w=-1, xdrop=50)
    for r, (q, t) in zip(res, ((e, e), (one, e), (one, one), (e, one))):
        assert not diff(r, po.extf2("oracle", q, t, 2, -4, 2, -1, 50), gu.FIELDS)


def test_sim_gg_family_golden(sim):
    """Every third committed case of the global family (tests/golden/gg_cases.npz: reference outputs of ksw_gg / ksw_gg2 / ksw_gg2_sse,
    bands -1 ... 500 and the band-cannot-reach-the-corner definition) through the product's ksw_gg / ksw_gg2 / ksw_gg2_sse."""
    gc = gu.GgCases()
    n = 0
    for k in gc.contract_cases()[::3]:
        c = gc.case(k)
        s, cg = sim.gg(c["func"], c["q"], c["t"], c["mat"], c["gq"], c["ge"], w=c["w"], with_cigar=c["with_cigar"])
        assert s == c["score"] and list(cg) == c["cigar"], (k, c["func"], c["w"], c["origin"], s, c["score"])
        n += 1
    assert n >= 250


def _one_shape_batch(seed, n, ql, tl, wild_at=()):
    q, t = synth.fixed_batch(seed, n, ql, tl, sub=0.05, ind=0.06)
    q, t = [np.array(x) for x in q], [np.array(x) for x in t]
    for k, i in enumerate(wild_at):                      # wildcards, alternately in the target and in the query
        if k % 2 == 0:
            t[i][len(t[i]) // 2] = 4                     # in the target: a streamed plan's arena is not scanned, the kernel reports it, fetch re-runs the pair
        else:
            q[i][len(q[i]) // 2] = 4                     # in the query: entry 4 of the packed kernels' column-profile table, scored in place
    return q, t


@pytest.mark.parametrize("flat", [False, True])
def test_sim_streamed_plans(sim, monkeypatch, flat):
    """Streamed plans (ksw2_host_plan.c "streamed plans") forced on for every plan that can: one-shape batches through
    the batch entry points, the arena in small pieces, the packed classes as queue launches (K2aQueueDesc) -- against the oracle on
    every pair: score-only and CIGAR classes, both gap models, an odd pair count, wildcard pairs (re-run in one batch), Z-drops with
    the deferred arg-max (inexact pairs re-run), and the fault hook (the last watermark never arrives: the launch gives up, the plan
    runs again behind its upload)."""
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")
    monkeypatch.setenv("KSW2AMD_STREAM", "1")
    monkeypatch.setenv("KSW2AMD_STREAM_PIECE_KB", "64")
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    cases = [(1501, 400, 420, 64, po.SCORE_ONLY, False, 30, (7, 800, 1500)), (1200, 500, 500, 100, 0, True, 100, ()),
             (1400, 700, 700, 300, po.SCORE_ONLY, False, 60, (3,)), (2400, 300, 330, 64, po.SCORE_ONLY | po.APPROX_MAX, False, -1, ())]
    for fault in (0, 1):
        monkeypatch.setenv("KSW2AMD_STREAM_FAULT", str(fault))
        for ci, (n, ql, tl, w, flag, dual, zd, wild) in enumerate(cases[:2] if fault else cases):
            qs, ts = _one_shape_batch(100 + ci, n, ql, tl, wild)
            for tn in (("0", None) if (wild and ci == 0 and not fault) else (None,)):      # (the old rule once: the first case without the fault hook)          # target wildcards handed back (KSW2AMD_TN=0) / scored in place (default, round 6)
                if tn is None:
                    monkeypatch.delenv("KSW2AMD_TN", raising=False)
                else:
                    monkeypatch.setenv("KSW2AMD_TN", tn)
                s0, r0 = sim.stream_stats(), sim.rerun_count()
                if flat:
                    fb = sim.make_flat_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=0, flag=flag)
                    res = fb.run_oneshot(dual)
                else:
                    res = (sim.extd_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, flag=flag) if dual else sim.extz_batch(qs, ts, mat, q, e, w=w, zdrop=zd, flag=flag))
                s1 = sim.stream_stats()
                streams = bool(flag & po.SCORE_ONLY)          # the queue builds of the kernels exist for the score-only classes
                assert (s1["streamed_plans"] > s0["streamed_plans"]) == streams, (ci, flat)
                assert (s1["aborted_runs"] > s0["aborted_runs"]) == bool(fault and streams), (ci, flat, fault)
                if wild and tn == "0":
                    assert sim.rerun_count() >= r0 + (len(wild) + 1) // 2      # the target wildcards (every other one) are handed back and re-run
                elif wild:
                    assert sim.rerun_count() == r0                             # ... or stay where they are
                for i in range(n):
                    exp = (ApproxLike.run(qs[i], ts[i], mat, w, flag) if flag & po.APPROX_MAX else
                           po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=0, flag=flag))
                    d = diff(exp, res[i], ["score"] if flag & po.APPROX_MAX else CMP)
                    assert not d, (ci, flat, fault, tn, i, d)


def _check_uniform_plans(lib, monkeypatch, shapes):
    """Uniform plans (ksw2_host_plan.c "uniform batches", round 5): a score-only batch of one shape and one set of parameters handed to
    the batch entry point -- the class from a two-pair probe, the records / task list / piece counts written on the device by rule
    (k2a_uniform_layout_kernel), one streamed launch, the host's per-pair arrays filled by the copy's workers.  Forced off
    (KSW2AMD_UNIFORM=0: the general path) against the default on every pair, both against the oracle on a sample; the plain and the
    deferred arg-max class, Z-drops (frozen books), wildcards in queries (scored in place) and in targets (handed back and re-run
    from the uniform plan's own host arrays), small pieces with a slowed-down upload, and the fault hook (the launch gives up, the
    plan is repeated unstreamed from the same device-built records)."""
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")
    mat = synth.simple_mat(5, 2, 4, -1)
    for si, (n, ql, tl, w, zd, defer) in enumerate(shapes):
        qs, ts = synth.fixed_batch(9100 + si, n, ql, tl, sub=0.05, ind=0.06, tail_random_frac=0.3, tail_pairs=0.2)
        qs, ts = [np.array(x) for x in qs], [np.array(x) for x in ts]
        qs[5][ql // 2] = 4; ts[n // 2][tl // 3] = 4; ts[n - 1][0] = 4
        fl = po.SCORE_ONLY | (po.EXTZ_ONLY if si % 2 else 0)

        def run(**env):
            for k in ("KSW2AMD_UNIFORM", "KSW2AMD_STREAM_PIECE_KB", "KSW2AMD_STREAM_SLEEP_US", "KSW2AMD_STREAM_FAULT", "KSW2AMD_STREAM_TIMEOUT_MS", "KSW2AMD_DEFER", "KSW2AMD_WIRE4", "KSW2AMD_WIRE2", "KSW2AMD_TN"):
                monkeypatch.delenv(k, raising=False)
            if defer is not None:
                monkeypatch.setenv("KSW2AMD_DEFER", str(defer))
            for k, v in env.items():
                monkeypatch.setenv(k, str(v))
            s0, r0 = lib.stream_stats(), lib.rerun_count()
            res = lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, end_bonus=7, flag=fl)
            s1 = lib.stream_stats()
            return res, s1["streamed_plans"] - s0["streamed_plans"], s1["aborted_runs"] - s0["aborted_runs"], lib.rerun_count() - r0

        off, ns, na, nr = run(KSW2AMD_UNIFORM=0)
        decoy = ([np.random.default_rng(3 + si).integers(0, 4, ql, dtype=np.uint8) for _ in range(n)], [np.random.default_rng(5 + si).integers(0, 4, tl, dtype=np.uint8) for _ in range(n)])
        keep = (qs, ts)
        qs, ts = decoy
        run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64)     # a decoy batch of the same shape through the same buffers first
        qs, ts = keep
        on, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_STREAM_SLEEP_US=100)
        assert ns == 1 and na == 0 and nr == 0, (si, ns, na, nr)       # ONE streamed plan; the target wildcards are rows of the packed kernel (round 6: K2aScoring.pk_tn1)
        old, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_TN=0)
        assert ns == 1 and na == 0 and nr == 4, (si, ns, na, nr)       # KSW2AMD_TN=0: the two tasks that hold one are handed back (both pairs of a task) ...
        assert not [i for i in range(n) if diff(old[i], on[i])], (si, "tn off")      # ... and come back with the same records
        b = lib.make_batch(qs, ts, mat, 4, 2, 0, 0, w=w, zdrop=zd, end_bonus=7, flag=fl)
        bad = [i for i in range(n) if diff(off[i], on[i])]
        assert not bad, (si, bad[:5], off[bad[0]], on[bad[0]])
        # the 4-bit wire format (two codes per byte in staging and upload, expanded by the wavefront that needs them) off: the same results;
        # a residue code above 15 does not fit it: the batch is repeated on the general path and comes back as the general path returns it
        w8, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_WIRE4=0)
        assert ns == 1 and na == 0 and not [i for i in range(n) if diff(off[i], w8[i])], (si, "wire4 off", ns, na)
        # ... and the 2-bit format (round 6, opt-in: four codes per byte, codes above 3 as escape entries in the pair's padding)
        w2, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_WIRE2=1)
        assert ns == 1 and na == 0 and not [i for i in range(n) if diff(off[i], w2[i])], (si, "wire2 on", ns, na)
        if si == 1:
            # escapes: runs of wildcards (one entry each), runs longer than an entry holds (255), the last slot, both sequences of a pair; then more
            # runs in a pair than its seven entries (the batch takes the general path: same records)
            keepq, keept = qs, ts
            qs, ts = [x.copy() for x in qs], [x.copy() for x in ts]
            ts[3][5:60] = 4; qs[3][0] = 4; qs[3][ql - 1] = 4; ts[3][tl - 1] = 4
            for k in range(7):
                ts[9][3 + 9 * k] = 4
            qs[11][10:ql - 5] = 4
            e_off, _, _, _ = run(KSW2AMD_UNIFORM=0)
            e_on, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_WIRE2=1)
            assert ns == 1 and not [i for i in range(n) if diff(e_off[i], e_on[i])], (si, "escapes")
            e_flt, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_WIRE2=1, KSW2AMD_STREAM_FAULT=1, KSW2AMD_STREAM_TIMEOUT_MS=20)      # the whole-arena expansion of a repeated run
            assert na == 1 and not [i for i in range(n) if diff(e_off[i], e_flt[i])], (si, "escapes, repeated run")
            e_tn, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_WIRE2=1, KSW2AMD_TN=0)                            # re-runs read the pairs back out of the staging copy (wire4_pair)
            assert nr > 0 and not [i for i in range(n) if diff(e_off[i], e_tn[i])], (si, "escapes, handed back")
            for k in range(8):
                qs[20][4 + 11 * k] = 4
            o_off, _, _, _ = run(KSW2AMD_UNIFORM=0)
            o_on, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_WIRE2=1)
            assert not [i for i in range(n) if diff(o_off[i], o_on[i])], (si, "more escapes than a pair's slot holds")
            for i in (3, 9, 11, 20):
                exp = po.align("oracle", "extz2", qs[i], ts[i], mat, 4, 2, w=w, zdrop=zd, end_bonus=7, flag=fl)
                assert not diff(exp, o_on[i], CMP), (si, "escapes vs oracle", i)
            qs, ts = keepq, keept
        if si == 0:
            keepq = qs[7].copy()
            qs[7][3] = 20
            fat_off, _, _, _ = run(KSW2AMD_UNIFORM=0)
            fat_on, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64)
            assert not [i for i in range(n) if diff(fat_off[i], fat_on[i])], (si, "code above 15")
            qs[7] = keepq
        qs, ts = decoy
        run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64)         # (the arena holds the decoy's bases again: what the repeated run reads must have been expanded by IT)
        qs, ts = keep
        flt, ns, na, nr = run(KSW2AMD_UNIFORM=1, KSW2AMD_STREAM_PIECE_KB=64, KSW2AMD_STREAM_FAULT=1, KSW2AMD_STREAM_TIMEOUT_MS=20)
        assert ns == 1 and na == 1, (si, ns, na)
        bad = [i for i in range(n) if diff(off[i], flt[i])]
        assert not bad, (si, "fault", bad[:5])
        for i in list(range(0, n, max(1, n // 24))) + [5, n // 2, n - 1]:
            exp = po.align("oracle", "extz2", qs[i], ts[i], mat, 4, 2, w=w, zdrop=zd, end_bonus=7, flag=fl)
            assert not diff(exp, on[i], CMP), (si, i, diff(exp, on[i], CMP))
        del b


def test_sim_uniform_plans(sim, monkeypatch):
    _check_uniform_plans(sim, monkeypatch, [(2048, 60, 64, 10, -1, None), (2050, 150, 140, 30, 40, 1), (2048, 90, 90, 70, 100, 0)])


class ApproxLike:
    """KSW_EZ_APPROX_MAX returns only the score (and the corner CIGAR): the exact computation's score without Z-drop."""
    @staticmethod
    def run(qq, tt, mat, w, flag):
        return po.align("oracle", "extz2", qq, tt, mat, 4, 2, w=w, zdrop=-1, end_bonus=0, flag=flag & ~po.APPROX_MAX)


def test_sim_target_wildcards_stay_packed(sim, monkeypatch):
    from tests.parity_util import check_target_wildcards
    check_target_wildcards(sim, monkeypatch.setenv, monkeypatch.delenv)
