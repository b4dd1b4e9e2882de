"""GPU: the HIP path through the C-ABI (libksw2_amd.so) against the oracle and the golden vectors. Bit-exact."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import ksw2_amd as ka
from ksw2_amd import synth
from oracle import pyoracle as po
from tests import golden_util as gu
from tests.parity_util import check_batch, cigar_score, diff, CMP_FIELDS

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _small_batches_stay_packed(monkeypatch):
    """Test batches are far too small to fill 1024 SIMDs; without this the host would send their packed-int16 candidates
    back to the int32 kernels (ksw2_host_plan.c, "0.4 wavefronts per SIMD") and the packed kernels would go untested."""
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")


@pytest.fixture(scope="module")
def lib():
    L = ka.library()                      # raises if the HIP library is missing: no fallback
    assert L.backend() == "hip:gfx950"
    assert L.device_count() >= 1
    return L


def test_t1q1_golden(lib):
    ka_ = gu.known_answers()
    _, ts = gu.read_fasta("t1.fa")
    _, qs = gu.read_fasta("q1.fa")
    mat = gu.simple_mat(5, 2, 4, 0)
    for k, rec in enumerate(ka_["t1q1"]):
        for flag in (0, po.RIGHT):
            exp = rec["ksw_extz/flag=%d" % flag]
            res = lib.extz(qs[k], ts[k], mat, 4, 2, flag=flag)
            for f in gu.FIELDS:
                assert res[f] == exp[f], (k, f)
            assert gu.cigar_string(res["cigar"]) == exp["cigar"]
            exp = rec["ksw_extd/flag=%d" % flag]
            res = lib.extd(qs[k], ts[k], mat, 4, 2, 13, 1, flag=flag)
            for f in gu.FIELDS:
                assert res[f] == exp[f], (k, f)
            assert gu.cigar_string(res["cigar"]) == exp["cigar"]
        for g in ("gg", "gg2", "gg2_sse"):
            s, c = lib.gg(g, qs[k], ts[k], mat, 4, 2, w=-1)
            assert s == rec["ksw_gg"]["score"] and gu.cigar_string(c) == rec["ksw_gg"]["cigar"]


def test_gg_family_golden(lib):
    """All committed cases of the global family on the GPU (tests/golden/gg_cases.npz, oracle/gen_golden_gg.py): ksw_gg, ksw_gg2 and
    ksw_gg2_sse of the reference on bands -1, |d|, |d| + 1, 20, 64, 500 (ksw_gg2 / ksw_gg2_sse where they equal ksw_gg), wildcards, score
    only and with CIGAR, and the band that cannot reach the corner (the library's definition: KSW_NEG_INF, no CIGAR; include/ksw2_amd.h)."""
    gc = gu.GgCases()
    n = {"gg": 0, "gg2": 0, "gg2_sse": 0, "contract": 0}
    for k in gc.contract_cases():
        c = gc.case(k)
        s, cg = lib.gg(c["func"], c["q"], c["t"], c["mat"], c["gq"], c["ge"], w=c["w"], with_cigar=c["with_cigar"])
        assert s == c["score"] and list(cg) == c["cigar"], (k, c["func"], c["w"], c["origin"], s, c["score"])
        n["contract" if c["origin"] else c["func"]] += 1
    assert n["gg"] > 200 and n["gg2"] > 200 and n["gg2_sse"] > 120 and n["contract"] > 100, n


def test_random_golden_cases(lib):
    """All 3600 committed random cases (outputs of the compiled reference), batched per (function, scoring)."""
    rc = gu.RandomCases()
    groups = {}
    for k in range(rc.n):
        c = rc.case(k)
        key = (c["func"], c["mat"].tobytes(), c["gq"], c["ge"], c["gq2"], c["ge2"])
        groups.setdefault(key, []).append(c)
    nchk = 0
    for (func, _, gq, ge, gq2, ge2), cs in groups.items():
        dual = "extd" in func
        scalar = not func.endswith("2_sse")
        qs, ts = [c["q"] for c in cs], [c["t"] for c in cs]
        w = np.array([c["w"] for c in cs]); zd = np.array([c["zdrop"] for c in cs])
        eb = np.array([c["end_bonus"] for c in cs])
        fl = np.array([c["flag"] | (po.GENERIC_SC if scalar else 0) for c in cs])
        if dual:
            res = lib.extd_batch(qs, ts, cs[0]["mat"], gq, ge, gq2, ge2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
        else:
            res = lib.extz_batch(qs, ts, cs[0]["mat"], gq, ge, w=w, zdrop=zd, end_bonus=eb, flag=fl)
        for c, r in zip(cs, res):
            exp = c["expect"]
            if scalar:
                if c["flag"] & po.EXTZ_ONLY:
                    # the scalar functions have no end bonus branch: compare through the scalar-named entry point
                    r = (lib.extd(c["q"], c["t"], c["mat"], gq, ge, gq2, ge2, w=c["w"], zdrop=c["zdrop"], flag=c["flag"]) if dual else
                         lib.extz(c["q"], c["t"], c["mat"], gq, ge, w=c["w"], zdrop=c["zdrop"], flag=c["flag"]))
                assert not diff(exp, r, gu.FIELDS + ["cigar"]), (func, c["w"], c["zdrop"], c["flag"])
            else:
                assert not diff(exp, r, gu.SSE_LOOSE_FIELDS), (func, c["w"], c["flag"])
                if r["max_t"] == exp["max_t"] and r["max_q"] == exp["max_q"]:
                    assert r["cigar"] == exp["cigar"]
            nchk += 1
    assert nchk == rc.n


@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("mode", [po.SCORE_ONLY, 0, po.RIGHT])
def test_ragged_batches_vs_oracle(lib, dual, mode):
    """Ragged lengths, every band class boundary, Z-drop on/off, extension flags, wildcards, several pairs per wavefront."""
    rng = np.random.Generator(np.random.PCG64(1234 + mode + 10 * dual))
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    tot = 0
    for rnd in range(6):
        n = 150
        pairs = synth.ragged_pairs(rng, n, 1, [120, 700, 2200][rnd % 3], sub=0.05, ind=0.12, n_rate=0.01 if rnd % 2 else 0.0)
        qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
        w = rng.choice([-1, 0, 1, 5, 20, 64, 68, 69, 100, 284, 285, 500, 536, 537, 1040], size=n)
        zd = rng.choice([-1, 50, 200, 400], size=n)
        eb = rng.choice([0, 10, 50], size=n)
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) |
                       (po.GENERIC_SC if rng.random() < 0.3 else 0) for _ in range(n)])
        k, _ = check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
        tot += k
    assert tot == 900


def test_cfg2_shape_subset(lib):
    """BASELINE config 2 shape (512 x 512, w=64, extz2 score-only): 4096 pairs, every pair checked."""
    n = 4096
    q, t = synth.fixed_batch(2, n, 512, 512, sub=0.05, ind=0.06)
    mat = synth.simple_mat(5, 2, 4, -1)
    k, _ = check_batch(lib, False, q, t, mat, 4, 2, 0, 0, w=64, zdrop=-1, flag=po.SCORE_ONLY)
    assert k == n


def test_cfg3_shape_subset(lib):
    """BASELINE config 3 shape (2048 x 2048, w=256, extd2, Z-drop 400, CIGAR): 256 pairs, 10 % with a random tail."""
    n = 256
    q, t = synth.fixed_batch(3, n, 2048, 2048, sub=0.05, ind=0.10, tail_random_frac=0.25, tail_pairs=0.10)
    mat = synth.simple_mat(5, 2, 4, -1)
    k, res = check_batch(lib, True, q, t, mat, 4, 2, 24, 1, w=256, zdrop=400, flag=0)
    assert k == n
    assert sum(r["zdropped"] for r in res) > 0          # the Z-drop path is exercised


def test_cfg2_full_size_properties(lib, monkeypatch):
    """BASELINE config 2 at its full size (65 536 pairs) through properties that do not need the oracle on every pair:
    the packed-int16 and the int32 kernels agree on every field of every pair, results do not depend on the position of a
    pair in the batch (reversed batch), and every 64th pair equals the oracle."""
    n = 65536
    q, t = synth.fixed_batch(2, n, 512, 512, sub=0.05, ind=0.06)
    mat = synth.simple_mat(5, 2, 4, -1)
    b = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=po.SCORE_ONLY)
    p = b.plan(False); assert p.packed_pairs() == n; p.run(); r1 = p.fetch_raw().copy(); p.close()
    b2 = lib.make_batch(q[::-1], t[::-1], mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=po.SCORE_ONLY)
    p = b2.plan(False); p.run(); r2 = p.fetch_raw().copy(); p.close()
    assert (r2[::-1] == r1).all()
    monkeypatch.setenv("KSW2AMD_NO_PK", "1")
    p = b.plan(False); assert p.packed_pairs() == 0; p.run(); r0 = p.fetch_raw().copy(); p.close()
    assert (r0 == r1).all()
    for i in range(0, n, 64):
        exp = po.align("oracle", "extz2", q[i], t[i], mat, 4, 2, w=64, zdrop=-1, flag=po.SCORE_ONLY)
        assert (exp["score"], exp["max"], exp["max_t"], exp["max_q"], exp["mqe"], exp["mte"]) == (r1[i][8], r1[i][0], r1[i][3], r1[i][2], r1[i][4], r1[i][6])


def test_cfg3_full_size_properties(lib, monkeypatch):
    """BASELINE config 3 at its full size (16 384 pairs, extd2, Z-drop, CIGAR): packed and int32 kernels agree on every pair
    (fields and CIGARs), every CIGAR spans exactly the aligned prefixes and re-scores to the reported score / maximum, Z-drop
    fires on the pairs with a random tail, and every 64th pair equals the oracle."""
    n = 16384
    q, t = synth.fixed_batch(3, n, 2048, 2048, sub=0.05, ind=0.10, tail_random_frac=0.25, tail_pairs=0.10)
    mat = synth.simple_mat(5, 2, 4, -1)
    res = lib.extd_batch(q, t, mat, 4, 2, 24, 1, w=256, zdrop=400, flag=0)
    monkeypatch.setenv("KSW2AMD_NO_PK", "1")
    res0 = lib.extd_batch(q, t, mat, 4, 2, 24, 1, w=256, zdrop=400, flag=0)
    ndrop = 0
    for i in range(n):
        r = res[i]
        assert r == res0[i], i
        sc, ql, tl = cigar_score(r["cigar"], q[i], t[i], mat, 5, 4, 2, 24, 1)
        if r["zdropped"]:
            ndrop += 1
            assert (ql, tl) == (r["max_q"] + 1, r["max_t"] + 1) and sc == r["max"], i
        else:
            assert (ql, tl) == (2048, 2048) and sc == r["score"], i
    assert 100 < ndrop < 0.2 * n, ndrop          # the pairs with a random tail: Z-drop fires on a few hundred of them
    for i in range(0, n, 64):
        exp = po.align("oracle", "extd2", q[i], t[i], mat, 4, 2, 24, 1, w=256, zdrop=400, flag=0)
        assert not diff(exp, res[i], CMP_FIELDS), i


def test_10k_banded(lib):
    """Headline shape of the north star: 10 000 x 10 000, w=500, zdrop=400, extz2, score-only and CIGAR."""
    n = 8
    q, t = synth.fixed_batch(6, n, 10000, 10000, sub=0.05, ind=0.06)
    mat = synth.simple_mat(5, 2, 4, -1)
    check_batch(lib, False, q, t, mat, 4, 2, 0, 0, w=500, zdrop=400, flag=po.SCORE_ONLY)
    check_batch(lib, False, q, t, mat, 4, 2, 0, 0, w=500, zdrop=400, flag=0)


def test_very_long_reads(lib):
    """Sizes beyond the packed kernels' 16-bit column index (32 000): 150 k x 150 k banded through the int32 resident class, and
    25 k x 25 k unbanded through the generation-serial class; score-only and CIGAR."""
    mat = synth.simple_mat(5, 2, 4, -1)
    q, t = synth.fixed_batch(9, 2, 150000, 150000, sub=0.04, ind=0.05)
    p = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=500, zdrop=400, flag=po.SCORE_ONLY).plan(False)
    assert p.packed_pairs() == 0
    p.close()
    check_batch(lib, False, q, t, mat, 4, 2, 0, 0, w=500, zdrop=400, flag=po.SCORE_ONLY)
    check_batch(lib, True, q, t, mat, 4, 2, 24, 1, w=500, zdrop=400, flag=0)
    q, t = synth.fixed_batch(10, 1, 25000, 25300, sub=0.04, ind=0.05)
    check_batch(lib, False, q, t, mat, 4, 2, 0, 0, w=-1, zdrop=-1, flag=po.RIGHT)
    # 50 k reads still fit the packed kernels (unsigned 16-bit column index, per-strip score bases)
    q, t = synth.fixed_batch(11, 4, 50000, 50011, sub=0.04, ind=0.05, tail_random_frac=0.2, tail_pairs=0.5)
    p = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=100, zdrop=300, flag=0).plan(True)
    assert p.packed_pairs() == 4
    p.close()
    check_batch(lib, True, q, t, mat, 4, 2, 24, 1, w=100, zdrop=300, flag=0)
    check_batch(lib, False, q, t, mat, 4, 2, 0, 0, w=100, zdrop=300, flag=po.SCORE_ONLY)


def test_edge_cases(lib):
    mat = synth.simple_mat(5, 2, 4, -1)
    one = np.array([1], dtype=np.uint8)
    # empty inputs -> reset record (ksw2_extz2_sse.c:57)
    r = lib.extz2(np.zeros(0, np.uint8), one, mat, 4, 2)
    assert (r["score"], r["max"], r["max_t"], r["n_cigar"], r["zdropped"]) == (ka.KSW_NEG_INF, 0, -1, 0, 0)
    # mismatch penalty larger than 2(q+e) -> reset record (ksw2_extz2_sse.c:78-82)
    r = lib.extz2(one, one, synth.simple_mat(5, 1, 20, -1), 4, 2)
    assert r["score"] == ka.KSW_NEG_INF and r["n_cigar"] == 0
    # 1 x 1
    for a, b in ((1, 1), (1, 2)):
        exp = po.align("oracle", "extz2", np.array([a], np.uint8), np.array([b], np.uint8), mat, 4, 2)
        res = lib.extz2(np.array([a], np.uint8), np.array([b], np.uint8), mat, 4, 2)
        assert not diff(exp, res)
    # band that cannot reach the corner: stop like the SSE kernels (zdropped, no score)
    rng = np.random.Generator(np.random.PCG64(5))
    t = rng.integers(0, 4, 300, dtype=np.uint8)
    qv = t[:100].copy()
    for qq, tt in ((qv, t), (t, qv)):
        exp = po.align("oracle", "extz2", qq, tt, mat, 4, 2, w=10)
        res = lib.extz2(qq, tt, mat, 4, 2, w=10)
        assert not diff(exp, res) and res["zdropped"] == 1 and res["score"] == ka.KSW_NEG_INF


def test_cigar_buffer_reuse(lib):
    """ez is reused across calls like cli.c does: capacity persists and grows by doubling from 4 (ksw2.h:116-119)."""
    rng = np.random.Generator(np.random.PCG64(9))
    mat = synth.simple_mat(5, 2, 4, -1)
    ez = ka.KswExtz()
    caps = []
    for L in (5, 200, 50):
        (qq, tt), = synth.ragged_pairs(rng, 1, L, L, sub=0.1, ind=0.2)
        r = lib.extz2(qq, tt, mat, 4, 2, ez=ez)
        exp = po.align("oracle", "extz2", qq, tt, mat, 4, 2)
        assert r["cigar"] == exp["cigar"]
        caps.append(r["m_cigar"])
        assert r["m_cigar"] >= r["n_cigar"] and (r["m_cigar"] & (r["m_cigar"] - 1)) == 0
    assert caps[2] == caps[1] >= caps[0]


def test_mt_pair_banded(lib):
    """MT-human x MT-orang with w=500 (golden: score -13510, CIGAR md5 c07fce86940f)."""
    ka_ = {(r["func"], r["w"], r.get("flag", 0), r.get("zdrop", -1)): r for r in gu.known_answers()["mt"]}
    _, ts = gu.read_fasta("MT-human.fa")
    _, qs = gu.read_fasta("MT-orang.fa")
    mat = gu.simple_mat(5, 2, 4, 0)
    exp = ka_[("ksw_extz", 500, 0, -1)]
    res = lib.extz(qs[0], ts[0], mat, 4, 2, w=500)
    assert (res["score"], res["max"], res["max_t"], res["max_q"]) == (exp["score"], exp["max"], exp["max_t"], exp["max_q"])
    s = gu.cigar_string(res["cigar"])
    assert hashlib.md5((s + "\n").encode()).hexdigest()[:12] == "c07fce86940f"
    exp = ka_[("ksw_extd", 500, 0, -1)]
    res = lib.extd(qs[0], ts[0], mat, 4, 2, 13, 1, w=500)
    assert res["score"] == exp["score"] and gu.cigar_string(res["cigar"]) == exp["cigar"]


def test_mt_pair_unbanded(lib):
    """BASELINE config 4 input: MT-human x MT-orang, w=-1, full global with CIGAR (generation-serial kernels).
    Golden: 16102 / 17054 / 16568 / 16024, CIGAR md5 ea0524d904ed (extz), df0e77e43f48 (extd), -r db8b671f4dbf."""
    ka_ = {(r["func"], r["w"], r.get("flag", 0), r.get("zdrop", -1)): r for r in gu.known_answers()["mt"]}
    _, ts = gu.read_fasta("MT-human.fa")
    _, qs = gu.read_fasta("MT-orang.fa")
    mat = gu.simple_mat(5, 2, 4, 0)
    for func, flag, md5 in (("extz", 0, "ea0524d904ed"), ("extz", po.RIGHT, "db8b671f4dbf"), ("extd", 0, "df0e77e43f48"),
                            ("extd", po.RIGHT, "8e2c9cfb877a")):
        exp = ka_[("ksw_" + func, -1, flag, -1)]
        res = lib.extz(qs[0], ts[0], mat, 4, 2, w=-1, flag=flag) if func == "extz" else lib.extd(qs[0], ts[0], mat, 4, 2, 13, 1, w=-1, flag=flag)
        for f in gu.FIELDS:
            assert res[f] == exp[f], (func, flag, f, res[f], exp[f])
        s = gu.cigar_string(res["cigar"])
        assert s == exp["cigar"] and hashlib.md5((s + "\n").encode()).hexdigest()[:12] == md5
    # replicated batch: every replica is computed and identical (config 4 replicates this pair 4096 x)
    n = 8
    res = lib.extz_batch([qs[0]] * n, [ts[0]] * n, mat, 4, 2, w=-1, zdrop=-1, flag=po.GENERIC_SC)
    exp = ka_[("ksw_extz", -1, 0, -1)]
    for r in res:
        assert r["score"] == exp["score"] and gu.cigar_string(r["cigar"]) == exp["cigar"]
    s, c = lib.gg("gg2_sse", qs[0], ts[0], mat, 4, 2, w=-1)
    assert s == 16102 and gu.cigar_string(c) == exp["cigar"]


@pytest.mark.parametrize("dual", [False, True])
def test_wide_band_ragged(lib, dual):
    """Bands beyond the resident kernels (w > 1040 on > 2048 rows): generation-serial class, Z-drop and flags."""
    rng = np.random.Generator(np.random.PCG64(77 + dual))
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    for mode in (po.SCORE_ONLY, 0, po.RIGHT):
        n = 24
        pairs = synth.ragged_pairs(rng, n, 2100, 5200, sub=0.05, ind=0.12, indel_mean=4.0, n_rate=0.005)
        qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
        w = rng.choice([-1, 1041, 1100, 2000, 3000], size=n)
        zd = rng.choice([-1, 200, 400, 2000], size=n)
        eb = rng.choice([0, 10, 50], size=n)
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) for _ in range(n)])
        check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)


def test_packed_and_int32_score_only_agree(lib, monkeypatch):
    """Config-2 shape through both score-only code paths: packed int16 (default) and int32 (KSW2AMD_NO_PK=1)."""
    n = 1024
    q, t = synth.fixed_batch(2, n, 512, 512, sub=0.05, ind=0.06, stream=3)
    mat = synth.simple_mat(5, 2, 4, -1)
    b = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=po.SCORE_ONLY)
    p = b.plan(False); assert p.packed_pairs() == n; p.run(); r1 = p.fetch_raw().copy(); p.close()
    monkeypatch.setenv("KSW2AMD_NO_PK", "1")
    p = b.plan(False); assert p.packed_pairs() == 0; p.run(); r0 = p.fetch_raw().copy(); p.close()
    assert (r0 == r1).all()
    k, _ = check_batch(lib, False, q, t, mat, 4, 2, 0, 0, w=64, zdrop=-1, flag=po.SCORE_ONLY, sample=range(0, n, 8))
    assert k == 128


@pytest.mark.parametrize("dual", [False, True])
def test_packed_fixed_shape_batches(lib, dual):
    """Packed-int16 class on the GPU: wildcards, per-pair Z-drop / flags, odd leftovers, all three geometries."""
    rng = np.random.Generator(np.random.PCG64(50 + dual))
    for rnd in range(18):
        mat, q, e, q2, e2 = [(synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (synth.simple_mat(5, 1, 3, 0), 5, 1, 20, 1),
                             (synth.simple_mat(5, 2, 4, -3), 4, 2, 13, 1)][rnd % 3]
        n = int(rng.integers(3, 60))
        ql = int(rng.integers(50, 900)); tl = max(1, ql + int(rng.integers(-30, 30)))
        w = int(rng.choice([20, 64, 68, 100, 284, 400, -1]))
        qs, ts = synth.fixed_batch(200 + rnd, n, ql, tl, sub=0.05, ind=0.08, tail_random_frac=0.3, tail_pairs=0.3)
        if rnd % 2:
            qs, ts = qs.copy(), ts.copy()
            qs[rng.random(qs.shape) < 0.01] = 4; ts[rng.random(ts.shape) < 0.01] = 4
        zd = rng.choice([-1, 30, 100, 400], size=n); eb = rng.choice([0, 10, 50], size=n)
        fl = np.array([po.SCORE_ONLY | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.GENERIC_SC if rnd % 3 == 0 else 0) for _ in range(n)])
        check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)


@pytest.mark.parametrize("dual", [False, True])
def test_packed_rebased_long_reads(lib, dual):
    """Reads whose absolute scores leave 16 bits: packed kernels with per-strip bases.  Fixed-shape batches, score-only
    and both traceback modes, bands up to the window limit, all-match / all-mismatch pairs, Z-drop on and off."""
    rng = np.random.Generator(np.random.PCG64(91 + dual))
    scs = [(synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (synth.simple_mat(5, 10, 12, 0), 12, 4, 40, 2),
           (synth.simple_mat(5, 1, 3, 0), 5, 1, 20, 1), (synth.simple_mat(5, 6, 9, -3), 9, 3, 30, 1)]
    wide = [[400, 500, 536, 560], [140, 170, 180], [500, 600, 700], [200, 250, 270]]
    npk = ntot = 0
    for rnd in range(12):
        mat, q, e, q2, e2 = scs[rnd % 4]
        n = int(rng.integers(3, 24))
        ql = int(rng.integers(4000, 21000)) if rnd % 4 in (0, 2) else int(rng.integers(1500, 6000))
        tl = ql + int(rng.integers(-60, 60))
        w = int(rng.choice([10, 20, 64, 68, 100, 150])) if rnd < 6 else int(rng.choice(wide[rnd % 4]))
        qs, ts = synth.fixed_batch(1900 + rnd, n, ql, tl, sub=0.05, ind=0.1, tail_random_frac=0.3, tail_pairs=0.3)
        qs, ts = qs.copy(), ts.copy()
        qs[0, :] = 0; ts[0, :] = 0
        qs[1, :] = 1; ts[1, :] = 2
        zd = rng.choice([-1, 100, 400, 2000], size=n)
        eb = rng.choice([0, 10, 50], size=n)
        mode = [po.SCORE_ONLY, 0, po.RIGHT][rnd % 3]
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) for _ in range(n)])
        p = lib.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl).plan(dual)
        npk += p.packed_pairs(); ntot += n
        p.close()
        check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl, sample=[0, 1] + list(range(2, n, 2)))
    assert npk > ntot // 2


def test_small_packed_class_leaves_the_pair_kernels(lib, monkeypatch):
    """A one-alignment-per-wavefront class of fewer reads than the device has SIMDs: every read to the solo kernel (a SIMD
    of its own); what the solo kernel cannot take (approximate mode here) goes back to the int32 kernels below 0.4 packed
    wavefronts per SIMD.  Same results either way."""
    n = 64
    q, t = synth.fixed_batch(2, n, 1500, 1500, sub=0.05, ind=0.06, stream=3)
    mat = synth.simple_mat(5, 2, 4, -1)
    for flag, kernel in ((po.SCORE_ONLY, "solo"), (po.SCORE_ONLY | po.APPROX_MAX, "int32")):
        monkeypatch.setenv("KSW2AMD_SIMDS", "0")
        b = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=200, zdrop=-1, flag=flag)                   # (64,8): one packed pair per wavefront
        p = b.plan(False); assert p.packed_pairs() == n and p.describe()[0]["kernel"] == "pk"; p.run(); r1 = p.fetch_raw().copy(); p.close()
        monkeypatch.delenv("KSW2AMD_SIMDS")
        p = b.plan(False); assert [d["kernel"] for d in p.describe()] == [kernel]; p.run(); r0 = p.fetch_raw().copy(); p.close()
        if flag & po.APPROX_MAX: assert (r0[:, 8] == r1[:, 8]).all()      # only the score exists in this mode (include/ksw2_amd.h)
        else: assert (r0 == r1).all()


def test_wide_alphabets(lib):
    """m > 5 residue types through the int32 kernels with the matrix in LDS; the single-call entry points too."""
    from tests.test_sim_parity import _wide_alphabet_cases
    rng = np.random.Generator(np.random.PCG64(18))
    for rnd in range(12):
        m, mat, qs, ts, w, zd, fl = _wide_alphabet_cases(rng, rnd)
        for dual in (False, True):
            check_batch(lib, dual, qs, ts, mat, 6, 2, 20, 1, w=w, zdrop=zd, flag=fl, m=m)
    m, mat, qs, ts, w, zd, fl = _wide_alphabet_cases(rng, 0)
    exp = po.align("oracle", "extz2", qs[0], ts[0], mat, 6, 2, w=-1, flag=po.GENERIC_SC, m=m)
    assert not diff(exp, lib.extz2(qs[0], ts[0], mat, 6, 2, w=-1, flag=po.GENERIC_SC, m=m))


def test_concurrent_host_threads(lib):
    """minimap2-style callers align from many host threads at once (threading): every thread has its own
    stream-less plan, staging and buffer cache; single calls and batches interleave and stay bit-exact."""
    import threading
    mat = synth.simple_mat(5, 2, 4, -1)
    errors = []

    def worker(tid):
        try:
            rng = np.random.Generator(np.random.PCG64(1000 + tid))
            for it in range(6):
                pairs = synth.ragged_pairs(rng, 12, 30, 900, sub=0.05, ind=0.1, n_rate=0.005 if tid % 2 else 0.0)
                qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
                fl = [0, po.RIGHT, po.SCORE_ONLY][(tid + it) % 3]
                check_batch(lib, bool(tid & 1), qs, ts, mat, 4, 2, 24, 1, w=int(rng.choice([-1, 20, 100])), zdrop=int(rng.choice([-1, 200])), flag=fl)
                exp = po.align("oracle", "extz2", qs[0], ts[0], mat, 4, 2, w=50, zdrop=100, flag=0)
                assert not diff(exp, lib.extz2(qs[0], ts[0], mat, 4, 2, w=50, zdrop=100, flag=0))
        except Exception as ex:                                  # noqa: BLE001 - reported below with the thread id
            errors.append((tid, repr(ex)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_cfg5_ont_like_mix(lib):
    """BASELINE config 5 shape: ragged ONT-like pairs, qlen in [300, 20000], 15 % indels, band 500, extd2 with
    Z-drop 400 and CIGAR.  48 pairs here (the oracle needs ~0.1 s per long pair); every field and CIGAR compared."""
    rng = np.random.Generator(np.random.PCG64(20260005))
    mat = synth.simple_mat(5, 2, 4, -1)
    qs, ts = [], []
    while len(qs) < 48:
        tl = int(rng.integers(300, 20001))
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        q = synth.mutate_one(t, rng, sub=0.03, ind=0.15, indel_mean=1.5)
        if abs(len(q) - tl) > 450 or len(q) < 300:
            continue
        qs.append(q); ts.append(t)
    k, res = check_batch(lib, True, qs, ts, mat, 4, 2, 24, 1, w=500, zdrop=400, flag=0)
    assert k == 48 and all(r["n_cigar"] > 0 for r in res)


def test_cfg4_replicas_sharded_plan(lib):
    """Config 4 replicates the MT pair; a plan larger than the device budget must be split transparently
    (KSW2AMD_MAX_BYTES forces the split with 6 replicas) and every replica must be identical."""
    import os
    _, ts = gu.read_fasta("MT-human.fa")
    _, qs = gu.read_fasta("MT-orang.fa")
    mat = gu.simple_mat(5, 2, 4, 0)
    os.environ["KSW2AMD_MAX_BYTES"] = str(400 << 20)          # ~2 replicas of 144 MB traceback per sub-batch
    try:
        res = lib.extz_batch([qs[0]] * 6, [ts[0]] * 6, mat, 4, 2, w=-1, zdrop=-1, flag=0)
    finally:
        del os.environ["KSW2AMD_MAX_BYTES"]
    import hashlib
    for r in res:
        assert (r["score"], r["max"], r["max_t"], r["max_q"]) == (16102, 17054, 16568, 16024)
        assert hashlib.md5((gu.cigar_string(r["cigar"]) + "\n").encode()).hexdigest()[:12] == "ea0524d904ed"


def test_cli_matches_reference_cli(lib):
    """tools/ksw2-test-amd against the reference's own ksw2-test (oracle/_ref, built from /root/reference/cli.c):
    byte-identical stdout on config 1 (test/t1.fa x test/q1.fa) and on the MT pair, scalar algorithms and options."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ours, ref = os.path.join(root, "tools", "ksw2-test-amd"), os.path.join(root, "oracle", "_ref", "ksw2-test")
    if not os.path.exists(ours):
        subprocess.run(["make", "-C", os.path.join(root, "tools")], check=True, capture_output=True)
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/ksw2-test not built (needs /root/reference)")
    d = os.path.join(root, "tests", "golden", "data")
    t1, q1, mh, mo = (os.path.join(d, f) for f in ("t1.fa", "q1.fa", "MT-human.fa", "MT-orang.fa"))
    runs = [(["-t", "extz"], t1, q1), (["-t", "extd"], t1, q1), (["-t", "extz", "-r"], t1, q1), (["-t", "extd", "-s"], t1, q1),
            (["-t", "gg"], t1, q1), (["-t", "gg2"], t1, q1), (["-t", "extz", "-A1", "-B3", "-O5", "-E1"], t1, q1), (["-t", "extd", "-a"], t1, q1),
            (["-t", "exts2_sse"], t1, q1), (["-t", "exts2_sse", "-r"], t1, q1), (["-t", "exts2_sse", "-z", "100"], t1, q1),
            (["-t", "extf2_sse"], t1, q1), (["-t", "extf2_sse", "-z", "30", "-w", "20"], t1, q1), (["-t", "extf2_sse", "-w", "500"], mh, mo),
            (["-t", "extz", "-w", "500"], mh, mo), (["-t", "extd", "-w", "500", "-r"], mh, mo), (["-t", "extz"], mh, mo), (["-t", "gg2", "-s"], mh, mo)]
    for opts, t, q in runs:
        a = subprocess.run([ours] + opts + [t, q], capture_output=True, text=True)
        b = subprocess.run([ref] + opts + [t, q], capture_output=True, text=True)
        assert a.returncode == 0, a.stderr
        assert a.stdout == b.stdout, (opts, a.stdout[:300], b.stdout[:300])
    # -K (results from the caller's pool, cli.c:177,206) and gzip-compressed input (cli.c:210-211)
    import gzip
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        gz = []
        for f in (t1, q1):
            gz.append(os.path.join(tmp, os.path.basename(f) + ".gz"))
            with open(f, "rb") as src, gzip.open(gz[-1], "wb") as dst:
                dst.write(src.read())
        for opts in (["-t", "extz", "-K"], ["-t", "extd2_sse", "-K"], ["-t", "exts2_sse", "-K"]):
            a = subprocess.run([ours] + opts + gz, capture_output=True, text=True, env=dict(os.environ, KSW2_TEST_POOL_STATS="1"))
            b = subprocess.run([ours] + opts[:2] + [t1, q1], capture_output=True, text=True)
            assert a.returncode == 0 and a.stdout == b.stdout and a.stdout.count("\n") == 5, (opts, a.stderr)
            assert "pool: " in a.stderr and " 0 krealloc" not in a.stderr, a.stderr       # the CIGARs really came from the pool
        a = subprocess.run([ours, "-t", "extz", "-K"] + gz, capture_output=True, text=True)
        b = subprocess.run([ref, "-t", "extz", "-K"] + gz, capture_output=True, text=True)
        assert a.stdout == b.stdout, (a.stdout[:300], b.stdout[:300])
    # the opt-in host path for tiny single calls (KSW2AMD_SMALL_CELLS): the per-pair mode's calls computed on the host, same bytes out,
    # with -K too (CIGARs from the caller's pool)
    small = dict(os.environ, KSW2AMD_SMALL_CELLS="1000000000")
    for opts, t, q in runs[:8] + [(["-t", "extz", "-w", "500"], mh, mo), (["-t", "extd", "-w", "500", "-r"], mh, mo), (["-t", "extd2_sse", "-K"], t1, q1)]:
        a = subprocess.run([ours] + opts + [t, q], capture_output=True, text=True, env=small)
        b = subprocess.run([ours] + opts + [t, q], capture_output=True, text=True)
        assert a.returncode == 0 and a.stdout == b.stdout, (opts, a.stdout[:300], b.stdout[:300])
    # batched mode prints the same lines as the per-pair mode
    a = subprocess.run([ours, "-t", "extz2_sse", "-b", t1, q1], capture_output=True, text=True)
    b = subprocess.run([ours, "-t", "extz2_sse", t1, q1], capture_output=True, text=True)
    assert a.returncode == 0 and a.stdout == b.stdout and a.stdout.count("\n") == 5


@pytest.mark.parametrize("big", [False, True])
def test_splice_aware_golden(lib, big, monkeypatch):
    """All 1200 ksw_exts2_sse cases produced by the compiled reference (tests/golden/exts_cases.npz), batched by scoring;
    once through the register-window kernels, once through the scratch-array kernel that takes diagonals of any length."""
    if big:
        monkeypatch.setenv("KSW2AMD_EXTS_BIG", "1")
    ec = gu.ExtsCases()
    groups = {}
    for k in range(ec.n):
        c = ec.case(k)
        groups.setdefault((c["mat"].tobytes(), c["gq"], c["ge"], c["gq2"], c["noncan"], c["junc_bonus"]), []).append(c)
    n = 0
    for (_, gq, ge, gq2, noncan, jb), cs in groups.items():
        res = lib.exts_batch([c["q"] for c in cs], [c["t"] for c in cs], cs[0]["mat"], gq, ge, gq2, noncan, zdrop=np.array([c["zdrop"] for c in cs]),
                             junc_bonus=jb, flag=np.array([c["flag"] for c in cs]), juncs=[c["junc"] for c in cs])
        for c, r in zip(cs, res):
            assert not diff(c["expect"], r, gu.FIELDS + ["cigar"]), (hex(c["flag"]), c["zdrop"], len(c["q"]), len(c["t"]))
            n += 1
    assert n == ec.n


def test_splice_aware_random_and_long(lib):
    from tests.test_sim_parity import _exts_cases, _intron_pair, check_exts_batch
    rng = np.random.Generator(np.random.PCG64(77))
    for rnd in range(30):
        check_exts_batch(lib, *_exts_cases(rng, rnd, 1400))
    mat = synth.simple_mat(5, 1, 2, 0)
    qs, ts = [], []
    for rnd in range(24):
        q, t = _intron_pair(rng, int(rng.integers(1500, 12000)))
        if rnd % 3 == 1:
            q, t = t, q
        qs.append(q)
        ts.append(t)
    flag = np.array([int(rng.choice([0, po.RIGHT, po.SCORE_ONLY, po.EXTZ_ONLY])) | po.SPLICE_FOR for _ in qs])
    zd = rng.choice([-1, 200, 1000], size=len(qs))
    check_exts_batch(lib, qs, ts, [None] * len(qs), mat, 2, 1, 32, 4, 0, flag, zd)
    r = lib.exts2(qs[0], ts[0], mat, 2, 1, 32, 4, flag=po.SPLICE_FOR)
    assert not diff(po.exts2("oracle", qs[0], ts[0], mat, 2, 1, 32, 4, flag=po.SPLICE_FOR), r, gu.FIELDS + ["cigar"])
    # diagonals beyond the largest register window (1472 cells): state in HBM
    big_q, big_t = [], []
    for tl in (2600, 4000, 3000):
        q, t = _intron_pair(rng, tl)
        big_q.append(np.concatenate([q, t[-2000:]])[:int(rng.integers(1600, 2400))])
        big_t.append(t)
    check_exts_batch(lib, big_q, big_t, [None] * 3, mat, 2, 1, 32, 4, 0, np.array([po.SPLICE_FOR, po.SPLICE_FOR | po.RIGHT, po.SCORE_ONLY]),
                     np.array([-1, 500, 200]))


def test_approx_max_mode_golden(lib):
    """KSW_EZ_APPROX_MAX alone on the three "...2_sse" functions: outputs of the compiled reference."""
    ac = gu.ApproxCases()
    for k in range(ac.n):
        c = ac.case(k)
        assert not diff(c["expect"], gu.ApproxCases.run(lib, c), gu.FIELDS + ["cigar"]), (k, c["func"], hex(c["flag"]))


def test_approx_max_packed_classes(lib):
    from tests.test_sim_parity import _approx_batches
    rng = np.random.Generator(np.random.PCG64(67))
    for rnd in range(18):
        mat, q, e, q2, e2, qs, ts, w, zd, eb, fl = _approx_batches(rng, rnd)
        for dual in (False, True):
            check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)


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
def test_solo_kernel(lib, dual, monkeypatch):
    """One alignment per wavefront on both register halves (k2a_fill_solo_kernel), every eligible alignment sent there."""
    monkeypatch.setenv("KSW2AMD_SOLO", "all")
    rng = np.random.Generator(np.random.PCG64(909 + dual))
    nsolo = ntot = 0
    for rnd in range(12):
        n = 40
        mat, q, e, q2, e2, qs, ts, w, zd, eb = _solo_cases(rng, rnd, n)
        mode = [po.SCORE_ONLY, 0, po.RIGHT][rnd % 3]
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) for _ in range(n)])
        p = lib.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl).plan(dual)
        nsolo += p.packed_pairs(); ntot += n
        p.close()
        check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
    assert nsolo > ntot * 3 // 4


def test_solo_long_reads_and_leftovers(lib, monkeypatch):
    """KSW2AMD_SOLO=1 on a mixed batch: same-shape pairs two per lane, unique shapes solo; 10-20 k reads with traceback."""
    monkeypatch.setenv("KSW2AMD_SOLO", "1")
    mat = synth.simple_mat(5, 2, 4, -1)
    qs, ts = synth.fixed_batch(33, 8, 3000, 2990, sub=0.05, ind=0.1)
    rng = np.random.Generator(np.random.PCG64(13))
    extra = synth.ragged_pairs(rng, 12, 8000, 20000, sub=0.05, ind=0.08)
    qs = [x for x in qs] + [p[0] for p in extra]; ts = [x for x in ts] + [p[1] for p in extra]
    for dual, flag in ((True, 0), (False, po.SCORE_ONLY), (True, po.RIGHT)):
        p = lib.make_batch(qs, ts, mat, 4, 2, 24, 1, w=300, zdrop=400, flag=flag).plan(dual)
        assert p.packed_pairs() == 20
        p.close()
        check_batch(lib, dual, qs, ts, mat, 4, 2, 24, 1, w=300, zdrop=400, flag=flag)


@pytest.mark.parametrize("state", ["auto", "win", "lds", "hbm"])
def test_linear_xdrop_golden(lib, state, monkeypatch):
    """All 2000 ksw_extf2_sse cases produced by the compiled reference (tests/golden/extf_cases.npz), batched by scoring.
    auto: the host's choice between register window and LDS; win: the register window wherever the band fits it; lds: every case through the LDS-state kernel; hbm: through
    the kernel that keeps U, V, S in HBM scratch (wide bands on targets over 21504 residues take it in production)."""
    if state == "hbm":
        monkeypatch.setenv("KSW2AMD_EXTF_HBM", "1")
    if state == "lds":
        monkeypatch.setenv("KSW2AMD_EXTF_LDS", "1")
    if state == "win":
        monkeypatch.setenv("KSW2AMD_EXTF_WIN", "1")
    fc = gu.ExtfCases()
    cases = [fc.case(k) for k in range(fc.n)]
    ndrop = 0
    for sc in sorted({(c["mch"], c["mis"], c["e"]) for c in cases}):
        sub = [c for c in cases if (c["mch"], c["mis"], c["e"]) == sc]
        res = lib.extf_batch([c["q"] for c in sub], [c["t"] for c in sub], *sc, w=[c["w"] for c in sub], xdrop=[c["xdrop"] for c in sub])
        for r, c in zip(res, sub):
            assert not diff(r, c["expect"], gu.FIELDS), (sc, len(c["q"]), len(c["t"]), c["w"], c["xdrop"])
            ndrop += r["zdropped"]
    assert ndrop > 200
    c = cases[7]
    assert not diff(lib.extf2(c["q"], c["t"], c["mch"], c["mis"], c["e"], c["w"], c["xdrop"]), c["expect"], gu.FIELDS)


def test_linear_xdrop_long_and_empty(lib):
    """Every state class against the oracle: targets of 1 k / 4 k / 20 k (LDS) and 30 k (HBM), banded and not, X-drop on and
    off, a diverging tail that triggers the drop; empty sequences."""
    rng = np.random.Generator(np.random.PCG64(606))
    qs, ts, ws, xs = [], [], [], []
    for tl, w, xd in ((1000, -1, -1), (1024, 100, 50), (1025, 33, -1), (4096, 500, 100), (5000, 64, -1), (20000, 200, 300), (21504, 16, -1),
                      (21505, 100, -1), (30000, 300, 200), (30000, 50, -1), (3000, 146, -1), (3000, 147, -1), (6000, 402, 400), (6000, 403, -1),
                      (30000, 403, -1), (22000, 1000, 300)):
        (q, t), = synth.ragged_pairs(rng, 1, tl, tl, sub=0.04, ind=0.02)
        t = t[:tl] if len(t) >= tl else np.concatenate([t, rng.integers(0, 4, tl - len(t)).astype(np.uint8)])
        if xd >= 0:
            q = np.concatenate([q[: len(q) * 2 // 3], rng.integers(0, 4, len(q) // 3).astype(np.uint8)])
        qs.append(q); ts.append(t); ws.append(w); xs.append(xd)
    res = lib.extf_batch(qs, ts, 2, -4, 2, w=ws, xdrop=xs)
    for i, r in enumerate(res):
        exp = po.extf2("oracle", qs[i], ts[i], 2, -4, 2, ws[i], xs[i])
        assert not diff(r, exp, gu.FIELDS), (i, len(qs[i]), len(ts[i]), ws[i], xs[i], {f: (r[f], exp[f]) for f in diff(r, exp, gu.FIELDS)})
    assert sum(r["zdropped"] for r in res) >= 2 and sum(not r["zdropped"] for r in res) >= 2
    e = np.zeros(0, np.uint8); one = np.array([2], np.uint8); five = (np.arange(5) % 4).astype(np.uint8)
    for q, t in ((e, e), (one, e), (e, one), (five, e), (e, five), (one, one)):
        assert not diff(lib.extf2(q, t, 2, -4, 2, -1, 50), po.extf2("oracle", q, t, 2, -4, 2, -1, 50), gu.FIELDS), (len(q), len(t))


@pytest.mark.parametrize("lds", ["0", "1"])
def test_row_state_in_registers_and_in_lds(lib, lds, monkeypatch):
    """Packed (64, 16) two-piece traceback and generation-serial single-gap traceback kernels with their per-row maxima in
    registers (KSW2AMD_LDSROWS=0) and in LDS (=1); the launcher's own choice depends on the number of tasks."""
    monkeypatch.setenv("KSW2AMD_LDSROWS", lds)
    rng = np.random.Generator(np.random.PCG64(2025))
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    npk = 0
    for rnd, (ql, w) in enumerate(((2500, 400), (9000, 500), (15000, 330))):
        qs, ts = synth.fixed_batch(800 + rnd, 6, ql, ql - 20, sub=0.05, ind=0.1, tail_random_frac=0.3, tail_pairs=0.3)
        zd = rng.choice([-1, 400, 2000], size=6)
        for mode in (0, po.RIGHT):
            fl = np.array([mode | (po.REV_CIGAR if rng.random() < 0.3 else 0) for _ in range(6)])
            p = lib.make_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, flag=fl).plan(True)
            npk += p.packed_pairs()
            p.close()
            check_batch(lib, True, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, flag=fl)
    assert npk >= 24
    pairs = synth.ragged_pairs(rng, 4, 2100, 5000, sub=0.05, ind=0.12, indel_mean=4.0)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    for flag in (0, po.RIGHT):
        check_batch(lib, False, qs, ts, mat, q, e, q2, e2, w=np.array([-1, 1500, 2000, -1]), zdrop=np.array([-1, 400, -1, 1000]), flag=flag)


@pytest.mark.parametrize("ldc", ["0", "1"])
def test_code_planes_in_registers_and_in_lds(lib, ldc, monkeypatch):
    """KSW2AMD_LDSCODES=0 / 1 forces the launcher's choice for the score-only packed kernels of the (64, 16) and (8, 18)
    geometries: target-code planes in registers / in LDS.  =1 runs k2a_fill_pk_kernel<G, C, false, 0, RB, NOMAX, 2> -- all eight
    g_fill_pk_ldscodes entries, among them the kernels the headline benchmark and config 2 time -- against the oracle: plain and
    re-based, exact and KSW_EZ_APPROX_MAX, Z-drop on / off, shapes in pairs, triples and singles; the plan reports the form it
    will launch."""
    from tests.test_sim_parity import _check_code_plane_forms
    monkeypatch.setenv("KSW2AMD_LDSCODES", ldc)
    monkeypatch.setenv("KSW2AMD_DEFER", "0")              # (the deferred arg-max kernels have their own forms: next test)
    for seed in (31, 32, 33):
        _check_code_plane_forms(lib, "ldscodes" if ldc == "1" else "registers", seed=seed)


def test_flat_batches_host_and_device_arena(lib):
    """ksw2amd_ext?_batch_flat / ksw2amd_plan_create_flat: one arena + offsets, from host memory and from a device-resident arena
    (what an RCCL-delivered shard is): every field and CIGAR equal to the pointer entry points; pairs with a wildcard code are
    reported by the packed kernels and re-run (host arena: where they lie; device arena: fetched back)."""
    from tests.test_sim_parity import _check_flat
    held = []

    def device_copy(arena):
        d = lib.device_copy(arena)
        held.append(d)
        return d, d

    try:
        assert _check_flat(lib, device_copy) == 6 * 2 * 36
    finally:
        for d in held:
            lib.device_free(d)


def test_flat_batch_full_size_config2(lib, monkeypatch):
    """Config 2 at its full size through the flat entry point from a page-locked arena, production occupancy rules: all 65 536
    results equal the pointer entry point's."""
    monkeypatch.delenv("KSW2AMD_SIMDS", raising=False)
    n = 65536
    q, t = synth.fast_fixed(2, n, 512, 512, sub=0.05, ind=0.06)
    mat = synth.simple_mat(5, 2, 4, -1)
    b = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=po.SCORE_ONLY)
    p = b.plan(False); p.run(); r1 = p.fetch_raw().copy(); p.close()
    fb = lib.make_flat_batch(q, t, mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=po.SCORE_ONLY)
    fb.register()
    try:
        p = fb.plan(False); p.run(); r2 = p.fetch_raw().copy(); p.close()
        ez = (ka.KswExtz * n)()
        lib._check(lib.lib.ksw2amd_extz_batch_flat(None, ctypes.byref(fb.sc), n, ctypes.byref(fb.flat), ez))
    finally:
        fb.unregister()
    assert np.array_equal(r1[:, :10], r2[:, :10])
    sc = np.array([ez[i].score for i in range(0, n, 7)])
    assert np.array_equal(sc, r1[::7, 8])


@pytest.mark.parametrize("defer", ["0", "1"])
def test_deferred_argmax_forced_on_and_off(lib, defer, monkeypatch):
    """KSW2AMD_DEFER=1 / 0: the exact score-only single-gap kernels that track row maxima without their columns, stream their
    checkpoints and leave max_q / mte_q to k2a_argmax_kernel (all four geometries, plain and re-based; alignments in which a Z-drop
    cannot be ruled out come back as inexact and are re-run), against the ordinary kernels and the oracle."""
    from tests.test_sim_parity import _check_deferred_argmax
    monkeypatch.setenv("KSW2AMD_DEFER", defer)
    for seed in (41, 42):
        _check_deferred_argmax(lib, defer == "1", seed=seed)


def test_headline_kernel_at_scale_unforced(lib, monkeypatch):
    """The headline workload's own launch, nothing forced: 3 200 pairs of 10 000 x 10 000, band 500, Z-drop 400, score only,
    under the production occupancy rules.  The host must take the deferred-arg-max kernels on its own; the whole batch equals,
    field by field, the ordinary kernels with their code planes in LDS (KSW2AMD_DEFER=0: the launcher's own choice at >= 1.5
    wavefronts per SIMD) and in registers (KSW2AMD_LDSCODES=0), and every 50th pair equals the oracle.  A fifth of the pairs get a
    random tail so that Z-drop fires (those are the ones the deferred kernels hand back)."""
    monkeypatch.delenv("KSW2AMD_SIMDS", raising=False)
    monkeypatch.delenv("KSW2AMD_LDSCODES", raising=False)
    monkeypatch.delenv("KSW2AMD_DEFER", raising=False)
    n = 3200
    q, t = synth.fast_fixed(6, n, 10000, 10000, sub=0.05, ind=0.06, tail_random_frac=0.25, tail_pairs=0.2)
    mat = synth.simple_mat(5, 2, 4, -1)
    b = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=500, zdrop=400, flag=po.SCORE_ONLY)
    p = b.plan(False)
    d = p.describe()
    assert len(d) == 1 and d[0]["kernel"] == "pk" and (d[0]["G"], d[0]["C"], d[0]["rebased"], d[0]["nomax"]) == (64, 16, 1, 0), d
    assert d[0]["form"] == "defer" and d[0]["tasks"] == n // 2, d
    r0 = lib.rerun_count()
    p.run(); r1 = p.fetch_raw().copy(); p.close()
    assert lib.rerun_count() == r0                               # nothing is handed back: the third pass (k2a_zscan_kernel) settles the frozen books on the device
    monkeypatch.setenv("KSW2AMD_DEFER", "0")
    p = b.plan(False)
    assert p.describe()[0]["form"] == "ldscodes"
    p.run(); r2 = p.fetch_raw().copy(); p.close()
    monkeypatch.setenv("KSW2AMD_LDSCODES", "0")
    p = b.plan(False)
    assert p.describe()[0]["form"] == "registers"
    p.run(); r0 = p.fetch_raw().copy(); p.close()
    assert np.array_equal(r0[:, :11], r1[:, :11]) and np.array_equal(r0[:, :11], r2[:, :11])
    assert 0 < int(r1[:, 1].sum()) < n                         # some pairs Z-dropped, most did not
    names = ["max", "zdropped", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "reach_end"]
    for i in range(0, n, 50):
        exp = po.align("oracle", "extz2", q[i], t[i], mat, 4, 2, w=500, zdrop=400, flag=po.SCORE_ONLY)
        assert all(int(r1[i, k]) == exp[f] for k, f in enumerate(names)), (i, exp, r1[i])


def test_automatic_kernel_choices_at_scale(lib, monkeypatch):
    """The host's and launcher's own choices, which need thousands of wavefronts to trigger: unique-shape long reads go to the
    solo kernel, a big two-piece traceback class takes the LDS row form (>= 1.5 per SIMD), a same-shape class of fewer reads
    than the device has SIMDs goes to the solo kernel too.
    A sample of each batch against the oracle."""
    monkeypatch.delenv("KSW2AMD_SIMDS", raising=False)          # the occupancy rules as in production
    rng = np.random.Generator(np.random.PCG64(4242))
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    n = 6144
    pairs = synth.ragged_pairs(rng, n, 3000, 5000, sub=0.05, ind=0.06)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    p = lib.make_batch(qs, ts, mat, q, e, q2, e2, w=300, zdrop=400, flag=po.SCORE_ONLY).plan(False)
    assert p.packed_pairs() > n * 9 // 10                      # solo tasks count as packed
    p.close()
    check_batch(lib, False, qs, ts, mat, q, e, q2, e2, w=300, zdrop=400, flag=po.SCORE_ONLY, sample=list(range(0, n, 211)))
    n = 3200
    qs, ts = synth.fixed_batch(903, n, 2000, 1990, sub=0.05, ind=0.1, tail_random_frac=0.3, tail_pairs=0.3)
    zd = rng.choice([-1, 400, 2000], size=n)
    check_batch(lib, True, qs, ts, mat, q, e, q2, e2, w=330, zdrop=zd, flag=0, sample=list(range(0, n, 97)))
    # a few hundred same-shape long reads: every read a wavefront (and a SIMD) of its own instead of two per wavefront
    n = 600
    qs, ts = synth.fixed_batch(904, n, 4000, 3990, sub=0.05, ind=0.08, tail_random_frac=0.3, tail_pairs=0.2)
    for dual, flag in ((False, po.SCORE_ONLY), (True, 0)):
        b = lib.make_batch(qs, ts, mat, q, e, q2, e2, w=400, zdrop=400, flag=flag)
        p = b.plan(dual)
        d = p.describe()
        assert len(d) == 1 and d[0]["kernel"] == "solo" and d[0]["tasks"] == n, d
        p.run(); r1 = p.fetch_raw().copy(); p.close()
        monkeypatch.setenv("KSW2AMD_SOLO", "0")
        p = b.plan(dual)
        assert p.describe()[0]["kernel"] != "solo"
        p.run(); r0 = p.fetch_raw().copy(); p.close()
        monkeypatch.delenv("KSW2AMD_SOLO")
        assert np.array_equal(r0[:, :11], r1[:, :11])
        check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=400, zdrop=400, flag=flag, sample=list(range(0, n, 41)))


def test_pairs_share_the_true_target_length(lib):
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
            check_batch(lib, dual, qs, ts, mat, 4, 2, 13, 1, w=5, zdrop=-1, end_bonus=10, flag=flag)


def test_eqx_golden(lib):
    """KSW_EZ_EQX (ksw2_extd2_sse.c:399-406): all committed runs of the reference, batched per scoring; plus single calls that
    reuse one ksw_extz_t (the =/X list is longer than the M list it replaces: the buffer must grow by the reference's rule)."""
    ec = gu.EqxCases()
    groups = {}
    for k in range(ec.n):
        c = ec.case(k)
        groups.setdefault((c["mat"].tobytes(), c["gq"], c["ge"], c["gq2"], c["ge2"]), []).append(c)
    n = 0
    for (_, gq, ge, gq2, ge2), cs in groups.items():
        res = lib.extd_batch([c["q"] for c in cs], [c["t"] for c in cs], cs[0]["mat"], gq, ge, gq2, ge2, w=np.array([c["w"] for c in cs]), zdrop=-1,
                             end_bonus=np.array([c["end_bonus"] for c in cs]), flag=np.array([c["flag"] for c in cs]))
        for c, r in zip(cs, res):
            bad, same = gu.EqxCases.check(c, r)
            assert not bad and same, (bad, c["flag"])
            n += 1
    assert n == ec.n >= 300
    ez = ka.KswExtz()
    for k in range(0, ec.n, 9):
        c = ec.case(k)
        r = lib.extd2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=-1, end_bonus=c["end_bonus"], flag=c["flag"], ez=ez)
        bad, _ = gu.EqxCases.check(c, r)
        assert not bad, (k, bad)
        assert r["m_cigar"] >= r["n_cigar"] and (r["m_cigar"] & (r["m_cigar"] - 1)) == 0
    ka._libc.free(ctypes.cast(ez.cigar, ctypes.c_void_p))


def test_50k_anchor(lib):
    """The README's 50 000 x 50 000 pair through the generation-serial kernels (unbanded: 49 generations of 1 024 rows) and the
    re-based packed kernels (band 500): the reference's answers, CIGAR by md5 (3 995 operations)."""
    ka_ = gu.known_answers()["t2q2_50k"]
    _, ts = gu.read_fasta("t2.fa.gz")
    _, qs = gu.read_fasta("q2.fa.gz")
    q, t = qs[0], ts[0]
    mat = gu.simple_mat(5, 2, 4, 0)
    for exp in ka_:
        if exp["func"] == "ksw_extz":
            res = lib.extz(q, t, mat, 4, 2, w=exp["w"], zdrop=exp["zdrop"], flag=exp["flag"])
            assert not diff(exp, res, gu.FIELDS), exp["func"]
            assert (res["score"], res["max"], res["max_t"], res["max_q"]) == (69932, 70010, 49962, 49999)
        elif exp["func"] == "ksw_extz2_sse":
            res = lib.extz2(q, t, mat, 4, 2, w=exp["w"], zdrop=exp["zdrop"], flag=exp["flag"])
            assert not diff(exp, res, gu.SSE_LOOSE_FIELDS), (exp["w"], exp["flag"])
            if not (exp["flag"] & po.SCORE_ONLY):
                assert hashlib.md5((gu.cigar_string(res["cigar"]) + "\n").encode()).hexdigest()[:12] == exp["cigar_md5_12"]
        else:
            res = lib.extd2(q, t, mat, 4, 2, 13, 1, w=exp["w"], zdrop=exp["zdrop"], flag=exp["flag"])
            assert not diff(exp, res, gu.SSE_LOOSE_FIELDS)


@pytest.mark.parametrize("mode", ["short", "long"])
def test_fuzz_slice(mode):
    """A time-boxed slice of tools/scripts/fuzz_gpu.py (the soak that found round 1's pairing bug): ragged and same-shape batches of
    all four functions against the oracle under every kernel-selection switch -- including none at all, i.e. the production
    occupancy rules (the script's environment does not carry this module's KSW2AMD_SIMDS=0) -- and through the worker pool."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("KSW2AMD_")}
    args = ["40", "20260002"] if mode == "short" else ["25", "20260003", "long"]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scripts", "fuzz_gpu.py")] + args, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("cigar,extra", [(0, {}), (1, {}), (0, {"KSW2AMD_COALESCE_WINDOW_US": "0"}), (1, {"KSW2AMD_COALESCE_SLOTS": "1"}),
                                         (0, {"KSW2AMD_COALESCE_SLOTS": "8", "KSW2AMD_COALESCE_PLAIN_STREAMS": "1"})],
                         ids=["score", "cigar", "no-window", "cigar-1slot", "8-plain-slots"])
def test_unchanged_threaded_caller_is_coalesced(cigar, extra):
    """tools/coalesce-bench: 64 host threads calling ksw_extz2_sse / ksw_extd2_sse one pair at a time (the minimap2 pattern).  The
    library batches concurrent calls behind the unchanged symbols; every call returns exactly what the batch entry point returns."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "coalesce-bench")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "tools"), "coalesce-bench"], check=True, capture_output=True)
    env = {k: v for k, v in os.environ.items() if not k.startswith("KSW2AMD_")}
    env.update(extra)                           # (the collection window off, one slot, eight slots on ordinary streams: same results)
    r = subprocess.run([exe, "64", "300", "512", "64", str(cigar)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-1000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["mismatches"] == 0 and d["calls"] == 64 * 300
    assert d["coalesced_calls"] > d["coalesced_batches"] > 0


def test_sse_compatible_mode(lib):
    """The opt-in SSE-compatible mode on the GPU (k2a_ssec_kernel): all 1500 golden cases of the unmodified ksw_extz2_sse /
    ksw_extd2_sse -- narrow bands whose blocks leak, anti-diagonal Z-drop, padded mte_q, KSW_EZ_APPROX_MAX with and without
    APPROX_DROP, swapped gap pieces -- every field and the CIGAR; mixed batches; the process-wide switch; long banded reads."""
    from tests import sse_compat_util as su
    assert su.check_golden(lib) >= 1500
    su.check_routing(lib)
    su.check_long(lib, n=6, length=6000, w=150)


def test_device_resident_ext_shards(lib):
    """X-drop and splice-aware shards that lie in device memory (what RCCL delivers on a receiving rank) through
    ksw2amd_ext?_batch_device -- device pointers per pair, one gather kernel into the plan's arena -- against the same shards from
    host memory: a few pairs on the calling thread, and 2 600 pairs through the worker pool's chunks."""
    from ksw2_amd import parallel
    from oracle.gen_golden_extf import noisy_pair
    from oracle.gen_golden_exts import spliced_pair
    rng = np.random.Generator(np.random.PCG64(77))
    for n in (9, 2600):
        fp = [noisy_pair(rng, int(rng.integers(40, 600)), k % 3) for k in range(n)]
        meta = np.zeros((n, parallel.META), dtype=np.int32)
        meta[:, 0], meta[:, 1], meta[:, 2], meta[:, 3], meta[:, 6] = [len(a) for a, _ in fp], [len(b) for _, b in fp], 50, 80, np.arange(n)
        seq = np.concatenate([a for a, _ in fp] + [b for _, b in fp])
        dptr = lib.device_copy(seq)
        rec, _ = parallel.align_flat(lib, "extf", None, meta, dict(mch=2, mis=-4, e=2), device_base=dptr)
        rec2, _ = parallel.align_flat(lib, "extf", seq, meta, dict(mch=2, mis=-4, e=2))
        lib.device_free(dptr)
        assert np.array_equal(rec, rec2), n
        for i in range(0, n, max(1, n // 12)):
            exp = po.extf2("oracle", fp[i][0], fp[i][1], 2, -4, 2, 50, 80)
            assert exp["score"] == rec[i, 0] and exp["max"] == rec[i, 1] and exp["max_t"] == rec[i, 2], (n, i)
        sp = [spliced_pair(rng, int(rng.integers(60, 260)), k % 5 == 4) for k in range(n)]
        meta = np.zeros((n, parallel.META), dtype=np.int32)
        meta[:, 0], meta[:, 1], meta[:, 3], meta[:, 6] = [len(x[0]) for x in sp], [len(x[1]) for x in sp], 200, np.arange(n)
        seq = np.concatenate([x[0] for x in sp] + [x[1] for x in sp])
        dptr = lib.device_copy(seq)
        ssc = dict(mat=synth.simple_mat(5, 1, 2, -1), q=2, e=1, q2=32, noncan=4)
        rec, cig = parallel.align_flat(lib, "exts", None, meta, ssc, device_base=dptr)
        rec2, cig2 = parallel.align_flat(lib, "exts", seq, meta, ssc)
        lib.device_free(dptr)
        assert np.array_equal(rec, rec2) and np.array_equal(cig, cig2), n


def test_linear_xdrop_group_form(lib, monkeypatch):
    """k2a_extf_grp_kernel on the GPU (tests/test_sim_parity._check_extf_group_form), and a batch of 4 099 extensions of 1 000 x 1 000
    at band 100 -- the bench workload's shape, a last wavefront with three empty groups -- against the position-per-lane kernels."""
    from tests.test_sim_parity import _check_extf_group_form
    _check_extf_group_form(lib, monkeypatch, rounds=60, maxlen=6000)
    from oracle.gen_golden_extf import noisy_pair
    rng = np.random.Generator(np.random.PCG64(5))
    qs, ts = zip(*[noisy_pair(rng, 1000, k % 3) for k in range(4099)])
    out = []
    for off in ("", "0"):
        monkeypatch.setenv("KSW2AMD_EXTF_GRP", off)
        out.append(lib.extf_batch(list(qs), list(ts), 2, -4, 2, w=100, xdrop=[-1 if k % 2 else 150 for k in range(4099)]))
    assert any(r["zdropped"] for r in out[0]) and not all(r["zdropped"] for r in out[0])
    for a, b in zip(*out):
        assert not diff(a, b, gu.FIELDS)


def test_linear_xdrop_wide_group_forms(lib, monkeypatch):
    """k2a_extf_grp_kernel<32> / <64> on the GPU (tests/test_sim_parity._check_extf_wide_group_forms), and 1 027 extensions of 3 000 x 3 000
    at bands 300 and 700 -- a last wavefront with an empty second group -- against the register-window / LDS kernels."""
    from tests.test_sim_parity import _check_extf_wide_group_forms
    _check_extf_wide_group_forms(lib, monkeypatch, rounds=30, maxlen=7000)
    from oracle.gen_golden_extf import noisy_pair
    rng = np.random.Generator(np.random.PCG64(6))
    base = [noisy_pair(rng, 3000, k % 3) for k in range(64)]
    qs, ts = [base[k % 64][0] for k in range(1027)], [base[k % 64][1] for k in range(1027)]
    for w in (300, 700):
        out = []
        for env in ("", "1"):
            monkeypatch.setenv("KSW2AMD_EXTF_GRP", env)
            out.append(lib.extf_batch(qs, ts, 2, -4, 2, w=w, xdrop=[-1 if k % 2 else 400 for k in range(1027)]))
        for a, b in zip(*out):
            assert not diff(a, b, gu.FIELDS)


def test_sse_compatible_register_form(lib, monkeypatch):
    """k2a_ssec_blk_kernel on the GPU (score-only SSE-compatible tasks, state in registers, H in an LDS ring): bands of 1 to 960
    positions, targets several rings long, both gap models, exact and approximate maxima, Z-drop -- against the oracle and against
    the position-per-lane kernel; the golden set with the form off; 10 k reads at band 500 against the position-per-lane kernel."""
    from tests import sse_compat_util as su
    assert su.check_register_form(lib, monkeypatch.setenv, rounds=9, long_len=5000) > 60
    monkeypatch.setenv("KSW2AMD_SSEC_BLK", "0")
    assert su.check_golden(lib) >= 1500
    mat = synth.simple_mat(5, 2, 4, -1)
    q, t = synth.fixed_batch(21, 24, 10000, 10000, sub=0.05, ind=0.06, tail_random_frac=0.3, tail_pairs=0.3)
    for dual, flag in ((False, po.SCORE_ONLY), (True, po.SCORE_ONLY), (False, po.SCORE_ONLY | po.APPROX_MAX | po.APPROX_DROP | po.EXTZ_ONLY), (False, 0), (True, po.RIGHT | po.EXTZ_ONLY)):
        fl = np.full(24, flag | ka.KSW2AMD_EZ_SSE_COMPAT)
        out = []
        for off in ("0", ""):
            monkeypatch.setenv("KSW2AMD_SSEC_BLK", off)
            out.append(lib.extd_batch(list(q), list(t), mat, 4, 2, 24, 1, w=500, zdrop=400, flag=fl) if dual else lib.extz_batch(list(q), list(t), mat, 4, 2, w=500, zdrop=400, flag=fl))
        assert any(r["zdropped"] for r in out[0]) and not all(r["zdropped"] for r in out[0])
        for a, b in zip(*out):
            assert not diff(a, b, gu.FIELDS + ([] if flag & po.SCORE_ONLY else ["cigar"]))


def test_packed_generation_serial(lib, monkeypatch):
    """The packed generation-serial class on the GPU (k2a_fill_pkmp_kernel: four wavefronts pipeline a task's generations, one
    workgroup barrier per 64 steps): same-shape batches whose band no resident geometry holds -- 2 to 25 generations, unbanded
    and wide bands, both gap models, gap alignment modes, Z-drop with diverging tails, odd task counts -- against the oracle;
    and against the int32 generation-serial kernels on a 25 k x 25 k pair."""
    mat = synth.simple_mat(5, 2, 4, -1)
    cases = [(2100, -1, False, 0, -1, 6), (2600, -1, False, po.SCORE_ONLY, -1, 10), (2200, 1100, True, 0, -1, 4), (2300, -1, True, po.RIGHT, 300, 6),
             (3000, 1300, False, po.RIGHT, -1, 2), (2301, -1, True, po.SCORE_ONLY, -1, 3), (2403, 1500, True, po.SCORE_ONLY, 300, 1), (4300, -1, False, po.EXTZ_ONLY, 400, 4), (5200, 1200, True, 0, 400, 4), (9000, -1, False, 0, -1, 2)]
    for L, w, dual, flag, zd, n in cases:
        q, t = synth.fixed_batch(9, (n + 1) // 2, L, L + 37, sub=0.05, ind=0.08, tail_random_frac=0.3 if zd >= 0 else 0.0, tail_pairs=0.5 if zd >= 0 else 0.0)
        qs, ts = [q[i // 2] for i in range(n)], [t[i // 2] for i in range(n)]
        p = lib.make_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag).plan(dual)
        assert p.packed_pairs() == n
        p.close()
        check_batch(lib, dual, qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=flag)
    q, t = synth.fixed_batch(10, 1, 25000, 25300, sub=0.04, ind=0.05)
    qs, ts = [q[0], q[0]], [t[0], t[0]]
    for flag in (0, po.SCORE_ONLY):
        a = lib.extz_batch(qs, ts, mat, 4, 2, w=-1, zdrop=-1, flag=flag)
        monkeypatch.setenv("KSW2AMD_NO_PKMP", "1")
        b = lib.extz_batch(qs, ts, mat, 4, 2, w=-1, zdrop=-1, flag=flag)
        monkeypatch.delenv("KSW2AMD_NO_PKMP")
        assert not diff(a[0], b[0], gu.FIELDS + ["cigar"]) and not diff(a[1], b[1], gu.FIELDS + ["cigar"])


def test_two_rank_nccl_scatter_gather(lib):
    """ksw2_amd/parallel.py over RCCL: two ranks on two GPUs, rank 0 scatters a ragged batch point to point and gathers records and
    CIGARs.  Needs two devices (the 1-GPU test box skips it; the gloo twin in tests/test_sharding_gloo.py runs everywhere)."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
import ksw2_amd as ka
from ksw2_amd import synth, parallel
from oracle import pyoracle as po
rank = int(os.environ["RANK"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", rank))
lib = ka.library(); lib.set_device(rank)
mat = synth.simple_mat(5, 2, 4, -1)
qs = ts = None
if rank == 0:
    rng = np.random.Generator(np.random.PCG64(3))
    pairs = synth.ragged_pairs(rng, 41, 50, 900, sub=0.05, ind=0.1)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
res = parallel.sharded(lib, "extd", qs, ts, dict(mat=mat, q=4, e=2, q2=24, e2=1), w=100, zdrop=200, flag=0)
ok = True
if rank == 0:
    ok = all(all(po.align("oracle", "extd2", qs[i], ts[i], mat, 4, 2, 24, 1, w=100, zdrop=200)[k] == res[i][k] for k in ka.FIELDS + ["cigar"]) for i in range(41))
# the splice-aware and the gap-linear X-drop batches shard the same way (host pointers on the receiving rank)
ss = st = None
if rank == 0:
    from oracle.gen_golden_exts import spliced_pair
    rng = np.random.Generator(np.random.PCG64(5))
    cases = [spliced_pair(rng, 300) for _ in range(13)]
    ss, st = [c[0] for c in cases], [c[1] for c in cases]
smat = synth.simple_mat(5, 1, 2, 0)
res = parallel.sharded(lib, "exts", ss, st, dict(mat=smat, q=2, e=1, q2=32, noncan=4), zdrop=-1, flag=ka.KSW_EZ_SPLICE_FOR)
if rank == 0:
    ok &= all(all(po.exts2("oracle", ss[i], st[i], smat, 2, 1, 32, 4, zdrop=-1, flag=po.SPLICE_FOR)[k] == res[i][k] for k in ka.FIELDS + ["cigar"]) for i in range(13))
fq = ft = None
if rank == 0:
    fq, ft = synth.fixed_batch(4, 33, 400, 410, sub=0.05, ind=0.03)
res = parallel.sharded(lib, "extf", fq, ft, dict(mch=2, mis=-4, e=2), w=40, zdrop=50)
if rank == 0:
    ok &= all(all(po.extf2("oracle", fq[i], ft[i], 2, -4, 2, 40, 50)[k] == res[i][k] for k in ka.FIELDS) for i in range(33))
    print("NCCL_SHARD_OK" if ok else "NCCL_SHARD_BAD")
dist.destroy_process_group()
''' % root
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", worker], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "NCCL_SHARD_OK" in outs[0][0], outs


def test_linear_xdrop_one_extension_per_lane(lib, monkeypatch):
    """k2a_extf_lane_kernel on the GPU: all 2000 reference cases of ksw_extf2_sse in batches of mixed shapes (groups of 64 with very
    different lengths and bands: divergent lanes), a same-shape batch large enough to fill a few wavefronts per SIMD, and the host's
    own choice of the form on a big narrow-band batch."""
    from tests.test_sim_parity import _check_extf_lane_forms
    monkeypatch.setenv("KSW2AMD_EXTF_LANE", "1")
    fc = gu.ExtfCases()
    cases = [fc.case(k) for k in range(fc.n)]
    _check_extf_lane_forms(lib, cases, monkeypatch)          # LDS rings of 32 / 48 / 64 rows and the HBM-scratch form, by plan diagnostics
    monkeypatch.delenv("KSW2AMD_EXTF_LANE")
    q, t = synth.fast_fixed(8, 200000, 200, 200, sub=0.05, ind=0.02)
    res = lib.extf_batch(list(q), list(t), 2, -4, 2, w=20, xdrop=40)          # 200 000 extensions, 21 positions in the band: the lane form by itself
    for i in range(0, 200000, 1999):
        assert not diff(res[i], po.extf2("oracle", q[i], t[i], 2, -4, 2, 20, 40), gu.FIELDS), i


def test_packed_kernels_on_generic_matrices(lib, monkeypatch):
    """KSW_EZ_GENERIC_SC through the packed kernels (round 5, column profiles: ksw2_lane_pk.h; the reference takes any matrix at its
    full rate, ksw2_extz2_sse.c:142-143): transition / transversion, asymmetric random, all-different 5 x 5 matrices and alphabets of
    four and three codes through the resident geometries (plain, re-based), the solo kernel, the generation-serial class and, forced,
    the deferred arg-max; score-only and both traceback modes, both gap models, Z-drops, wildcards in the query -- every field and
    CIGAR against the oracle, and the plan's description asserts that no pair left the packed kernels for its matrix."""
    from tests.test_sim_parity import _check_generic_packed
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")                    # (small test batches: keep the occupancy rules from demoting them)
    _check_generic_packed(lib)
    monkeypatch.setenv("KSW2AMD_DEFER", "1")
    assert "pk-defer" in _check_generic_packed(lib)


def test_frozen_books_with_wildcards(lib, monkeypatch):
    """tests/test_sim_parity.py::_check_frozen_books_with_wildcards on the device (round 5's fuzz find: the third pass of the deferred
    arg-max must keep the fill's wildcard report)."""
    from tests.test_sim_parity import _check_frozen_books_with_wildcards
    _check_frozen_books_with_wildcards(lib, monkeypatch)


def test_target_wildcards_stay_packed(lib, monkeypatch):
    """tests/parity_util.py::check_target_wildcards on the device (round 6): a target's wildcard code is a row of every packed kernel
    family -- nothing demoted to int32, nothing handed back -- with KSW2AMD_TN=0 and a matrix whose wildcard row varies on the old rule."""
    from tests.parity_util import check_target_wildcards
    check_target_wildcards(lib, monkeypatch.setenv, monkeypatch.delenv, scale=3)


def test_uniform_plans(lib, monkeypatch):
    """tests/test_sim_parity.py::_check_uniform_plans on the device: small forced batches of three shapes (plain class, deferred arg-max
    forced on and off, Z-drops, wildcards, the fault hook), then config 2 at full size through the default routing: ONE uniform
    streamed plan, every record against the general path's."""
    from tests.test_sim_parity import _check_uniform_plans
    _check_uniform_plans(lib, monkeypatch, [(2048, 60, 64, 10, -1, None), (4098, 300, 290, 30, 40, 1), (2048, 700, 700, 300, 100, 1), (3000, 512, 512, 64, 100, 0)])
    for k in ("KSW2AMD_UNIFORM", "KSW2AMD_STREAM_PIECE_KB", "KSW2AMD_STREAM_SLEEP_US", "KSW2AMD_STREAM_FAULT", "KSW2AMD_STREAM_TIMEOUT_MS", "KSW2AMD_DEFER", "KSW2AMD_SIMDS"):
        monkeypatch.delenv(k, raising=False)
    q, t = synth.fast_fixed(2, 65536, 512, 512, sub=0.05, ind=0.06)
    mat = synth.simple_mat(5, 2, 4, -1)
    b = lib.make_batch(q, t, mat, 4, 2, 0, 0, w=64, zdrop=-1, end_bonus=0, flag=po.SCORE_ONLY)
    s0 = lib.stream_stats()
    ez1, v1 = _raw_batch(lib, False, b)
    assert lib.stream_stats()["streamed_plans"] == s0["streamed_plans"] + 1      # the default routing took the uniform plan
    monkeypatch.setenv("KSW2AMD_UNIFORM", "0")
    ez0, v0 = _raw_batch(lib, False, b)
    try:
        for f in _EZ_DT.names:
            if f not in ("cigar", "m_cigar"):
                assert (v0[f] == v1[f]).all(), (f, np.flatnonzero(v0[f] != v1[f])[:5])
        for i in range(0, 65536, 4099):
            exp = po.align("oracle", "extz2", q[i], t[i], mat, 4, 2, w=64, zdrop=-1, end_bonus=0, flag=po.SCORE_ONLY)
            assert exp["score"] == v1["score"][i] and exp["max"] == (v1["max_zd"][i] & 0x7fffffff) and exp["max_q"] == v1["max_q"][i] and exp["mte_q"] == v1["mte_q"][i], i
    finally:
        _free_raw(ez0, v0); _free_raw(ez1, v1)


def test_lane_primitives_match_their_simulator_twins(tmp_path):
    """tools/probe/lane_ops_probe.hip: every register primitive of the packed kernels (inline asm / builtins: v_perm_b32 with every
    selector byte value, v_bitop3_b32 0xe4, v_pk_mad_i16, v_pk_maximum3_f16, v_pk_ashrrev_i16 ...) on the device against the C twin the
    lock-step simulator runs, 4 M operand triples each, bit for bit.  The CPU test tier stands on those twins."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    twin, obj, exe = str(tmp_path / "twin.o"), str(tmp_path / "probe.o"), str(tmp_path / "probe")
    subprocess.run(["g++", "-O2", "-std=c++17", "-c", "-o", twin, os.path.join(root, "tools/probe/lane_ops_twin.cpp")], check=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", obj, os.path.join(root, "tools/probe/lane_ops_probe.hip")], check=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-o", exe, obj, twin], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all primitives agree" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("flat", [False, True])
def test_streamed_plans_forced_on_and_off(lib, monkeypatch, flat):
    """Streamed plans (ksw2_host_plan.c "streamed plans"): one launch per packed class, started under the
    upload, whose wavefronts wait in front of their task for the arena's pieces it lies in (k2a_queue_wait).  Forced on (KSW2AMD_STREAM=1) with small pieces and a
    slowed-down upload, so the wavefronts really wait for their watermarks; against the forced-off run on every pair and the oracle on
    a sample; then the fault hook with a 20 ms timeout: the launch must give up, the plan must be run again behind its upload and still
    return the same results (a kernel of this library never spins without a bound).  Score-only classes incl. the deferred arg-max,
    a CIGAR class, both gap models, wildcard pairs (reported by the kernels, re-run in one batch), an odd pair count."""
    mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
    cases = [(20001, 500, 520, 64, po.SCORE_ONLY, False, 40, (11, 9000, 20000)), (6000, 1200, 1200, 300, po.SCORE_ONLY, False, 100, ()),
             (4000, 600, 600, 100, 0, True, 100, (5,)), (3000, 3000, 3000, 500, po.SCORE_ONLY, False, 400, ())]
    for ci, (n, ql, tl, w, flag, dual, zd, wild) in enumerate(cases):
        qs, ts = synth.fixed_batch(300 + ci, n, ql, tl, sub=0.05, ind=0.06)
        qs, ts = [np.array(x) for x in qs], [np.array(x) for x in ts]
        for i in wild:
            ts[i][tl // 3] = 4
        if ci == 3:                                        # diverging tails: Z-drops, inexact pairs of the deferred arg-max
            rng = np.random.Generator(np.random.PCG64(5))
            for i in range(0, n, 5):
                qs[i][ql // 2:] = rng.integers(0, 4, ql - ql // 2, dtype=np.uint8)

        def run(**env):
            for k in ("KSW2AMD_STREAM", "KSW2AMD_STREAM_PIECE_KB", "KSW2AMD_STREAM_SLEEP_US", "KSW2AMD_STREAM_FAULT", "KSW2AMD_STREAM_TIMEOUT_MS"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, str(v))
            s0 = lib.stream_stats()
            if flat:
                fb = lib.make_flat_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=0, flag=flag)
                r = fb.run_oneshot(dual)
            else:
                r = lib.extd_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, flag=flag) if dual else lib.extz_batch(qs, ts, mat, q, e, w=w, zdrop=zd, flag=flag)
            s1 = lib.stream_stats()
            return r, s1["streamed_plans"] - s0["streamed_plans"], s1["aborted_runs"] - s0["aborted_runs"]

        off, ns, na = run(KSW2AMD_STREAM=0)
        assert ns == 0 and na == 0
        streams = bool(flag & po.SCORE_ONLY)              # the queue builds of the kernels exist for the score-only classes; a CIGAR batch keeps ordinary launches behind its pieces
        # A decoy batch of the same shapes but other bases goes through the same path first: the device arena, the staging and the
        # caches then hold ITS bytes, and a wavefront that read anything that was not uploaded for this batch -- too early, or out of
        # a stale cache line -- would differ.  (Running the same batch twice hides exactly that: round 4's missing acquire behind the
        # watermark poll passed such a test and failed the fuzz script.)
        keep = (qs, ts)
        qs, ts = [np.random.default_rng(7 + ci).integers(0, 4, len(x), dtype=np.uint8) for x in keep[0]], [np.random.default_rng(9 + ci).integers(0, 4, len(x), dtype=np.uint8) for x in keep[1]]
        run(KSW2AMD_STREAM=1, KSW2AMD_STREAM_PIECE_KB=512, KSW2AMD_STREAM_SLEEP_US=200)
        qs, ts = keep
        on, ns, na = run(KSW2AMD_STREAM=1, KSW2AMD_STREAM_PIECE_KB=512, KSW2AMD_STREAM_SLEEP_US=200)
        assert (ns >= 1) == streams and na == 0, (ci, ns, na)
        bad = [i for i in range(n) if diff(off[i], on[i])]
        assert not bad, (ci, flat, bad[:5])
        qs, ts = [np.random.default_rng(17 + ci).integers(0, 4, len(x), dtype=np.uint8) for x in keep[0]], [np.random.default_rng(19 + ci).integers(0, 4, len(x), dtype=np.uint8) for x in keep[1]]
        run(KSW2AMD_STREAM=1)                              # (decoy once more, default pieces, nothing slowed down)
        qs, ts = keep
        on2, ns, na = run(KSW2AMD_STREAM=1)
        bad = [i for i in range(n) if diff(off[i], on2[i])]
        assert not bad, (ci, flat, "default pieces", bad[:5])
        if streams and ci in (1, 3):
            # ... behind a plan of much longer reads: the recycled result records and checkpoint blocks then hold ITS rows and offsets, and an
            # arg-max pass that ran for tasks the aborted fill never started would index megabytes past this plan's blocks (round 4's advice)
            bq, bt = synth.fixed_batch(900 + ci, max(64, n // 16), ql * 4, tl * 4, sub=0.05, ind=0.06)
            qs, ts = [np.array(x) for x in bq], [np.array(x) for x in bt]
            run(KSW2AMD_STREAM=1)
            qs, ts = keep
        flt, ns, na = run(KSW2AMD_STREAM=1, KSW2AMD_STREAM_PIECE_KB=512, KSW2AMD_STREAM_FAULT=1, KSW2AMD_STREAM_TIMEOUT_MS=20)
        assert (ns >= 1 and na >= 1) == streams, (ci, ns, na)
        bad = [i for i in range(n) if diff(off[i], flt[i])]
        assert not bad, (ci, flat, "fault", bad[:5])
        for i in list(range(0, n, max(1, n // 40))) + list(wild):
            exp = po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=0, flag=flag)
            assert not diff(exp, on[i]), (ci, flat, i)


_EZ_DT = np.dtype({"names": ["max_zd", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "m_cigar", "n_cigar", "reach_end", "cigar"],
                   "formats": ["<u4"] + ["<i4"] * 10 + ["<u8"], "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32, 36, 40, 48], "itemsize": 56})


def _raw_batch(lib, dual, batch):
    """One call of the batch entry point into a ctypes record array (no Python objects per pair: for batches of 10^5 pairs with CIGARs);
    returns (ez, structured numpy view of the records)."""
    ez = (ka.KswExtz * batch.n)()
    f = lib.lib.ksw2amd_extd_batch if dual else lib.lib.ksw2amd_extz_batch
    lib._check(f(None, ctypes.byref(batch.sc), batch.n, batch.pairs, ez))
    return ez, np.frombuffer(ez, dtype=_EZ_DT)


def _free_raw(ez, view):
    for p in view["cigar"]:
        if p:
            ka._libc.free(ctypes.c_void_p(int(p)))


def _cigar_words(view, i):
    n = int(view["n_cigar"][i])
    return np.ctypeslib.as_array(ctypes.cast(int(view["cigar"][i]), ctypes.POINTER(ctypes.c_uint32)), (n,)) if n else np.zeros(0, np.uint32)


def test_cfg4_full_size_properties(lib):
    """BASELINE config 4 at its full size: MT-human x MT-orang replicated 4 096 x, unbanded extz2 with CIGAR, through ksw2amd_extz_batch
    (four device-filling plans of the packed generation-serial class).  Every replica's record equals the reference's known answer
    (score 16102, max 17054 at (16568, 16024)) and every CIGAR equals the first replica's word for word, whose md5 is the
    golden ea0524d904ed."""
    ka_ = gu.known_answers()
    exp = [r for r in ka_["mt"] if r["func"] == "ksw_extz" and r["w"] == -1 and r.get("flag", 0) == 0 and r.get("zdrop", -1) == -1][0]
    assert exp["cigar_md5_12"] == "ea0524d904ed"
    _, ts = gu.read_fasta("MT-human.fa")
    _, qs = gu.read_fasta("MT-orang.fa")
    n = 4096
    q1, t1 = np.ascontiguousarray(qs[0]), np.ascontiguousarray(ts[0])
    lib.release_cache()
    b = lib.make_batch([q1] * n, [t1] * n, gu.simple_mat(5, 2, 4, 0), 4, 2, 0, 0, w=-1, zdrop=-1, end_bonus=0, flag=0)
    ez, v = _raw_batch(lib, False, b)
    try:
        for f, name in (("score", "score"), ("max_q", "max_q"), ("max_t", "max_t"), ("mqe", "mqe"), ("mqe_t", "mqe_t"), ("mte", "mte"), ("mte_q", "mte_q"),
                        ("reach_end", "reach_end"), ("n_cigar", "n_cigar")):
            assert (v[f] == exp[name]).all(), (name, np.flatnonzero(v[f] != exp[name])[:5])
        assert ((v["max_zd"] & 0x7fffffff) == exp["max"]).all() and ((v["max_zd"] >> 31) == exp["zdropped"]).all()
        c0 = _cigar_words(v, 0).copy()
        s0 = gu.cigar_string([int(x) for x in c0])
        assert hashlib.md5((s0 + "\n").encode()).hexdigest()[:12] == "ea0524d904ed"
        for i in range(1, n):
            assert np.array_equal(_cigar_words(v, i), c0), i
    finally:
        _free_raw(ez, v)
        lib.release_cache()


def test_cfg5_share_properties(lib, monkeypatch):
    """BASELINE config 5 at the per-GPU share it states (1 M pairs over 8 GPUs = 125 000 ONT-like pairs, qlen in [300, 20 000], 15 %
    indels, band 500, extd2, Z-drop 400, CIGAR; production occupancy rules: seven device-filling plans, the solo kernel's rules, the
    LPT order).  The default launch, KSW2AMD_SOLO=0 and KSW2AMD_NO_PK=1 (the int32 kernels) agree on every record and every CIGAR; every
    CIGAR spans exactly the prefixes its record names and re-scores to the reported score / maximum (oracle/kso_cigar_score); every
    500th pair equals the oracle in every field."""
    monkeypatch.delenv("KSW2AMD_SIMDS", raising=False)
    lib.release_cache()                                 # the buffers the earlier tests' plans left in the thread's cache (config 4: 140 GB of direction codes)
    n = 125000
    qs, ts = synth.fast_ragged(5, n, 300, 20000, sub=0.03, ind=0.15, maxdiff=450)
    mat = synth.simple_mat(5, 2, 4, -1)
    b = lib.make_batch(qs, ts, mat, 4, 2, 24, 1, w=500, zdrop=400, end_bonus=0, flag=0)
    ez0, v0 = _raw_batch(lib, True, b)
    olib = po.oracle_lib()
    olib.kso_cigar_score.restype = ctypes.c_int
    olib.kso_cigar_score.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    try:
        ndrop = 0
        qu, tu = ctypes.c_int(0), ctypes.c_int(0)
        for i in range(n):
            sc = olib.kso_cigar_score(int(v0["n_cigar"][i]), int(v0["cigar"][i]), len(qs[i]), qs[i].ctypes.data, len(ts[i]), ts[i].ctypes.data, 5, mat.ctypes.data,
                                      4, 2, 24, 1, ctypes.byref(qu), ctypes.byref(tu))
            if v0["max_zd"][i] >> 31:
                ndrop += 1
                assert (qu.value, tu.value) == (v0["max_q"][i] + 1, v0["max_t"][i] + 1) and sc == (v0["max_zd"][i] & 0x7fffffff), i
            else:
                assert (qu.value, tu.value) == (len(qs[i]), len(ts[i])) and sc == v0["score"][i], i
        assert ndrop < n // 20, ndrop
        for env in ({"KSW2AMD_SOLO": "0"}, {"KSW2AMD_NO_PK": "1"}):
            for k, val in env.items():
                monkeypatch.setenv(k, val)
            ez1, v1 = _raw_batch(lib, True, b)
            try:
                for f in _EZ_DT.names:
                    if f not in ("cigar", "m_cigar"):
                        assert (v0[f] == v1[f]).all(), (env, f, np.flatnonzero(v0[f] != v1[f])[:5])
                for i in range(n):
                    assert np.array_equal(_cigar_words(v0, i), _cigar_words(v1, i)), (env, i)
            finally:
                _free_raw(ez1, v1)
            for k in env:
                monkeypatch.delenv(k)
        for i in range(0, n, 500):
            exp = po.align("oracle", "extd2", qs[i], ts[i], mat, 4, 2, 24, 1, w=500, zdrop=400, end_bonus=0, flag=0)
            got = dict(score=int(v0["score"][i]), max=int(v0["max_zd"][i] & 0x7fffffff), zdropped=int(v0["max_zd"][i] >> 31), max_q=int(v0["max_q"][i]), max_t=int(v0["max_t"][i]),
                       mqe=int(v0["mqe"][i]), mqe_t=int(v0["mqe_t"][i]), mte=int(v0["mte"][i]), mte_q=int(v0["mte_q"][i]), reach_end=int(v0["reach_end"][i]),
                       n_cigar=int(v0["n_cigar"][i]), cigar=[int(x) for x in _cigar_words(v0, i)])
            assert not diff(exp, got, CMP_FIELDS), i
    finally:
        _free_raw(ez0, v0)
        lib.release_cache()


def test_set_devices_two_worker_sets_on_one_gpu(lib, monkeypatch):
    """ksw2amd_set_devices on real hardware (the in-process multi-GPU form a C caller would use; tests/test_host_pipeline.py runs it on
    simulated devices): the device list { 0, 0 } -- two entries, so the pool shards the batch with its several-devices rules (finer
    chunks, three per worker) -- must return what the single-device call returns, on a config-3 shaped batch (one shape, extd2, Z-drop,
    CIGAR) and a config-5 shaped one (ragged, band 500), every field and CIGAR, with samples against the oracle."""
    monkeypatch.delenv("KSW2AMD_SIMDS", raising=False)
    mat = synth.simple_mat(5, 2, 4, -1)
    q3, t3 = synth.fixed_batch(3, 4096, 2048, 2048, sub=0.05, ind=0.10, tail_random_frac=0.25, tail_pairs=0.10)
    q5, t5 = synth.fast_ragged(5, 3000, 300, 20000, sub=0.03, ind=0.15, maxdiff=450)
    one3 = lib.extd_batch(q3, t3, mat, 4, 2, 24, 1, w=256, zdrop=400, flag=0)
    one5 = lib.extd_batch(q5, t5, mat, 4, 2, 24, 1, w=500, zdrop=400, flag=0)
    s0 = lib.host_stats()
    lib.set_devices([0, 0])
    try:
        two3 = lib.extd_batch(q3, t3, mat, 4, 2, 24, 1, w=256, zdrop=400, flag=0)
        two5 = lib.extd_batch(q5, t5, mat, 4, 2, 24, 1, w=500, zdrop=400, flag=0)
    finally:
        lib.set_devices([])
        lib.release_cache()
    assert lib.host_stats()["pool_batches"] >= s0["pool_batches"] + 2          # both went through the pool's multi-device path
    for a, b, name in ((one3, two3, "cfg3"), (one5, two5, "cfg5")):
        bad = [i for i in range(len(a)) if diff(a[i], b[i])]
        assert not bad, (name, bad[:5])
    for i in range(0, 4096, 512):
        assert not diff(po.align("oracle", "extd2", q3[i], t3[i], mat, 4, 2, 24, 1, w=256, zdrop=400, flag=0), two3[i]), i
    for i in range(0, 3000, 500):
        assert not diff(po.align("oracle", "extd2", q5[i], t5[i], mat, 4, 2, 24, 1, w=500, zdrop=400, flag=0), two5[i]), i
