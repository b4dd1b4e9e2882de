"""CPU: the oracle (oracle/ksw2_oracle.c) against every golden vector produced by the compiled reference.

This is what pins the oracle: the known-answer table of section 4.2 and 3600 seeded
random cases, all outputs of the unmodified reference (oracle/gen_golden.py).
"""
import hashlib

import numpy as np
import pytest

from oracle import pyoracle as po
from tests import golden_util as gu


def _check(res, exp, fields):
    for f in fields:
        assert res[f] == exp[f], (f, res[f], exp[f])


def test_struct_layout():
    import ctypes
    assert ctypes.sizeof(po.Ez) == 56 and po.Ez.cigar.offset == 48 and po.Ez.score.offset == 28


def test_t1q1_known_answers():
    ka = gu.known_answers()
    _, ts = gu.read_fasta("t1.fa")
    _, qs = gu.read_fasta("q1.fa")
    mat = gu.simple_mat(5, 2, 4, 0)
    expected_extz = [(-2, 4, 1, 1, "2M1D"), (-12, 2, 0, 0, "2D7M2D4M4D"),
                     (12, 48, 35, 33, "5M2D27M6D7M2D4M3D3M3D2M2D6M"), (-18, 0, -1, -1, "11D4M"), (8, 10, 4, 4, "34M")]
    for k, rec in enumerate(ka["t1q1"]):
        q, t = qs[k], ts[k]
        for flag in (0, po.RIGHT):
            for func in ("extz", "extd"):
                exp = rec["ksw_%s/flag=%d" % (func, flag)]
                res = po.align("oracle", func, q, t, mat, 4, 2, 13, 1, flag=flag)
                _check(res, exp, gu.FIELDS)
                assert gu.cigar_string(res["cigar"]) == exp["cigar"]
        # SURVEY.md section 4.2 literal table (ksw2-test -t extz test/t1.fa test/q1.fa)
        res = po.align("oracle", "extz", q, t, mat, 4, 2)
        sc, mx, mt, mq, cg = expected_extz[k]
        assert (res["score"], res["max"], res["max_t"], res["max_q"], gu.cigar_string(res["cigar"])) == (sc, mx, mt, mq, cg)
        for g in ("gg", "gg2"):
            s, c = po.global_align("oracle", g, q, t, mat, 4, 2, w=-1)
            assert s == rec["ksw_" + g]["score"] and gu.cigar_string(c) == rec["ksw_" + g]["cigar"]
        # gg2_sse agrees with gg on these inputs, so the same oracle call covers it
        assert rec["ksw_gg2_sse"]["score"] == rec["ksw_gg"]["score"] and rec["ksw_gg2_sse"]["cigar"] == rec["ksw_gg"]["cigar"]
        key = "ksw_extz/A1B9O16E1w10"
        if key in rec:
            res = po.align("oracle", "extz", q, t, gu.simple_mat(5, 1, 9, 0), 16, 1, w=10)
            _check(res, rec[key], gu.FIELDS)


def test_random_cases_scalar_contract():
    rc = gu.RandomCases()
    n = {"ksw_extz": 0, "ksw_extd": 0}
    for k in range(rc.n):
        c = rc.case(k)
        if c["func"] not in n:
            continue
        func = c["func"][4:]
        res = po.align("oracle", func, c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=c["zdrop"],
                       flag=c["flag"])
        _check(res, c["expect"], gu.FIELDS + ["cigar"])
        # the "...2" contract with explicit matrix scoring and no end bonus is the same function
        res2 = po.align("oracle", func + "2", c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"],
                        zdrop=c["zdrop"], end_bonus=0, flag=c["flag"] | po.GENERIC_SC)
        if not (c["flag"] & po.EXTZ_ONLY):
            _check(res2, c["expect"], gu.FIELDS + ["cigar"])
        n[c["func"]] += 1
    assert min(n.values()) > 500


def test_random_cases_sse_signature():
    """end_bonus / reach_end / implicit wildcard scoring: compared with the SSE kernels on loose bands."""
    rc = gu.RandomCases()
    n = 0
    for k in range(rc.n):
        c = rc.case(k)
        if not c["func"].endswith("2_sse"):
            continue
        func = "extz2" if "extz" in c["func"] else "extd2"
        res = po.align("oracle", func, c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=c["zdrop"],
                       end_bonus=c["end_bonus"], flag=c["flag"])
        _check(res, c["expect"], gu.SSE_LOOSE_FIELDS)
        if res["max_t"] == c["expect"]["max_t"] and res["max_q"] == c["expect"]["max_q"]:
            assert res["cigar"] == c["expect"]["cigar"]          # F4: CIGAR follows the max cell under EXTZ_ONLY
        n += 1
    assert n > 500


@pytest.mark.parametrize("idx", range(10))
def test_mt_pair(idx):
    """16.5 kb mitochondrial pair; ~1-2 s per setting on one core."""
    ka = gu.known_answers()["mt"]
    _, ts = gu.read_fasta("MT-human.fa")
    _, qs = gu.read_fasta("MT-orang.fa")
    exp = ka[idx]
    mat = gu.simple_mat(5, 2, 4, 0)
    res = po.align("oracle", exp["func"][4:], qs[0], ts[0], mat, 4, 2, 13, 1, w=exp["w"], zdrop=exp["zdrop"], flag=exp["flag"])
    _check(res, exp, gu.FIELDS)
    s = gu.cigar_string(res["cigar"])
    assert s == exp["cigar"]
    assert hashlib.md5((s + "\n").encode()).hexdigest()[:12] == exp["cigar_md5_12"]


def test_mt_survey_anchors():
    """The literal anchors of BASELINE.md section 4 (independent of the JSON file's content)."""
    ka = {(r["func"], r["w"], r.get("flag", 0), r.get("zdrop", -1)): r for r in gu.known_answers()["mt"]}
    r = ka[("ksw_extz", -1, 0, -1)]
    assert (r["score"], r["max"], r["max_t"], r["max_q"], r["cigar_md5_12"]) == (16102, 17054, 16568, 16024, "ea0524d904ed")
    r = ka[("ksw_extd", -1, 0, -1)]
    assert (r["score"], r["max"], r["max_t"], r["max_q"], r["cigar_md5_12"]) == (17127, 17614, 16568, 16024, "df0e77e43f48")
    r = ka[("ksw_extz", 500, 0, -1)]
    assert (r["score"], r["max"], r["cigar_md5_12"]) == (-13510, 2, "c07fce86940f")
    assert ka[("ksw_extz", -1, po.RIGHT, -1)]["cigar_md5_12"] == "db8b671f4dbf"
    assert ka[("ksw_extd", -1, po.RIGHT, -1)]["cigar_md5_12"] == "8e2c9cfb877a"


def test_band_cells():
    assert po.band_cells(512, 512, 64) == 61888          # SURVEY section 8 config sizes
    assert po.band_cells(2048, 2048, 256) == 984832
    assert po.band_cells(10000, 10000, 500) == 9759500
    assert po.band_cells(16499, 16569, -1) == 16499 * 16569


def test_approx_max_mode_matches_reference_golden():
    """KSW_EZ_APPROX_MAX alone: the reference returns only the score and the corner CIGAR (tests/golden/approx_cases.npz)."""
    ac = gu.ApproxCases()
    assert ac.n >= 400
    for k in range(ac.n):
        c = ac.case(k)
        r = gu.ApproxCases.run("oracle", c)
        for f in gu.FIELDS + ["cigar"]:
            assert r[f] == c["expect"][f], (k, c["func"], hex(c["flag"]), f, r[f], c["expect"][f])


def test_eqx_cases_pin_the_oracle():
    """KSW_EZ_EQX: all 400 committed runs of the reference's ksw_extd2_sse (buffers pre-sized so that its ksw_cigar2eqx never
    reallocates; oracle/gen_golden_eqx.py), =/X CIGARs word for word."""
    ec = gu.EqxCases()
    assert ec.n >= 300
    nx = 0
    for k in range(ec.n):
        c = ec.case(k)
        r = po.align("oracle", "extd2", c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=-1, end_bonus=c["end_bonus"],
                     flag=c["flag"])
        bad, same = gu.EqxCases.check(c, r)
        assert not bad, (k, bad, c["flag"])
        assert same
        nx += sum(1 for x in r["cigar"] if (x & 0xf) == 8)
        assert all((x & 0xf) != 0 for x in r["cigar"])           # no M left
    assert nx > 1000


def test_50k_anchor():
    """The README's 50 000 x 50 000 pair (test/t2.fa.gz x test/q2.fa.gz; SURVEY 4.2: 69932 / 70010 / 49962 / 49999), unbanded score-only
    through the scalar contract and banded through the SSE signature."""
    ka = gu.known_answers()["t2q2_50k"]
    _, ts = gu.read_fasta("t2.fa.gz")
    _, qs = gu.read_fasta("q2.fa.gz")
    assert len(ts[0]) == len(qs[0]) == 50000
    mat = gu.simple_mat(5, 2, 4, 0)
    exp = [r for r in ka if r["func"] == "ksw_extz"][0]
    assert (exp["score"], exp["max"], exp["max_t"], exp["max_q"]) == (69932, 70010, 49962, 49999)
    res = po.align("oracle", "extz", qs[0], ts[0], mat, 4, 2, w=-1, flag=po.SCORE_ONLY)
    _check(res, exp, gu.FIELDS)
    exp = [r for r in ka if r["func"] == "ksw_extz2_sse" and r["w"] == 500][0]
    res = po.align("oracle", "extz2", qs[0], ts[0], mat, 4, 2, w=500, zdrop=400, flag=po.SCORE_ONLY)
    _check(res, exp, gu.SSE_LOOSE_FIELDS)


def test_gg_family_golden():
    """ksw_gg / ksw_gg2 / ksw_gg2_sse (ksw2_gg.c:6-102, ksw2_gg2.c:4-114, ksw2_gg2_sse.c:11-126): 900 committed cases -- bands -1, |d|,
    |d| + 1, 20, 64, 500, wildcards, score only and with CIGAR -- pin kso_gg / kso_gg2; ksw_gg2 / ksw_gg2_sse cases only where the
    reference's function agrees with its own ksw_gg (recorded in the fixture); w < |d| is the library's definition (KSW_NEG_INF, no CIGAR)."""
    gc = gu.GgCases()
    assert gc.n >= 600
    cnt = {"gg": 0, "gg2": 0, "gg2_sse": 0, "contract": 0}
    for k in gc.contract_cases():
        c = gc.case(k)
        s, cg = po.global_align("oracle", "gg" if c["func"] == "gg" else "gg2", c["q"], c["t"], c["mat"], c["gq"], c["ge"], w=c["w"], with_cigar=c["with_cigar"])
        assert s == c["score"], (k, c["func"], c["w"], s, c["score"])
        assert cg == c["cigar"], (k, c["func"], c["w"])
        cnt["contract" if c["origin"] else c["func"]] += 1
    assert cnt["gg"] > 200 and cnt["gg2"] > 200 and cnt["gg2_sse"] > 120 and cnt["contract"] > 100, cnt
    # what the fixture records about the reference itself: its scalar ksw_gg2 leaves the exact band on a few w <= 2 cases
    # (cells outside the band get difference values 0, not -infinity: ksw2_gg2.c:36-41), its SSE twin on about a third
    differ = {"gg2": [], "gg2_sse": []}
    for k in range(gc.n):
        c = gc.case(k)
        if not c["agree"]:
            differ[c["func"]].append(c["w"])
    assert 0 < len(differ["gg2"]) < 10 and all(0 <= w <= 2 for w in differ["gg2"]), differ["gg2"]
    assert len(differ["gg2_sse"]) > 50
