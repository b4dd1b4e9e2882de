"""The oracle's restatement of the SSE kernels' own results (oracle/ksw2_oracle_sse.c) against golden vectors produced by the
unmodified reference (tests/golden/sse_cases.npz, oracle/gen_golden_sse.py): every ksw_extz_t field and the CIGAR, on narrow
bands, Z-drop, KSW_EZ_APPROX_MAX / APPROX_DROP, swapped gap pieces -- and live against oracle/_ref where it exists."""
import numpy as np
import pytest

from oracle import pyoracle as po
from tests import golden_util as gu

ALL = gu.FIELDS + ["cigar"]


def test_sse_image_oracle_matches_golden():
    sc = gu.SseCases()
    assert sc.n >= 1500
    kinds = {"leak": 0, "zdrop": 0, "approx": 0, "approx_drop": 0, "cigar": 0}
    for k in range(sc.n):
        c = sc.case(k)
        got = po.align("oracle", "extd2_sse" if c["dual"] else "extz2_sse", c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"],
                       w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        bad = [f for f in ALL if got[f] != c["expect"][f]]
        assert not bad, (k, bad, c["dual"], c["w"], c["zdrop"], hex(c["flag"]), len(c["q"]), len(c["t"]))
        kinds["zdrop"] += c["expect"]["zdropped"]
        kinds["approx"] += bool(c["flag"] & po.APPROX_MAX)
        kinds["approx_drop"] += bool(c["flag"] & po.APPROX_DROP)
        kinds["cigar"] += c["expect"]["n_cigar"] > 0
        if not (c["flag"] & po.APPROX_MAX) and 0 <= c["w"] <= 20:
            exact = po.align("oracle", "extd2" if c["dual"] else "extz2", c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"],
                             w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
            kinds["leak"] += exact["score"] != got["score"] or exact["max"] != got["max"]
    # the set really contains what the mode is for: results that differ from the exact-band contract, drops, both approximate modes
    assert kinds["leak"] > 20 and kinds["zdrop"] > 100 and kinds["approx"] > 200 and kinds["approx_drop"] > 100 and kinds["cigar"] > 500, kinds


@pytest.mark.skipif(po.ref_lib() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_sse_image_oracle_matches_reference_live():
    from ksw2_amd import synth
    rng = np.random.Generator(np.random.PCG64(99))
    mats = [(2, 4, -1, 4, 2, 24, 1), (2, 4, 0, 24, 1, 4, 2), (1, 2, 0, 2, 1, 32, 0)]
    for it in range(300):
        a, b, scn, gq, ge, gq2, ge2 = mats[it % 3]
        mat = po.simple_mat(5, a, b, scn)
        (q, t), = synth.ragged_pairs(rng, 1, 1, 300, sub=0.15 * rng.random(), ind=0.25 * rng.random(), n_rate=0.02 if it % 5 == 0 else 0.0)
        w, zd = int(rng.choice([-1, 2, 9, 33, 100])), int(rng.choice([-1, 20, 100]))
        flag = int(rng.choice([0, po.SCORE_ONLY, po.RIGHT, po.APPROX_MAX, po.APPROX_MAX | po.APPROX_DROP, po.EXTZ_ONLY, po.APPROX_MAX | po.APPROX_DROP | po.EXTZ_ONLY | po.RIGHT]))
        for func in ("extz2_sse", "extd2_sse"):
            r = po.align("ref", func, q, t, mat, gq, ge, gq2, ge2, w=w, zdrop=zd, end_bonus=7, flag=flag)
            o = po.align("oracle", func, q, t, mat, gq, ge, gq2, ge2, w=w, zdrop=zd, end_bonus=7, flag=flag)
            assert all(r[f] == o[f] for f in ALL), (func, w, zd, hex(flag), [f for f in ALL if r[f] != o[f]])
