#!/usr/bin/env python3
"""Agreement of every ksw_extz_t field between this library's exact-band / row-wise Z-drop contract and the reference's real
SSE kernels (ksw_extz2_sse / ksw_extd2_sse), on the committed cases the compiled reference produced (SURVEY.md section 0,
F1-F4; tests/golden/random_cases.npz "...2_sse" cases, eqx_cases.npz, approx_cases.npz).

  python tools/scripts/sse_agreement.py oracle          the CPU oracle (bit-identical to the HIP path by the parity tests)
  python tools/scripts/sse_agreement.py hip [out.json]  the HIP library on the GPU box
  python tools/scripts/sse_agreement.py sim             the simulator build
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import golden_util as gu          # noqa: E402

FIELDS = gu.FIELDS + ["cigar"]


def runner(which):
    if which == "oracle":
        from oracle import pyoracle as po

        def run(func, c, **kw):
            return po.align("oracle", func, c["q"], c["t"], c["mat"], c["gq"], c["ge"], c.get("gq2"), c.get("ge2"), **kw)
        return run
    import ksw2_amd as ka
    lib = ka.library() if which == "hip" else ka.Library(os.path.join(ROOT, "tests", "sim", "libksw2_amd_sim.so"))

    def run(func, c, **kw):
        if func == "extd2":
            return lib.extd2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], **kw)
        return lib.extz2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], **kw)
    return run


def tally(tab, name, exp, res):
    t = tab.setdefault(name, {"cases": 0, "all_fields_equal": 0, "differ": {f: 0 for f in FIELDS}})
    t["cases"] += 1
    bad = [f for f in FIELDS if exp[f] != (list(res[f]) if f == "cigar" else res[f])]
    for f in bad:
        t["differ"][f] += 1
    t["all_fields_equal"] += not bad


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "oracle"
    run = runner(which)
    tab = {}
    rc = gu.RandomCases()
    for k in range(rc.n):
        c = rc.case(k)
        if not c["func"].endswith("2_sse"):
            continue
        func = "extd2" if "extd" in c["func"] else "extz2"
        r = run(func, c, w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        tally(tab, "%s, loose band, no Z-drop (random_cases.npz)" % c["func"], c["expect"], r)
    ec = gu.EqxCases()
    for k in range(ec.n):
        c = ec.case(k)
        r = run("extd2", c, w=c["w"], zdrop=-1, end_bonus=c["end_bonus"], flag=c["flag"])
        tally(tab, "ksw_extd2_sse + KSW_EZ_EQX (eqx_cases.npz)", c["expect"], r)
    ac = gu.ApproxCases()
    for k in range(ac.n):
        c = ac.case(k)
        if c["func"] == 2:
            continue
        c2 = dict(c, mat=gu.simple_mat(5, 2, 4, -1), gq=4, ge=2, gq2=24, ge2=1)
        r = run("extd2" if c["func"] else "extz2", c2, w=-1, zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"])
        tally(tab, "%s + KSW_EZ_APPROX_MAX (approx_cases.npz)" % ("ksw_extd2_sse" if c["func"] else "ksw_extz2_sse"), c["expect"], r)
    for t in tab.values():
        t["agreement"] = {f: round(1.0 - t["differ"][f] / t["cases"], 5) for f in FIELDS}
    out = {"library": which, "fields": FIELDS, "sets": tab}
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    for name, t in tab.items():
        print("%-62s %5d cases, %5d identical in every field; differing: %s" % (name, t["cases"], t["all_fields_equal"],
              ", ".join("%s %d" % (f, n) for f, n in t["differ"].items() if n) or "none"))
    return out


if __name__ == "__main__":
    main()
