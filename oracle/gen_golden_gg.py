#!/usr/bin/env python3
"""Golden vectors for the global family: ksw_gg (ksw2_gg.c:6-102), ksw_gg2 (ksw2_gg2.c:4-114), ksw_gg2_sse (ksw2_gg2_sse.c:11-126)
from the UNMODIFIED reference compiled into oracle/_ref/libksw2ref.so.

Run in the build container only (needs /root/reference):   python oracle/gen_golden_gg.py
Output (committed, data only): tests/golden/gg_cases.npz -- inputs, the reference's score and CIGAR per case, and per case
  origin = 0  the reference's own output on a band that reaches the corner (w >= |tlen - qlen|, or w < 0);
  origin = 1  the band cannot reach the corner (w < |tlen - qlen|): the reference's three functions return three different,
              undefined answers there (gg -> KSW_NEG_INF / garbage CIGAR, gg2 -> a finite number, gg2_sse another),
              so the expected value is the LIBRARY'S DEFINITION (include/ksw2_amd.h: KSW_NEG_INF, no CIGAR), not a reference output;
  agree  = 1  the function returned exactly what ksw_gg returns on the same input.  ksw_gg2_sse's 16-position blocks leak across
              narrow bands (a third of these cases), and the scalar ksw_gg2 itself is not exact there either: it
              gives cells just outside the band the difference values 0 instead of -infinity (ksw2_gg2.c:36-41), which at w = 1, 2
              lets a few paths score higher than the exact band allows (5 of 253 cases here).  Such outputs are NOT the contract
              (the library's ksw_gg2 / ksw_gg2_sse are the exact band = ksw_gg); the parity tests use agree = 1 cases only and the
              fixture records which those are.
Bands: w in { -1, |d|, |d| + 1, 20, 64, 500 } with d = tlen - qlen; wildcards in a fifth of the cases (the global functions score
with the matrix as given); score only (all three CIGAR pointers NULL: ksw_gg and ksw_gg2 accept that, ksw2_gg.c:17, ksw2_gg2.c:18)
and with CIGAR.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po          # noqa: E402
from ksw2_amd import synth                  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FUNCS = ["gg", "gg2", "gg2_sse"]
MATS = [(2, 4, 0, 4, 2), (1, 9, 0, 4, 2), (1, 9, -1, 16, 2), (2, 4, -3, 4, 2), (2, 4, -1, 6, 1), (1, 2, 0, 2, 1)]   # a, b, sc_n, q, e


def main(n_cases=900, seed=20260401):
    rng = np.random.Generator(np.random.PCG64(seed))
    seqs, seq_off = [], [0]
    params, cigs, cig_off = [], [], [0]
    it = 0
    stats = {"contract": 0, "sse_agree": 0, "sse_differ": 0, "gg2_agree": 0, "gg2_differ": 0}
    while len(params) < n_cases:
        it += 1
        a, b, sc_n, q, e = MATS[it % len(MATS)]
        mat = po.simple_mat(5, a, b, sc_n)
        func = FUNCS[it % 3]
        big = it % 40 == 0
        qq, tt = synth.ragged_pairs(rng, 1, 1, 1200 if big else 320, sub=0.03 + 0.12 * rng.random(), ind=0.25 * rng.random(),
                                    indel_mean=1.5 if it % 4 else 6.0, n_rate=0.02 if it % 5 == 0 else 0.0)[0]
        d = abs(len(tt) - len(qq))
        wsel = (it // 3) % 7
        w = [-1, d, d + 1, 20, 64, 500, d - 1][wsel]
        with_cigar = 1 if func == "gg2_sse" else (it // 21) % 2            # ksw_gg2_sse dereferences its CIGAR pointers (ksw2_gg2_sse.c:123)
        origin, agree = 0, 1
        if w >= 0 and w < d:
            # undefined in the reference: the library's definition (score = KSW_NEG_INF, no CIGAR), reference not consulted
            origin = 1
            score, cg = po.NEG_INF, []
            stats["contract"] += 1
        elif wsel == 6:
            continue                                                       # d == 0: w = -1 is the unbanded case, already covered
        else:
            score, cg = po.global_align("ref", func, qq, tt, mat, q, e, w=w, with_cigar=bool(with_cigar))
            if func != "gg":
                s0, c0 = po.global_align("ref", "gg", qq, tt, mat, q, e, w=w, with_cigar=True)
                agree = int(s0 == score and (not with_cigar or c0 == cg))
                stats[("sse_" if func == "gg2_sse" else "gg2_") + ("agree" if agree else "differ")] += 1
        seqs += [qq, tt]
        seq_off += [seq_off[-1] + len(qq), seq_off[-1] + len(qq) + len(tt)]
        params.append([FUNCS.index(func), a, b, sc_n, q, e, w, with_cigar, origin, agree, score])
        cigs.append(np.array(cg, dtype=np.uint32))
        cig_off.append(cig_off[-1] + len(cg))
    out = os.path.join(GOLD, "gg_cases.npz")
    np.savez_compressed(out, seq=np.concatenate(seqs), seq_off=np.array(seq_off, dtype=np.int64), params=np.array(params, dtype=np.int32),
                        cigar=np.concatenate(cigs), cigar_off=np.array(cig_off, dtype=np.int64),
                        param_names=np.array(["func", "a", "b", "sc_n", "q", "e", "w", "with_cigar", "origin", "agree", "score"]),
                        funcs=np.array(["ksw_gg", "ksw_gg2", "ksw_gg2_sse"]))
    print("gg cases:", len(params), stats, "bytes:", os.path.getsize(out))


if __name__ == "__main__":
    if po.ref_lib() is None and not po.build_ref("/root/reference"):
        sys.exit("reference sources not available: golden vectors can only be regenerated in the build container")
    main()
