"""Golden vectors for the splice-aware extension (ksw_exts2_sse): inputs + outputs of the UNMODIFIED reference.

Run in the build container only (needs oracle/_ref/libksw2ref.so, i.e. /root/reference):   python oracle/gen_golden_exts.py
Output (committed, data only): tests/golden/exts_cases.npz -- seeded cases: sequences, junction annotation, parameters,
all ksw_extz_t fields and CIGAR words as returned by the reference's ksw_exts2_sse (gcc -O2 -msse4.1).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FIELDS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]   # = tests/golden_util.FIELDS
SCORINGS = [(1, 2, 0, 2, 1, 32, 4), (2, 4, -1, 4, 2, 24, 5), (1, 3, 0, 2, 1, 20, 9), (2, 5, -1, 3, 1, 40, 7), (1, 1, 0, 1, 1, 10, 3)]
MODES = [0, po.SCORE_ONLY, po.RIGHT, po.EXTZ_ONLY, po.REV_CIGAR, po.GENERIC_SC, po.RIGHT | po.REV_CIGAR, po.EXTZ_ONLY | po.RIGHT]
SPLICE = [0, po.SPLICE_FOR, po.SPLICE_REV, po.SPLICE_FOR | po.SPLICE_FLANK, po.SPLICE_FOR | po.SPLICE_REV, po.SPLICE_REV | po.SPLICE_FLANK]


def spliced_pair(rng, tl, low_complexity=False):
    """A target with an optional GT..AG (or CT..AC) intron and a noisy copy of its exons as the query."""
    if low_complexity:                          # two-letter sequences: many equal-score cells, the tie rules matter
        t = (rng.integers(0, 2, tl, dtype=np.uint8) * 2).astype(np.uint8)
        q = (rng.integers(0, 2, max(1, tl + int(rng.integers(-20, 20))), dtype=np.uint8) * 2).astype(np.uint8)
        return q, t
    t = rng.integers(0, 4, tl, dtype=np.uint8)
    q = t.copy()
    if tl > 80 and rng.random() < 0.8:
        a = int(rng.integers(10, tl // 2))
        b = int(rng.integers(a + 20, min(tl - 5, a + 20 + tl // 2)))
        kind = rng.random()
        if kind < 0.5:
            t[a], t[a + 1], t[b - 2], t[b - 1] = 2, 3, 0, 2          # GT ... AG
        elif kind < 0.75:
            t[a], t[a + 1], t[b - 2], t[b - 1] = 1, 3, 0, 1          # CT ... AC (reverse strand)
        q = np.concatenate([t[:a], t[b:]])
    mask = rng.random(len(q)) < 0.05
    q[mask] = rng.integers(0, 4, int(mask.sum()), dtype=np.uint8)
    if len(q) > 30 and rng.random() < 0.5:
        k = int(rng.integers(3, len(q) - 10))
        q = np.delete(q, slice(k, k + int(rng.integers(1, 6))))
    if rng.random() < 0.2:
        q[rng.random(len(q)) < 0.02] = 4
    if rng.random() < 0.2:
        t = t.copy()
        t[rng.random(len(t)) < 0.01] = 4
    if len(q) == 0:
        q = np.array([1], dtype=np.uint8)
    return q.astype(np.uint8), t.astype(np.uint8)


def main(n_cases=1200, seed=20260002):
    rng = np.random.Generator(np.random.PCG64(seed))
    seqs, seq_off, params, expect, cigs, cig_off = [], [0], [], [], [], [0]
    for it in range(n_cases):
        a, b, sc_n, q, e, q2, noncan = SCORINGS[it % len(SCORINGS)]
        tl = int(rng.integers(1, 1500 if it % 40 == 0 else 400))
        qq, tt = spliced_pair(rng, tl, low_complexity=(it % 7 == 3))
        mat = po.simple_mat(5, a, b, sc_n)
        flag = MODES[(it // 5) % len(MODES)] | SPLICE[(it // 3) % len(SPLICE)]
        zdrop = [-1, 20, 100, 400][(it // 11) % 4]
        junc = np.zeros(len(tt), dtype=np.uint8)
        jb = 0
        if it % 4 == 1:
            junc = (rng.integers(0, 16, len(tt), dtype=np.uint8) * (rng.random(len(tt)) < 0.05)).astype(np.uint8)
            jb = int(rng.integers(1, 6))
        res = po.exts2("ref", qq, tt, mat, q, e, q2, noncan, zdrop=zdrop, junc_bonus=jb, flag=flag, junc=junc if jb else None)
        seqs += [qq, tt, junc]
        o = seq_off[-1]
        seq_off += [o + len(qq), o + len(qq) + len(tt), o + len(qq) + 2 * len(tt)]
        params.append([a, b, sc_n, q, e, q2, noncan, zdrop, jb, flag])
        expect.append([res[f] for f in FIELDS])
        cigs += res["cigar"]
        cig_off.append(len(cigs))
    np.savez_compressed(os.path.join(GOLD, "exts_cases.npz"), seq=np.concatenate(seqs).astype(np.uint8), seq_off=np.array(seq_off, dtype=np.int64),
                        params=np.array(params, dtype=np.int32), expect=np.array(expect, dtype=np.int64),
                        cigar=np.array(cigs, dtype=np.uint32), cigar_off=np.array(cig_off, dtype=np.int64))
    print("wrote", n_cases, "cases,", os.path.getsize(os.path.join(GOLD, "exts_cases.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
