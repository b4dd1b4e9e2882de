"""Golden vectors for the gap-linear X-drop extension (ksw_extf2_sse): inputs + outputs of the UNMODIFIED reference.

Run in the build container only (needs oracle/_ref/libksw2ref.so, i.e. /root/reference):   python oracle/gen_golden_extf.py
Output (committed, data only): tests/golden/extf_cases.npz -- seeded cases: sequences, parameters and all ksw_extz_t fields
as returned by the reference's ksw_extf2_sse (gcc -O2 -msse4.1).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FIELDS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]   # = tests/golden_util.FIELDS
SCORINGS = [(1, -2, 1), (2, -4, 2), (1, -1, 1), (2, -3, 1), (3, -5, 2), (1, -3, 2), (5, -4, 3), (2, 4, 2), (10, -12, 6), (1, -2, 0)]
BANDS = [-1, 0, 1, 2, 5, 10, 15, 16, 17, 31, 32, 50, 100, 400, 2000]
XDROPS = [-1, 0, 5, 20, 50, 100, 1000]


def noisy_pair(rng, tl, kind):
    """Target and a query derived from it: substitutions, a few indels, optional unrelated tail / truncation / wildcards."""
    if kind == 3:                                # two-letter sequences: long runs of ties in the greedy cell choice
        t = (rng.integers(0, 2, tl, dtype=np.uint8) * 3).astype(np.uint8)
        return (rng.integers(0, 2, max(1, tl + int(rng.integers(-20, 20))), dtype=np.uint8) * 3).astype(np.uint8), t
    if kind == 4:                                # unrelated
        return rng.integers(0, 4, max(1, int(rng.integers(1, 2 * tl + 2))), dtype=np.uint8), rng.integers(0, 4, tl, dtype=np.uint8)
    t = rng.integers(0, 4, tl, dtype=np.uint8)
    q = t.copy()
    mask = rng.random(tl) < rng.choice([0.0, 0.02, 0.1, 0.3])
    q[mask] = rng.integers(0, 4, int(mask.sum()), dtype=np.uint8)
    for _ in range(int(rng.integers(0, 4))):
        if len(q) > 10:
            k = int(rng.integers(1, len(q) - 1))
            if rng.random() < 0.5:
                q = np.delete(q, slice(k, k + int(rng.integers(1, 6))))
            else:
                q = np.insert(q, k, rng.integers(0, 4, int(rng.integers(1, 6)), dtype=np.uint8))
    if rng.random() < 0.3:
        q = np.concatenate([q, rng.integers(0, 4, int(rng.integers(1, 80)), dtype=np.uint8)])
    if rng.random() < 0.2:
        q = q[:max(1, int(rng.integers(1, len(q) + 1)))]
    if rng.random() < 0.15:
        q = q.copy(); q[rng.random(len(q)) < 0.03] = 4
        t = t.copy(); t[rng.random(len(t)) < 0.03] = 4
    if len(q) == 0:
        q = np.array([1], dtype=np.uint8)
    return q.astype(np.uint8), t.astype(np.uint8)


def main(n_cases=2000, seed=20260003):
    rng = np.random.Generator(np.random.PCG64(seed))
    seqs, seq_off, params, expect = [], [0], [], []
    for it in range(n_cases):
        mch, mis, e = SCORINGS[it % len(SCORINGS)]
        tl = int(rng.integers(1, 4000 if it % 50 == 0 else 600 if it % 5 == 0 else 130))
        if it % 97 == 0:
            tl = int(rng.choice([16, 32, 64, 128, 256]))          # padded length = length
        q, t = noisy_pair(rng, tl, it % 7)
        w = int(BANDS[(it // 3) % len(BANDS)])
        xd = int(XDROPS[(it // 7) % len(XDROPS)])
        res = po.extf2("ref", q, t, mch, mis, e, w, xd)
        seqs += [q, t]
        o = seq_off[-1]
        seq_off += [o + len(q), o + len(q) + len(t)]
        params.append([mch, mis, e, w, xd])
        expect.append([res[f] for f in FIELDS])
    np.savez_compressed(os.path.join(GOLD, "extf_cases.npz"), seq=np.concatenate(seqs).astype(np.uint8), seq_off=np.array(seq_off, dtype=np.int64),
                        params=np.array(params, dtype=np.int32), expect=np.array(expect, dtype=np.int64))
    print("wrote", n_cases, "cases,", os.path.getsize(os.path.join(GOLD, "extf_cases.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
