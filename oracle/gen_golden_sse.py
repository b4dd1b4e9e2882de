"""Golden vectors for the SSE-compatible mode: inputs + every ksw_extz_t field + CIGAR of the UNMODIFIED ksw_extz2_sse /
ksw_extd2_sse (oracle/_ref) on cases where the SSE kernels differ from the scalar contract -- narrow bands whose 16-position
blocks leak, Z-drop per anti-diagonal (F2), mte_q from the padded range (F3), tie order of the maximum (F4),
KSW_EZ_APPROX_MAX with and without KSW_EZ_APPROX_DROP (ksw2_extz2_sse.c:270-286, ksw2_extd2_sse.c:366-382), bands that cannot
reach the corner, swapped gap pieces (ksw2_extd2_sse.c:78), wildcards with and without KSW_EZ_GENERIC_SC, end_bonus.
KSW_EZ_EQX is left to eqx_cases.npz (the reference needs preallocated buffers there).

Run in the build container only (needs oracle/_ref):   python oracle/gen_golden_sse.py   ->   tests/golden/sse_cases.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po                       # noqa: E402
from ksw2_amd import synth                              # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FIELDS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]   # = tests/golden_util.FIELDS
MATS = [(2, 4, -1, 4, 2, 24, 1), (1, 9, 0, 4, 2, 24, 1), (2, 4, 0, 4, 2, 13, 1), (2, 4, -3, 6, 1, 30, 0), (1, 2, 0, 2, 1, 32, 0),
        (2, 4, 0, 4, 2, 4, 2), (2, 4, 0, 24, 1, 4, 2)]   # a, b, sc_n, q, e, q2, e2 (the last one: pieces the callee swaps)


def main(n_cases=1500, seed=20260005):
    rng = np.random.Generator(np.random.PCG64(seed))
    seqs, seq_off, params, expect, cigs, cig_off = [], [0], [], [], [], [0]
    for it in range(n_cases):
        a, b, sc_n, gq, ge, gq2, ge2 = MATS[it % len(MATS)]
        mat = po.simple_mat(5, a, b, sc_n)
        maxlen = int(rng.choice([40, 150, 400, 900])) if it % 50 else 2500
        (q, t), = synth.ragged_pairs(rng, 1, 1, maxlen, sub=0.02 + 0.15 * rng.random(), ind=0.25 * rng.random(),
                                     indel_mean=1.5 if it % 3 else 5.0, n_rate=0.02 if it % 6 == 0 else 0.0)
        w = int(rng.choice([-1, 1, 3, 8, 20, 64, 200, 1000]))
        zd = int(rng.choice([-1, 10, 40, 100, 400]))
        eb = int(rng.choice([0, 5, 50]))
        flag = 0
        for bit, pr in ((po.SCORE_ONLY, 0.3), (po.RIGHT, 0.3), (po.GENERIC_SC, 0.2), (po.APPROX_MAX, 0.3), (po.EXTZ_ONLY, 0.3), (po.REV_CIGAR, 0.3)):
            if rng.random() < pr:
                flag |= bit
        if flag & po.APPROX_MAX and rng.random() < 0.6:
            flag |= po.APPROX_DROP
        dual = it % 2
        res = po.align("ref", "extd2_sse" if dual else "extz2_sse", q, t, mat, gq, ge, gq2, ge2, w=w, zdrop=zd, end_bonus=eb, flag=flag)
        seqs += [q, t]
        seq_off += [seq_off[-1] + len(q), seq_off[-1] + len(q) + len(t)]
        params.append([a, b, sc_n, gq, ge, gq2, ge2, w, zd, eb, flag, dual])
        expect.append([res[f] for f in FIELDS])
        cigs += res["cigar"]
        cig_off.append(len(cigs))
    np.savez_compressed(os.path.join(GOLD, "sse_cases.npz"), seq=np.concatenate(seqs).astype(np.uint8), seq_off=np.array(seq_off, dtype=np.int64),
                        params=np.array(params, dtype=np.int32), expect=np.array(expect, dtype=np.int64),
                        cigar=np.array(cigs, dtype=np.uint32), cigar_off=np.array(cig_off, dtype=np.int64))
    ex = np.array(expect)
    print("wrote", n_cases, "cases,", os.path.getsize(os.path.join(GOLD, "sse_cases.npz")) // 1024, "KiB; zdropped", int(ex[:, 8].sum()),
          "with CIGAR", int((ex[:, 10] > 0).sum()))


if __name__ == "__main__":
    if po.ref_lib() is None and not po.build_ref():
        sys.exit("reference sources not available: golden vectors can only be regenerated in the build container")
    main()
