"""Golden vectors for KSW_EZ_EQX on ksw_extd2_sse (=/X instead of M, ksw2_extd2_sse.c:399-406 / ksw2.h:163-182): inputs +
outputs of the UNMODIFIED reference, with RIGHT / REV_CIGAR / EXTZ_ONLY / GENERIC_SC / end_bonus mixed in.

The reference's ksw_cigar2eqx drops ksw_push_cigar's return value (ksw2.h:171-176): a realloc that moves the block leaves it
writing through a stale pointer.  Every case therefore hands the reference a CIGAR buffer that is already large enough
(2 * (qlen + tlen) + 16 words, m_cigar set accordingly), so no reallocation happens and its output is well defined.
Loose bands (w = -1 or far wider than the length difference) and no Z-drop, like the other "...2_sse" cases.
Note the reference walks a REV_CIGAR (end -> start) list from the START of both sequences when it splits M runs; that is what
it returns and what is pinned here.

Run in the build container only (needs oracle/_ref):   python oracle/gen_golden_eqx.py   ->   tests/golden/eqx_cases.npz
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po                       # noqa: E402
from ksw2_amd import synth                              # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
FIELDS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]   # = tests/golden_util.FIELDS
MATS = [(2, 4, -1, 4, 2, 24, 1), (1, 9, 0, 4, 2, 24, 1), (2, 4, 0, 4, 2, 13, 1), (2, 4, -3, 6, 1, 30, 0)]   # a, b, sc_n, q, e, q2, e2


def ref_extd2_prealloc(q, t, mat, gq, ge, gq2, ge2, w, end_bonus, flag):
    lib = po.ref_lib()
    q, t = np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    cap = 2 * (len(q) + len(t)) + 16
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    ez = po.Ez()
    ez.cigar = ctypes.cast(libc.malloc(4 * cap), ctypes.POINTER(ctypes.c_uint32))
    ez.m_cigar = cap
    lib.ksw_extd2_sse(None, len(q), q.ctypes.data_as(po._u8p), len(t), t.ctypes.data_as(po._u8p), 5, mat.ctypes.data_as(po._i8p), gq, ge, gq2, ge2,
                      w, -1, end_bonus, flag, ez)
    assert ez.m_cigar == cap, "the reference reallocated: its EQX output would be undefined"
    return po._ez_to_dict(ez)


def main(n_cases=400, seed=20260004):
    rng = np.random.Generator(np.random.PCG64(seed))
    seqs, seq_off, params, expect, cigs, cig_off = [], [0], [], [], [], [0]
    extra = [0, po.RIGHT, po.REV_CIGAR, po.EXTZ_ONLY, po.EXTZ_ONLY | po.REV_CIGAR, po.RIGHT | po.REV_CIGAR, po.RIGHT | po.EXTZ_ONLY, po.GENERIC_SC,
             po.GENERIC_SC | po.REV_CIGAR]
    for it in range(n_cases):
        a, b, sc_n, gq, ge, gq2, ge2 = MATS[it % len(MATS)]
        mat = po.simple_mat(5, a, b, sc_n)
        flag = po.EQX | extra[it % len(extra)]
        eb = int(rng.choice([0, 5, 30, 100]))
        (q, t), = synth.ragged_pairs(rng, 1, 1, 900 if it % 40 == 0 else 260, sub=0.02 + 0.12 * rng.random(), ind=0.2 * rng.random(),
                                     indel_mean=1.5 if it % 3 else 5.0, n_rate=0.02 if it % 6 == 0 else 0.0)
        w = -1 if it % 2 else 400 + abs(len(q) - len(t))
        res = ref_extd2_prealloc(q, t, mat, gq, ge, gq2, ge2, w, eb, flag)
        seqs += [q, t]
        seq_off += [seq_off[-1] + len(q), seq_off[-1] + len(q) + len(t)]
        params.append([a, b, sc_n, gq, ge, gq2, ge2, w, eb, flag])
        expect.append([res[f] for f in FIELDS])
        cigs += res["cigar"]
        cig_off.append(len(cigs))
    np.savez_compressed(os.path.join(GOLD, "eqx_cases.npz"), seq=np.concatenate(seqs).astype(np.uint8), seq_off=np.array(seq_off, dtype=np.int64),
                        params=np.array(params, dtype=np.int32), expect=np.array(expect, dtype=np.int64),
                        cigar=np.array(cigs, dtype=np.uint32), cigar_off=np.array(cig_off, dtype=np.int64))
    ops = np.array(cigs, dtype=np.uint32) & 0xf
    print("wrote", n_cases, "cases,", os.path.getsize(os.path.join(GOLD, "eqx_cases.npz")) // 1024, "KiB; ops =", int((ops == 7).sum()), "X", int((ops == 8).sum()),
          "M", int((ops == 0).sum()))


if __name__ == "__main__":
    if po.ref_lib() is None and not po.build_ref():
        sys.exit("reference sources not available: golden vectors can only be regenerated in the build container")
    main()
