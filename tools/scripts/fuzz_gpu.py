"""Randomised GPU-vs-oracle soak (GPU box):  python tools/scripts/fuzz_gpu.py [seconds] [seed] [long]
("long": few long reads per batch -- re-based, LDS-row, solo and generation-serial classes on 5-20 k sequences.)
Ragged batches of every function on the path under the kernel-selection switches; stops at the first difference."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402
from oracle import pyoracle as po        # noqa: E402
from oracle.gen_golden_extf import noisy_pair          # noqa: E402
from oracle.gen_golden_exts import spliced_pair        # noqa: E402
from tests.parity_util import check_batch, diff        # noqa: E402
from tests import golden_util as gu                    # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
LONG = len(sys.argv) > 3
rng = np.random.Generator(np.random.PCG64(seed))
lib = ka.Library(os.environ["KSW2AMD_FUZZ_LIB"]) if os.environ.get("KSW2AMD_FUZZ_LIB") else ka.library()      # the simulator build, for reproducing on the CPU
ENVS = [{}, {"KSW2AMD_SOLO": "1"}, {"KSW2AMD_SOLO": "all"}, {"KSW2AMD_LDSROWS": "0"}, {"KSW2AMD_LDSROWS": "1"}, {"KSW2AMD_NO_PK": "1"},
        {"KSW2AMD_SIMDS": "0"}, {"KSW2AMD_EXTF_WIN": "1"}, {"KSW2AMD_EXTF_LDS": "1"}, {"KSW2AMD_EXTS_REG": "1"},
        {"KSW2AMD_POOL_MIN": "8", "KSW2AMD_THREADS": "3"}, {"KSW2AMD_POOL_MIN": "4", "KSW2AMD_SIMDS": "0"}, {"KSW2AMD_NO_PKMP": "1"},
        {"KSW2AMD_LDSCODES": "0", "KSW2AMD_SIMDS": "0"}, {"KSW2AMD_LDSCODES": "1", "KSW2AMD_SIMDS": "0"},
        {"KSW2AMD_LDSCODES": "1", "KSW2AMD_LDSROWS": "1"}, {"KSW2AMD_DEFER": "1", "KSW2AMD_SIMDS": "0"}, {"KSW2AMD_DEFER": "0"},
        {"KSW2AMD_DEFER": "1"}, {"KSW2AMD_SOLO": "0"}, {"KSW2AMD_EXTF_LANE": "1", "KSW2AMD_EXTF_RING": "1"}, {"KSW2AMD_EXTF_LANE": "1", "KSW2AMD_EXTF_RING": "0"},
        # streamed plans (round 4): every plan that can, the arena in small pieces; with the packed kernels forced for small test batches
        {"KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "64", "KSW2AMD_SIMDS": "0"}, {"KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "256", "KSW2AMD_DEFER": "1", "KSW2AMD_SIMDS": "0"},
        {"KSW2AMD_STREAM": "0"},
        # round 5: the SSE-compatible score-only tasks and the narrow-band X-drop extensions through their other kernels; two copy lanes
        {"KSW2AMD_SSEC_BLK": "0"}, {"KSW2AMD_EXTF_GRP": "0"}, {"KSW2AMD_EXTF_GRP": "1", "KSW2AMD_EXTF_LDS": "1"}, {"KSW2AMD_EXTF_GRP": "1"}, {"KSW2AMD_EXTF_GRP": "2", "KSW2AMD_EXTF_WIN": "1"},
        {"KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "64", "KSW2AMD_STREAM_LANES": "2", "KSW2AMD_SIMDS": "0"},
        # uniform plans (one shape, one parameter set, >= 2 048 pairs: the `uniform` rounds below build such batches; a POOL_MIN here would switch the route off)
        {"KSW2AMD_UNIFORM": "1", "KSW2AMD_SIMDS": "0"}, {"KSW2AMD_UNIFORM": "1", "KSW2AMD_WIRE4": "0", "KSW2AMD_SIMDS": "0"}, {"KSW2AMD_UNIFORM": "1", "KSW2AMD_WIRE2": "1", "KSW2AMD_SIMDS": "0"},
        # round 6: target wildcards as rows of the packed kernels (default) against the rule before it
        {"KSW2AMD_TN": "0"}, {"KSW2AMD_TN": "0", "KSW2AMD_SIMDS": "0", "KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "64"}]
KEYS = sorted({k for e in ENVS for k in e})
t0 = time.time()
rounds = pairs = 0
while time.time() - t0 < budget:
    env = ENVS[rounds % len(ENVS)]
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    kind = rounds % 2 if LONG else rounds % 4
    if "KSW2AMD_UNIFORM" in env and not LONG:
        # one shape, one parameter set, score only, >= 2 048 pairs: the uniform plan (device-built records, 4-bit wire format or not), with
        # now and then a wildcard run in a target, a query wildcard, and a residue code above 15 (does not fit the wire format: general path)
        n = int(rng.integers(1024, 1200)) * 2                             # (an even count: the odd one out of a packed class ends the uniform route)
        ql = int(rng.integers(30, 260)); tl = max(1, ql + int(rng.integers(-20, 20)))
        w_, zd_ = int(rng.choice([8, 20, 64, 300])), int(rng.choice([-1, 40, 200]))
        a, b, q, e = [(2, 4, 4, 2), (1, 3, 4, 1)][int(rng.integers(2))]
        mat = synth.simple_mat(5, a, b, int(rng.choice([0, -1, -3])))
        qs, ts = synth.fixed_batch(int(rng.integers(1 << 30)), n, ql, tl, sub=0.05, ind=0.08, tail_random_frac=0.3, tail_pairs=0.2)
        qs, ts = [np.array(x, dtype=np.uint8) for x in qs], [np.array(x, dtype=np.uint8) for x in ts]
        for i in rng.integers(0, n, size=int(rng.integers(0, 6))):
            ln = int(rng.integers(1, min(tl, 30) + 1)); at = int(rng.integers(0, tl - ln + 1)); ts[int(i)][at:at + ln] = 4
        for i in rng.integers(0, n, size=int(rng.integers(0, 4))):
            qs[int(i)][int(rng.integers(ql))] = 4
        fl_ = po.SCORE_ONLY | (po.EXTZ_ONLY if rng.random() < 0.3 else 0)
        s0 = lib.stream_stats()["streamed_plans"]
        res = lib.extz_batch(qs, ts, mat, q, e, w=w_, zdrop=zd_, end_bonus=int(rng.choice([0, 7])), flag=fl_)
        assert lib.stream_stats()["streamed_plans"] == s0 + 1, ("uniform route not taken", env, n, ql, tl, w_)
        os.environ["KSW2AMD_UNIFORM"] = "0"
        ref = lib.extz_batch(qs, ts, mat, q, e, w=w_, zdrop=zd_, end_bonus=0, flag=fl_)
        os.environ["KSW2AMD_UNIFORM"] = "1"
        got = lib.extz_batch(qs, ts, mat, q, e, w=w_, zdrop=zd_, end_bonus=0, flag=fl_)
        badi = [i for i in range(n) if diff(ref[i], got[i])]
        assert not badi, ("uniform vs general path", env, n, ql, tl, w_, zd_, badi[:4])
        for i in [int(x) for x in rng.integers(0, n, size=32)] + [0, n - 1]:
            exp = po.align("oracle", "extz2", qs[i], ts[i], mat, q, e, w=w_, zdrop=zd_, end_bonus=0, flag=fl_)
            assert not diff(exp, got[i]), ("uniform vs oracle", env, i, ql, tl, w_, zd_, diff(exp, got[i]))
        if rng.random() < 0.3:                                              # a code the wire format cannot hold
            qs[3][1] = 20
            os.environ["KSW2AMD_UNIFORM"] = "0"
            ref = lib.extz_batch(qs, ts, mat, q, e, w=w_, zdrop=zd_, end_bonus=0, flag=fl_)
            os.environ["KSW2AMD_UNIFORM"] = "1"
            got = lib.extz_batch(qs, ts, mat, q, e, w=w_, zdrop=zd_, end_bonus=0, flag=fl_)
            assert not [i for i in range(n) if diff(ref[i], got[i])], ("uniform, code above 15", env)
        pairs += n
        rounds += 1
        continue
    if kind < 2:                                                          # extz2 / extd2
        dual = bool(kind)
        a, b, q, e, q2, e2 = [(2, 4, 4, 2, 24, 1), (1, 3, 4, 1, 24, 1), (2, 4, 4, 2, 13, 1), (2, 5, 5, 3, 20, 2)][int(rng.integers(4))]
        mat = synth.simple_mat(5, a, b, int(rng.choice([0, -1, -3])))
        n = int(rng.integers(2, 9)) if LONG else int(rng.integers(4, 40))
        hi = int(rng.choice([5000, 9000, 14000, 20000])) if LONG else int(rng.choice([150, 700, 2500, 7000]))
        if rng.random() < 0.5:                                             # same-shape batch: the packed pairs
            ql = int(rng.integers(hi // 2 if LONG else 20, hi)); tl = max(1, ql + int(rng.integers(-40, 40)))
            qs, ts = synth.fixed_batch(int(rng.integers(1 << 30)), n, ql, tl, sub=0.05, ind=0.1, tail_random_frac=0.3, tail_pairs=0.3)
            qs, ts = list(qs), list(ts)
            if rng.random() < 0.35:                                        # wildcard runs in some targets: rows of the packed kernels (round 6), or int32 / handed back (KSW2AMD_TN=0)
                ts = [np.array(x, dtype=np.uint8) for x in ts]
                for k in range(0, n, int(rng.integers(1, 4))):
                    ln = int(rng.integers(1, min(len(ts[k]), 50) + 1)); at = int(rng.integers(0, len(ts[k]) - ln + 1)); ts[k][at:at + ln] = 4
        else:
            pr = synth.ragged_pairs(rng, n, hi // 3 if LONG else 1, hi, sub=0.05, ind=0.12, n_rate=0.01 if rng.random() < 0.2 else 0.0)
            qs, ts = [p[0] for p in pr], [p[1] for p in pr]
        w = rng.choice([100, 284, 285, 330, 500, 536, 537, 800, 1040, 1041, 3000] + ([-1] if hi <= 9000 else []), size=n) if LONG else \
            rng.choice([-1, 0, 1, 5, 16, 20, 64, 68, 100, 284, 285, 330, 500, 536, 537, 1040, 1041], size=n)
        zd = rng.choice([-1, 50, 200, 400, 2000], size=n)
        eb = rng.choice([0, 10, 50], size=n)
        mode = int(rng.choice([po.SCORE_ONLY, 0, po.RIGHT]))
        fl = np.array([mode | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0) |
                       (po.GENERIC_SC if rng.random() < 0.2 else 0) | (po.EQX if dual and rng.random() < 0.2 else 0) for _ in range(n)])
        if not LONG and rng.random() < 0.3:                                 # the SSE kernels' own results: opt-in flag, or the APPROX_DROP route
            wn = rng.choice([-1, 0, 1, 2, 5, 9, 16, 33, 64, 100, 300, 700, 959, 960], size=n)
            fl2 = np.array([int(f) & ~po.EQX | (po.EQX if dual and rng.random() < 0.1 and not (f & po.SCORE_ONLY) else 0) |
                            int(rng.choice([ka.KSW2AMD_EZ_SSE_COMPAT, ka.KSW2AMD_EZ_SSE_COMPAT | po.APPROX_MAX, po.APPROX_MAX | po.APPROX_DROP])) for f in fl])
            res = lib.extd_batch(qs, ts, mat, q, e, q2, e2, w=wn, zdrop=zd, end_bonus=eb, flag=fl2) if dual else \
                lib.extz_batch(qs, ts, mat, q, e, w=wn, zdrop=zd, end_bonus=eb, flag=fl2)
            for i, r in enumerate(res):
                exp = po.align("oracle", "extd2_sse" if dual else "extz2_sse", qs[i], ts[i], mat, q, e, q2, e2, w=int(wn[i]), zdrop=int(zd[i]),
                               end_bonus=int(eb[i]), flag=int(fl2[i]) & ~ka.KSW2AMD_EZ_SSE_COMPAT)
                assert not diff(r, exp), ("sse-compatible", env, dual, len(qs[i]), len(ts[i]), int(wn[i]), int(zd[i]), hex(int(fl2[i])))
            pairs += n
            rounds += 1
            continue
        if rng.random() < 0.15:                                             # KSW_EZ_APPROX_MAX alone: the no-maximum kernels
            fl = np.array([int(f) & ~po.EQX | po.APPROX_MAX for f in fl])
        if rng.random() < 0.25:                                             # the flat entry (one arena + offsets, unscanned upload) on the same pairs
            if rng.random() < 0.5:                                          # ... with a wildcard somewhere: reported by the packed kernels, re-run
                k = int(rng.integers(n)); qs[k] = qs[k].copy(); qs[k][int(rng.integers(len(qs[k])))] = 4
            fb = lib.make_flat_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
            ref = lib.extd_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl) if dual else lib.extz_batch(qs, ts, mat, q, e, w=w, zdrop=zd, end_bonus=eb, flag=fl)
            got = fb.run_oneshot(dual)
            for i in range(n):
                if diff(ref[i], got[i]):
                    import pickle
                    pickle.dump(dict(qs=qs, ts=ts, mat=mat, sc=(q, e, q2, e2), w=w, zd=zd, eb=eb, fl=fl, dual=dual), open("/tmp/fuzz_fail.pkl", "wb"))
                assert not diff(ref[i], got[i]), ("flat", env, dual, i, len(qs[i]), len(ts[i]), int(w[i]), hex(int(fl[i])), diff(ref[i], got[i]))
        try:
            check_batch(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl)
        except AssertionError:
            print("failing round", rounds, "env", env, "dual", dual, "scoring", (a, b, q, e, q2, e2))
            import pickle
            pickle.dump(dict(qs=qs, ts=ts, mat=mat, sc=(q, e, q2, e2), w=w, zd=zd, eb=eb, fl=fl, dual=dual), open("/tmp/fuzz_fail.pkl", "wb"))
            raise
        pairs += n
    elif kind == 2:                                                       # extf2
        n = int(rng.integers(4, 40))
        prs = [noisy_pair(rng, int(rng.integers(1, int(rng.choice([200, 1500, 6000])))), int(rng.integers(7))) for _ in range(n)]
        w = rng.choice([-1, 0, 3, 16, 33, 100, 146, 147, 158, 159, 160, 300, 402, 403, 414, 415, 416, 700, 900, 926, 927, 928], size=n)
        xd = rng.choice([-1, 10, 60, 500], size=n)
        mch, mis, e = [(1, -2, 1), (2, -4, 2), (4, -6, 3)][int(rng.integers(3))]
        res = lib.extf_batch([p[0] for p in prs], [p[1] for p in prs], mch, mis, e, w=w, xdrop=xd)
        for i, r in enumerate(res):
            exp = po.extf2("oracle", prs[i][0], prs[i][1], mch, mis, e, int(w[i]), int(xd[i]))
            assert not diff(r, exp, gu.FIELDS), ("extf", env, len(prs[i][0]), len(prs[i][1]), int(w[i]), int(xd[i]))
        pairs += n
    else:                                                                 # exts2
        n = int(rng.integers(4, 24))
        prs = [spliced_pair(rng, int(rng.integers(1, int(rng.choice([300, 1200])))), low_complexity=rng.random() < 0.15) for _ in range(n)]
        mat = po.simple_mat(5, 1, 2, 0)
        fl = [int(rng.choice([0, po.SPLICE_FOR, po.SPLICE_REV | po.RIGHT, po.SPLICE_FOR | po.EXTZ_ONLY, po.SCORE_ONLY | po.SPLICE_FOR])) for _ in range(n)]
        zd = [int(rng.choice([-1, 30, 200])) for _ in range(n)]
        res = lib.exts_batch([p[0] for p in prs], [p[1] for p in prs], mat, 2, 1, 32, 4, zdrop=zd, flag=fl)
        for i, r in enumerate(res):
            exp = po.exts2("oracle", prs[i][0], prs[i][1], mat, 2, 1, 32, 4, zdrop=zd[i], flag=fl[i])
            assert not diff(r, exp), ("exts", env, len(prs[i][0]), len(prs[i][1]), hex(fl[i]), zd[i])
        pairs += n
    rounds += 1
print("fuzz ok: %d rounds, %d alignments, %.0f s, seed %d" % (rounds, pairs, time.time() - t0, seed))
