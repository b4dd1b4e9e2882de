#!/usr/bin/env python3
"""Small random batches through streamed plans (forced) against the same batches unstreamed, many rounds per setting; counts the rounds
with a difference.  GPU box.   usage: stream_stress.py [rounds]"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402
from tests.parity_util import diff       # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
SET = [{"KSW2AMD_STREAM_PIECE_KB": "64"}, {"KSW2AMD_STREAM_PIECE_KB": "256"}, {"KSW2AMD_STREAM_PIECE_KB": "8192"},
       {"KSW2AMD_STREAM_PIECE_KB": "64", "KSW2AMD_NO_PARCOPY": "1"}, {"KSW2AMD_STREAM_PIECE_KB": "64", "KSW2AMD_STREAM_SLEEP_US": "50"},
       {"KSW2AMD_STREAM_PIECE_KB": "64", "KSW2AMD_DEFER": "1"}]
for extra in SET:
    bad = tot = 0
    first = None
    rng = np.random.Generator(np.random.PCG64(11))
    for r in range(rounds):
        n = int(rng.integers(4, 40))
        hi = int(rng.choice([150, 700, 2500, 7000]))
        dual = bool(r & 1)
        if rng.random() < 0.5:
            ql = int(rng.integers(20, hi)); tl = max(1, ql + int(rng.integers(-40, 40)))
            qs, ts = synth.fixed_batch(int(rng.integers(1 << 30)), n, ql, tl, sub=0.05, ind=0.1, tail_random_frac=0.3, tail_pairs=0.3)
            qs, ts = list(qs), list(ts)
        else:
            pr = synth.ragged_pairs(rng, n, 1, hi, sub=0.05, ind=0.12)
            qs, ts = [p[0] for p in pr], [p[1] for p in pr]
        w = rng.choice([-1, 0, 5, 20, 64, 100, 284, 500], size=n)
        zd = rng.choice([-1, 50, 200, 400], size=n)
        fl = np.full(n, 1 | (8 if rng.random() < 0.2 else 0))
        flat = rng.random() < 0.4
        out = []
        for env in ({"KSW2AMD_STREAM": "0"}, dict(extra, KSW2AMD_STREAM="1")):
            for k in list(os.environ):
                if k.startswith("KSW2AMD_"):
                    del os.environ[k]
            os.environ["KSW2AMD_SIMDS"] = "0"
            os.environ.update(env)
            if flat:
                fb = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=0, flag=fl)
                out.append(fb.run_oneshot(dual))
            else:
                out.append(lib.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=fl) if dual else lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, flag=fl))
        d = [i for i in range(n) if diff(out[0][i], out[1][i])]
        tot += 1
        if d:
            bad += 1
            if first is None:
                first = (r, n, dual, flat, d[:4], [(len(qs[i]), len(ts[i])) for i in d[:4]], sum(len(a) + len(b) for a, b in zip(qs, ts)))
    print(extra, "rounds with a difference: %d / %d" % (bad, tot), "first:", first, lib.stream_stats(), flush=True)
