#!/usr/bin/env python3
"""Replay the batch tools/scripts/fuzz_gpu.py dumped at its first difference (/tmp/fuzz_fail.pkl) under a list of environments,
several times each: is the difference deterministic, and which switch does it follow?   usage: fuzz_replay.py [pkl]"""
import os
import pickle
import sys

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from oracle import pyoracle as po        # noqa: E402
from tests.parity_util import diff       # noqa: E402

d = pickle.load(open(sys.argv[1] if len(sys.argv) > 1 else "/tmp/fuzz_fail.pkl", "rb"))
lib = ka.Library(os.environ["KSW2AMD_FUZZ_LIB"]) if os.environ.get("KSW2AMD_FUZZ_LIB") else ka.library()
qs, ts, mat, (q, e, q2, e2), w, zd, eb, fl, dual = d["qs"], d["ts"], d["mat"], d["sc"], d["w"], d["zd"], d["eb"], d["fl"], d["dual"]
n = len(qs)
print("n", n, "dual", dual, "lens", [(len(a), len(b)) for a, b in zip(qs, ts)][:12], "w", list(w)[:12], "flags", [hex(int(x)) for x in fl][:12])
exp = [po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, q, e, q2, e2, w=int(w[i]), zdrop=int(zd[i]), end_bonus=int(eb[i]), flag=int(fl[i])) for i in range(n)]
for env in ({"KSW2AMD_STREAM": "0", "KSW2AMD_SIMDS": "0"}, {"KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "64", "KSW2AMD_SIMDS": "0"},
            {"KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "64", "KSW2AMD_SIMDS": "0", "KSW2AMD_NO_PARCOPY": "1"},
            {"KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "4096", "KSW2AMD_SIMDS": "0"},
            {"KSW2AMD_STREAM": "1", "KSW2AMD_STREAM_PIECE_KB": "256", "KSW2AMD_DEFER": "1", "KSW2AMD_SIMDS": "0"},
            {"KSW2AMD_STREAM": "0", "KSW2AMD_DEFER": "1", "KSW2AMD_SIMDS": "0"}):
    for k in list(os.environ):
        if k.startswith("KSW2AMD_") and k != "KSW2AMD_FUZZ_LIB":
            del os.environ[k]
    os.environ.update(env)
    # alternate the two entries (their arenas are laid out differently): whatever a run reads that was not uploaded for IT differs from what it should read
    counts = {"pointer": [0, 0], "flat": [0, 0]}
    rng = np.random.Generator(np.random.PCG64(5))
    for rep in range(16):
        # different data in between, so that neither staging nor device buffers hold this batch's bytes from an earlier repetition
        qq = [rng.integers(0, 4, len(x), dtype=np.uint8) for x in qs]
        tt = [rng.integers(0, 4, len(x), dtype=np.uint8) for x in ts]
        (lib.extd_batch if dual else lib.extz_batch)(qq, tt, mat, *((q, e, q2, e2) if dual else (q, e)), w=w, zdrop=zd, end_bonus=eb, flag=fl)
        lib.make_flat_batch(qq, tt, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl).run_oneshot(dual)
        for flat in (False, True):
            if flat:
                res = lib.make_flat_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl).run_oneshot(dual)
            else:
                res = lib.extd_batch(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl) if dual else lib.extz_batch(qs, ts, mat, q, e, w=w, zdrop=zd, end_bonus=eb, flag=fl)
            bad = [i for i in range(n) if diff(exp[i], res[i])]
            counts["flat" if flat else "pointer"][0] += 1
            counts["flat" if flat else "pointer"][1] += bool(bad)
    print(env, "runs with a difference / runs:", counts, lib.stream_stats(), flush=True)
