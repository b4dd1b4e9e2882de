#!/usr/bin/env python3
"""Streamed plans against the chunked path on the same inputs (GPU box, or KSW2AMD_CHECK_LIB=tests/sim/libksw2_amd_sim.so):
one-shape batches through the batch entry points with KSW2AMD_STREAM = 0 / 1 / unset, small pieces, a slowed-down upload (the
kernel really waits), and the fault hook (the launch times out, the plan is repeated).  Every field of every pair must agree."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ksw2_amd as ka                     # noqa: E402
from ksw2_amd import synth                # noqa: E402

L = ka.Library(os.environ["KSW2AMD_CHECK_LIB"]) if os.environ.get("KSW2AMD_CHECK_LIB") else ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
SIM = L.backend() == "sim"


def run(q, t, w, flag, dual, flat, **env):
    for k in list(os.environ):
        if k.startswith("KSW2AMD_STREAM"):
            del os.environ[k]
    os.environ.update({k: str(v) for k, v in env.items()})
    s0 = L.stream_stats()
    t0 = time.perf_counter()
    if flat:
        fb = L.make_flat_batch(q, t, mat, 4, 2, 24, 1, w=w, zdrop=100, end_bonus=0, flag=flag)
        fb.register()
        r = fb.run_oneshot(dual)
        fb.unregister()
    elif dual:
        r = L.extd_batch(q, t, mat, 4, 2, 24, 1, w=w, zdrop=100, flag=flag)
    else:
        r = L.extz_batch(q, t, mat, 4, 2, w=w, zdrop=100, flag=flag)
    dt = time.perf_counter() - t0
    s1 = L.stream_stats()
    return r, {k: s1[k] - s0[k] for k in s1}, dt


bad_total = 0
cases = [(65536, 512, 512, 64, 1, False), (9000, 2000, 2100, 300, 1, False), (20001, 500, 520, 64, 1, True), (8192, 1000, 1000, 100, 0x09, False)]
if SIM:
    cases = [(c[0] // 16, c[1] // 2, c[2] // 2, c[3], c[4], c[5]) for c in cases]
for (n, ql, tl, w, flag, dual) in cases:
    q, t = synth.fixed_batch(3, n, ql, tl, sub=0.05, ind=0.06)
    dq, dt = synth.fixed_batch(4, n, ql, tl, sub=0.05, ind=0.06)       # decoy: same shapes, other bases -- run before every streamed run, so that arenas,
    for flat in (False, True):                                         # staging and caches hold ITS bytes (a repeated batch hides stale or early reads)
        r0, s0, d0 = run(q, t, w, flag, dual, flat, KSW2AMD_STREAM=0)
        outs = []
        for name, env in (("auto", {}), ("forced small pieces", dict(KSW2AMD_STREAM=1, KSW2AMD_STREAM_PIECE_KB=256)),
                          ("slow upload", dict(KSW2AMD_STREAM=1, KSW2AMD_STREAM_PIECE_KB=1024, KSW2AMD_STREAM_SLEEP_US=300)),
                          ("fault", dict(KSW2AMD_STREAM=1, KSW2AMD_STREAM_FAULT=1, KSW2AMD_STREAM_TIMEOUT_MS=20))):
            run(dq, dt, w, flag, dual, flat, **env)
            outs.append((name, run(q, t, w, flag, dual, flat, **env)))
        for name, (r, s, d) in outs:
            bad = [i for i in range(n) if any(r0[i][f] != r[i][f] for f in ka.FIELDS + ["cigar"])]
            bad_total += len(bad)
            print("n=%d %dx%d w=%d flag=%#x dual=%d flat=%d  %-20s mismatches %d  %s  %.1f ms (chunked %.1f ms)" % (n, ql, tl, w, flag, dual, flat, name, len(bad), s, d * 1e3, d0 * 1e3), flush=True)
print("TOTAL MISMATCHES", bad_total)
sys.exit(1 if bad_total else 0)
