#!/usr/bin/env python3
"""Per-step wall time of one batch entry point on a one-shape batch (config 2 by default): pointer entry and flat entry, streamed or
not (KSW2AMD_STREAM), to see the steady state and any drift.  usage: stream_steps.py [n qlen tlen w steps]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ksw2_amd as ka                     # noqa: E402
from ksw2_amd import synth                # noqa: E402

a = [int(x) for x in sys.argv[1:]]
n, ql, tl, w, steps = (a + [65536, 512, 512, 64, 40][len(a):])[:5]
L = ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
q, t = synth.fast_fixed(2, n, ql, tl, sub=0.05, ind=0.06)
cells = n * (ql * (2 * w + 1) - w * (w + 1)) if w < ql else n * ql * tl
b = L.make_batch(q, t, mat, 4, 2, 24, 1, w=w, zdrop=-1, end_bonus=0, flag=1)
fb = L.make_flat_batch(q, t, mat, 4, 2, 24, 1, w=w, zdrop=-1, end_bonus=0, flag=1)
fb.register()
ez = (ka.KswExtz * n)()
only = os.environ.get("STEPS_ONLY", "")
order = os.environ.get("STEPS_ORDER", "pointer,flat").split(",")
calls = dict((("pointer", lambda: L.lib.ksw2amd_extz_batch(None, ctypes.byref(b.sc), b.n, b.pairs, ez)),
                   ("flat", lambda: L.lib.ksw2amd_extz_batch_flat(None, ctypes.byref(fb.sc), fb.n, ctypes.byref(fb.flat), ez))))
for name in order:
    call = calls[name]
    if only and name != only:
        continue
    ts = []
    for k in range(steps):
        t0 = time.perf_counter()
        rc = call()
        ts.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0, L.last_error()
    ts = np.array(ts)
    print("%-8s ms/step: first %s ... median %.3f  min %.3f  max(after 5) %.3f  -> %.0f GCUPS at the median   %s" % (
        name, " ".join("%.2f" % x for x in ts[:6]), np.median(ts[5:]), ts[5:].min(), ts[5:].max(), cells / np.median(ts[5:]) / 1e6, L.stream_stats()), flush=True)
    print("         every 4th: " + " ".join("%.2f" % x for x in ts[::4]))
fb.unregister()
