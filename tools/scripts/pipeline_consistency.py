"""GPU box: the pooled / chunked batch paths against the single-plan path (KSW2AMD_THREADS=0) on the same inputs, every pair, and
against the oracle on a sample.  Uniform batches whose sizes are not multiples of a device fill, ragged batches, all four functions.
usage: python tools/scripts/pipeline_consistency.py [seed]"""
import os
import sys
import numpy as np
sys.path.insert(0, '.')
import ksw2_amd as ka                         # noqa: E402
from ksw2_amd import synth                    # noqa: E402
from oracle import pyoracle as po             # noqa: E402
from tests.parity_util import diff            # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
rng = np.random.Generator(np.random.PCG64(seed))
lib = ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
total = 0


def both(fn):
    os.environ.pop("KSW2AMD_THREADS", None)
    a = fn()
    os.environ["KSW2AMD_THREADS"] = "0"
    b = fn()
    os.environ.pop("KSW2AMD_THREADS", None)
    return a, b


for n, L, w, dual, flag in [(2049, 300, 20, False, po.SCORE_ONLY), (5000, 300, 20, False, 0), (12345, 150, 16, True, 0), (70001, 120, 10, False, po.SCORE_ONLY),
                            (4097, 2100, 300, True, 0), (9000, 2100, 300, False, po.SCORE_ONLY), (3000, 6000, 500, False, po.SCORE_ONLY), (2500, 6000, 500, False, 0)]:
    qs, ts = synth.fixed_batch(int(rng.integers(1 << 30)), n, L, L, sub=0.05, ind=0.08, tail_random_frac=0.2, tail_pairs=0.1)
    qs, ts = list(qs), list(ts)
    run = lambda: (lib.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=200, end_bonus=0, flag=flag) if dual else
                   lib.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=200, end_bonus=0, flag=flag))
    s0 = lib.host_stats()
    a, b = both(run)
    s1 = lib.host_stats()
    bad = [i for i in range(n) if diff(a[i], b[i])]
    assert not bad, ("pooled vs single plan", n, L, w, dual, flag, bad[:5])
    fb = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=200, end_bonus=0, flag=flag)        # the flat entry, pooled, page-locked arena
    fb.register()
    c = fb.run_oneshot(dual)
    fb.unregister()
    bad = [i for i in range(n) if diff(a[i], c[i])]
    assert not bad, ("flat entry vs pointer entry", n, L, w, dual, flag, bad[:5])
    for i in rng.choice(n, size=min(n, 60 if L > 1000 else 200), replace=False):
        exp = po.align("oracle", "extd" if dual else "extz", qs[i], ts[i], mat, 4, 2, 24, 1, w=w, zdrop=200, end_bonus=0, flag=flag)
        assert not diff(a[i], exp), ("oracle", n, L, w, dual, flag, int(i))
    total += n
    print("uniform n=%d L=%d w=%d dual=%d flag=%#x: ok, %d chunks" % (n, L, w, dual, flag, s1["pool_chunks"] - s0["pool_chunks"]), flush=True)

for n, hi, dual, flag in [(3000, 800, False, 0), (20000, 400, True, po.SCORE_ONLY), (6000, 3000, True, 0)]:
    pr = synth.ragged_pairs(rng, n, 30, hi, sub=0.05, ind=0.1, n_rate=0.002)
    qs, ts = [p[0] for p in pr], [p[1] for p in pr]
    wv = rng.choice([-1, 10, 64, 200], size=n)
    run = lambda: (lib.extd_batch(qs, ts, mat, 4, 2, 24, 1, w=wv, zdrop=150, end_bonus=5, flag=flag) if dual else
                   lib.extz_batch(qs, ts, mat, 4, 2, w=wv, zdrop=150, end_bonus=5, flag=flag))
    a, b = both(run)
    bad = [i for i in range(n) if diff(a[i], b[i])]
    assert not bad, ("ragged pooled vs single plan", n, hi, dual, flag, bad[:5])
    c = lib.make_flat_batch(qs, ts, mat, 4, 2, 24, 1, w=wv, zdrop=150, end_bonus=5, flag=flag).run_oneshot(dual)      # wildcards (n_rate): re-runs
    bad = [i for i in range(n) if diff(a[i], c[i])]
    assert not bad, ("ragged flat entry vs pointer entry", n, hi, dual, flag, bad[:5])
    for i in rng.choice(n, size=150, replace=False):
        exp = po.align("oracle", "extd" if dual else "extz", qs[i], ts[i], mat, 4, 2, 24, 1, w=int(wv[i]), zdrop=150, end_bonus=5, flag=flag)
        assert not diff(a[i], exp), ("oracle ragged", n, hi, dual, flag, int(i))
    total += n
    print("ragged n=%d hi=%d dual=%d flag=%#x: ok" % (n, hi, dual, flag), flush=True)
print("pipeline consistency ok: %d alignments, seed %d" % (total, seed))
