"""Latency of small one-shape batches under the kernel-selection switches (GPU box):  python tools/scripts/latency_probe.py [len] [band]
n pairs of len x len through the batch entry point, 300 calls each, ms per call (host + launch + kernel + fetch: the kernel's latency
is what differs between the rows).  What the single-pair callers' coalesced batches (16-64 pairs of one shape) should run on."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
ENVS = [("default", {}), ("PK_FIRST=1", {"KSW2AMD_PK_FIRST": "1"}), ("PK_FIRST=2 (-> solo by the SIMD rule)", {"KSW2AMD_PK_FIRST": "2"}),
        ("PK_FIRST=2 SIMDS=0 (pk(64,8) pairs)", {"KSW2AMD_PK_FIRST": "2", "KSW2AMD_SIMDS": "0"}),
        ("PK_FIRST=3 SIMDS=0 (pk(64,16) pairs)", {"KSW2AMD_PK_FIRST": "3", "KSW2AMD_SIMDS": "0"}),
        ("NO_PK=1 (int32)", {"KSW2AMD_NO_PK": "1"}), ("SOLO=all", {"KSW2AMD_SOLO": "all"})]
KEYS = sorted({k for _, e in ENVS for k in e})
for flag, what in ((ka.KSW_EZ_SCORE_ONLY, "score only"), (0, "CIGAR")):
    for n in (1, 2, 16, 64, 256):
        q, t = synth.fixed_batch(5, n, L, L, sub=0.05, ind=0.02)
        ref = None
        row = []
        for name, env in ENVS:
            for k in KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            for _ in range(20):
                r = lib.extz_batch(q, t, mat, 4, 2, w=W, zdrop=-1, flag=flag)
            t0 = time.perf_counter()
            for _ in range(300):
                r = lib.extz_batch(q, t, mat, 4, 2, w=W, zdrop=-1, flag=flag)
            ms = (time.perf_counter() - t0) / 300 * 1e3
            key = [(x["score"], x["max"], x["max_q"], x["max_t"], tuple(x["cigar"])) for x in r]
            assert ref is None or key == ref, (name, n)
            ref = key
            row.append("%s %.3f" % (name.split(" ")[0], ms))
        print("%s, n = %3d, %d x %d, w = %d:  %s" % (what, n, L, L, W, " | ".join(row)))
