"""Where does one read per wavefront (solo kernel) beat two per wavefront (packed pairs) and the int32 kernels?
Long reads, w = 500, by batch size; same-shape batches (pairs possible) and unique shapes (self-paired / solo / int32).
GPU box:  python tools/scripts/solo_crossover.py"""
import os
import sys

import numpy as np

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402

lib = ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
rng = np.random.Generator(np.random.PCG64(1))


def run(qs, ts, flag, dual, env):
    for k in ("KSW2AMD_SOLO", "KSW2AMD_NO_PK", "KSW2AMD_SIMDS", "KSW2AMD_DEFER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    p = lib.make_batch(qs, ts, mat, 4, 2, 24, 1, w=500, zdrop=400, flag=flag).plan(dual)
    p.run(); p.timing()
    ms = []
    for _ in range(3):
        p.run(); ms.append(p.timing()[1])
    d = p.describe()
    r = (float(np.mean(ms)), p.cells() / np.mean(ms) / 1e6, d)
    p.close()
    return r


for shape in ("same", "unique"):
    for flag, dual, name in ((1, False, "score"), (0, False, "cigar"), (0, True, "cigar-dual")):
        for n in (256, 512, 1024, 1536, 2048, 3072, 4096, 8192):
            if shape == "same":
                qs, ts = synth.fixed_batch(7, n, 10000, 10000, sub=0.05, ind=0.06)
            else:
                pairs = synth.ragged_pairs(rng, n, 8000, 12000, sub=0.05, ind=0.06)
                qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
            out = []
            for lab, env in (("auto", {}), ("packed", {"KSW2AMD_SOLO": "0", "KSW2AMD_SIMDS": "0"}), ("solo", {"KSW2AMD_SOLO": "all", "KSW2AMD_SIMDS": "0"}), ("int32", {"KSW2AMD_NO_PK": "1"})):
                ms, g, d = run(qs, ts, flag, dual, env)
                out.append("%s %7.2f ms %6.0f" % (lab, ms, g))
                if lab == "auto":
                    out.append("[" + "; ".join("%s(%s,%s) x%s" % (x.get("kernel"), x.get("G"), x.get("C"), x.get("tasks")) for x in d) + "]")
            print("%-6s %-10s n=%5d  %s" % (shape, name, n, "  ".join(out)), flush=True)
