"""Does the end-to-end time of a batch drift over many calls?  (GPU box)  python tools/scripts/step_drift_probe.py [workload] [calls]
Config 2 through the pointer batch entry, wall time per call averaged over blocks of 50 calls, plus the library's counters."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import bench                              # noqa: E402
import ksw2_amd                           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 600
lib = ksw2_amd.library()
job = bench.Job(lib, name, bench.WORKLOADS[name], 0)
for flat in (False, True):
    if flat and not job.flat_ready():
        break
    for _ in range(5):
        job.e2e_step(flat=flat)
    out = []
    for b in range(calls // 50):
        t0 = time.perf_counter()
        for _ in range(50):
            job.e2e_step(flat=flat)
        out.append((time.perf_counter() - t0) / 50 * 1e3)
    print(name, "flat" if flat else "pointers", "ms per call, blocks of 50:", " ".join("%.2f" % x for x in out), "| host stats", lib.host_stats())
