import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import ksw2_amd as ka
from oracle.gen_golden_extf import noisy_pair
L = ka.Library()
rng = np.random.Generator(np.random.PCG64(5))
base = [noisy_pair(rng, 1000, k % 3) for k in range(64)]
qs = [base[k % 64][0] for k in range(16384)]; ts = [base[k % 64][1] for k in range(16384)]
b = L.make_linear_batch(qs, ts, 2, -4, 2, w=100, xdrop=-1)
ez = (ka.KswExtz * b.n)()
def run():
    L._check(L.lib.ksw2amd_extf_batch(None, *b.par, b.n, b.pairs, ez))
for _ in range(3): run()
t0 = time.perf_counter()
for _ in range(10): run()
print("batch ms/step", (time.perf_counter() - t0) * 100)
for _ in range(2):
    t0 = time.perf_counter(); p = b.plan(); t1 = time.perf_counter(); p.run(); p.sync() if hasattr(p, "sync") else None; t2 = time.perf_counter(); r = p.fetch_raw(); t3 = time.perf_counter(); p.close()
    print("plan create %.2f run %.2f fetch %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
