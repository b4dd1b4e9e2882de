"""End-to-end (host buffers in, ksw_extz_t out) rates of the one-shot entry points on config 2; run on the GPU box.
Reported in DESIGN.md for information -- bench.py's `value` is the HBM-resident rate.

  python tools/scripts/pcie_inclusive.py            one calling thread, plus phases and single-call latency
  python tools/scripts/pcie_inclusive.py 8          8 host threads, each aligning its own slice of the batch
"""
import ctypes
import sys
import threading
import time

sys.path.insert(0, '.')
import ksw2_amd as ka                    # noqa: E402
from ksw2_amd import synth               # noqa: E402

CELLS = 4055891968
N = 65536
lib = ka.library()
mat = synth.simple_mat(5, 2, 4, -1)
q, t = synth.fixed_batch(2, N, 512, 512)
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 1

if nthreads == 1:
    b = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=1)
    ez = (ka.KswExtz * N)()
    for it in range(3):
        t0 = time.perf_counter()
        rc = lib.lib.ksw2amd_extz_batch(None, ctypes.byref(b.sc), b.n, b.pairs, ez)
        t1 = time.perf_counter()
        print('one-shot ksw2amd_extz_batch cfg2: %.1f ms -> %.1f GCUPS (pack + H2D + kernel + D2H)' % ((t1 - t0) * 1e3, CELLS / (t1 - t0) / 1e9))
    t0 = time.perf_counter(); p = b.plan(False); t1 = time.perf_counter(); p.run(); r = p.fetch_raw(); t2 = time.perf_counter()
    print('plan_create %.1f ms, run+fetch_raw %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    one_q, one_t = q[0], t[0]
    lib.extz2(one_q, one_t, mat, 4, 2, w=64, flag=1)
    t0 = time.perf_counter()
    for _ in range(50):
        lib.extz2(one_q, one_t, mat, 4, 2, w=64, flag=1)
    print('single ksw_extz2_sse call: %.3f ms' % ((time.perf_counter() - t0) / 50 * 1e3))
else:
    per = N // nthreads
    batches = [lib.make_batch(q[i * per:(i + 1) * per], t[i * per:(i + 1) * per], mat, 4, 2, 24, 1, w=64, zdrop=-1, flag=1) for i in range(nthreads)]
    ezs = [(ka.KswExtz * per)() for _ in range(nthreads)]
    reps, rounds = 6, 4
    bar = threading.Barrier(nthreads + 1)

    def work(i):                                # a pool thread: buffers, pinned staging and stream stay cached between calls
        for _ in range(rounds):
            bar.wait()
            for _ in range(reps):               # ctypes releases the GIL inside the call
                lib.lib.ksw2amd_extz_batch(None, ctypes.byref(batches[i].sc), batches[i].n, batches[i].pairs, ezs[i])
            bar.wait()

    th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    for x in th:
        x.start()
    for rnd in range(rounds):
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = (time.perf_counter() - t0) / reps
        print('%d host threads x %d pairs: %.1f ms per %d pairs -> %.1f GCUPS end to end' % (nthreads, per, dt * 1e3, per * nthreads, CELLS * (per * nthreads / N) / dt / 1e9))
    for x in th:
        x.join()
