import sys, time, ctypes, numpy as np
sys.path.insert(0, '.')
import ksw2_amd as ka
from ksw2_amd import synth
lib = ka.library()
mat = synth.simple_mat(5,2,4,-1)
q,t = synth.fixed_batch(2, 65536, 512, 512)
b = lib.make_batch(q,t,mat,4,2,24,1,w=64,zdrop=-1,flag=1)
ez = (ka.KswExtz*65536)()
for it in range(3):
    t0=time.perf_counter(); rc = lib.lib.ksw2amd_extz_batch(None, ctypes.byref(b.sc), b.n, b.pairs, ez); t1=time.perf_counter()
    print('one-shot ksw2amd_extz_batch cfg2: %.1f ms -> %.1f GCUPS (pack + H2D + kernel + D2H)'%((t1-t0)*1e3, 4055891968/(t1-t0)/1e9))
# phases
t0=time.perf_counter(); p=b.plan(False); t1=time.perf_counter(); p.run(); r=p.fetch_raw(); t2=time.perf_counter()
print('plan_create %.1f ms, run+fetch_raw %.1f ms'%((t1-t0)*1e3,(t2-t1)*1e3))
# single call latency
one_q, one_t = q[0], t[0]
lib.extz2(one_q, one_t, mat, 4, 2, w=64, flag=1)
t0=time.perf_counter()
for _ in range(50): lib.extz2(one_q, one_t, mat, 4, 2, w=64, flag=1)
print('single ksw_extz2_sse call: %.3f ms'%((time.perf_counter()-t0)/50*1e3))
