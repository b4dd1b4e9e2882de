import sys, numpy as np
sys.path.insert(0,'.')
import ksw2_amd as ka
from ksw2_amd import synth
lib = ka.library(); mat = synth.simple_mat(5,2,4,-1)
for n, ln in ((4096, 4000), (2048, 6000)):
    q, t = synth.fixed_batch(9, n, ln, ln, sub=0.05, ind=0.05)
    for dual, flag in ((False, 0), (False, 1)):
        p = lib.make_batch(q, t, mat, 4, 2, 24, 1, w=-1, zdrop=-1, flag=flag).plan(dual)
        p.run(); p.timing(); ms=[]
        for _ in range(3):
            p.run(); ms.append(p.timing()[1])
        print("mp probe n=%d len=%d flag=%d packed=%d %.2f ms %.1f GCUPS" % (n, ln, flag, p.packed_pairs(), np.mean(ms), p.cells()/np.mean(ms)/1e6))
        p.close()
