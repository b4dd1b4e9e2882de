"""CPU, world_size 2, gloo: the multi-GPU front-end (ksw2_amd/parallel.py) scatters a batch from rank 0, each rank
aligns its shard (simulator build of the library stands in for the per-rank GPU), rank 0 gathers; results must equal
the oracle in the original order.  On the GPU box the same code runs over RCCL with the HIP library."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import ksw2_amd as ka
from ksw2_amd import synth, parallel
from oracle import pyoracle as po
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
lib = ka.Library(os.path.join(%(root)r, "tests", "sim", "libksw2_amd_sim.so"))
rank = dist.get_rank()
mat, q, e, q2, e2 = synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1
ok = True
for dual in (False, True):
    qs = ts = w = zd = fl = None
    if rank == 0:
        rng = np.random.Generator(np.random.PCG64(17 + dual))
        pairs = synth.ragged_pairs(rng, 37, 1, 400, sub=0.05, ind=0.12)
        qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
        w = rng.choice([-1, 5, 64, 100], size=37); zd = rng.choice([-1, 100], size=37)
        fl = rng.choice([0, po.SCORE_ONLY, po.RIGHT, po.EXTZ_ONLY], size=37)
    res = parallel.sharded_align(lib, dual, qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=5, flag=fl)
    if rank == 0:
        for i in range(37):
            exp = po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, q, e, q2, e2, w=int(w[i]), zdrop=int(zd[i]), end_bonus=5, flag=int(fl[i]))
            for k in ("score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar", "cigar"):
                ok &= exp[k] == res[i][k]
    else:
        assert res is None
# splice-aware and gap-linear X-drop batches shard the same way; fixed-length batches as 2-D arrays
qs = ts = js = zd = fl = None
if rank == 0:
    from oracle.gen_golden_exts import spliced_pair
    rng = np.random.Generator(np.random.PCG64(5))
    cases = [spliced_pair(rng, 200) for _ in range(9)]
    qs, ts = [c[0] for c in cases], [c[1] for c in cases]
    js = [None if i %% 3 else rng.integers(0, 4, len(ts[i])).astype(np.uint8) for i in range(9)]
    zd = rng.choice([-1, 60], size=9); fl = rng.choice([ka.KSW_EZ_SPLICE_FOR, ka.KSW_EZ_SPLICE_REV, ka.KSW_EZ_SPLICE_FOR | po.SCORE_ONLY], size=9)
smat = synth.simple_mat(5, 1, 2, 0)
res = parallel.sharded(lib, "exts", qs, ts, dict(mat=smat, q=2, e=1, q2=32, noncan=4, junc_bonus=3), zdrop=zd, flag=fl, juncs=js)
if rank == 0:
    for i in range(9):
        exp = po.exts2("oracle", qs[i], ts[i], smat, 2, 1, 32, 4, zdrop=int(zd[i]), junc_bonus=3, flag=int(fl[i]), junc=js[i])
        ok &= all(exp[k] == res[i][k] for k in ka.FIELDS + ["cigar"])
q2d = t2d = None
if rank == 0:
    q2d, t2d = synth.fixed_batch(4, 11, 150, 160, sub=0.05, ind=0.03)
res = parallel.sharded(lib, "extf", q2d, t2d, dict(mch=2, mis=-4, e=2), w=25, zdrop=30)
if rank == 0:
    for i in range(11):
        exp = po.extf2("oracle", q2d[i], t2d[i], 2, -4, 2, 25, 30)
        ok &= all(exp[k] == res[i][k] for k in ka.FIELDS)
# a shard that arrived in device memory (what RCCL delivers on a receiving rank) is aligned where it lies: the simulator's
# "device" memory stands in, through the same ksw2amd_ext?_batch_flat(on_device = 1) call
rng = np.random.Generator(np.random.PCG64(23 + rank))
pairs = synth.ragged_pairs(rng, 9, 20, 300, sub=0.05, ind=0.1)
pq, pt = [p[0] for p in pairs], [p[1] for p in pairs]
pq[2] = pq[2].copy(); pq[2][3] = 4                      # a wildcard: reported by the packed kernel, re-run from the device arena
meta = np.zeros((9, parallel.META), dtype=np.int32)
meta[:, 0], meta[:, 1], meta[:, 2], meta[:, 3], meta[:, 5], meta[:, 6] = [len(x) for x in pq], [len(x) for x in pt], 64, 100, 0, np.arange(9)
seq = np.concatenate(pq + pt)
dptr = lib.device_copy(seq)
rec, cig = parallel.align_flat(lib, "extd", None, meta, dict(mat=mat, q=q, e=e, q2=q2, e2=e2), device_base=dptr)
rec2, cig2 = parallel.align_flat(lib, "extd", seq, meta, dict(mat=mat, q=q, e=e, q2=q2, e2=e2))
lib.device_free(dptr)
assert np.array_equal(rec, rec2) and np.array_equal(cig, cig2), rank
for i in range(9):
    exp = po.align("oracle", "extd2", pq[i], pt[i], mat, q, e, q2, e2, w=64, zdrop=100)
    assert exp["score"] == rec[i, 0] and exp["n_cigar"] == rec[i, 10], (rank, i)
# ... and so are X-drop and splice-aware shards (ksw2amd_ext?_batch_device: device pointers per pair, gathered into the plan's arena by a kernel)
from oracle.gen_golden_extf import noisy_pair
from oracle.gen_golden_exts import spliced_pair
fp = [noisy_pair(rng, int(rng.integers(40, 400)), k %% 3) for k in range(7)]
fmeta = np.zeros((7, parallel.META), dtype=np.int32)
fmeta[:, 0], fmeta[:, 1], fmeta[:, 2], fmeta[:, 3], fmeta[:, 6] = [len(a) for a, _ in fp], [len(b) for _, b in fp], 40, 60, np.arange(7)
fseq = np.concatenate([a for a, _ in fp] + [b for _, b in fp])
dptr = lib.device_copy(fseq)
frec, _ = parallel.align_flat(lib, "extf", None, fmeta, dict(mch=2, mis=-4, e=2), device_base=dptr)
frec2, _ = parallel.align_flat(lib, "extf", fseq, fmeta, dict(mch=2, mis=-4, e=2))
lib.device_free(dptr)
assert np.array_equal(frec, frec2), rank
for i in range(7):
    exp = po.extf2("oracle", fp[i][0], fp[i][1], 2, -4, 2, 40, 60)
    assert exp["score"] == frec[i, 0] and exp["max"] == frec[i, 1] and exp["max_t"] == frec[i, 2], (rank, i)
sp = [spliced_pair(rng, int(rng.integers(60, 200)), k) for k in range(5)]
smeta = np.zeros((5, parallel.META), dtype=np.int32)
smeta[:, 0], smeta[:, 1], smeta[:, 3], smeta[:, 5], smeta[:, 6] = [len(x[0]) for x in sp], [len(x[1]) for x in sp], 200, 0, np.arange(5)
sseq = np.concatenate([x[0] for x in sp] + [x[1] for x in sp])
dptr = lib.device_copy(sseq)
ssc = dict(mat=synth.simple_mat(5, 1, 2, -1), q=2, e=1, q2=32, noncan=4)
srec, scig = parallel.align_flat(lib, "exts", None, smeta, ssc, device_base=dptr)
srec2, scig2 = parallel.align_flat(lib, "exts", sseq, smeta, ssc)
lib.device_free(dptr)
assert np.array_equal(srec, srec2) and np.array_equal(scig, scig2) and srec[:, 10].min() > 0, rank
# an empty shard (more ranks than pairs) must work too
res = parallel.sharded_align(lib, False, [np.array([1, 2], np.uint8)] if rank == 0 else None, [np.array([1, 2], np.uint8)] if rank == 0 else None, mat, q, e)
if rank == 0:
    ok &= res[0]["score"] == 4
    print("SHARD_OK" if ok else "SHARD_BAD")
dist.destroy_process_group()
'''


def test_two_rank_scatter_gather(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "sim")], check=True, capture_output=True)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SHARD_OK" in outs[0][0], outs


def test_lpt_partition_balances():
    from ksw2_amd.parallel import lpt_partition
    rng = np.random.Generator(np.random.PCG64(1))
    costs = rng.integers(1, 1000, size=500)
    shards = lpt_partition(costs, 8)
    assert sorted(np.concatenate(shards).tolist()) == list(range(500))
    loads = np.array([costs[s].sum() for s in shards])
    assert loads.max() - loads.min() <= costs.max()
