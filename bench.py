#!/usr/bin/env python3
"""bench.py -- GCUPS of the banded extension hot path on N MI355X (one process per GPU).

Headline workload (north_star): 10 000 x 10 000 extensions, band 500, Z-drop 400, ksw_extz2_sse semantics, score only.

A "step" = ONE call of the drop-in batch entry point (ksw2amd_extz_batch / ksw2amd_extd_batch) on one batch of synthetic
pairs that sit in ordinary host memory: pack into pinned staging, H2D, fill (+ traceback) kernels, D2H, ksw_extz_t assembly.
`value` is therefore transfer-inclusive ("wall seconds incl. H2D/D2H"); the library overlaps the phases
of successive chunks of the batch on its worker threads / streams.  The HBM-resident kernel rate of the same batch
(ksw2amd_plan_run only, HIP events on the launch stream) is reported next to it as `value_hbm_resident` and prices the
roofline.  The batch is sized so that the K = 20 steps of the driver's run take a few seconds.

The other configurations of BASELINE.json (config 2, 3, 4 at 4 096 replicas, 5 with channel-derived target lengths) and the
10 k case with CIGAR / at the n = 1 024 of SURVEY 8d run in the same process and are reported in the `also` array.

Launch:  python bench.py [--gpus 1 --steps K --warmup W]
         python bench.py --gpus N            (starts its own N ranks through torch.distributed.run)
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import ksw2_amd                      # noqa: E402
from ksw2_amd import synth           # noqa: E402

# algorithmic integer ops per cell and VALU peak (256 CU x 4 SIMD x 32 lanes x 2.4 GHz x 2 for
# packed int16; gfx950 has no packed int8 add/max).  The int32 lane-op peak is half of that.
OPS_PER_CELL = {("extz", True): 15, ("extz", False): 22, ("extd", True): 28, ("extd", False): 42,
                ("exts", True): 21, ("exts", False): 31,     # extz + the long-gap state: 3 more values per cell, 3 more decisions
                ("extf", True): 7}                            # score compare/select, add, two maxima, two subtractions
VALU_PEAK_PK16 = 157.3e12
HBM_PEAK = 8.0e12
SO = ksw2_amd.KSW_EZ_SCORE_ONLY

WORKLOADS = {
    # north_star's 10k x 10k banded extension.  49 152 pairs per step (0.96 GB of sequence, 4.8e11 cells): ~0.16 s per step.
    "10k": dict(idx=6, n=49152, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO, sub=0.05, ind=0.06),
    # the same at the n = 1 024 SURVEY 8d names: 512 packed wavefronts cannot fill 1 024 SIMDs, the host takes the int32 kernels
    "10k-n1024": dict(idx=6, n=1024, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO, sub=0.05, ind=0.06),
    "10k-cigar": dict(idx=6, n=4096, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=0, sub=0.05, ind=0.06),
    "cfg2": dict(idx=2, n=65536, qlen=512, tlen=512, w=64, zdrop=-1, dual=False, flag=SO, sub=0.05, ind=0.06),
    "cfg3": dict(idx=3, n=16384, qlen=2048, tlen=2048, w=256, zdrop=400, dual=True, flag=0, sub=0.05, ind=0.10, tail_frac=0.25, tail_pairs=0.10),
    # config 4: MT-human x MT-orang (tests/golden/data), full global extz2 with CIGAR, 4 096 replicas: 590 GB of direction
    # bits, so the batch entry point runs it as several plans (no single resident plan: transfer-inclusive figure only)
    "cfg4": dict(idx=4, n=4096, qlen=16499, tlen=16569, w=-1, zdrop=-1, dual=False, flag=0, mt=True, resident_n=1024),
    # the same score-only (diagnostics: what the packed generation-serial kernel does without the traceback stream)
    "cfg4-so": dict(idx=4, n=4096, qlen=16499, tlen=16569, w=-1, zdrop=-1, dual=False, flag=SO, mt=True, resident_n=1024),
    # config 5: ONT-like mix, query length uniform in [300, 20000], 3 % substitutions + 15 % indels, target length from the
    # channel (|tlen - qlen| <= 450), band 500, extd2 with Z-drop 400 and CIGAR; 16 384 pairs = one eighth of the per-GPU share
    # of the 1 M pair config (which shards over 8 GPUs)
    "cfg5": dict(idx=5, n=16384, qlen=0, tlen=0, w=500, zdrop=400, dual=True, flag=0, sub=0.03, ind=0.15, ragged=True),
    # config 5 at the per-GPU share BASELINE.json states: 1 M pairs over 8 GPUs = 125 000 pairs per GPU (1.26e12 cells, 2.5 GB of
    # sequence, ~1.3 TB of direction bytes: the batch entry point runs it as several device-filling plans); resident figure on a slice
    "cfg5-share": dict(idx=5, n=125000, qlen=0, tlen=0, w=500, zdrop=400, dual=True, flag=0, sub=0.03, ind=0.15, ragged=True, resident_n=16384),
    # splice-aware extension and gap-linear X-drop extension (N3): see DESIGN.md sections 3.6 / 3.7
    "exts": dict(idx=7, n=16384, qlen=400, tlen=1500, w=-1, zdrop=-1, dual=False, flag=0, sub=0.03, ind=0.0, splice=True),
    "extf": dict(idx=8, n=16384, qlen=1000, tlen=1000, w=100, zdrop=-1, dual=False, flag=SO, sub=0.05, ind=0.01, linear=True),
    # ... at bands past the four-per-wavefront form: two / one extension per wavefront (32 / 64 lanes each)
    "extf-w300": dict(idx=8, n=8192, qlen=2000, tlen=2000, w=300, zdrop=-1, dual=False, flag=SO, sub=0.05, ind=0.01, linear=True),
    "extf-w900": dict(idx=8, n=4096, qlen=4000, tlen=4000, w=900, zdrop=-1, dual=False, flag=SO, sub=0.05, ind=0.01, linear=True),
    # the headline's shape where extensions DO drop: a fifth of the pairs get the last quarter of the query replaced by random bases, so the
    # Z-drop fires there -- what the deferred arg-max costs when it has to hand alignments back is in this line
    "10k-zdrop": dict(idx=6, n=49152, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO, sub=0.05, ind=0.06, tail_frac=0.25, tail_pairs=0.20),
    # ... and where 1 % of the pairs hold a wildcard base: those leave the packed kernels (they score match / mismatch only)
    "10k-N": dict(idx=6, n=49152, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO, sub=0.05, ind=0.06, wild_pairs=0.01),
    # ... and where 1 % of the TARGETS carry a run of 1-50 wildcard bases (reference genomes hold N runs; ksw2_extz2_sse.c:125-140 scores them at the
    # full SIMD rate): what the packed kernels' target-wildcard rule costs (rules 3 and 9)
    "10k-tN": dict(idx=6, n=49152, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO, sub=0.05, ind=0.06, twild_pairs=0.01),
    # ... and under a scoring matrix without match / mismatch structure (KSW_EZ_GENERIC_SC; transitions -2, transversions -4): the reference
    # takes any matrix at its full rate (ksw2_extz2_sse.c:142-143), and since round 5 so do the packed kernels (column profiles)
    "10k-generic": dict(idx=6, n=49152, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO | ksw2_amd.KSW_EZ_GENERIC_SC, sub=0.05, ind=0.06, tstv=True),
    # ... and in the SSE-compatible mode (the reference's SSE kernels' own results): every pair through k2a_ssec_kernel
    "10k-ssec": dict(idx=6, n=1024, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO, sub=0.05, ind=0.06, sse=True),
    # ... at four wavefronts per SIMD instead of one (the kernel holds one alignment per wavefront: 1 024 pairs are one wavefront per SIMD,
    # which issues at half the rate a SIMD reaches with four) and with KSW_EZ_APPROX_MAX (one followed cell instead of H: the reference's fast mode)
    "10k-ssec-n4096": dict(idx=6, n=4096, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO, sub=0.05, ind=0.06, sse=True),
    "10k-ssec-approx": dict(idx=6, n=4096, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO | 0x08, sub=0.05, ind=0.06, sse=True),
    # KSW_EZ_APPROX_MAX alone -- the reference's fastest mode (README.md:104-105; score and corner CIGAR only) -- under the default contract: the
    # packed kernels without maximum tracking (NOMAX).  (With KSW_EZ_APPROX_DROP the heuristic is defined by the SSE data flow:
    # `10k-ssec-approx` above.)
    "10k-approx": dict(idx=6, n=49152, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=SO | 0x08, sub=0.05, ind=0.06),
    # ... and with the CIGAR (the direction bytes of the reference's SSE kernel, its own walk)
    "10k-ssec-cigar": dict(idx=6, n=1024, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=0, sub=0.05, ind=0.06, sse=True),
}
ALSO_DEFAULT = ["10k-n1024", "10k-cigar", "cfg2", "cfg3", "cfg5", "cfg4", "cfg5-share", "10k-zdrop", "10k-N", "10k-tN", "10k-generic", "exts", "extf", "extf-w300", "extf-w900", "10k-approx", "10k-ssec", "10k-ssec-n4096", "10k-ssec-approx", "10k-ssec-cigar"]
# pairs of each workload's last timed batch that are compared with the oracle outside the clock (the MT pair costs ~1 s per pair on the host)
PARITY_PAIRS = {"cfg4": 16, "cfg4-so": 4, "cfg5": 16, "cfg5-share": 16}
# N > 1: the configurations BASELINE.json quotes for several GPUs at their per-GPU share (config 4: 4 096 replicas / 8)
ALSO_MULTI = {"cfg5-share": None, "cfg4": 512}
SCORING = dict(a=2, b=4, sc_n=-1, q=4, e=2, q2=24, e2=1)
LINEAR_SCORING = dict(mch=2, mis=-4, e=2)
SPLICE_SCORING = dict(a=1, b=2, sc_n=0, q=2, e=1, q2=32, noncan=4)


def make_spliced(wl, rank, n):
    """Fixed-shape spliced pairs: target = exon | GT intron AG | exon with flanks, query = the two exons with substitutions."""
    rng = synth.rng_for(wl["idx"], rank)
    tl, ql = wl["tlen"], wl["qlen"]
    t = rng.integers(0, 4, size=(n, tl), dtype=np.uint8)
    ex1 = ql // 2
    a, b = 50, tl - 50 - (ql - ex1)                    # exon 1 = [a, a+ex1), exon 2 = [b, b+ql-ex1)
    t[:, a + ex1], t[:, a + ex1 + 1], t[:, b - 2], t[:, b - 1] = 2, 3, 0, 2
    q = np.concatenate([t[:, a:a + ex1], t[:, b:b + ql - ex1]], axis=1).copy()
    mism = rng.random(q.shape) < wl["sub"]
    q[mism] = (q[mism] + rng.integers(1, 4, size=int(mism.sum()), dtype=np.uint8)) & 3
    return q, t


def make_batch(wl, rank, n):
    """Synthetic pairs of the workload (tools/synth: one xorshift64* stream per pair, seed 20260001 + config index); rank r
    takes pairs [r * n, (r + 1) * n) of the stream, so every rank aligns different data."""
    if wl.get("splice"):
        return make_spliced(wl, rank, n)
    if wl.get("ragged"):
        return synth.fast_ragged(wl["idx"], n, 300, 20000, sub=wl["sub"], ind=wl["ind"], maxdiff=450, first=rank * n)
    if wl.get("mt"):
        from tests import golden_util as gu
        _, ts = gu.read_fasta("MT-human.fa")
        _, qs = gu.read_fasta("MT-orang.fa")
        q1, t1 = np.ascontiguousarray(qs[0]), np.ascontiguousarray(ts[0])
        return [q1] * n, [t1] * n                        # same bytes; every replica is still packed, uploaded and computed
    q, t = synth.fast_fixed(wl["idx"], n, wl["qlen"], wl["tlen"], sub=wl["sub"], ind=wl["ind"],
                            tail_random_frac=wl.get("tail_frac", 0.0), tail_pairs=wl.get("tail_pairs", 0.0), first=rank * n)
    if wl.get("wild_pairs"):                             # one wildcard base (code 4) somewhere in the query of every 1 / wild_pairs-th pair
        rng = np.random.Generator(np.random.PCG64(wl["idx"] * 1000003 + rank))
        step = max(1, int(round(1.0 / wl["wild_pairs"])))
        for i in range(step // 2, n, step):
            q[i, int(rng.integers(wl["qlen"]))] = 4
    if wl.get("twild_pairs"):                            # a run of 1-50 wildcard bases somewhere in the target of every 1 / twild_pairs-th pair
        rng = np.random.Generator(np.random.PCG64(wl["idx"] * 1000033 + rank))
        step = max(1, int(round(1.0 / wl["twild_pairs"])))
        for i in range(step // 2, n, step):
            ln = int(rng.integers(1, 51))
            at = int(rng.integers(wl["tlen"] - ln))
            t[i, at:at + ln] = 4
    return q, t


def cells_of_rows(qlen, rows, w):
    """In-band cells of target rows [0, rows) for arrays of (qlen, rows, w): the closed form of ksw2_host_plan.c::band_cells."""
    qlen, rows, w = (np.asarray(x, dtype=np.int64) for x in (qlen, rows, w))
    T = np.minimum(qlen + w, rows)
    a = qlen - 1 - w
    na = np.where(a < 0, 0, np.minimum(a + 1, T))
    nb = np.minimum(w + 1, T)
    sum_en = na * (na - 1) // 2 + na * w + (T - na) * (qlen - 1)
    sum_st = (T - nb) * (T - 1 + nb) // 2 - (T - nb) * w
    return np.where(T <= 0, 0, sum_en - sum_st + T)


_CPU_SAMPLES = {}


def cpu_call_key(wl, mat):
    """What distinguishes one CPU sample from another: the reference function, the shape, the parameters, the reference's own flag bits, the
    matrix and what shapes the DATA's run time (random tails = early Z-drops, the ragged channel).  `10k`, `10k-N`, `10k-n1024`, `10k-ssec*`
    time the same reference call on the same stream of pairs: one sample serves them all (the driver's run was mostly these samples)."""
    return (bool(wl["dual"]), bool(wl.get("splice")), bool(wl.get("linear")), wl["idx"], wl["qlen"], wl["tlen"], wl["w"], wl["zdrop"], wl["flag"] & 0xffff,
            bytes(np.asarray(mat, dtype=np.int8)), wl.get("tail_frac", 0.0), wl.get("tail_pairs", 0.0), bool(wl.get("ragged")), bool(wl.get("mt")),
            wl.get("sub"), wl.get("ind"))


def cpu_baseline(wl, q, t, mat, seconds=10.0, share=True):
    """cpu_baseline_run, once per distinct reference call (cpu_call_key); a shared sample says which workload took it (`shared_from`)."""
    key = cpu_call_key(wl, mat)
    if share and key in _CPU_SAMPLES:
        return dict(_CPU_SAMPLES[key][1], shared_from=_CPU_SAMPLES[key][0])
    res = cpu_baseline_run(wl, q, t, mat, seconds)
    _CPU_SAMPLES.setdefault(key, (wl.get("name", "?"), res))
    return res


def cpu_baseline_run(wl, q, t, mat, seconds=10.0):
    """Reference ksw_extz2_sse / ksw_extd2_sse (oracle/_ref, gcc -O2 -msse4.1) on this box's host cores: a bounded sample of
    the same batch, 1 thread and all cores, pthread loop in oracle/cpu_bench.c."""
    from oracle import pyoracle as po
    olib = po.oracle_lib()
    ref = po.ref_lib()
    kind = "reference" if ref is not None else "port"
    name = ("ksw_extd2_sse" if wl["dual"] else "ksw_extz2_sse") if ref is not None else ("kso_extd2_km" if wl["dual"] else "kso_extz2_km")
    if wl.get("splice"):
        name = "ksw_exts2_sse" if ref is not None and hasattr(ref, "ksw_exts2_sse") else "kso_exts2_km"
        kind = "reference" if name.startswith("ksw_") else "port"
    if wl.get("linear"):
        name = "ksw_extf2_sse" if ref is not None and hasattr(ref, "ksw_extf2_sse") else "kso_extf2_km"
        kind = "reference" if name.startswith("ksw_") else "port"
    fn = ctypes.cast(getattr(ref if name.startswith("ksw_") else olib, name), ctypes.c_void_p)
    olib.kso_cpu_bench.restype = ctypes.c_long
    olib.kso_cpu_bench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int8, ctypes.c_void_p, ctypes.c_int8, ctypes.c_int8,
                                   ctypes.c_int8, ctypes.c_int8, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    S = SCORING
    mode = 3 if wl.get("linear") else 2 if wl.get("splice") else int(wl["dual"])
    if wl.get("linear"):
        S = dict(q=LINEAR_SCORING["mch"], e=LINEAR_SCORING["mis"], q2=LINEAR_SCORING["e"], e2=0)
    if wl.get("splice"):
        S = dict(q=SPLICE_SCORING["q"], e=SPLICE_SCORING["e"], q2=SPLICE_SCORING["q2"], e2=SPLICE_SCORING["noncan"])
    # a bounded sample: the first pairs of the batch (the loop wraps around if the CPU gets through them)
    ns = min(len(q), 4096)
    qa = [np.ascontiguousarray(q[i]) for i in range(ns)]
    ta = [np.ascontiguousarray(t[i]) for i in range(ns)]
    qp = np.array([x.ctypes.data for x in qa], dtype=np.int64)
    tp = np.array([x.ctypes.data for x in ta], dtype=np.int64)
    ql = np.array([len(x) for x in qa], dtype=np.int32)
    tl = np.array([len(x) for x in ta], dtype=np.int32)
    wv = np.where(wl["w"] < 0, np.maximum(ql, tl), wl["w"]).astype(np.int64)
    cells = cells_of_rows(ql, tl, np.minimum(wv, np.maximum(ql, tl))).astype(np.float64)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {}
    for threads in sorted({1, ncores}):
        el = ctypes.c_double(0)
        done = olib.kso_cpu_bench(fn, mode, threads, seconds, ns, qp.ctypes.data, tp.ctypes.data, ql.ctypes.data, tl.ctypes.data, None,
                                  5, mat.ctypes.data, S["q"], S["e"], S["q2"], S["e2"], wl["w"], wl["zdrop"], wl["flag"] & 0xffff, ctypes.byref(el))      # (the reference's own flag bits only)
        # pairs are taken in order (index mod ns): cells of the pairs actually aligned
        full, rem = divmod(int(done), ns)
        c = full * cells.sum() + cells[:rem].sum()
        out[threads] = (done, el.value, c / el.value / 1e9)
    n1, dt1, g1 = out[1]
    what = "reference %s gcc -O2 -msse4.1" % name if kind == "reference" else "oracle scalar port (no reference build here)"
    res = {"value": round(g1, 4), "unit": "GCUPS", "cores": 1, "kind": kind,
           "sample": "%d pairs of the batch, %.1f s, %s" % (n1, dt1, what), "pairs_per_s": round(n1 / dt1, 1)}
    if ncores > 1:
        nn, dtn, gn = out[ncores]
        res["all_cores"] = {"value": round(gn, 4), "cores": ncores, "pairs_per_s": round(nn / dtn, 1)}
    return res


def recorded_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE
    need their own rocprofv3 --pmc runs, so they cannot be sampled inside this process); None if never recorded."""
    import glob
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.json" % workload)))
    if not cand:
        return None, None
    d = json.load(open(cand[-1]))
    return d["derived"].get("hbm_bytes_gfx950_corrected"), os.path.relpath(cand[-1], ROOT)


class Job:
    """One workload on this rank: host batch, the library's batch object, and the two measurements."""

    def __init__(self, lib, name, wl, rank, n_override=None, approx=False, sse=False):
        self.lib, self.name, self.rank, self.sse = lib, name, rank, sse
        if approx:
            wl = dict(wl, flag=wl["flag"] | 0x08)
        sse = sse or bool(wl.get("sse"))
        self.sse = sse
        if sse:                                  # the SSE kernels' own results: every pair through k2a_ssec_kernel
            wl = dict(wl, flag=wl["flag"] | ksw2_amd.KSW2AMD_EZ_SSE_COMPAT)
        self.wl = wl = dict(wl, name=name)
        self.n = n_override or wl["n"]
        t0 = time.perf_counter()
        self.q, self.t = make_batch(wl, rank, self.n)
        self.gen_s = time.perf_counter() - t0
        S = SCORING
        if wl.get("splice"):
            P = SPLICE_SCORING
            self.mat = synth.simple_mat(5, P["a"], P["b"], P["sc_n"])
            self.wl = wl = dict(wl, flag=wl["flag"] | ksw2_amd.KSW_EZ_SPLICE_FOR)
            self.batch = lib.make_splice_batch(list(self.q), list(self.t), self.mat, P["q"], P["e"], P["q2"], P["noncan"], zdrop=wl["zdrop"], flag=wl["flag"])
            self.kind = "exts"
        elif wl.get("linear"):
            P = LINEAR_SCORING
            self.mat = synth.simple_mat(5, P["mch"], -P["mis"], 0)          # for the CPU leg's signature only
            self.batch = lib.make_linear_batch(list(self.q), list(self.t), P["mch"], P["mis"], P["e"], w=wl["w"], xdrop=wl["zdrop"])
            self.kind = "extf"
        else:
            self.mat = synth.simple_mat(5, S["a"], S["b"], 0 if wl.get("mt") else S["sc_n"])
            if wl.get("tstv"):                   # A<->G and C<->T (codes 0<->2, 1<->3) cost 2, the other substitutions 4
                self.mat = np.array([[2, -4, -2, -4, -1], [-4, 2, -4, -2, -1], [-2, -4, 2, -4, -1], [-4, -2, -4, 2, -1], [-1, -1, -1, -1, -1]], dtype=np.int8).reshape(-1)
            self.batch = lib.make_batch(self.q, self.t, self.mat, S["q"], S["e"], S["q2"], S["e2"], w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"])
            self.kind = "extd" if wl["dual"] else "extz"
        self.score_only = bool(wl["flag"] & SO)
        ql, tl = np.asarray(self.batch.qlen if hasattr(self.batch, "qlen") else [len(x) for x in self.q]), \
            np.asarray(self.batch.tlen if hasattr(self.batch, "tlen") else [len(x) for x in self.t])
        self.qlen, self.tlen = ql.astype(np.int64), tl.astype(np.int64)
        mx = np.maximum(self.qlen, self.tlen)
        self.weff = np.where((wl["w"] < 0) | (wl["w"] > mx), mx, wl["w"]) if self.kind in ("extz", "extd", "extf") else mx
        self.cells = int(cells_of_rows(self.qlen, self.tlen, self.weff).sum()) if self.kind != "exts" else int((self.qlen * self.tlen).sum())
        self.ez = None
        self.fbatch = None

    def flat_ready(self, register=True):
        """The same pairs as ONE arena + offsets (ksw2amd_flat_t), page-locked once like a caller's reusable read buffer."""
        if self.kind not in ("extz", "extd") or self.sse:
            return False
        if self.fbatch is None:
            wl, S = self.wl, SCORING
            self.fbatch = self.lib.make_flat_batch(self.q, self.t, self.mat, S["q"], S["e"], S["q2"], S["e2"], w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"])
            if register and not os.environ.get("KSW2_BENCH_NO_REGISTER"):
                self.fbatch.register()
        return True

    def flat_done(self):
        if self.fbatch is not None:
            self.fbatch.unregister()
            self.fbatch = None

    # ---- transfer-inclusive: one call of the batch entry point per step
    def e2e_step(self, flat=False):
        b, L = self.batch, self.lib
        if self.ez is None:
            self.ez = (ksw2_amd.KswExtz * max(self.n, 1))()     # reused across steps like a caller would: CIGAR buffers are recycled
        if flat:
            fb = self.fbatch
            f = L.lib.ksw2amd_extd_batch_flat if self.kind == "extd" else L.lib.ksw2amd_extz_batch_flat
            L._check(f(None, ctypes.byref(fb.sc), fb.n, ctypes.byref(fb.flat), self.ez))
        elif self.kind == "exts":
            L._check(L.lib.ksw2amd_exts_batch(None, ctypes.byref(b.sc), b.n, b.pairs, self.ez))
        elif self.kind == "extf":
            L._check(L.lib.ksw2amd_extf_batch(None, *b.par, b.n, b.pairs, self.ez))
        else:
            f = L.lib.ksw2amd_extd_batch if self.kind == "extd" else L.lib.ksw2amd_extz_batch
            L._check(f(None, ctypes.byref(b.sc), b.n, b.pairs, self.ez))

    # ---- correctness evidence attached to the number: outside the clock, a few pairs of the batch the timed loop just aligned
    def decoy_then_step(self, flat):
        """Score-only one-shape batches (what the batch entry points stream): one step on a DECOY batch of the same shapes
        but other bases, then one more step on the real batch -- outside the clock.  The timed loop aligns the same batch K times, so the
        device arena, the staging buffers and the caches hold its bytes from the step before: a kernel that read a sequence before its
        upload had landed, or out of a stale cache line, would still return the right answer there.  After a decoy it would not."""
        if self.kind not in ("extz", "extd") or not self.score_only or not isinstance(self.q, np.ndarray) or self.sse:
            return False
        rng = np.random.Generator(np.random.PCG64(12345))
        dq = rng.integers(0, 4, size=self.q.shape, dtype=np.uint8)
        dt = rng.integers(0, 4, size=self.t.shape, dtype=np.uint8)
        wl, S = self.wl, SCORING
        keep = (self.batch, self.fbatch)
        if flat:
            self.fbatch = self.lib.make_flat_batch(dq, dt, self.mat, S["q"], S["e"], S["q2"], S["e2"], w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"])
            self.fbatch.register()
        else:
            self.batch = self.lib.make_batch(dq, dt, self.mat, S["q"], S["e"], S["q2"], S["e2"], w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"])
        try:
            self.e2e_step(flat=flat)
        finally:
            if flat:
                self.fbatch.unregister()
            self.batch, self.fbatch = keep
        self.e2e_step(flat=flat)
        return True

    def parity_sample(self, k=16, flat=False):
        """Compare k evenly spaced results of the last e2e_step (self.ez, i.e. what the timed calls returned) with the oracle:
        every ksw_extz_t field and the CIGAR.  Returns a JSON-able verdict."""
        from oracle import pyoracle as po
        if self.ez is None or self.n == 0:
            return {"pairs": 0, "result": "not run"}
        decoyed = self.decoy_then_step(flat)
        wl, S = self.wl, SCORING
        idx = sorted({int(x) for x in np.linspace(0, self.n - 1, min(k, self.n))})
        fields = ksw2_amd.FIELDS + ([] if self.kind == "extf" else ["cigar"])
        bad = []
        for i in idx:
            got = ksw2_amd.ez_to_dict(self.ez[i])
            if self.kind == "exts":
                P = SPLICE_SCORING
                exp = po.exts2("oracle", self.q[i], self.t[i], self.mat, P["q"], P["e"], P["q2"], P["noncan"], zdrop=wl["zdrop"], flag=wl["flag"])
            elif self.kind == "extf":
                P = LINEAR_SCORING
                exp = po.extf2("oracle", self.q[i], self.t[i], P["mch"], P["mis"], P["e"], wl["w"], wl["zdrop"])
            elif self.sse:                     # the oracle's restatement of the SSE kernels' memory image (oracle/ksw2_oracle_sse.c, pinned by 1 500 reference vectors)
                exp = po.align("oracle", "extd2_sse" if self.kind == "extd" else "extz2_sse", self.q[i], self.t[i], self.mat, S["q"], S["e"], S["q2"], S["e2"],
                               w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"] & ~ksw2_amd.KSW2AMD_EZ_SSE_COMPAT)
            else:
                exp = po.align("oracle", "extd2" if self.kind == "extd" else "extz2", self.q[i], self.t[i], self.mat, S["q"], S["e"], S["q2"], S["e2"],
                               w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"])
            d = [f for f in fields if exp[f] != got[f]]
            if d:
                bad.append({"pair": i, "fields": d})
        return {"pairs": len(idx), "result": "ok" if not bad else "MISMATCH", "checked": "all ksw_extz_t fields" + ("" if self.kind == "extf" else " + CIGAR") +
                (" of one more step of the timed batch, run behind a decoy batch of the same shapes (other bases), vs oracle/" if decoyed else
                 " of the timed batch's last step vs oracle/") + " (CPU restatement pinned to the compiled reference)", **({"mismatches": bad[:4]} if bad else {})}

    def free_ez(self):
        if self.ez is not None:
            for i in range(self.n):
                if self.ez[i].cigar:
                    ksw2_amd._libc.free(ctypes.cast(self.ez[i].cigar, ctypes.c_void_p))
            self.ez = None

    # ---- HBM-resident: plan_run only, HIP events on the launch stream
    def resident(self, steps, warmup, stream, min_seconds=0.0):
        wl = self.wl
        nres = wl.get("resident_n")
        if os.environ.get("KSW2_BENCH_ONE_DEVICE") and int(os.environ.get("WORLD_SIZE", "1")) > 1:
            # the host-contention rehearsal (N ranks on ONE GPU): the ranks' resident plans share one device's memory
            nres = max(256, (nres or self.n) // int(os.environ["WORLD_SIZE"]))
        b = self.batch
        if nres and nres < self.n:          # a plan of the whole batch does not fit one device: a slice of it
            S = SCORING
            b = self.lib.make_batch(self.q[:nres], self.t[:nres], self.mat, S["q"], S["e"], S["q2"], S["e2"], w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"])
            b.qlen, b.tlen = self.qlen[:nres], self.tlen[:nres]
        plan = b.plan() if self.kind in ("exts", "extf") else b.sse_plan(wl["dual"]) if self.sse else b.plan(wl["dual"])
        cells = plan.cells()
        for _ in range(warmup):
            plan.run(stream)
        if warmup:
            plan.timing()
        fill_ms, total_ms = [], []
        t0 = time.perf_counter()
        k = 0
        while k < steps or time.perf_counter() - t0 < min_seconds:
            plan.run(stream)
            f, tot = plan.timing()              # blocks on the step's last event
            fill_ms.append(f)
            total_ms.append(tot)
            k += 1
        wall = time.perf_counter() - t0
        raw = plan.fetch_raw()
        rows = raw[:, 11].astype(np.int64)
        nn = b.n
        if self.kind == "exts":
            done = cells
        elif self.kind == "extf":
            # rows_done = anti-diagonals completed before the X-drop (k2a_extf_finish); cells of those diagonals, per distinct shape
            done = 0
            shapes = {}
            for i in range(nn):
                shapes.setdefault((int(self.qlen[i]), int(self.tlen[i]), int(self.weff[i])), []).append(i)
            for (ql_, tl_, w_), idx in shapes.items():
                r = np.arange(ql_ + tl_ - 1, dtype=np.int64)
                lo = np.maximum(np.maximum(0, r - ql_ + 1), (r - w_ + 1) >> 1)
                hi = np.minimum(np.minimum(tl_ - 1, r), (r + w_) >> 1)
                cum = np.concatenate([[0], np.cumsum(np.maximum(hi - lo + 1, 0))])
                tot = int(cells_of_rows(ql_, tl_, w_))
                nd = np.clip(rows[idx], 0, len(r))
                done += int((cum[nd] * (tot / max(int(cum[-1]), 1))).sum())      # scaled to the row-band cell count `cells` is quoted in
        else:
            done = int(cells_of_rows(self.qlen[:nn], np.minimum(rows, self.tlen[:nn]), self.weff[:nn]).sum())
        res = dict(n=nn, cells=cells, steps=k, wall_s=wall, kernel_ms=float(np.mean(total_ms)), fill_ms=float(np.mean(fill_ms)),
                   cells_done=done, zdropped=int(raw[:, 1].sum()), packed_pairs=plan.packed_pairs(), device_bytes=plan.device_bytes(),
                   kernels=plan.describe())
        plan.close()
        return res


def describe(job, world):
    wl = job.wl
    func = {"extf": "extf2 gap-linear X-drop", "exts": "exts2 splice-aware", "extd": "extd2 dual-gap", "extz": "extz2 affine"}[job.kind]
    shape = "qlen in [300,20000] tlen from the channel" if wl.get("ragged") else "qlen=%d tlen=%d" % (wl["qlen"], wl["tlen"])
    return "%s: %d pairs/step/GPU, %s band=%d zdrop=%d %s %s" % (job.name, job.n, shape, wl["w"], wl["zdrop"], func,
                                                                 ("score-only" if job.score_only else "CIGAR") + (" APPROX_MAX" if wl["flag"] & 0x08 else "") +
                                                                 (" SSE-compatible mode" if getattr(job, "sse", False) else ""))


def roofline_of(job, res, workload_key):
    ops = OPS_PER_CELL[(job.kind, job.score_only)]
    kern_s = res["kernel_ms"] * 1e-3
    achieved = res["cells"] * ops / kern_s
    seq_bytes = int(job.qlen[:res["n"]].sum() + job.tlen[:res["n"]].sum())
    alg_bytes = seq_bytes + 56 * res["n"] + (0 if job.score_only else res["cells"] // (1 if job.kind == "extd" else 2))
    # the PMC passes under profiles/ are of the plain workloads: a different kernel runs with --approx / --sse-compat or a resized batch
    plain = not os.environ.get("KSW2AMD_NO_PKMP") and not (getattr(job, "sse", False) and not WORKLOADS[workload_key].get("sse")) and not (job.wl["flag"] & 0x08 and not WORKLOADS[workload_key]["flag"] & 0x08) and res["n"] == (WORKLOADS[workload_key].get("resident_n") or WORKLOADS[workload_key]["n"])
    traffic, src = recorded_traffic(workload_key) if plain else (None, None)
    defer = any(c.get("form") == "defer" for c in res.get("kernels", []))
    out = _roofline_dict(job, res, ops, kern_s, achieved, alg_bytes, traffic, src)
    if defer and traffic:
        # the deferred arg-max kernels write, once, 8 bytes per lane and step (512 B per wavefront-step) as checkpoints for the second
        # pass: that stream is by design, not a re-read; at ~1.4 TB/s it is far from the HBM bound
        out["traffic_note"] = "write-once checkpoint stream of the deferred arg-max (DESIGN.md 3.11): 512 B per wavefront-step; sequences + results are the algorithmic bytes"
    return out


def _kernel_line(c):
    """One line of ksw2amd_plan_describe as a short string (extz / extd classes carry a geometry, the others their own fields)."""
    if "G" in c:
        return "%s(%d,%d) gaps=%d %s%s%s %s x%d" % (c["kernel"], c["G"], c["C"], c["gaps"], c["mode"], " rebased" if c["rebased"] else "",
                                                    " nomax" if c["nomax"] else "", c["form"], c["tasks"])
    return "%s %s x%d" % (c["kernel"], " ".join("%s=%s" % (k, v) for k, v in c.items() if k not in ("kernel", "tasks")), c["tasks"])


def _roofline_dict(job, res, ops, kern_s, achieved, alg_bytes, traffic, src):
    return {"bound": "valu", "achieved": round(achieved / 1e12, 4), "peak": VALU_PEAK_PK16 / 1e12, "unit": "Tiop/s",
            "frac": round(achieved / VALU_PEAK_PK16, 5), "traffic": traffic, "traffic_source": src,
            # rocprofv3 --pmc cannot run inside this process: `traffic` is replayed from the newest committed PMC summary of this workload
            "traffic_measured_in": ("%s (round %s; separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload ... --resident-only`, "
                                    "tools/scripts/profile_round.sh)" % (src, os.path.basename(src).split("_")[0].lstrip("r")[:1])) if src else None,
            "ops_per_cell": ops, "kernel_ms": round(res["kernel_ms"], 4), "fill_kernel_ms": round(res["fill_ms"], 4),
            "kernel_gcups": round(res["cells"] / kern_s / 1e9, 2), "pairs_per_launch": res["n"], "cells_per_launch": res["cells"],
            "kernel_gcups_cells_filled": round(res["cells_done"] / kern_s / 1e9, 2),
            "early_stop_fraction": round(1.0 - res["cells_done"] / max(res["cells"], 1), 5), "zdropped_pairs": res["zdropped"],
            "kernels": [_kernel_line(c) for c in res.get("kernels", [])],
            "algorithmic_bytes": alg_bytes, "hbm_algorithmic_GBps": round(alg_bytes / kern_s / 1e9, 2), "hbm_peak_GBps": HBM_PEAK / 1e9}


def dtype_of(job, res):
    if job.kind == "extf":
        return "u8 (wrapping, one position per lane)"
    if getattr(job, "sse", False):
        return "i8 differences + int32 H (the SSE kernels' data flow, one position per lane)"
    npk = res["packed_pairs"]
    return "int16x2 (packed, two alignments per lane)" if npk == res["n"] else "int32" if npk == 0 else "int16x2 + int32"


def scatter_gather_leg(lib, job, rank, world, barrier, red_dev, steps=3):
    """ksw2_amd/parallel.py::sharded on the headline workload: rank 0 owns `world` x n pairs in host memory; one step = LPT
    partition + point-to-point scatter (RCCL) + every rank's batch call + gather of records and CIGARs back to rank 0."""
    import torch
    import torch.distributed as dist
    from ksw2_amd import parallel
    wl = job.wl
    n = min(job.n, 8192)
    q = t = None
    if rank == 0:
        q, t = make_batch(wl, 0, n * world)
    S = SCORING
    sc = dict(mat=job.mat, q=S["q"], e=S["e"], q2=S["q2"], e2=S["e2"])
    kw = dict(w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"], raw=True)
    parallel.sharded(lib, job.kind, q, t, sc, **kw)                  # warm-up
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = parallel.sharded(lib, job.kind, q, t, sc, **kw)
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    if rank != 0:
        return None
    ql = np.full(n * world, wl["qlen"]) if not wl.get("ragged") else np.array([len(x) for x in q])
    tl = np.full(n * world, wl["tlen"]) if not wl.get("ragged") else np.array([len(x) for x in t])
    cells = float(parallel.band_cells(ql, tl, np.full(len(ql), wl["w"])).sum())
    return {"value": round(cells * steps / dt / 1e9, 2), "unit": "GCUPS", "pairs_per_step": n * world, "steps": steps,
            "ms_per_step": round(dt / steps * 1e3, 3), "records_checked": int(out[0].shape[0]),
            "what": "rank 0 holds the batch: LPT partition, torch.distributed point-to-point scatter, per-rank batch call, gather to rank 0"}


def pin_to_gpu(dev, local_rank, local_world):
    """Put this rank (and so every thread the library starts later) on the cores next to its GPU: /sys/bus/pci/devices/<bdf>/
    local_cpulist of the device's PCI function, shared evenly by the ranks whose GPUs hang off the same NUMA node.  Eight ranks x (six
    batch workers + up to 24 gather threads + pinned staging) on one 256-thread host is where weak scaling bleeds first.  Returns what
    it did for the JSON line; no /sys entry (or KSW2_BENCH_NO_PIN=1): nothing is changed."""
    info = {"pinned": False}
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(base + "/numa_node").read().strip())
        cpus = []
        for part in open(base + "/local_cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        info.update({"pci": bdf, "numa_node": node, "local_cpus": len(cpus)})
        # ranks whose GPU reports the same local CPU list split it (device order = rank order on one node)
        same = [d for d in range(local_world) if _local_cpulist(d) == cpus]
        if os.environ.get("KSW2_BENCH_NO_PIN") or not cpus or dev not in same:
            return info
        k, share = same.index(dev), max(1, len(cpus) // len(same))
        mine = cpus[k * share:(k + 1) * share] or cpus
        os.sched_setaffinity(0, mine)
        info.update({"pinned": True, "cpus": len(mine), "cpu_first": mine[0], "cpu_last": mine[-1]})
    except Exception as exc:                              # no such attribute / file: leave the affinity alone
        info["note"] = "%s: %s" % (type(exc).__name__, exc)
    return info


def _local_cpulist(dev):
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev)
        out = []
        for part in open("/sys/bus/pci/devices/%04x:%02x:%02x.0/local_cpulist" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)).read().strip().split(","):
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
        return out
    except Exception:
        return None


def lpt_imbalance(world):
    """Config 5 as BASELINE.json states it (1 M ONT-like pairs) cut for `world` ranks by ksw2_amd/parallel.py::lpt_partition on the
    exact band cells: the largest share over the mean share -- what the rank-0-holds-the-batch form of the multi-GPU path would lose
    to imbalance.  Lengths only (no sequences are generated)."""
    from ksw2_amd import parallel
    wl = WORKLOADS["cfg5-share"]
    n = 1000000
    ql, tl = synth.ragged_lengths(wl["idx"], n, 300, 20000, sub=wl["sub"], ind=wl["ind"], maxdiff=450)
    cost = parallel.band_cells(ql, tl, np.full(n, wl["w"]))
    parts = parallel.lpt_partition(cost, world)
    share = np.array([float(cost[idx].sum()) for idx in parts])
    return {"pairs": n, "ranks": world, "max_over_mean_cells": round(float(share.max() / share.mean()), 6), "cells_total": float(cost.sum())}


# ---------------------------------------------------------------------------------------------------------------------------
# The ONE line the driver parses.  Round 5's line carried every workload's prose and grew to 25.6 KB: the driver could not parse
# it.  The line now holds the contract's keys, short `roofline` / `cpu_baseline` objects and one object of <= 9 short keys per
# `also` workload; everything else (kernel lists, host pipeline counters, parity detail, per-workload CPU samples, sentences)
# goes to DETAIL_FILE.  tests/test_bench_units.py::test_final_line_stays_small pins the size.
DETAIL_FILE = "bench_detail.json"
LINE_BUDGET = 6000
ROOFLINE_LINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "ops_per_cell", "kernel_ms", "fill_kernel_ms",
                      "kernel_gcups", "pairs_per_launch", "cells_per_launch", "algorithmic_bytes")


def _sig(x, digits=4):
    """Numbers of the compact array: `digits` significant digits are what a run-to-run spread of +-10 % leaves meaningful."""
    if x is None or isinstance(x, (str, bool)):
        return x
    return float("%.*g" % (digits, float(x)))


def compact_also(entry):
    """One `also` workload as the short object the line carries: w = workload, v = GCUPS end to end (transfer-inclusive), flat = the flat-arena
    entry, res = HBM-resident kernel rate, frac = its fraction of the VALU roofline, ms = kernel ms per launch, cpu1 / cpuN = the reference's own
    function on 1 thread / all host cores (GCUPS), par = parity sample (pointer entry, flat entry)."""
    if "error" in entry:
        return {"w": entry["name"], "err": str(entry["error"])[:80]}
    o = {"w": entry["name"], "v": _sig(entry["value"]), "flat": _sig(entry.get("value_flat_arena")), "res": _sig(entry["value_hbm_resident"]),
         "frac": _sig(entry["roofline"]["frac"], 3), "ms": _sig(entry["roofline"]["kernel_ms"])}
    cb = entry.get("cpu_baseline")
    if cb:
        o["cpu1"] = _sig(cb["value"], 3)
        if "all_cores" in cb:
            o["cpuN"] = _sig(cb["all_cores"]["value"], 3)
    par = entry["parity_sample"].split(" ")[0]
    if entry.get("parity_sample_flat_arena") not in (None, par):
        par = "MISMATCH"
    o["par"] = par
    return {k: v for k, v in o.items() if v is not None}


def compact_line(out, detail_path):
    """The driver's line from the full record (see above); `out` itself is what DETAIL_FILE holds."""
    keep = ("metric", "value", "unit", "value_hbm_resident", "value_flat_arena", "pairs_per_s", "n_gpus", "steps", "warmup", "ms_per_step",
            "timed_region_s", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in keep if k in out}
    cfg = out["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "cells_per_step_per_gpu", "setup_priming_batches", "parallelism") if k in cfg}
    if "per_rank" in cfg:                                    # N > 1: every rank's own rate; affinity and host-thread times are in the detail file
        line["config"]["per_rank_gcups"] = [r["gcups"] for r in cfg["per_rank"]]
    if "rank0_scatter_gather" in cfg:
        sg = cfg["rank0_scatter_gather"]
        line["config"]["rank0_scatter_gather"] = {k: sg[k] for k in ("value", "pairs_per_step", "ms_per_step", "records_checked")}
    if "cfg5_lpt_imbalance" in cfg and "max_over_mean_cells" in cfg["cfg5_lpt_imbalance"]:
        line["config"]["cfg5_lpt_max_over_mean"] = cfg["cfg5_lpt_imbalance"]["max_over_mean_cells"]
    line["roofline"] = {k: out["roofline"][k] for k in ROOFLINE_LINE_KEYS if k in out["roofline"]}
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "pairs_per_s", "all_cores") if k in cb}
    for k in ("gpu_over_cpu_1thread", "gpu_over_cpu_all_cores", "parity_sample", "parity_sample_flat_arena", "parity_failed"):
        if k in out:
            line[k] = out[k]
    if out.get("also"):
        line["also_keys"] = "w=workload v=GCUPS(e2e) flat=flat-arena entry res=HBM-resident frac=of VALU roofline ms=kernel ms/launch cpu1/cpuN=reference CPU 1 thread/all cores par=parity"
        line["also"] = [compact_also(a) for a in out["also"]]
    line["detail"] = detail_path
    return line


def write_detail(out):
    """The full record next to the script (and under gpurun_out/ when that exists, so it comes back from a GPU box)."""
    path = os.path.join(ROOT, DETAIL_FILE)
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        god = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(god):
            with open(os.path.join(god, DETAIL_FILE), "w") as f:
                json.dump(out, f, indent=1)
    except OSError as exc:                                   # a read-only checkout: the line still goes out
        return "unwritten (%s)" % type(exc).__name__
    return DETAIL_FILE


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="10k", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="override pairs per step and GPU (parity/debug only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--also-cpu-seconds", type=float, default=3.0, help="CPU sample per `also` workload (1 thread, then all cores); 0 = none")
    ap.add_argument("--also", default=None, help="comma-separated workloads for the `also` array ('' = none; default: the other configs at N = 1)")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--resident-only", action="store_true", help="profiling: only the HBM-resident kernel loop (what rocprofv3 should see)")
    ap.add_argument("--approx", action="store_true", help="OR KSW_EZ_APPROX_MAX into the flags (score + corner CIGAR only, as in the reference)")
    ap.add_argument("--sse-compat", action="store_true", help="OR KSW2AMD_EZ_SSE_COMPAT into the flags: the SSE kernels' own results through the SSE-compatible kernels")
    args = ap.parse_args()

    # N > 1 without a launcher: start our own ranks BEFORE anything touches a GPU (a process that has initialised HIP is never
    # re-exec'ed); the child ranks print the JSON line, we pass their exit code on
    if args.gpus > 1 and "RANK" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    # KSW2_BENCH_PLUMBING_LIB=<tests/sim/libksw2_amd_sim.so>: the CPU test tier runs this script's multi-rank plumbing (rendezvous,
    # barriers, reductions, the JSON line) against the lock-step simulator build with a handful of pairs.  The line then says so
    # ("data": "... NOT A MEASUREMENT"); no GPU, no product library, no number anyone should read.
    plumbing = os.environ.get("KSW2_BENCH_PLUMBING_LIB")
    if not plumbing and not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    # KSW2_BENCH_BACKEND=gloo + KSW2_BENCH_ONE_DEVICE=1 exist only to exercise the multi-rank code path on a 1-GPU box
    backend = "gloo" if plumbing else os.environ.get("KSW2_BENCH_BACKEND", "nccl")
    dev = 0 if os.environ.get("KSW2_BENCH_ONE_DEVICE") else local_rank
    if not plumbing:
        torch.cuda.set_device(dev)
    # before the library starts any thread: this rank's threads next to its GPU (reported per rank in config.per_rank)
    affinity = pin_to_gpu(dev, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world))) if not plumbing and world > 1 else {"pinned": False}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    if plumbing:
        if not (0 < args.pairs <= 256):
            sys.exit("KSW2_BENCH_PLUMBING_LIB needs --pairs <= 256: the simulator is a test vehicle")
        lib = ksw2_amd.Library(plumbing)
        stream = None
    else:
        lib = ksw2_amd.library()
        lib.set_device(dev)
        stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        if not plumbing:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not plumbing:
            torch.cuda.synchronize()

    # ------------------------------------------------------------------ headline
    job = Job(lib, args.workload, WORKLOADS[args.workload], rank, args.pairs or None, approx=args.approx, sse=args.sse_compat)
    if args.resident_only:
        res = job.resident(args.steps, args.warmup, stream)
        if rank == 0:
            print(json.dumps({"resident_only": True, "workload": describe(job, world), "roofline": roofline_of(job, res, args.workload)}))
        return
    # Library set-up, outside the W warm-up steps and the K timed ones: the batch entry points size their per-thread staging and device
    # buffer caches on first use (page-locked host memory: slow to allocate), which takes about three batches to settle.  With W >= 3
    # (the driver runs W = 5) nothing is added; with a smaller W the missing batches run here, untimed, and the line says so.
    priming = max(0, 3 - args.warmup)
    for _ in range(priming):
        job.e2e_step()
    for _ in range(args.warmup):
        job.e2e_step()
    stats0 = lib.host_stats()
    phase0 = lib.host_phase_ms()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.e2e_step()
    barrier()
    dt = time.perf_counter() - t0
    stats1 = lib.host_stats()
    phase1 = lib.host_phase_ms()
    # host-thread milliseconds per step of this rank's batch entry calls (summed over its worker threads): plan creation = pack / gather /
    # uploads issued; launch calls; waiting for the device + fetch + ksw_extz_t assembly.  In config.per_rank: which side a slow rank was slow on.
    host_ms = {k: round((phase1[k] - phase0[k]) / max(args.steps, 1), 3) for k in ("create_ms", "launch_ms", "wait_fetch_ms")}
    host_ms["plans_per_step"] = round((phase1["plans"] - phase0["plans"]) / max(args.steps, 1), 2)
    per_rank = None
    if world > 1:
        mine = torch.tensor([float(job.cells), dt], dtype=torch.float64, device=red_dev)
        allr = [torch.zeros(2, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(allr, mine)                                     # load balance: every rank's cells and its own loop time
        per_rank = [{"rank": r, "cells_per_step": float(x[0].item()), "loop_seconds": round(float(x[1].item()), 4),
                     "gcups": round(float(x[0].item()) * args.steps / float(x[1].item()) / 1e9, 1)} for r, x in enumerate(allr)]
        aff = [None] * world
        dist.all_gather_object(aff, affinity)                           # NUMA node / cores every rank ran on
        hm = [None] * world
        dist.all_gather_object(hm, host_ms)
        for r in range(world):
            per_rank[r]["affinity"] = aff[r]
            per_rank[r]["host_thread_ms_per_step"] = hm[r]
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cc = torch.tensor([float(job.cells)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        cells_all = float(cc.item())
        pairs_all = job.n * world
    else:
        cells_all, pairs_all = float(job.cells), job.n
    parity = job.parity_sample(16) if rank == 0 else None      # outside the timed region: the results the timed calls returned
    # the same K steps through the flat entry point (one arena + offsets, page-locked once): what a caller that owns its read
    # buffer would use.  Timed exactly like the loop above.
    dt_flat, parity_flat = None, None
    if job.flat_ready():
        for _ in range(max(1, min(args.warmup, 2))):
            job.e2e_step(flat=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            job.e2e_step(flat=True)
        barrier()
        dt_flat = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_flat], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_flat = float(tt.item())
        parity_flat = job.parity_sample(8, flat=True) if rank == 0 else None
        job.flat_done()
    job.free_ez()
    lib.release_cache()
    barrier()
    # north_star's form of the multi-GPU path, next to the host-side sharding above ("report both"): rank 0 holds
    # the whole batch, LPT-partitions it, scatters sequences + parameters and gathers records + CIGARs over RCCL
    sg = scatter_gather_leg(lib, job, rank, world, barrier, red_dev) if world > 1 and job.kind in ("extz", "extd") else None
    res = job.resident(args.steps, max(1, min(args.warmup, 3)), stream)
    lib.release_cache()

    out = None
    if rank == 0:
        rl = roofline_of(job, res, args.workload)
        rl["note"] = ("integer-VALU bound (no dense contraction, SURVEY 8d); peak = the guide's vector peak 256CU x 4SIMD x 32 lanes x 2.4GHz x 2 "
                      "(packed int16); measured with tools/probe/valu_rate.hip (profiles/r1d_valu_rate.txt): v_pk_*_i16, v_max_i32, v_bfi issue one "
                      "wave64 instruction per 4 cycles per SIMD (add/sub/xor/bitop3: 2), i.e. 39.3 T lane-instr/s = 78.6 T 16-bit ops/s, half of `peak`")
        value = cells_all * args.steps / dt / 1e9
        out = {
            "metric": "GCUPS (DP cells/s) + pairs/s at fixed (qlen,tlen,band)",
            "value": round(value, 3), "unit": "GCUPS",
            "value_definition": "exact-band cells of all pairs / wall time of `steps` calls of the batch entry point on host-memory inputs: "
                                "pack + H2D + kernels + D2H + ksw_extz_t assembly (SURVEY 8d)",
            "value_hbm_resident": rl["kernel_gcups"],
            "value_flat_arena": round(cells_all * args.steps / dt_flat / 1e9, 3) if dt_flat else None,
            "value_flat_arena_definition": "the same steps through ksw2amd_ext?_batch_flat: the batch as one page-locked host arena + offsets (no per-pair gather, "
                                           "no host pass over the bytes); transfer-inclusive like `value`" if dt_flat else None,
            "pairs_per_s": round(pairs_all * args.steps / dt, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "timed_region_s": round(dt, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_of(job, res),
            "data": "synthetic" if not plumbing else "synthetic -- PLUMBING TEST ON THE CPU SIMULATOR, NOT A MEASUREMENT",
            "config": {"workload": describe(job, world), "cells_per_step_per_gpu": job.cells,
                       "setup_priming_batches": priming,      # untimed library set-up batches in front of the W warm-up steps (0 when W >= 3)
                       "host_pipeline": {k: stats1[k] - stats0[k] for k in stats1},
                       "parallelism": "pairs sharded over %d GPU(s), one process per GPU, no collective in the data path" % world},
            "roofline": rl,
            "parity_sample": parity["result"], "parity_detail": parity,
        }
        if dt_flat:
            out["parity_sample_flat_arena"] = parity_flat["result"]
        if sg:
            out["config"]["rank0_scatter_gather"] = sg
        if per_rank:
            out["config"]["per_rank"] = per_rank
        else:
            out["config"]["host_thread_ms_per_step"] = host_ms
        if world > 1 and not plumbing:
            try:
                out["config"]["cfg5_lpt_imbalance"] = lpt_imbalance(world)
            except Exception as exc:
                out["config"]["cfg5_lpt_imbalance"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        # the reference on this box's host cores (rank 0, N = 1 only), BEFORE the `also` loop: workloads that make the same reference call
        # share this sample (cpu_call_key) instead of timing it again
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(job.wl, job.q, job.t, job.mat, seconds=args.cpu_seconds)
            if out["cpu_baseline"]["value"]:
                out["gpu_over_cpu_1thread"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
                if "all_cores" in out["cpu_baseline"]:
                    out["gpu_over_cpu_all_cores"] = round(out["value"] / out["cpu_baseline"]["all_cores"]["value"], 1)
    # ------------------------------------------------------------------ the other configurations (N = 1: one run covers them all)
    names = ALSO_DEFAULT if args.also is None else [x for x in args.also.split(",") if x]
    if args.no_also or world > 1 or args.pairs or args.approx or args.sse_compat or args.workload != "10k":
        names = [] if args.also is None else names
    if world > 1 and args.also is None and not (args.no_also or args.pairs or args.approx or args.sse_compat or args.workload != "10k"):
        names = list(ALSO_MULTI)                                 # the multi-GPU configurations at their per-GPU share, every rank its own slice
    also = []
    mismatch = parity is not None and parity["result"] == "MISMATCH"
    if dt_flat and rank == 0 and parity_flat is not None and parity_flat["result"] == "MISMATCH":
        mismatch = True

    def all_ok(ok):
        """N > 1: every rank must get through a set-up step before any of them enters a collective -- a rank that failed alone (an
        allocation, say) would otherwise leave the others waiting in the barrier until the RCCL timeout."""
        if world == 1:
            return ok
        f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item() > 0.5)

    for name in names:
        if name == args.workload:
            continue
        j, err = None, None
        try:
            j = Job(lib, name, WORKLOADS[name], rank, ALSO_MULTI.get(name) if world > 1 else None)
        except Exception as exc:
            err = "%s: %s" % (type(exc).__name__, exc)
        if not all_ok(j is not None):
            also.append({"name": name, "workload": name, "error": err or "set-up failed on another rank"})
            lib.release_cache()
            continue
        try:
            big = j.cells > 5e11                                 # seconds per step: one warm-up, two timed steps
            if world > 1:                                        # (step counts from the SAME figure on every rank)
                bt = torch.tensor([float(j.cells)], dtype=torch.float64, device=red_dev)
                dist.all_reduce(bt, op=dist.ReduceOp.MAX)
                big = float(bt.item()) > 5e11
            def timed(flat):
                """>= 3 steps and >= 1.5 s (big workloads: 2 steps) of the batch entry point; N > 1: between barriers, the slowest rank's time,
                every rank's cells."""
                for _ in range(1 if big else 2):                 # warm-up: buffers, streams, worker threads (every worker's first chunk
                    j.e2e_step(flat=flat)                        # allocates its device buffers)
                barrier()
                t0 = time.perf_counter()
                kk = 0
                while kk < (2 if big else 3) or (world == 1 and not big and time.perf_counter() - t0 < 1.5):
                    j.e2e_step(flat=flat)
                    kk += 1
                barrier()
                el = time.perf_counter() - t0
                cells = float(j.cells)
                if world > 1:
                    red = torch.tensor([el, -cells], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(red, op=dist.ReduceOp.MAX)           # slowest rank
                    el = float(red[0].item())
                    tot = torch.tensor([cells], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
                    cells = float(tot.item())
                return kk, el, cells
            k, edt, cells_sum = timed(False)
            jpar = j.parity_sample(PARITY_PAIRS.get(name, 8))
            vflat = None
            if j.flat_ready():
                kf, fdt, _ = timed(True)
                vflat = round(cells_sum * kf / fdt / 1e9, 2)
                jparf = j.parity_sample(max(4, PARITY_PAIRS.get(name, 8) // 4), flat=True)
                mismatch = mismatch or jparf["result"] == "MISMATCH"
                j.flat_done()
            j.free_ez()
            lib.release_cache()
            r = j.resident(3, 1, stream, min_seconds=1.0)
            lib.release_cache()
            rr = roofline_of(j, r, name)
            mismatch = mismatch or jpar["result"] == "MISMATCH"
            also.append({"name": name, "workload": describe(j, world), "n_gpus": world, "value": round(cells_sum * k / edt / 1e9, 2), "value_flat_arena": vflat, "value_hbm_resident": rr["kernel_gcups"],
                         **({"parity_sample_flat_arena": jparf["result"]} if vflat else {}),
                         "unit": "GCUPS", "pairs_per_s": round(j.n * world * k / edt, 1), "steps": k, "ms_per_step": round(edt / k * 1e3, 3),
                         "dtype": dtype_of(j, r), "parity_sample": "%s (%d pairs)" % (jpar["result"], jpar["pairs"]),
                         "roofline": {x: rr[x] for x in ("frac", "kernel_ms", "fill_kernel_ms", "ops_per_cell", "pairs_per_launch", "kernels",
                                                         "kernel_gcups_cells_filled", "early_stop_fraction", "zdropped_pairs", "traffic", "traffic_source",
                                                         "traffic_measured_in")},
                         # SURVEY 8d "per config": the reference's own function on this box's host cores, a bounded sample of this batch
                         **({"cpu_baseline": cpu_baseline(j.wl, j.q, j.t, j.mat, seconds=args.also_cpu_seconds)} if not args.no_cpu and args.also_cpu_seconds > 0 else {})})
            del j
        except Exception as exc:                                  # one workload must not take the headline with it
            if world > 1:
                raise                                             # ... but with several ranks a lone failure inside the collectives cannot be contained: fail loudly
            also.append({"name": name, "workload": name, "error": "%s: %s" % (type(exc).__name__, exc)})
            lib.release_cache()
    if rank == 0:
        if also:
            out["also"] = also
        if mismatch:
            out["parity_failed"] = True                           # a timed batch whose sampled results differ from the oracle: the number is void
        line = compact_line(out, write_detail(out))
        text = json.dumps(line, separators=(",", ":"))
        if len(text) > LINE_BUDGET:                               # never again a line the driver cannot parse: shed the optional parts, loudly
            sys.stderr.write("bench.py: final line %d > %d characters, dropping the `also` array (it is in %s)\n" % (len(text), LINE_BUDGET, DETAIL_FILE))
            line.pop("also", None)
            line.pop("also_keys", None)
            text = json.dumps(line, separators=(",", ":"))
        print(text)
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and mismatch:
        sys.exit(3)


if __name__ == "__main__":
    main()
