#!/usr/bin/env python3
"""bench.py -- GCUPS of the banded extension hot path on N MI355X (one process per GPU).

A "step" = one pass of the fill (+ traceback) kernels over one resident batch of synthetic pairs.
Default workload = BASELINE.json configs[1]: 65 536 pairs, qlen = tlen = 512, band 64, extz2 affine,
score-only.  Inputs are resident in HBM before the timed region (ksw2amd_plan_create uploads them);
`value` is whole-job GCUPS = exact-band DP cells of all ranks / max-over-ranks wall time.

Launch:  python bench.py [--gpus 1 --steps K --warmup W]
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
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

WORKLOADS = {
    # name: (config index, n pairs, qlen, tlen, w, zdrop, dual, flag, sub, ind, tail_frac, tail_pairs)
    "cfg2": dict(idx=2, n=65536, qlen=512, tlen=512, w=64, zdrop=-1, dual=False, flag=ksw2_amd.KSW_EZ_SCORE_ONLY, sub=0.05, ind=0.06),
    "cfg3": dict(idx=3, n=16384, qlen=2048, tlen=2048, w=256, zdrop=400, dual=True, flag=0, sub=0.05, ind=0.10, tail_frac=0.25, tail_pairs=0.10),
    # north_star's 10k x 10k banded extension.  4096 pairs per GPU: one alignment (or packed pair) per wavefront, and
    # 1024 SIMDs need a few wavefronts each (1024 pairs leave half of them empty; the host then picks the int32 kernels)
    "10k": dict(idx=6, n=4096, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=ksw2_amd.KSW_EZ_SCORE_ONLY, sub=0.05, ind=0.06),
    "10k-cigar": dict(idx=6, n=4096, qlen=10000, tlen=10000, w=500, zdrop=400, dual=False, flag=0, sub=0.05, ind=0.06),
    # config 4: MT-human x MT-orang (tests/golden/data), full global extz2 with CIGAR, replicated; 1024 replicas per GPU here:
    # 147 GB of traceback (144 MB per pair at 4 bits per cell), one wavefront per SIMD.  4096 replicas do not fit one GPU at
    # once; the batch entry points split such batches.
    "cfg4": dict(idx=4, n=1024, qlen=16499, tlen=16569, w=-1, zdrop=-1, dual=False, flag=0, mt=True),
    # config 5: ONT-like mix, target length uniform in [300, 20000] (64 length buckets), 3 % substitutions + 15 % indels, band 500,
    # extd2 with Z-drop 400 and CIGAR; 16384 pairs per GPU here (the full config shards 1 M pairs over 8 GPUs)
    "cfg5": dict(idx=5, n=16384, qlen=0, tlen=0, w=500, zdrop=400, dual=True, flag=0, sub=0.03, ind=0.15, ragged=True),
    # splice-aware extension: 16384 spliced pairs per GPU, 400-base query = two exons around a 1000-base GT..AG
    # intron of a 1500-base target, unbanded, forward signals, CIGAR with N; the reference CLI's splice scoring
    "exts": dict(idx=7, n=16384, qlen=400, tlen=1500, w=-1, zdrop=-1, dual=False, flag=0, sub=0.03, ind=0.0, splice=True),
    # gap-linear X-drop extension: 16384 pairs per GPU, 1000 x 1000, band 100, no drop (every anti-diagonal runs)
    "extf": dict(idx=8, n=16384, qlen=1000, tlen=1000, w=100, zdrop=-1, dual=False, flag=ksw2_amd.KSW_EZ_SCORE_ONLY, sub=0.05, ind=0.01, linear=True),
}
SCORING = dict(a=2, b=4, sc_n=-1, q=4, e=2, q2=24, e2=1)
LINEAR_SCORING = dict(mch=2, mis=-4, e=2)
SPLICE_SCORING = dict(a=1, b=2, sc_n=0, q=2, e=1, q2=32, noncan=4)


def make_ragged(wl, rank, n):
    """Length-bucketed ragged batch: every bucket is a fixed-shape batch from the vectorised channel; the query keeps the
    length the channel produced on average (tlen * (1 + ind/2 * (mean_ins - mean_del)) ~ tlen), so |tlen - qlen| << band."""
    rng = synth.rng_for(wl["idx"], 1000 + rank)
    nb = 64
    lens = np.sort(rng.integers(300, 20001, size=nb))
    per = [n // nb + (1 if b < n % nb else 0) for b in range(nb)]
    qs, ts = [], []
    for b in range(nb):
        if per[b] == 0:
            continue
        q, t = synth.fixed_batch(wl["idx"], per[b], int(lens[b]), int(lens[b]), sub=wl["sub"], ind=wl["ind"], stream=rank * 100 + b)
        qs += list(q)
        ts += list(t)
    return qs, ts


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


def make_batch(wl, rank, n_override=None):
    n = n_override or wl["n"]
    if wl.get("splice"):
        return make_spliced(wl, rank, n)
    if wl.get("ragged"):
        return make_ragged(wl, rank, n)
    if wl.get("mt"):
        from tests import golden_util as gu
        _, ts = gu.read_fasta("MT-human.fa")
        _, qs = gu.read_fasta("MT-orang.fa")
        return np.repeat(qs[0][None, :], n, axis=0), np.repeat(ts[0][None, :], n, axis=0)
    q, t = synth.fixed_batch(wl["idx"], n, wl["qlen"], wl["tlen"], sub=wl["sub"], ind=wl["ind"],
                             tail_random_frac=wl.get("tail_frac", 0.0), tail_pairs=wl.get("tail_pairs", 0.0), stream=rank)
    return q, t


def cpu_baseline(wl, q, t, mat, seconds=10.0):
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
    olib.kso_cpu_bench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int8, ctypes.c_void_p, ctypes.c_int8, ctypes.c_int8,
                                   ctypes.c_int8, ctypes.c_int8, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    if wl.get("ragged"):
        return {"value": None, "unit": "GCUPS", "cores": 0, "kind": "skipped", "sample": "ragged workload: CPU baseline loop needs fixed shapes"}
    cells_pair = synth.band_cells(wl["qlen"], wl["tlen"], wl["w"])
    S = SCORING
    mode = 3 if wl.get("linear") else 2 if wl.get("splice") else int(wl["dual"])
    if wl.get("linear"):
        S = dict(q=LINEAR_SCORING["mch"], e=LINEAR_SCORING["mis"], q2=LINEAR_SCORING["e"], e2=0)
    if wl.get("splice"):
        S = dict(q=SPLICE_SCORING["q"], e=SPLICE_SCORING["e"], q2=SPLICE_SCORING["q2"], e2=SPLICE_SCORING["noncan"])
    qa, ta = np.ascontiguousarray(q), np.ascontiguousarray(t)
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = {}
    for threads in sorted({1, ncores}):
        el = ctypes.c_double(0)
        done = olib.kso_cpu_bench(fn, mode, threads, seconds, len(qa), wl["qlen"], wl["tlen"], qa.ctypes.data, ta.ctypes.data,
                                  5, mat.ctypes.data, S["q"], S["e"], S["q2"], S["e2"], wl["w"], wl["zdrop"], wl["flag"], ctypes.byref(el))
        out[threads] = (done, el.value, done * cells_pair / el.value / 1e9)
    n1, dt1, g1 = out[1]
    what = "reference %s, gcc -O2 -msse4.1, exact-max mode" % name if kind == "reference" else "oracle int32 scalar port (reference artefact absent: not comparable)"
    res = {"value": round(g1, 4), "unit": "GCUPS", "cores": 1, "kind": kind,
           "sample": "%d pairs of the same batch in %.1f s on 1 thread; %s" % (n1, dt1, what), "pairs_per_s": round(n1 / dt1, 1)}
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
    return d["derived"]["hbm_bytes_gfx950_corrected"], os.path.relpath(cand[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--pairs", type=int, default=0, help="override pairs per GPU (parity/debug only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--approx", action="store_true", help="OR KSW_EZ_APPROX_MAX into the flags (score + corner CIGAR only, as in the reference)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    # KSW2_BENCH_BACKEND=gloo + KSW2_BENCH_ONE_DEVICE=1 exist only to exercise the multi-rank code path on a 1-GPU box
    backend = os.environ.get("KSW2_BENCH_BACKEND", "nccl")
    dev = 0 if os.environ.get("KSW2_BENCH_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    lib = ksw2_amd.library()
    lib.set_device(dev)
    wl = WORKLOADS[args.workload]
    if args.approx:
        wl = dict(wl, flag=wl["flag"] | 0x08)
    S = SCORING
    q, t = make_batch(wl, rank, args.pairs or None)
    n = len(q)
    if wl.get("splice"):
        P = SPLICE_SCORING
        mat = synth.simple_mat(5, P["a"], P["b"], P["sc_n"])
        wl = dict(wl, flag=wl["flag"] | ksw2_amd.KSW_EZ_SPLICE_FOR)
        batch = lib.make_splice_batch(list(q), list(t), mat, P["q"], P["e"], P["q2"], P["noncan"], zdrop=wl["zdrop"], flag=wl["flag"])
        plan = batch.plan()
    elif wl.get("linear"):
        P = LINEAR_SCORING
        mat = synth.simple_mat(5, P["mch"], -P["mis"], 0)          # for the CPU leg's signature only
        batch = lib.make_linear_batch(list(q), list(t), P["mch"], P["mis"], P["e"], w=wl["w"], xdrop=wl["zdrop"])
        plan = batch.plan()
    else:
        mat = synth.simple_mat(5, S["a"], S["b"], 0 if wl.get("mt") else S["sc_n"])
        batch = lib.make_batch(q, t, mat, S["q"], S["e"], S["q2"], S["e2"], w=wl["w"], zdrop=wl["zdrop"], end_bonus=0, flag=wl["flag"])
        plan = batch.plan(wl["dual"])                 # packs and uploads: inputs resident in HBM from here on
    cells = plan.cells()
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan.run(stream)
    barrier()
    fill_ms, total_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(stream)
        # per-launch device time from HIP events recorded on this stream by the library (blocks on the step's last event)
        f, tot = plan.timing()
        fill_ms.append(f)
        total_ms.append(tot)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cc = torch.tensor([cells], dtype=torch.float64, device=red_dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        cells_all = float(cc.item())
        pairs_all = n * world
    else:
        cells_all, pairs_all = float(cells), n

    if rank == 0:
        score_only = bool(wl["flag"] & ksw2_amd.KSW_EZ_SCORE_ONLY)
        ops = OPS_PER_CELL[("extf" if wl.get("linear") else "exts" if wl.get("splice") else "extd" if wl["dual"] else "extz", score_only)]
        kern_ms = float(np.mean(total_ms))
        fill_only_ms = float(np.mean(fill_ms))
        achieved = cells * ops / (kern_ms * 1e-3)
        seq_bytes = sum(len(x) for x in q) + sum(len(x) for x in t) if wl.get("ragged") else n * (wl["qlen"] + wl["tlen"])
        alg_bytes = seq_bytes + 56 * n + (0 if score_only else cells // (1 if wl["dual"] else 2))
        traffic, traffic_src = recorded_traffic(args.workload) if not args.pairs else (None, None)
        npk = plan.packed_pairs()
        dtype = "u8 (wrapping, one position per lane)" if wl.get("linear") else "int16x2 (packed, two alignments per lane)" if npk == n else "int32" if npk == 0 else "int16x2 + int32"
        func = "extf2 gap-linear X-drop" if wl.get("linear") else "exts2 splice-aware" if wl.get("splice") else "extd2 dual-gap" if wl["dual"] else "extz2 affine"
        out = {
            "metric": "GCUPS (DP cells/s) + pairs/s at fixed (qlen,tlen,band)",
            "value": round(cells_all * args.steps / dt / 1e9, 3), "unit": "GCUPS",
            "pairs_per_s": round(pairs_all * args.steps / dt, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": "%s: %d pairs/GPU, qlen=%d tlen=%d band=%d zdrop=%d %s %s" % (
                args.workload, n, wl["qlen"], wl["tlen"], wl["w"], wl["zdrop"], func,
                ("score-only" if score_only else "CIGAR") + (" APPROX_MAX" if args.approx else "")), "cells_per_gpu": cells, "parallelism": "pairs sharded over %d GPU(s), no collective" % world},
            "roofline": {"bound": "valu", "achieved": round(achieved / 1e12, 4), "peak": VALU_PEAK_PK16 / 1e12, "unit": "Tiop/s",
                         "frac": round(achieved / VALU_PEAK_PK16, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "ops_per_cell": ops, "kernel_ms": round(kern_ms, 4), "fill_kernel_ms": round(fill_only_ms, 4),
                         "kernel_gcups": round(cells / (kern_ms * 1e-3) / 1e9, 2),
                         "algorithmic_bytes": alg_bytes, "hbm_algorithmic_GBps": round(alg_bytes / (kern_ms * 1e-3) / 1e9, 2), "hbm_peak_GBps": HBM_PEAK / 1e9,
                         "note": "integer-VALU bound (no dense contraction, SURVEY 8d); peak = the guide's vector peak 256CU x 4SIMD x 32 lanes x 2.4GHz x 2 "
                                 "(packed int16); measured with tools/probe/valu_rate.hip (profiles/r1d_valu_rate.txt): v_pk_*_i16, v_max_i32, v_bfi issue one "
                                 "wave64 instruction per 4 cycles per SIMD (add/sub/xor/bitop3: 2), so the attainable rate for this recurrence is 39.3 T "
                                 "lane-instr/s; the fill kernels run at 4.35-4.45 cycles per instruction"},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(wl, q, t, mat, seconds=args.cpu_seconds)
            if out["cpu_baseline"]["value"]:
                out["gpu_over_cpu_1thread"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    plan.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
