"""CPU: bench.py's pieces that do not need a GPU -- the closed-form cell counts, the fast synthetic generator, and a dry run of
one Job (transfer-inclusive step + resident loop + roofline record) on the simulator build."""
import os
import subprocess

import numpy as np
import pytest

import bench
import ksw2_amd as ka
from ksw2_amd import synth

SIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sim")


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", SIM_DIR], check=True, capture_output=True)
    return ka.Library(os.path.join(SIM_DIR, "libksw2_amd_sim.so"))


def test_cells_of_rows_matches_definition():
    rng = np.random.default_rng(5)
    for _ in range(300):
        ql, tl, w = int(rng.integers(1, 90)), int(rng.integers(1, 90)), int(rng.integers(0, 100))
        assert int(bench.cells_of_rows(ql, tl, min(w, max(ql, tl)))) == synth.band_cells(ql, tl, w)
        rows = int(rng.integers(0, tl + 1))
        assert int(bench.cells_of_rows(ql, rows, min(w, max(ql, tl)))) == (synth.band_cells(ql, rows, min(w, max(ql, tl))) if rows else 0)


def test_fast_generator_is_a_function_of_the_pair_index():
    q, t = synth.fast_fixed(6, 40, 700, 650, sub=0.05, ind=0.06)
    q2, t2 = synth.fast_fixed(6, 10, 700, 650, sub=0.05, ind=0.06, first=30)
    assert (q[30:] == q2).all() and (t[30:] == t2).all() and q.max() <= 3 and t.max() <= 3
    qs, ts = synth.fast_ragged(5, 50, 300, 2000, maxdiff=100)
    qs2, ts2 = synth.fast_ragged(5, 20, 300, 2000, maxdiff=100, first=30)
    assert all((a == b).all() for a, b in zip(qs[30:], qs2)) and all((a == b).all() for a, b in zip(ts[30:], ts2))
    assert all(300 <= len(a) <= 2000 and abs(len(a) - len(b)) <= 100 for a, b in zip(qs, ts))
    # the channel leaves most of the read alignable: the oracle's global score is clearly positive
    from oracle import pyoracle as po
    r = po.align("oracle", "extd2", qs[0], ts[0], synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1, w=200, zdrop=-1, flag=po.SCORE_ONLY)
    assert r["score"] > len(qs[0]) // 2


@pytest.mark.parametrize("name", ["10k", "cfg3", "cfg5", "cfg4", "exts", "extf"])
def test_job_dry_run_on_the_simulator(sim, name, monkeypatch):
    wl = dict(bench.WORKLOADS[name])
    if name in ("10k", "cfg3"):
        wl.update(qlen=300, tlen=300, w=40)
    if name == "cfg4":
        monkeypatch.setattr(bench, "make_batch", lambda wl_, rank, n: synth.fast_fixed(4, n, 500, 520))
        wl.update(resident_n=3)
    if name == "cfg5":
        monkeypatch.setattr(bench, "make_batch", lambda wl_, rank, n: synth.fast_ragged(5, n, 100, 400, maxdiff=60))
        wl.update(w=80)
    if name in ("exts", "extf"):
        wl.update(qlen=120, tlen=300 if name == "exts" else 120, w=-1 if name == "exts" else 30)
    j = bench.Job(sim, name, wl, 0, n_override=6)
    j.e2e_step()
    j.e2e_step()
    assert j.ez[0].score > ka.KSW_NEG_INF or j.ez[0].max_zd >> 31
    j.free_ez()
    r = j.resident(2, 1, None)
    rl = bench.roofline_of(j, r, name)
    assert r["cells"] > 0 and 0 <= rl["early_stop_fraction"] <= 1 and rl["kernel_gcups"] >= 0 and r["kernel_ms"] > 0      # (the simulator's rate rounds to 0.00 on a busy box)
    if name not in ("exts",):
        assert r["cells"] == (j.cells if name != "cfg4" else int(bench.cells_of_rows(j.qlen[:3], j.tlen[:3], j.weff[:3]).sum()))
    assert bench.describe(j, 1).startswith(name) and bench.dtype_of(j, r)


def test_bench_two_ranks_plumbing(sim, tmp_path):
    """`python bench.py --gpus 2` end to end on the CPU tier: the script launches its own two ranks through torch.distributed.run
    (before anything could touch a GPU), they rendezvous over gloo, time K steps between barriers, reduce MAX time / SUM cells,
    run the rank-0 scatter/gather leg (ksw2_amd/parallel.py) and rank 0 prints ONE well-formed JSON line.  The simulator build stands
    in for the per-rank GPU (KSW2_BENCH_PLUMBING_LIB): the line says that it is not a measurement."""
    import json
    import sys
    root = os.path.dirname(SIM_DIR.rstrip("/")).rsplit("/tests", 1)[0]
    env = dict(os.environ, KSW2_BENCH_PLUMBING_LIB=os.path.join(SIM_DIR, "libksw2_amd_sim.so"), MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg3", "--pairs", "24",
                        "--no-cpu", "--no-also"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "GCUPS"
    assert d["value"] > 0 and d["parity_sample"] == "ok" and "NOT A MEASUREMENT" in d["data"]
    assert len(lines[0]) < bench.LINE_BUDGET and len(d["config"]["per_rank_gcups"]) == 2
    assert d["config"]["rank0_scatter_gather"]["records_checked"] == 48
    assert {"bound", "achieved", "peak", "frac", "traffic"} <= set(d["roofline"])
    # everything the line leaves out is in the side file it names
    full = json.load(open(os.path.join(root, d["detail"])))
    pr = full["config"]["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["cells_per_step"] > 0 for x in pr)
    assert full["value"] == d["value"] and "kernels" in full["roofline"] and "parity_detail" in full


def test_recorded_traffic_names_the_latest_profile_of_the_same_workload():
    """`roofline.traffic` replays the PMC summary under profiles/ whose name sorts last for exactly this workload: records of
    attempts that were not kept (another kernel) carry another workload tag and must not be picked up."""
    for wl, kern in (("10k-cigar", "k2a_fill_pk_kernel"), ("exts", "k2a_exts_kernel"), ("extf-w900", "k2a_extf_grp_kernel"), ("10k-ssec", "k2a_ssec_blk_kernel")):
        traffic, src = bench.recorded_traffic(wl)
        assert traffic and src and src.endswith("_%s_pmc.json" % wl), (wl, src)
        import json
        assert kern in json.load(open(os.path.join(bench.ROOT, src)))["derived"]["dominant_kernel"], (wl, src)
    assert bench.recorded_traffic("no-such-workload") == (None, None)


def _fake_full_record(n_also):
    """A record of the size a default run produces: the headline with every detail key and `n_also` workloads with their prose."""
    rl = {"bound": "valu", "achieved": 74.9123, "peak": 157.3, "unit": "Tiop/s", "frac": 0.47623, "traffic": 137451702221.7143,
          "traffic_source": "profiles/r6zz_10k-ssec-approx_pmc.json", "traffic_measured_in": "x" * 330, "ops_per_cell": 15, "kernel_ms": 96.0612,
          "fill_kernel_ms": 95.2511, "kernel_gcups": 4993.71, "pairs_per_launch": 49152, "cells_per_launch": 479698944000,
          "kernel_gcups_cells_filled": 4993.71, "early_stop_fraction": 0.0, "zdropped_pairs": 0, "kernels": ["pk(64,16) gaps=1 exact rebased defer x384"] * 3,
          "algorithmic_bytes": 985792512, "hbm_algorithmic_GBps": 10.26, "hbm_peak_GBps": 8000.0, "traffic_note": "y" * 160, "note": "z" * 380}
    cb = {"value": 2.5356, "unit": "GCUPS", "cores": 1, "kind": "reference", "sample": "2592 pairs of the batch, 10.0 s, reference ksw_extz2_sse gcc -O2 -msse4.1",
          "pairs_per_s": 259.2, "all_cores": {"value": 59.2621, "cores": 256, "pairs_per_s": 6072.3}}
    also = [{"name": "10k-ssec-approx", "workload": "w" * 130, "n_gpus": 1, "value": 1234.56, "value_flat_arena": 1300.12, "value_hbm_resident": 1721.33,
             "parity_sample_flat_arena": "ok", "unit": "GCUPS", "pairs_per_s": 12345.6, "steps": 3, "ms_per_step": 123.456, "dtype": "d" * 40,
             "parity_sample": "ok (8 pairs)", "roofline": dict(rl), "cpu_baseline": dict(cb)} for _ in range(n_also)]
    also.append({"name": "cfg4", "workload": "cfg4", "error": "RuntimeError: " + "e" * 300})
    return {"metric": "GCUPS (DP cells/s) + pairs/s at fixed (qlen,tlen,band)", "value": 4809.123, "unit": "GCUPS", "value_definition": "v" * 200,
            "value_hbm_resident": 4993.71, "value_flat_arena": 4901.2, "value_flat_arena_definition": "f" * 200, "pairs_per_s": 492345.1, "n_gpus": 8,
            "steps": 20, "warmup": 5, "ms_per_step": 99.7481, "timed_region_s": 1.995, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16x2 (packed, two alignments per lane)", "data": "synthetic",
            "config": {"workload": "10k: 49152 pairs/step/GPU, qlen=10000 tlen=10000 band=500 zdrop=400 extz2 affine score-only", "cells_per_step_per_gpu": 479698944000,
                       "setup_priming_batches": 0, "host_pipeline": {"k%d" % i: i for i in range(30)}, "parallelism": "p" * 90,
                       "per_rank": [{"rank": r, "cells_per_step": 4.79e11, "loop_seconds": 1.99, "gcups": 4801.2, "affinity": {"a": "b" * 100},
                                     "host_thread_ms_per_step": {"c": 1.0}} for r in range(8)],
                       "rank0_scatter_gather": {"value": 1.0, "unit": "GCUPS", "pairs_per_step": 8, "steps": 3, "ms_per_step": 1.0, "records_checked": 8, "what": "s" * 150},
                       "cfg5_lpt_imbalance": {"pairs": 1000000, "ranks": 8, "max_over_mean_cells": 1.000001, "cells_total": 1e13}},
            "roofline": rl, "parity_sample": "ok", "parity_detail": {"pairs": 16, "result": "ok", "checked": "c" * 200}, "parity_sample_flat_arena": "ok",
            "cpu_baseline": cb, "gpu_over_cpu_1thread": 1896.6, "gpu_over_cpu_all_cores": 81.1, "also": also}


def test_final_line_stays_small():
    """the driver parses ONE line and keeps an 8 081-character tail; round 5's 25.6 KB line left the round unmeasured.
    A full-size record (8 ranks, 24 `also` workloads -- six more than the default list) must give a line under 6 000 characters that still carries
    the contract's keys, `roofline`, `cpu_baseline` and one short object per workload."""
    import json
    out = _fake_full_record(24)
    line = bench.compact_line(out, bench.DETAIL_FILE)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.LINE_BUDGET <= 6000, len(text)
    assert len(bench.ALSO_DEFAULT) <= 24
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "parity_sample"):
        assert k in line, k
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert line["config"]["workload"].startswith("10k") and len(line["config"]["per_rank_gcups"]) == 8
    assert all(len(a) <= 9 and all(not isinstance(v, str) or len(v) <= 90 for v in a.values()) for a in line["also"])
    assert line["also"][0] == {"w": "10k-ssec-approx", "v": 1235.0, "flat": 1300.0, "res": 1721.0, "frac": 0.476, "ms": 96.06, "cpu1": 2.54, "cpuN": 59.3, "par": "ok"}
    assert "err" in line["also"][-1]
    # a flat-arena mismatch must not hide behind the pointer entry's "ok"
    out["also"][0]["parity_sample_flat_arena"] = "MISMATCH"
    assert bench.compact_also(out["also"][0])["par"] == "MISMATCH"


def test_cpu_samples_are_shared_per_reference_call():
    """The workloads that time the same reference call on the same stream of pairs share one CPU sample (cpu_call_key)."""
    W, mat = bench.WORKLOADS, synth.simple_mat(5, 2, 4, -1)
    k = lambda n, **kw: bench.cpu_call_key(dict(W[n], name=n, **kw), mat)
    assert k("10k") == k("10k-n1024") == k("10k-N") == k("10k-ssec") == k("10k-ssec-n4096")
    assert k("10k-cigar") == k("10k-ssec-cigar") != k("10k")
    assert len({k("10k"), k("10k-zdrop"), k("10k-ssec-approx"), k("cfg2"), k("cfg3"), k("cfg5"), k("extf"), k("extf-w300")}) == 8
    assert k("10k") != bench.cpu_call_key(dict(W["10k-generic"], name="g"), np.arange(25, dtype=np.int8))
