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
    pr = d["config"]["per_rank"]
    assert [x["rank"] for x in pr] == [0, 1] and all(x["cells_per_step"] > 0 for x in pr)
    assert d["config"]["rank0_scatter_gather"]["records_checked"] == 48
    assert {"bound", "achieved", "peak", "frac", "traffic"} <= set(d["roofline"])


def test_recorded_traffic_names_the_latest_profile_of_the_same_workload():
    """`roofline.traffic` replays the PMC summary under profiles/ whose name sorts last for exactly this workload: records of
    attempts that were not kept (another kernel) carry another workload tag and must not be picked up."""
    for wl, kern in (("10k-cigar", "k2a_fill_pk_kernel"), ("exts", "k2a_exts_kernel"), ("extf-w900", "k2a_extf_grp_kernel"), ("10k-ssec", "k2a_ssec_blk_kernel")):
        traffic, src = bench.recorded_traffic(wl)
        assert traffic and src and src.endswith("_%s_pmc.json" % wl), (wl, src)
        import json
        assert kern in json.load(open(os.path.join(bench.ROOT, src)))["derived"]["dominant_kernel"], (wl, src)
    assert bench.recorded_traffic("no-such-workload") == (None, None)
