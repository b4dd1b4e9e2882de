"""CPU: the host-side machinery around the kernels -- the chunked worker pool behind the batch entry points (one device and
several), the coalescing of concurrent single-pair calls, the error handler -- on the simulator build of the product's own
ksw2_host_*.c, against the oracle.  Results must not depend on how a batch is cut or on who runs it."""
import ctypes
import os
import subprocess
import threading

import numpy as np
import pytest

import ksw2_amd as ka
from ksw2_amd import synth
from oracle import pyoracle as po
from tests.parity_util import check_batch, diff, CMP_FIELDS

SIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sim")


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", SIM_DIR], check=True, capture_output=True)
    L = ka.Library(os.path.join(SIM_DIR, "libksw2_amd_sim.so"))
    assert L.backend() == "sim"
    return L


def _ragged(seed, n, hi=260):
    rng = np.random.Generator(np.random.PCG64(seed))
    pairs = synth.ragged_pairs(rng, n, 1, hi, sub=0.05, ind=0.12, n_rate=0.01)
    qs, ts = [p[0] for p in pairs], [p[1] for p in pairs]
    w = rng.choice([-1, 3, 20, 64, 100], size=n)
    zd = rng.choice([-1, 50, 200], size=n)
    fl = np.array([rng.choice([0, po.SCORE_ONLY, po.RIGHT]) | (po.EXTZ_ONLY if rng.random() < 0.3 else 0) | (po.REV_CIGAR if rng.random() < 0.3 else 0)
                   for _ in range(n)])
    return qs, ts, w, zd, fl


@pytest.mark.parametrize("dual", [False, True])
def test_pooled_batch_equals_oracle(sim, dual, monkeypatch):
    """A batch cut into chunks for the worker threads (forced by KSW2AMD_POOL_MIN) gives every pair the oracle's result."""
    monkeypatch.setenv("KSW2AMD_POOL_MIN", "16")
    monkeypatch.setenv("KSW2AMD_THREADS", "3")
    qs, ts, w, zd, fl = _ragged(11 + dual, 120)
    mat = synth.simple_mat(5, 2, 4, -1)
    s0 = sim.host_stats()
    n, _ = check_batch(sim, dual, qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=5, flag=fl)
    s1 = sim.host_stats()
    assert n == 120
    assert s1["pool_batches"] == s0["pool_batches"] + 1 and s1["pool_chunks"] >= s0["pool_chunks"] + 3


def test_pooled_batch_several_devices(sim, monkeypatch):
    """ksw2amd_set_devices: the same counter shards the chunks over the workers of several (here: simulated) devices."""
    monkeypatch.setenv("KSW2AMD_SIM_DEVICES", "4")
    monkeypatch.setenv("KSW2AMD_POOL_MIN", "16")
    monkeypatch.setenv("KSW2AMD_THREADS", "2")
    assert sim.device_count() == 4
    sim.set_devices([0, 1, 2, 3])
    try:
        qs, ts, w, zd, fl = _ragged(21, 160)
        mat = synth.simple_mat(5, 2, 4, -1)
        s0 = sim.host_stats()
        n, _ = check_batch(sim, True, qs, ts, mat, 4, 2, 24, 1, w=w, zdrop=zd, flag=fl)
        assert n == 160
        assert sim.host_stats()["pool_chunks"] >= s0["pool_chunks"] + 8
        with pytest.raises(ka.Ksw2Error):
            sim.set_devices([0, 7])
    finally:
        sim.set_devices([])
        sim.release_cache()


@pytest.mark.parametrize("score_only,expect_chunks", [(True, 4), (False, 4)])
def test_uniform_batch_is_cut_at_device_fills(sim, monkeypatch, score_only, expect_chunks):
    """A batch of one shape is cut into chunks that are multiples of a device fill (ksw2_host_pool.c::uniform_chunks); here a
    simulated device of 4 SIMDs: (8 lanes x 18 rows) / (16 x 8) geometry -> 64 / 32 pairs per unit, doubled until three workers
    have at most two chunks each -> four chunks of 1 024 pairs."""
    monkeypatch.setenv("KSW2AMD_SIM_SIMDS", "4")
    monkeypatch.setenv("KSW2AMD_SIMDS", "0")
    monkeypatch.setenv("KSW2AMD_THREADS", "3")
    monkeypatch.setenv("KSW2AMD_STREAM", "0")            # the chunked path is what this test is about (a streamed batch is one plan)
    n = 4096
    qs, ts = synth.fixed_batch(77, n, 1000, 1000, sub=0.05, ind=0.1)
    mat = synth.simple_mat(5, 2, 4, -1)
    fl = po.SCORE_ONLY if score_only else 0
    s0 = sim.host_stats()
    m, _ = check_batch(sim, False, list(qs), list(ts), mat, 4, 2, 24, 1, w=8, zdrop=-1, flag=fl)
    s1 = sim.host_stats()
    assert m == n
    assert s1["pool_batches"] == s0["pool_batches"] + 1
    assert s1["pool_chunks"] == s0["pool_chunks"] + expect_chunks


def test_big_plan_copies_its_sequences_on_the_pool(sim, monkeypatch):
    """A plan of 32 MB or more created on the caller's thread has the pool's threads share the sequence copy
    (ksw2_host_pool.c::parallel_copy); every pair must still see its own bytes (a wildcard in one pair only, narrow band)."""
    monkeypatch.setenv("KSW2AMD_THREADS", "3")
    n, L = 3400, 5000
    qs, ts = synth.fixed_batch(91, n, L, L, sub=0.03, ind=0.02)
    qs, ts = [np.array(q) for q in qs], [np.array(t) for t in ts]
    qs[1234][100] = 4                                   # a wildcard: that pair alone leaves the packed class
    mat = synth.simple_mat(5, 2, 4, -1)
    b = sim.make_batch(qs, ts, mat, 4, 2, 24, 1, w=1, zdrop=-1, end_bonus=0, flag=po.SCORE_ONLY)
    s0 = sim.host_stats()
    plan = b.plan(False)
    plan.run()
    res = plan.fetch()
    plan.close()
    assert sim.host_stats()["pool_batches"] == s0["pool_batches"]          # the copy job is not a batch
    for i in list(range(0, n, 97)) + [1233, 1234, 1235, n - 1]:
        exp = po.align("oracle", "extz", qs[i], ts[i], mat, 4, 2, 24, 1, w=1, zdrop=-1, end_bonus=0, flag=po.SCORE_ONLY)
        assert not diff(res[i], exp), i


def test_pool_off_and_inline_paths_agree(sim, monkeypatch):
    qs, ts, w, zd, fl = _ragged(31, 60)
    mat = synth.simple_mat(5, 2, 4, -1)
    monkeypatch.setenv("KSW2AMD_THREADS", "0")
    a = sim.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, flag=fl)
    monkeypatch.setenv("KSW2AMD_THREADS", "4")
    monkeypatch.setenv("KSW2AMD_POOL_MIN", "8")
    b = sim.extz_batch(qs, ts, mat, 4, 2, w=w, zdrop=zd, flag=fl)
    assert all(not diff(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("env", [{}, {"KSW2AMD_COALESCE_CROWD": "1", "KSW2AMD_COALESCE_WINDOW_US": "3000"},
                                 {"KSW2AMD_COALESCE_CROWD": "1", "KSW2AMD_COALESCE_WINDOW_US": "500", "KSW2AMD_COALESCE_SLOTS": "1"},
                                 {"KSW2AMD_COALESCE_CROWD": "1", "KSW2AMD_COALESCE_WINDOW_US": "20000", "KSW2AMD_COALESCE_SLOTS": "2"}],
                         ids=["default", "crowd", "crowd-1slot", "crowd-2slots-long-window"])
def test_concurrent_single_calls_coalesce(sim, env, monkeypatch):
    """Many host threads inside ksw_extz2_sse / ksw_extd2_sse at once (a minimap2-style pool): the library batches them;
    every caller still gets exactly its own result, CIGAR included."""
    nthreads, per = 12, 25
    for k, v in env.items():                    # the collection window of a crowd (a leader waits for the callers the last batches held)
        monkeypatch.setenv(k, v)
    sim.lib.ksw2amd_reload_env()
    mat = synth.simple_mat(5, 2, 4, -1)
    work = []
    for t in range(nthreads):
        qs, ts, w, zd, fl = _ragged(100 + t, per, hi=150)
        work.append((qs, ts, w, zd, fl))
    out = [[None] * per for _ in range(nthreads)]
    start = threading.Barrier(nthreads)

    def run(t):
        qs, ts, w, zd, fl = work[t]
        start.wait()
        for i in range(per):
            if (t + i) % 2:
                out[t][i] = sim.extd2(qs[i], ts[i], mat, 4, 2, 24, 1, w=int(w[i]), zdrop=int(zd[i]), end_bonus=7, flag=int(fl[i]))
            else:
                out[t][i] = sim.extz2(qs[i], ts[i], mat, 4, 2, w=int(w[i]), zdrop=int(zd[i]), end_bonus=7, flag=int(fl[i]))

    s0 = sim.host_stats()
    th = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    s1 = sim.host_stats()
    assert s1["coalesced_calls"] > s0["coalesced_calls"] and s1["coalesced_batches"] - s0["coalesced_batches"] < s1["coalesced_calls"] - s0["coalesced_calls"]
    for t in range(nthreads):
        qs, ts, w, zd, fl = work[t]
        for i in range(per):
            dual = bool((t + i) % 2)
            exp = po.align("oracle", "extd2" if dual else "extz2", qs[i], ts[i], mat, 4, 2, 24, 1, w=int(w[i]), zdrop=int(zd[i]), end_bonus=7, flag=int(fl[i]))
            assert not diff(exp, out[t][i], CMP_FIELDS), (t, i)


def test_failing_call_returns_reset_ez_without_a_handler(sim, capfd):
    """No handler installed: the failing ksw2-named call does not abort -- it resets *ez, counts the failure, keeps the message and
    says so on stderr."""
    before = sim.error_count()
    ez = ka.KswExtz()
    ez.score = 123
    mat = synth.simple_mat(5, 2, 4, -1)
    sim.lib.ksw_extd2_sse(None, 5, None, 5, None, 5, mat.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), 4, 2, 24, 1, -1, -1, 0, 0, ctypes.byref(ez))
    assert sim.error_count() == before + 1 and ez.score == ka.KSW_NEG_INF and ez.n_cigar == 0
    assert "NULL" in sim.last_error()
    assert "ksw_extd2_sse failed" in capfd.readouterr().err


def test_error_handler_replaces_abort(sim, monkeypatch):
    """A failing ksw2-named call (here: a scoring matrix with more than 127 residue types is impossible through int8_t m, so
    an invalid device is used instead) calls the installed handler and returns with ez reset instead of aborting."""
    seen = []
    sim.set_error_handler(lambda f, c, m: seen.append((f, c, m)))
    try:
        before = sim.error_count()
        # NULL sequence pointers with positive lengths: plan_create refuses (KSW2AMD_E_PARAM)
        ez = ka.KswExtz()
        mat = synth.simple_mat(5, 2, 4, -1)
        sim.lib.ksw_extz2_sse(None, 5, None, 5, None, 5, mat.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), 4, 2, -1, -1, 0, 0, ctypes.byref(ez))
        assert sim.error_count() == before + 1
        assert seen and seen[0][0] == "ksw_extz2_sse" and seen[0][1] == -2 and "NULL" in seen[0][2]
        assert ez.score == ka.KSW_NEG_INF and ez.n_cigar == 0
    finally:
        sim.set_error_handler(None)
