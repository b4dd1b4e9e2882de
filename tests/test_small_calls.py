"""The opt-in host path for tiny single calls (ksw2amd_set_small_call_cells, ksw2_host_single.c: small_pair): the library's own scalar
code for single-pair calls below a cell count, on the calling thread.  It must return exactly what the kernels return: every
committed reference vector of the scalar functions, the "...2_sse" signatures' flag semantics (end bonus, EXTZ_ONLY, REV_CIGAR,
EQX, early rejects), the global functions, and random pairs against the oracle AND against the library's device path.

CPU tier: through the simulator build (tests/sim), whose host side is the product's own ksw2_host_*.c -- the code under test -- with
the lock-step simulator standing in for the device path it is compared with.  GPU tier: the same checks on libksw2_amd.so."""
import os
import subprocess

import numpy as np
import pytest

import ksw2_amd as ka
from ksw2_amd import synth
from oracle import pyoracle as po
from tests import golden_util as gu
from tests.parity_util import diff


SIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sim")


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-C", SIM_DIR], check=True, capture_output=True)
    L = ka.Library(os.path.join(SIM_DIR, "libksw2_amd_sim.so"))
    assert L.backend() == "sim"
    return L


@pytest.fixture(scope="module")
def lib():
    L = ka.library()                      # raises if the HIP library is missing: no fallback
    assert L.backend() == "hip:gfx950"
    return L


def _golden(lib):
    rc = gu.RandomCases()
    lib.set_small_call_cells(1 << 40)
    n0 = lib.small_call_count()
    n = 0
    try:
        for k in range(rc.n):
            c = rc.case(k)
            dual = "extd" in c["func"]
            if c["func"].endswith("2_sse"):
                r = (lib.extd2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"],
                               flag=c["flag"]) if dual else
                     lib.extz2(c["q"], c["t"], c["mat"], c["gq"], c["ge"], w=c["w"], zdrop=c["zdrop"], end_bonus=c["end_bonus"], flag=c["flag"]))
                assert not diff(c["expect"], r, gu.SSE_LOOSE_FIELDS), (k, c["func"], c["w"], c["zdrop"], hex(c["flag"]))
            else:
                r = (lib.extd(c["q"], c["t"], c["mat"], c["gq"], c["ge"], c["gq2"], c["ge2"], w=c["w"], zdrop=c["zdrop"], flag=c["flag"]) if dual else
                     lib.extz(c["q"], c["t"], c["mat"], c["gq"], c["ge"], w=c["w"], zdrop=c["zdrop"], flag=c["flag"]))
                assert not diff(c["expect"], r, gu.FIELDS + ["cigar"]), (k, c["func"], c["w"], c["zdrop"], hex(c["flag"]))
            n += 1
        assert n == rc.n and lib.small_call_count() - n0 >= n * 9 // 10          # (empty sequences are rejected before either path)
    finally:
        lib.set_small_call_cells(0)


def _random_against_both_paths(lib, rounds, seed):
    """random pairs, every flag combination the entry points take: small path == device path == oracle"""
    rng = np.random.Generator(np.random.PCG64(seed))
    scs = [(synth.simple_mat(5, 2, 4, -1), 4, 2, 24, 1), (synth.simple_mat(5, 1, 3, 0), 4, 1, 24, 1), (synth.simple_mat(5, 2, 4, -3), 4, 2, 13, 1),
           (synth.simple_mat(5, 2, 5, -1), 5, 3, 20, 2), (synth.simple_mat(5, 2, 4, -1), 24, 1, 4, 2)]
    fields = gu.FIELDS + ["cigar"]
    for rnd in range(rounds):
        mat, q, e, q2, e2 = scs[rnd % len(scs)]
        hi = [30, 120, 400][rnd % 3]
        (qs, ts), = synth.ragged_pairs(rng, 1, 1, hi, sub=0.06, ind=0.12, n_rate=0.02 if rnd % 5 == 0 else 0.0)
        if rnd % 7 == 0:
            ts = np.concatenate([ts, rng.integers(0, 4, int(rng.integers(1, 200))).astype(np.uint8)])
        w = int(rng.choice([-1, 0, 1, 3, 8, 20, 64, 100, 1000]))
        zd = int(rng.choice([-1, 10, 50, 400]))
        eb = int(rng.choice([0, 5, 50]))
        fl = int(rng.choice([0, po.SCORE_ONLY, po.RIGHT, po.RIGHT | po.SCORE_ONLY])) | (po.EXTZ_ONLY if rng.random() < 0.4 else 0) | \
            (po.REV_CIGAR if rng.random() < 0.3 else 0) | (po.GENERIC_SC if rng.random() < 0.3 else 0)
        dual = bool(rnd & 1)
        if dual and rng.random() < 0.3:
            fl |= po.EQX
        out = []
        for cells in (0, 1 << 40):
            lib.set_small_call_cells(cells)
            n0 = lib.small_call_count()
            try:
                r = (lib.extd2(qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl) if dual else
                     lib.extz2(qs, ts, mat, q, e, w=w, zdrop=zd, end_bonus=eb, flag=fl))
            finally:
                lib.set_small_call_cells(0)
            assert (lib.small_call_count() - n0) == (1 if cells else 0)
            out.append(r)
        assert not diff(out[0], out[1], fields), (rnd, dual, w, zd, eb, hex(fl), len(qs), len(ts))
        if not (fl & po.EQX):
            exp = po.align("oracle", "extd2" if dual else "extz2", qs, ts, mat, q, e, q2, e2, w=w, zdrop=zd, end_bonus=eb, flag=fl) if dual else \
                po.align("oracle", "extz2", qs, ts, mat, q, e, w=w, zdrop=zd, end_bonus=eb, flag=fl)
            assert not diff(exp, out[1], fields), (rnd, dual, w, zd, eb, hex(fl))


def _global_and_threshold(lib):
    ka_ = gu.known_answers()
    _, ts = gu.read_fasta("t1.fa")
    _, qs = gu.read_fasta("q1.fa")
    mat = gu.simple_mat(5, 2, 4, 0)
    lib.set_small_call_cells(1 << 40)
    try:
        for k, rec in enumerate(ka_["t1q1"]):
            n0 = lib.small_call_count()
            for g in ("gg", "gg2", "gg2_sse"):
                sc_, c = lib.gg(g, qs[k], ts[k], mat, 4, 2, w=-1)
                assert sc_ == rec["ksw_gg"]["score"] and gu.cigar_string(c) == rec["ksw_gg"]["cigar"], (k, g)
            sc_, _ = lib.gg("gg2", qs[k], ts[k], mat, 4, 2, w=-1, with_cigar=False)
            assert sc_ == rec["ksw_gg"]["score"] and lib.small_call_count() - n0 == 4
        # the threshold is a cell count of the exact band: 100 x 100, w = 10 has 1 990 cells
        q, t = synth.fixed_batch(5, 1, 100, 100, sub=0.05, ind=0.05)
        for cells, small in ((1989, 0), (1990, 1)):
            lib.set_small_call_cells(cells)
            n0 = lib.small_call_count()
            lib.extz2(q[0], t[0], mat, 4, 2, w=10, zdrop=-1, flag=0)
            assert lib.small_call_count() - n0 == small
        # approximate modes and the batch entry points never take it
        lib.set_small_call_cells(1 << 40)
        n0 = lib.small_call_count()
        lib.extz2(q[0], t[0], mat, 4, 2, w=10, zdrop=-1, flag=po.SCORE_ONLY | po.APPROX_MAX)
        lib.make_batch(list(q), list(t), mat, 4, 2, 24, 1, w=10, zdrop=-1, flag=0).run_oneshot(False)
        assert lib.small_call_count() == n0
    finally:
        lib.set_small_call_cells(0)


def test_small_calls_golden_vectors(sim):
    _golden(sim)


def test_small_calls_random_pairs_both_paths_and_oracle(sim):
    _random_against_both_paths(sim, 400, 31)


def test_small_calls_global_functions_threshold_and_exclusions(sim):
    _global_and_threshold(sim)


@pytest.mark.gpu
def test_small_calls_on_the_gpu_library(lib):
    _golden(lib)
    _random_against_both_paths(lib, 250, 32)
    _global_and_threshold(lib)
