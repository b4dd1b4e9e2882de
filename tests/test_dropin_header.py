"""CPU: a caller compiled against the REFERENCE's own ksw2.h links and runs against our library unchanged (struct layout,
prototypes, flag values, CIGAR ownership).  Needs /root/reference at compile time only; linked against the simulator build."""
import os
import subprocess

import numpy as np
import pytest

from ksw2_amd import synth
from oracle import pyoracle as po
from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "ksw2.h")), reason="reference header not present")
@pytest.mark.parametrize("kalloc", [False, True])
def test_caller_built_against_reference_header(tmp_path, kalloc):
    """kalloc=True: the caller is built with -DHAVE_KALLOC plus the reference's kalloc.c (compiled where it lies) and passes
    its pool as `km`; the library finds the process's krealloc (executable linked with -rdynamic) and grows the CIGAR there."""
    sim_dir = os.path.join(ROOT, "tests", "sim")
    subprocess.run(["make", "-C", sim_dir], check=True, capture_output=True)
    exe = str(tmp_path / "caller")
    extra = ["-DHAVE_KALLOC", os.path.join(REF, "kalloc.c"), "-rdynamic"] if kalloc else []
    subprocess.run(["gcc", "-O1", "-Wall", "-I" + REF, os.path.join(ROOT, "tests", "dropin", "caller.c")] + extra + ["-o", exe,
                    "-L" + sim_dir, "-lksw2_amd_sim", "-Wl,-rpath," + sim_dir], check=True, capture_output=True)
    rng = np.random.Generator(np.random.PCG64(12))
    mat = synth.simple_mat(5, 2, 4, -1)
    lines, expect = [], []
    for k in range(40):
        (q, t), = synth.ragged_pairs(rng, 1, 5, 400, sub=0.05, ind=0.1, n_rate=0.01 if k % 4 == 0 else 0.0)
        algo = ["extz2", "extd2", "exts2", "gg2", "extf2"][k % 5]
        w = int(rng.choice([-1, 50, 500])) if algo != "gg2" else -1
        zd = int(rng.choice([-1, 100])) if algo != "gg2" else -1
        flag = int(rng.choice([0, po.RIGHT, po.EXTZ_ONLY, po.REV_CIGAR])) if algo != "gg2" else 0
        lines.append("%s %d %d %d %s %s" % (algo, w, zd, flag, "".join("ACGTN"[c] for c in q), "".join("ACGTN"[c] for c in t)))
        if algo == "extz2":
            e = po.align("oracle", "extz2", q, t, mat, 4, 2, w=w, zdrop=zd, end_bonus=10, flag=flag)
        elif algo == "extd2":
            e = po.align("oracle", "extd2", q, t, mat, 4, 2, 24, 1, w=w, zdrop=zd, end_bonus=10, flag=flag)
        elif algo == "extf2":
            e = po.extf2("oracle", q, t, 2, -4, 2, w, zd)
        elif algo == "exts2":
            e = po.exts2("oracle", q, t, mat, 4, 2, 32, 4, zdrop=zd, flag=flag | po.SPLICE_FOR)
        else:
            s, c = po.global_align("oracle", "gg2", q, t, mat, 4, 2, w=-1)
            e = dict(score=s, max=0, max_t=-1, max_q=-1, mqe=po.NEG_INF if hasattr(po, "NEG_INF") else -0x40000000, mqe_t=-1,
                     mte=-0x40000000, mte_q=-1, zdropped=0, reach_end=0, n_cigar=len(c), cigar=c)
        expect.append(e)
    out = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(out) == len(expect)
    for line, e, src in zip(out, expect, lines):
        v = [int(x) for x in line.split()]
        got = dict(zip(gu.FIELDS, v[:11]))
        got["cigar"] = v[11:]
        for f in gu.FIELDS + ["cigar"]:
            if src.startswith("gg2") and f in ("mqe", "mte"):
                continue
            assert got[f] == e[f], (src[:40], f, got[f], e[f])
