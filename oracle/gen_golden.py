#!/usr/bin/env python3
"""Generate tests/golden/* from the UNMODIFIED reference compiled into oracle/_ref/libksw2ref.so.

Run in the build container only (needs /root/reference):   python oracle/gen_golden.py
Outputs (committed, data only -- inputs and the reference's outputs):
  tests/golden/data/{t1,q1,MT-human,MT-orang}.fa   test inputs shipped by the reference (MIT, see NOTICE there)
  tests/golden/known_answers.json                   SURVEY.md section 4.2 table, regenerated (ksw2-test CLI defaults)
  tests/golden/random_cases.npz                     seeded random cases: sequences, parameters, reference outputs
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po          # noqa: E402
from ksw2_amd import synth                  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
NT4 = np.full(256, 4, dtype=np.uint8)
for ch, v in zip("ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
    NT4[ord(ch)] = v


def read_fasta(path):
    names, seqs, cur = [], [], []
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            if names:
                seqs.append("".join(cur))
            names.append(line[1:].split()[0])
            cur = []
        elif line:
            cur.append(line)
    seqs.append("".join(cur))
    return names, seqs


def encode(s):
    return NT4[np.frombuffer(s.encode(), dtype=np.uint8)]


FIELDS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]


def pack(res):
    d = {k: res[k] for k in FIELDS}
    d["cigar"] = po.cigar_string(res["cigar"])
    d["cigar_md5_12"] = hashlib.md5((d["cigar"] + "\n").encode()).hexdigest()[:12]
    return d


def known_answers():
    os.makedirs(os.path.join(GOLD, "data"), exist_ok=True)
    for f in ("t1.fa", "q1.fa", "MT-human.fa", "MT-orang.fa"):
        shutil.copy(os.path.join(REF, "test", f), os.path.join(GOLD, "data", f))
    with open(os.path.join(GOLD, "data", "NOTICE"), "w") as fh:
        fh.write("t1.fa q1.fa MT-human.fa MT-orang.fa t2.fa.gz q2.fa.gz: test inputs distributed with lh3/ksw2 (test/), MIT licence,\n"
                 "Copyright (c) 2018- Dana-Farber Cancer Institute, 2017-2018 Broad Institute, Inc.  Data only.\n")
    out = {"scoring": {"a": 2, "b": 4, "q": 4, "e": 2, "q2": 13, "e2": 1, "sc_n": 0},
           "note": "ksw2-test defaults (cli.c:162): mat = 5x5 a=2 b=-4 N=0; -O4,13 -E2,1; w=-1 zdrop=-1",
           "t1q1": [], "mt": []}
    mat = po.simple_mat(5, 2, 4, 0)
    tn, ts = read_fasta(os.path.join(REF, "test", "t1.fa"))
    qn, qs = read_fasta(os.path.join(REF, "test", "q1.fa"))
    for k in range(len(tn)):
        t, q = encode(ts[k]), encode(qs[k])
        rec = {"tname": tn[k], "qname": qn[k]}
        for func in ("extz", "extd", "extz2", "extd2"):
            for flag in (0, po.RIGHT):
                key = {"extz": "ksw_extz", "extd": "ksw_extd", "extz2": "ksw_extz2_sse", "extd2": "ksw_extd2_sse"}[func]
                rec["%s/flag=%d" % (key, flag)] = pack(po.align("ref", func, q, t, mat, 4, 2, 13, 1, flag=flag))
        # the t1.fa:9 regression setting: -A1 -B9 -O16 -E1 -w10 (scalar is memory-unsafe there when tlen-1-w > qlen: F6)
        mat19 = po.simple_mat(5, 1, 9, 0)
        if abs(len(t) - len(q)) <= 10:
            rec["ksw_extz/A1B9O16E1w10"] = pack(po.align("ref", "extz", q, t, mat19, 16, 1, w=10))
        rec["ksw_extz2_sse/A1B9O16E1w10"] = pack(po.align("ref", "extz2", q, t, mat19, 16, 1, w=10))
        for g in ("gg", "gg2", "gg2_sse"):
            sc, cg = po.global_align("ref", g, q, t, mat, 4, 2, w=-1)
            rec["ksw_%s" % g] = {"score": sc, "cigar": po.cigar_string(cg)}
        out["t1q1"].append(rec)
    _, ts = read_fasta(os.path.join(REF, "test", "MT-human.fa"))
    _, qs = read_fasta(os.path.join(REF, "test", "MT-orang.fa"))
    t, q = encode(ts[0]), encode(qs[0])
    for func, w, flag, zdrop in [("extz", -1, 0, -1), ("extz", -1, po.RIGHT, -1), ("extd", -1, 0, -1), ("extd", -1, po.RIGHT, -1),
                                 ("extz", 500, 0, -1), ("extd", 500, 0, -1), ("extz", 500, 0, 100), ("extd", 500, 0, 400),
                                 ("extz", -1, po.SCORE_ONLY, -1), ("extz", -1, po.EXTZ_ONLY | po.REV_CIGAR, 400)]:
        r = pack(po.align("ref", func, q, t, mat, 4, 2, 13, 1, w=w, zdrop=zdrop, flag=flag))
        r.update({"func": "ksw_" + func, "w": w, "flag": flag, "zdrop": zdrop})
        out["mt"].append(r)
        print("MT", func, w, flag, zdrop, r["score"], r["max"], r["max_t"], r["max_q"], len(r["cigar"]), r["cigar_md5_12"])
    sc, cg = po.global_align("ref", "gg", q, t, mat, 4, 2, w=-1)
    out["mt"].append({"func": "ksw_gg", "w": -1, "score": sc, "cigar": po.cigar_string(cg),
                      "cigar_md5_12": hashlib.md5((po.cigar_string(cg) + "\n").encode()).hexdigest()[:12]})
    json.dump(out, open(os.path.join(GOLD, "known_answers.json"), "w"), indent=1)


def read_fasta_gz(path):
    import gzip
    seqs, cur = [], []
    for line in gzip.open(path, "rt"):
        line = line.strip()
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
                cur = []
        elif line:
            cur.append(line)
    seqs.append("".join(cur))
    return seqs


def t2q2_anchor():
    """The README's 50 000 x 50 000 data set (test/t2.fa.gz x test/q2.fa.gz, README.md:96-107; SURVEY 4.2: 69932 / 70010 / 49962 /
    49999): the two gzip files are copied as data, the reference's answers go into known_answers.json["t2q2_50k"]."""
    for f in ("t2.fa.gz", "q2.fa.gz"):
        shutil.copy(os.path.join(REF, "test", f), os.path.join(GOLD, "data", f))
        os.chmod(os.path.join(GOLD, "data", f), 0o644)
    t = encode(read_fasta_gz(os.path.join(REF, "test", "t2.fa.gz"))[0])
    q = encode(read_fasta_gz(os.path.join(REF, "test", "q2.fa.gz"))[0])
    mat = po.simple_mat(5, 2, 4, 0)
    out = []
    for func, flag, w, zd in [("extz", po.SCORE_ONLY, -1, -1), ("extz2", po.SCORE_ONLY, -1, -1), ("extz2", 0, -1, -1), ("extd2", po.SCORE_ONLY, -1, -1),
                              ("extz2", po.SCORE_ONLY, 500, 400)]:
        r = pack(po.align("ref", func, q, t, mat, 4, 2, 13, 1, w=w, zdrop=zd, flag=flag))
        if len(r["cigar"]) > 2000:
            r["cigar_len"] = len(r["cigar"])
            r["cigar"] = None                       # the md5 pins it
        r.update({"func": {"extz": "ksw_extz", "extz2": "ksw_extz2_sse", "extd2": "ksw_extd2_sse"}[func], "flag": flag, "w": w, "zdrop": zd})
        out.append(r)
        print("50k", func, flag, w, zd, r["score"], r["max"], r["max_t"], r["max_q"], r["n_cigar"], r["cigar_md5_12"])
    ka = json.load(open(os.path.join(GOLD, "known_answers.json")))
    ka["t2q2_50k"] = out
    json.dump(ka, open(os.path.join(GOLD, "known_answers.json"), "w"), indent=1)


MATS = [  # (a, b, sc_n, q, e, q2, e2)
    (2, 4, 0, 4, 2, 24, 1), (1, 9, 0, 4, 2, 24, 1), (1, 9, -1, 16, 2, 41, 1), (2, 4, -3, 4, 2, 13, 1), (2, 4, -1, 4, 2, 24, 1),
]
FLAGSETS = [0, po.SCORE_ONLY, po.RIGHT, po.EXTZ_ONLY | po.REV_CIGAR, po.SCORE_ONLY | po.RIGHT, po.EXTZ_ONLY | po.REV_CIGAR | po.RIGHT,
            po.EXTZ_ONLY]


def random_cases(n_cases=3600, seed=88172645463325252 % (2 ** 63)):
    """Scalar contract cases (ksw_extz / ksw_extd with the scoring matrix as given) and
    '...2_sse'-signature cases on loose bands (end_bonus / reach_end / implicit wildcard scoring)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    seqs, seq_off = [], [0]
    params, expect, cigs, cig_off = [], [], [], [0]
    FUNCS = ["extz", "extd", "extz2", "extd2"]
    it = 0
    while len(params) < n_cases:
        it += 1
        a, b, sc_n, q, e, q2, e2 = MATS[it % len(MATS)]
        mat = po.simple_mat(5, a, b, sc_n)
        func = FUNCS[it % 4] if it % 9 else FUNCS[2 + it % 2]
        flag = FLAGSETS[(it // 4) % len(FLAGSETS)]
        n_rate = 0.02 if it % 5 == 0 else 0.0
        big = it % 50 == 0
        qq, tt = synth.ragged_pairs(rng, 1, 1, 900 if big else 300, sub=0.03 + 0.12 * rng.random(), ind=0.25 * rng.random(),
                                    indel_mean=1.5 if it % 3 else 6.0, n_rate=n_rate)[0]
        if func in ("extz", "extd"):
            w = [-1, 1, 3, 5, 8, 10, 20, 64][(it // 28) % 8]
            zdrop = [-1, 30, 50, 100][(it // 7) % 4]
            end_bonus = 0
            if w >= 0 and abs(len(qq) - len(tt)) > w:
                continue                     # scalar reference is undefined there
        else:
            # SSE kernels are only comparable on loose bands without Z-drop events;
            # implicit wildcard scoring applies (no GENERIC_SC) on every other case
            w = [-1, 400][it % 2] if not big else -1
            zdrop = -1
            end_bonus = [0, 5, 20, 100][(it // 3) % 4]
            if it % 2:
                flag |= po.GENERIC_SC
            if (a, q) == (1, 16):
                continue
        res = po.align("ref", func, qq, tt, mat, q, e, q2, e2, w=w, zdrop=zdrop, end_bonus=end_bonus, flag=flag)
        seqs += [qq, tt]
        seq_off += [seq_off[-1] + len(qq), seq_off[-1] + len(qq) + len(tt)]
        params.append([FUNCS.index(func), a, b, sc_n, q, e, q2, e2, w, zdrop, end_bonus, flag])
        expect.append([res[k] for k in FIELDS])
        cigs.append(np.array(res["cigar"], dtype=np.uint32))
        cig_off.append(cig_off[-1] + len(res["cigar"]))
    np.savez_compressed(os.path.join(GOLD, "random_cases.npz"),
                        seq=np.concatenate(seqs), seq_off=np.array(seq_off, dtype=np.int64),
                        params=np.array(params, dtype=np.int32), expect=np.array(expect, dtype=np.int32),
                        cigar=np.concatenate(cigs) if cigs else np.zeros(0, np.uint32), cigar_off=np.array(cig_off, dtype=np.int64),
                        fields=np.array(FIELDS), param_names=np.array(["func", "a", "b", "sc_n", "q", "e", "q2", "e2", "w", "zdrop",
                                                                       "end_bonus", "flag"]),
                        funcs=np.array(["ksw_extz", "ksw_extd", "ksw_extz2_sse", "ksw_extd2_sse"]))
    print("random cases:", len(params), "bytes:", os.path.getsize(os.path.join(GOLD, "random_cases.npz")))


if __name__ == "__main__":
    if po.ref_lib() is None and not po.build_ref(REF):
        sys.exit("reference sources not available: golden vectors can only be regenerated in the build container")
    known_answers()
    t2q2_anchor()
    random_cases()
