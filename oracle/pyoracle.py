"""ctypes access to the test oracle (oracle/libksw2_oracle.so) and, when it has been built, to the
unmodified reference (oracle/_ref/libksw2ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg -- never by the product package ksw2_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libksw2_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libksw2ref.so")

NEG_INF = -0x40000000
SCORE_ONLY, RIGHT, GENERIC_SC, APPROX_MAX, APPROX_DROP = 0x01, 0x02, 0x04, 0x08, 0x10
EXTZ_ONLY, REV_CIGAR, EQX = 0x40, 0x80, 0x800


class Ez(ctypes.Structure):
    """ksw_extz_t (ksw2.h:33-42): word0 = max:31 | zdropped<<31, cigar pointer at offset 48."""
    _fields_ = [("max_zd", ctypes.c_uint32), ("max_q", ctypes.c_int), ("max_t", ctypes.c_int),
                ("mqe", ctypes.c_int), ("mqe_t", ctypes.c_int), ("mte", ctypes.c_int), ("mte_q", ctypes.c_int),
                ("score", ctypes.c_int), ("m_cigar", ctypes.c_int), ("n_cigar", ctypes.c_int),
                ("reach_end", ctypes.c_int), ("cigar", ctypes.POINTER(ctypes.c_uint32))]


assert ctypes.sizeof(Ez) == 56 and Ez.cigar.offset == 48

_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]

_u8p = ctypes.POINTER(ctypes.c_uint8)
_i8p = ctypes.POINTER(ctypes.c_int8)
_i8 = ctypes.c_int8
_int = ctypes.c_int


def build_oracle():
    subprocess.run(["make", "-C", HERE, "libksw2_oracle.so"], check=True, capture_output=True)


def build_ref(ref_dir="/root/reference"):
    """Compile the reference where its sources lie (only possible in the build container)."""
    if not os.path.isdir(ref_dir):
        return False
    subprocess.run(["make", "-C", HERE, "ref", "REF=" + ref_dir], check=True, capture_output=True)
    return True


_cache = {}


def oracle_lib():
    if "o" not in _cache:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        lib = ctypes.CDLL(ORACLE_SO)
        lib.kso_extz.argtypes = [_int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, _int, _int, ctypes.POINTER(Ez)]
        lib.kso_extd.argtypes = [_int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _int, _int, ctypes.POINTER(Ez)]
        lib.kso_extz2.argtypes = [_int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, _int, _int, _int, ctypes.POINTER(Ez)]
        lib.kso_extd2.argtypes = [_int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _int, _int, _int, ctypes.POINTER(Ez)]
        lib.kso_extz2_sse.argtypes = lib.kso_extz2.argtypes
        lib.kso_extd2_sse.argtypes = lib.kso_extd2.argtypes
        gg = [_int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, ctypes.POINTER(_int), ctypes.POINTER(_int),
              ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
        lib.kso_gg.argtypes = gg
        lib.kso_gg.restype = _int
        lib.kso_gg2.argtypes = gg
        lib.kso_gg2.restype = _int
        lib.kso_band_cells.argtypes = [_int, _int, _int]
        lib.kso_band_cells.restype = ctypes.c_int64
        lib.kso_exts2.argtypes = [_int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _i8, _int, _u8p, ctypes.POINTER(Ez)]
        lib.kso_extf2.argtypes = [_int, _u8p, _int, _u8p, _i8, _i8, _i8, _int, _int, ctypes.POINTER(Ez)]
        lib.kso_long_thres.argtypes = [_int, _int, _int]
        lib.kso_long_thres.restype = _int
        _cache["o"] = lib
    return _cache["o"]


def ref_lib():
    """The compiled reference, or None when oracle/_ref has not been built (e.g. fresh clone)."""
    if "r" not in _cache:
        lib = None
        if os.path.exists(REF_SO):
            lib = ctypes.CDLL(REF_SO)
            km = ctypes.c_void_p
            lib.ksw_extz.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, _int, _int, ctypes.POINTER(Ez)]
            lib.ksw_extd.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _int, _int, ctypes.POINTER(Ez)]
            lib.ksw_extz2_sse.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, _int, _int, _int, ctypes.POINTER(Ez)]
            lib.ksw_extd2_sse.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _int, _int, _int, ctypes.POINTER(Ez)]
            gg = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, ctypes.POINTER(_int), ctypes.POINTER(_int),
                  ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
            if hasattr(lib, "ksw_exts2_sse"):
                lib.ksw_exts2_sse.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _i8, _int, _u8p, ctypes.POINTER(Ez)]
            if hasattr(lib, "ksw_extf2_sse"):
                lib.ksw_extf2_sse.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8, _i8, _int, _int, ctypes.POINTER(Ez)]
            for name in ("ksw_gg", "ksw_gg2", "ksw_gg2_sse"):
                getattr(lib, name).argtypes = gg
                getattr(lib, name).restype = _int
        _cache["r"] = lib
    return _cache["r"]


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _ez_to_dict(ez, free_cigar=True):
    n = ez.n_cigar
    cig = [int(ez.cigar[i]) for i in range(n)] if n > 0 else []
    d = dict(max=int(ez.max_zd & 0x7fffffff), zdropped=int(ez.max_zd >> 31), max_q=ez.max_q, max_t=ez.max_t,
             mqe=ez.mqe, mqe_t=ez.mqe_t, mte=ez.mte, mte_q=ez.mte_q, score=ez.score, reach_end=ez.reach_end,
             n_cigar=n, cigar=cig)
    if free_cigar and ez.cigar:
        _libc.free(ctypes.cast(ez.cigar, ctypes.c_void_p))
    return d


def simple_mat(m=5, a=2, b=4, sc_n=0):
    """m x m matrix: +a on the diagonal, -b elsewhere, last row/column (wildcard) = sc_n (cli.c:36-48)."""
    mat = np.full((m, m), -abs(b), dtype=np.int8)
    np.fill_diagonal(mat, abs(a))
    mat[m - 1, :] = sc_n
    mat[:, m - 1] = sc_n
    return mat.reshape(-1).copy()


def align(which, func, query, target, mat, q, e, q2=None, e2=None, w=-1, zdrop=-1, end_bonus=0, flag=0, m=None):
    """Run one alignment through the oracle (which='oracle') or the compiled reference (which='ref').

    func: 'extz' | 'extd' | 'extz2' | 'extd2'  (ref: ksw_extz / ksw_extd / ksw_extz2_sse / ksw_extd2_sse)
    Returns a dict with every ksw_extz_t field and the CIGAR words as a list.
    """
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    if m is None:
        m = int(round(len(mat) ** 0.5))
    ez = Ez()
    matp = mat.ctypes.data_as(_i8p)
    if which == "oracle":
        lib = oracle_lib()
        if func == "extz":
            lib.kso_extz(len(query), _p8(query), len(target), _p8(target), m, matp, q, e, w, zdrop, flag, ez)
        elif func == "extd":
            lib.kso_extd(len(query), _p8(query), len(target), _p8(target), m, matp, q, e, q2, e2, w, zdrop, flag, ez)
        elif func == "extz2":
            lib.kso_extz2(len(query), _p8(query), len(target), _p8(target), m, matp, q, e, w, zdrop, end_bonus, flag, ez)
        elif func == "extd2":
            lib.kso_extd2(len(query), _p8(query), len(target), _p8(target), m, matp, q, e, q2, e2, w, zdrop, end_bonus, flag, ez)
        elif func == "extz2_sse":                    # the SSE kernels as they are (leaky band, anti-diagonal Z-drop, APPROX modes)
            lib.kso_extz2_sse(len(query), _p8(query), len(target), _p8(target), m, matp, q, e, w, zdrop, end_bonus, flag, ez)
        elif func == "extd2_sse":
            lib.kso_extd2_sse(len(query), _p8(query), len(target), _p8(target), m, matp, q, e, q2, e2, w, zdrop, end_bonus, flag, ez)
        else:
            raise ValueError(func)
    else:
        lib = ref_lib()
        if lib is None:
            raise RuntimeError("oracle/_ref/libksw2ref.so not built (make -C oracle ref)")
        if func == "extz":
            lib.ksw_extz(None, len(query), _p8(query), len(target), _p8(target), m, matp, q, e, w, zdrop, flag, ez)
        elif func == "extd":
            lib.ksw_extd(None, len(query), _p8(query), len(target), _p8(target), m, matp, q, e, q2, e2, w, zdrop, flag, ez)
        elif func in ("extz2", "extz2_sse"):
            lib.ksw_extz2_sse(None, len(query), _p8(query), len(target), _p8(target), m, matp, q, e, w, zdrop, end_bonus, flag, ez)
        elif func in ("extd2", "extd2_sse"):
            lib.ksw_extd2_sse(None, len(query), _p8(query), len(target), _p8(target), m, matp, q, e, q2, e2, w, zdrop, end_bonus, flag, ez)
        else:
            raise ValueError(func)
    return _ez_to_dict(ez)


def global_align(which, func, query, target, mat, q, e, w=-1, with_cigar=True, m=None):
    """func: 'gg' | 'gg2' | 'gg2_sse' (oracle: gg2_sse == gg2).  Returns (score, cigar list)."""
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    if m is None:
        m = int(round(len(mat) ** 0.5))
    matp = mat.ctypes.data_as(_i8p)
    mc, nc = _int(0), _int(0)
    cig = ctypes.POINTER(ctypes.c_uint32)()
    if which == "oracle":
        f = getattr(oracle_lib(), "kso_gg" if func == "gg" else "kso_gg2")
        args = (len(query), _p8(query), len(target), _p8(target), m, matp, q, e, w)
    else:
        f = getattr(ref_lib(), "ksw_" + func)
        args = (None, len(query), _p8(query), len(target), _p8(target), m, matp, q, e, w)
    if with_cigar:
        score = f(*args, ctypes.byref(mc), ctypes.byref(nc), ctypes.byref(cig))
    else:
        score = f(*args, None, None, None)
    out = [int(cig[i]) for i in range(nc.value)]
    if cig:
        _libc.free(ctypes.cast(cig, ctypes.c_void_p))
    return int(score), out


def band_cells(qlen, tlen, w):
    return int(oracle_lib().kso_band_cells(qlen, tlen, w))


def cigar_string(cigar):
    return "".join("%d%s" % (c >> 4, "MIDN===X"[c & 0xf] if (c & 0xf) < 4 else {7: "=", 8: "X"}[c & 0xf]) for c in cigar)


SPLICE_FOR, SPLICE_REV, SPLICE_FLANK = 0x100, 0x200, 0x400


def exts2(which, query, target, mat, q, e, q2, noncan, zdrop=-1, junc_bonus=0, flag=0, junc=None, m=None):
    """ksw_exts2_sse (which='ref') or its restatement kso_exts2 (which='oracle'); dict of ksw_extz_t fields + CIGAR."""
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    mat = np.ascontiguousarray(mat, dtype=np.int8)
    if m is None:
        m = int(round(len(mat) ** 0.5))
    jp = None
    if junc is not None:
        junc = np.ascontiguousarray(junc, dtype=np.uint8)
        jp = _p8(junc)
    ez = Ez()
    matp = mat.ctypes.data_as(_i8p)
    if which == "oracle":
        oracle_lib().kso_exts2(len(query), _p8(query), len(target), _p8(target), m, matp, q, e, q2, noncan, zdrop, junc_bonus, flag, jp, ez)
    else:
        lib = ref_lib()
        if lib is None or not hasattr(lib, "ksw_exts2_sse"):
            raise RuntimeError("oracle/_ref/libksw2ref.so not built (make -C oracle ref)")
        lib.ksw_exts2_sse(None, len(query), _p8(query), len(target), _p8(target), m, matp, q, e, q2, noncan, zdrop, junc_bonus, flag, jp, ez)
    return _ez_to_dict(ez)


def extf2(which, query, target, mch, mis, e, w=-1, xdrop=-1):
    """ksw_extf2_sse (which='ref') or its restatement kso_extf2 (which='oracle'); dict of ksw_extz_t fields."""
    query = np.ascontiguousarray(query, dtype=np.uint8)
    target = np.ascontiguousarray(target, dtype=np.uint8)
    ez = Ez()
    if which == "oracle":
        oracle_lib().kso_extf2(len(query), _p8(query), len(target), _p8(target), mch, mis, e, w, xdrop, ez)
    else:
        lib = ref_lib()
        if lib is None or not hasattr(lib, "ksw_extf2_sse"):
            raise RuntimeError("oracle/_ref/libksw2ref.so not built (make -C oracle ref)")
        lib.ksw_extf2_sse(None, len(query), _p8(query), len(target), _p8(target), mch, mis, e, w, xdrop, ez)
    return _ez_to_dict(ez)
