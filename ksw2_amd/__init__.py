"""ksw2_amd -- ctypes binding of libksw2_amd.so, the MI355X (gfx950) implementation of ksw2's banded
extension / global alignment hot path (C-ABI in include/ksw2_amd.h).

The binding mirrors the reference's C interface (ksw2.h:61-90): same function names, argument order and
meaning, same ksw_extz_t fields.  There is no CPU fallback: if the HIP library is missing or no GPU is
usable the calls raise / abort loudly.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_SO = os.path.join(HERE, "libksw2_amd.so")

KSW_NEG_INF = -0x40000000
KSW_EZ_SCORE_ONLY, KSW_EZ_RIGHT, KSW_EZ_GENERIC_SC, KSW_EZ_APPROX_MAX, KSW_EZ_APPROX_DROP = 0x01, 0x02, 0x04, 0x08, 0x10
KSW_EZ_EXTZ_ONLY, KSW_EZ_REV_CIGAR, KSW_EZ_EQX = 0x40, 0x80, 0x800
KSW2AMD_EZ_SSE_COMPAT = 0x20000000      # per-pair opt-in: the SSE kernels' own results (include/ksw2_amd.h)

FIELDS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]


class KswExtz(ctypes.Structure):
    """ksw_extz_t (ksw2.h:33-42)."""
    _fields_ = [("max_zd", ctypes.c_uint32), ("max_q", ctypes.c_int), ("max_t", ctypes.c_int),
                ("mqe", ctypes.c_int), ("mqe_t", ctypes.c_int), ("mte", ctypes.c_int), ("mte_q", ctypes.c_int),
                ("score", ctypes.c_int), ("m_cigar", ctypes.c_int), ("n_cigar", ctypes.c_int),
                ("reach_end", ctypes.c_int), ("cigar", ctypes.POINTER(ctypes.c_uint32))]


class Scoring(ctypes.Structure):
    _fields_ = [("m", ctypes.c_int32), ("mat", ctypes.POINTER(ctypes.c_int8)),
                ("q", ctypes.c_int8), ("e", ctypes.c_int8), ("q2", ctypes.c_int8), ("e2", ctypes.c_int8)]


class SpliceScoring(ctypes.Structure):
    _fields_ = [("m", ctypes.c_int32), ("mat", ctypes.POINTER(ctypes.c_int8)),
                ("q", ctypes.c_int8), ("e", ctypes.c_int8), ("q2", ctypes.c_int8), ("noncan", ctypes.c_int8), ("junc_bonus", ctypes.c_int8)]


class SplicePair(ctypes.Structure):
    _fields_ = [("query", ctypes.c_void_p), ("target", ctypes.c_void_p), ("junc", ctypes.c_void_p), ("qlen", ctypes.c_int32),
                ("tlen", ctypes.c_int32), ("zdrop", ctypes.c_int32), ("flag", ctypes.c_int32)]


class LinearPair(ctypes.Structure):
    """ksw2amd_fpair_t: the per-call arguments of ksw_extf2_sse."""
    _fields_ = [("query", ctypes.c_void_p), ("target", ctypes.c_void_p), ("qlen", ctypes.c_int32), ("tlen", ctypes.c_int32),
                ("w", ctypes.c_int32), ("xdrop", ctypes.c_int32)]


class Pair(ctypes.Structure):
    _fields_ = [("query", ctypes.c_void_p), ("target", ctypes.c_void_p), ("qlen", ctypes.c_int32), ("tlen", ctypes.c_int32),
                ("w", ctypes.c_int32), ("zdrop", ctypes.c_int32), ("end_bonus", ctypes.c_int32), ("flag", ctypes.c_int32)]


class Flat(ctypes.Structure):
    """ksw2amd_flat_t: one arena + offsets (include/ksw2_amd.h)."""
    _fields_ = [("base", ctypes.c_void_p), ("qoff", ctypes.c_void_p), ("toff", ctypes.c_void_p), ("qlen", ctypes.c_void_p), ("tlen", ctypes.c_void_p),
                ("w", ctypes.c_void_p), ("zdrop", ctypes.c_void_p), ("end_bonus", ctypes.c_void_p), ("flag", ctypes.c_void_p),
                ("w_all", ctypes.c_int32), ("zdrop_all", ctypes.c_int32), ("end_bonus_all", ctypes.c_int32), ("flag_all", ctypes.c_int32),
                ("on_device", ctypes.c_int32)]


_u8p = ctypes.POINTER(ctypes.c_uint8)
_i8p = ctypes.POINTER(ctypes.c_int8)
_i8 = ctypes.c_int8
_int = ctypes.c_int
_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]

EXPORTS = ["ksw_extz2_sse", "ksw_extd2_sse", "ksw_gg2", "ksw_gg2_sse", "ksw_extz", "ksw_extd", "ksw_gg",
           "ksw_extz2_sse41", "ksw_extz2_sse2", "ksw_extd2_sse41", "ksw_extd2_sse2",
           "ksw2amd_last_error", "ksw2amd_backend", "ksw2amd_device_count", "ksw2amd_set_device", "ksw2amd_release_cache",
           "ksw2amd_extz_batch", "ksw2amd_extd_batch", "ksw2amd_plan_create", "ksw2amd_plan_run", "ksw2amd_plan_fetch",
           "ksw2amd_plan_destroy", "ksw2amd_plan_timing", "ksw2amd_plan_cells", "ksw2amd_plan_device_bytes", "ksw2amd_plan_packed_pairs",
           "ksw2amd_plan_fetch_raw", "ksw_exts2_sse", "ksw_exts2_sse41", "ksw_exts2_sse2", "ksw2amd_exts_batch", "ksw2amd_exts_plan_create",
           "ksw_extf2_sse", "ksw2amd_extf_batch", "ksw2amd_extf_plan_create",
           "ksw2amd_set_devices", "ksw2amd_set_error_handler", "ksw2amd_error_count", "ksw2amd_host_stats",
           "ksw2amd_set_sse_compat", "ksw2amd_sse_plan_create", "ksw2amd_plan_describe", "ksw2amd_reload_env",
           "ksw2amd_extz_batch_flat", "ksw2amd_extd_batch_flat", "ksw2amd_plan_create_flat", "ksw2amd_host_register", "ksw2amd_host_unregister",
           "ksw2amd_device_alloc", "ksw2amd_device_free", "ksw2amd_device_upload", "ksw2amd_device_download", "ksw2amd_rerun_count",
           "ksw2amd_set_small_call_cells", "ksw2amd_small_call_count", "ksw2amd_stream_stats", "ksw2amd_host_phase_us", "ksw2amd_exts_batch_device", "ksw2amd_extf_batch_device"]
# entry points whose behaviour depends on KSW2AMD_* switches: the library reads its environment once per process, so this binding
# re-reads it in front of each of them (tests and A/B scripts flip switches inside one process)
_ENV_ENTRIES = ["ksw_extz2_sse", "ksw_extd2_sse", "ksw_gg2", "ksw_gg2_sse", "ksw_extz", "ksw_extd", "ksw_gg", "ksw_extz2_sse41",
                "ksw_extz2_sse2", "ksw_extd2_sse41", "ksw_extd2_sse2", "ksw_exts2_sse", "ksw_exts2_sse41", "ksw_exts2_sse2", "ksw_extf2_sse",
                "ksw2amd_extz_batch", "ksw2amd_extd_batch", "ksw2amd_exts_batch", "ksw2amd_extf_batch", "ksw2amd_exts_batch_device", "ksw2amd_extf_batch_device", "ksw2amd_plan_create",
                "ksw2amd_sse_plan_create", "ksw2amd_exts_plan_create", "ksw2amd_extf_plan_create", "ksw2amd_plan_run",
                "ksw2amd_plan_describe", "ksw2amd_extz_batch_flat", "ksw2amd_extd_batch_flat", "ksw2amd_plan_create_flat"]
ERROR_FN = ctypes.CFUNCTYPE(None, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p)
KSW_EZ_SPLICE_FOR, KSW_EZ_SPLICE_REV, KSW_EZ_SPLICE_FLANK = 0x100, 0x200, 0x400


class Ksw2Error(RuntimeError):
    pass


def ez_to_dict(ez, free_cigar=False):
    n = ez.n_cigar
    d = dict(max=int(ez.max_zd & 0x7fffffff), zdropped=int(ez.max_zd >> 31), max_q=ez.max_q, max_t=ez.max_t, mqe=ez.mqe,
             mqe_t=ez.mqe_t, mte=ez.mte, mte_q=ez.mte_q, score=ez.score, reach_end=ez.reach_end, n_cigar=n,
             m_cigar=ez.m_cigar, cigar=[int(ez.cigar[i]) for i in range(n)] if n > 0 else [])
    if free_cigar and ez.cigar:
        _libc.free(ctypes.cast(ez.cigar, ctypes.c_void_p))
    return d


def cigar_string(cigar):
    return "".join("%d%s" % (c >> 4, "MIDN"[c & 0xf] if (c & 0xf) < 4 else {7: "=", 8: "X"}[c & 0xf]) for c in cigar)


class Library:
    """One loaded libksw2_amd.so (tests/sim loads its simulator build through the same class)."""

    def __init__(self, path=DEFAULT_SO):
        if not os.path.exists(path):
            raise Ksw2Error("%s not found: build it with `make -C ksw2_amd/csrc` (or __graft_entry__.build()); "
                            "there is no CPU fallback" % path)
        self.path = path
        L = self.lib = ctypes.CDLL(path)
        km = ctypes.c_void_p
        ezp = ctypes.POINTER(KswExtz)
        z2 = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, _int, _int, _int, ezp]
        d2 = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _int, _int, _int, ezp]
        for name in ("ksw_extz2_sse", "ksw_extz2_sse41", "ksw_extz2_sse2"):
            getattr(L, name).argtypes = z2
            getattr(L, name).restype = None
        for name in ("ksw_extd2_sse", "ksw_extd2_sse41", "ksw_extd2_sse2"):
            getattr(L, name).argtypes = d2
            getattr(L, name).restype = None
        L.ksw_extz.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, _int, _int, ezp]
        L.ksw_extz.restype = None
        L.ksw_extd.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _int, _int, ezp]
        L.ksw_extd.restype = None
        gg = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _int, ctypes.POINTER(_int), ctypes.POINTER(_int),
              ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
        for name in ("ksw_gg", "ksw_gg2", "ksw_gg2_sse"):
            getattr(L, name).argtypes = gg
            getattr(L, name).restype = _int
        s2 = [km, _int, _u8p, _int, _u8p, _i8, _i8p, _i8, _i8, _i8, _i8, _int, _i8, _int, _u8p, ezp]
        for name in ("ksw_exts2_sse", "ksw_exts2_sse41", "ksw_exts2_sse2"):
            getattr(L, name).argtypes = s2
            getattr(L, name).restype = None
        L.ksw2amd_exts_batch.argtypes = [km, ctypes.POINTER(SpliceScoring), _int, ctypes.POINTER(SplicePair), ezp]
        L.ksw2amd_exts_batch_device.argtypes = [km, ctypes.POINTER(SpliceScoring), _int, ctypes.POINTER(SplicePair), ezp]
        L.ksw2amd_exts_plan_create.argtypes = [ctypes.POINTER(SpliceScoring), _int, ctypes.POINTER(SplicePair)]
        L.ksw2amd_exts_plan_create.restype = ctypes.c_void_p
        L.ksw_extf2_sse.argtypes = [km, _int, _u8p, _int, _u8p, _i8, _i8, _i8, _int, _int, ezp]
        L.ksw_extf2_sse.restype = None
        L.ksw2amd_extf_batch.argtypes = [km, _i8, _i8, _i8, _int, ctypes.POINTER(LinearPair), ezp]
        L.ksw2amd_extf_batch_device.argtypes = [km, _i8, _i8, _i8, _int, ctypes.POINTER(LinearPair), ezp]
        L.ksw2amd_extf_plan_create.argtypes = [_i8, _i8, _i8, _int, ctypes.POINTER(LinearPair)]
        L.ksw2amd_extf_plan_create.restype = ctypes.c_void_p
        L.ksw2amd_last_error.restype = ctypes.c_char_p
        L.ksw2amd_backend.restype = ctypes.c_char_p
        L.ksw2amd_set_device.argtypes = [_int]
        bt = [km, ctypes.POINTER(Scoring), _int, ctypes.POINTER(Pair), ezp]
        L.ksw2amd_extz_batch.argtypes = bt
        L.ksw2amd_extd_batch.argtypes = bt
        L.ksw2amd_plan_create.argtypes = [_int, ctypes.POINTER(Scoring), _int, ctypes.POINTER(Pair)]
        L.ksw2amd_plan_create.restype = ctypes.c_void_p
        L.ksw2amd_sse_plan_create.argtypes = [_int, ctypes.POINTER(Scoring), _int, ctypes.POINTER(Pair)]
        L.ksw2amd_sse_plan_create.restype = ctypes.c_void_p
        L.ksw2amd_set_sse_compat.argtypes = [_int]
        L.ksw2amd_set_sse_compat.restype = None
        L.ksw2amd_plan_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ksw2amd_plan_fetch.argtypes = [ctypes.c_void_p, km, ezp]
        L.ksw2amd_plan_destroy.argtypes = [ctypes.c_void_p]
        L.ksw2amd_plan_destroy.restype = None
        L.ksw2amd_plan_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        L.ksw2amd_plan_cells.argtypes = [ctypes.c_void_p]
        L.ksw2amd_plan_cells.restype = ctypes.c_int64
        L.ksw2amd_plan_device_bytes.argtypes = [ctypes.c_void_p]
        L.ksw2amd_plan_device_bytes.restype = ctypes.c_int64
        L.ksw2amd_plan_packed_pairs.argtypes = [ctypes.c_void_p]
        L.ksw2amd_plan_packed_pairs.restype = ctypes.c_int64
        L.ksw2amd_plan_fetch_raw.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
        L.ksw2amd_set_devices.argtypes = [_int, ctypes.POINTER(_int)]
        L.ksw2amd_set_error_handler.argtypes = [ERROR_FN, ctypes.c_void_p]
        L.ksw2amd_set_error_handler.restype = None
        L.ksw2amd_error_count.restype = ctypes.c_long
        L.ksw2amd_release_cache.restype = None
        L.ksw2amd_host_stats.argtypes = [ctypes.POINTER(ctypes.c_int64)]
        L.ksw2amd_host_stats.restype = None
        L.ksw2amd_host_phase_us.argtypes = [ctypes.POINTER(ctypes.c_int64)]
        L.ksw2amd_host_phase_us.restype = None
        L.ksw2amd_plan_describe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, _int]
        fl = [km, ctypes.POINTER(Scoring), _int, ctypes.POINTER(Flat), ezp]
        L.ksw2amd_extz_batch_flat.argtypes = fl
        L.ksw2amd_extd_batch_flat.argtypes = fl
        L.ksw2amd_plan_create_flat.argtypes = [_int, ctypes.POINTER(Scoring), _int, ctypes.POINTER(Flat)]
        L.ksw2amd_plan_create_flat.restype = ctypes.c_void_p
        L.ksw2amd_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.ksw2amd_host_unregister.argtypes = [ctypes.c_void_p]
        L.ksw2amd_device_alloc.argtypes = [ctypes.c_size_t]
        L.ksw2amd_device_alloc.restype = ctypes.c_void_p
        L.ksw2amd_device_free.argtypes = [ctypes.c_void_p]
        L.ksw2amd_device_free.restype = None
        L.ksw2amd_device_upload.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.ksw2amd_device_download.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.ksw2amd_rerun_count.restype = ctypes.c_int64
        L.ksw2amd_set_small_call_cells.argtypes = [ctypes.c_int64]
        L.ksw2amd_set_small_call_cells.restype = None
        L.ksw2amd_small_call_count.restype = ctypes.c_long
        L.ksw2amd_stream_stats.restype = None
        L.ksw2amd_reload_env.restype = None
        reload_env = L.ksw2amd_reload_env

        def with_env(fn):
            def call(*a):
                reload_env()
                return fn(*a)
            return call
        for name in _ENV_ENTRIES:
            setattr(L, name, with_env(getattr(L, name)))

    # ---- info
    def backend(self):
        return self.lib.ksw2amd_backend().decode()

    def device_count(self):
        return int(self.lib.ksw2amd_device_count())

    def set_device(self, dev):
        self._check(self.lib.ksw2amd_set_device(int(dev)))

    def last_error(self):
        return self.lib.ksw2amd_last_error().decode()

    def set_devices(self, devices):
        """ksw2amd_set_devices: the batch entry points shard over these devices ([] = the calling thread's device)."""
        arr = (_int * max(len(devices), 1))(*devices)
        self._check(self.lib.ksw2amd_set_devices(len(devices), arr))

    def set_error_handler(self, fn):
        """ksw2amd_set_error_handler: fn(func_name, code, message) instead of abort() when a ksw2-named call fails; None restores abort."""
        self._err_cb = ERROR_FN(lambda f, c, m, u: fn(f.decode(), c, m.decode())) if fn else ctypes.cast(None, ERROR_FN)
        self.lib.ksw2amd_set_error_handler(self._err_cb, None)

    def error_count(self):
        return int(self.lib.ksw2amd_error_count())

    def release_cache(self):
        self.lib.ksw2amd_release_cache()

    def set_sse_compat(self, on):
        """ksw2amd_set_sse_compat: process-wide, every extz2 / extd2 call returns what the reference's SSE kernels return."""
        self.lib.ksw2amd_set_sse_compat(1 if on else 0)

    def device_copy(self, array):
        """ksw2amd_device_alloc + ksw2amd_device_upload: a device copy of a numpy array; returns its address (free with device_free)."""
        a = np.ascontiguousarray(array)
        d = self.lib.ksw2amd_device_alloc(a.nbytes + 16)
        if not d:
            raise Ksw2Error("device_alloc failed: " + self.last_error())
        self._check(self.lib.ksw2amd_device_upload(d, a.ctypes.data, a.nbytes))
        return d

    def device_free(self, d):
        self.lib.ksw2amd_device_free(d)

    def set_small_call_cells(self, cells):
        """ksw2amd_set_small_call_cells: single calls of at most `cells` band cells run on the calling thread (0 = never, the default)."""
        self.lib.ksw2amd_set_small_call_cells(int(cells))

    def small_call_count(self):
        return int(self.lib.ksw2amd_small_call_count())

    def rerun_count(self):
        """ksw2amd_rerun_count: pairs that a fetch ran again through the ordinary kernels (flat wildcard pairs, deferred arg-max)."""
        return int(self.lib.ksw2amd_rerun_count())

    def stream_stats(self):
        """ksw2amd_stream_stats -> dict(streamed_plans, aborted_runs): plans whose batch ran as one launch started under its upload (wavefronts wait for their pieces)."""
        out = (ctypes.c_int64 * 2)()
        self.lib.ksw2amd_stream_stats(out)
        return dict(streamed_plans=int(out[0]), aborted_runs=int(out[1]))

    def host_phase_ms(self):
        """ksw2amd_host_phase_us -> dict(create_ms, launch_ms, wait_fetch_ms, plans): host-thread time of the batch entry points, summed over threads."""
        out = (ctypes.c_int64 * 4)()
        self.lib.ksw2amd_host_phase_us(out)
        return dict(create_ms=out[0] / 1000.0, launch_ms=out[1] / 1000.0, wait_fetch_ms=out[2] / 1000.0, plans=int(out[3]))

    def host_stats(self):
        """ksw2amd_host_stats -> dict(pool_batches, pool_chunks, coalesced_calls, coalesced_batches)."""
        out = (ctypes.c_int64 * 4)()
        self.lib.ksw2amd_host_stats(out)
        return dict(zip(("pool_batches", "pool_chunks", "coalesced_calls", "coalesced_batches"), (int(x) for x in out)))

    def _check(self, rc):
        if rc != 0:
            raise Ksw2Error("libksw2_amd error %d: %s" % (rc, self.last_error()))

    # ---- single-pair calls, signatures of ksw2.h:61-71
    @staticmethod
    def _seq(a):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        return a, a.ctypes.data_as(_u8p)

    def extz2(self, query, target, mat, q, e, w=-1, zdrop=-1, end_bonus=0, flag=0, m=None, ez=None):
        """ksw_extz2_sse(km=NULL, ...) -> dict of ksw_extz_t fields (+ CIGAR list)."""
        qa, qp = self._seq(query)
        ta, tp = self._seq(target)
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        m = int(round(len(mat) ** 0.5)) if m is None else m
        own = ez is None
        ez = KswExtz() if own else ez
        self.lib.ksw_extz2_sse(None, len(qa), qp, len(ta), tp, m, mat.ctypes.data_as(_i8p), q, e, w, zdrop, end_bonus, flag, ez)
        return ez_to_dict(ez, free_cigar=own)

    def extd2(self, query, target, mat, q, e, q2, e2, w=-1, zdrop=-1, end_bonus=0, flag=0, m=None, ez=None):
        qa, qp = self._seq(query)
        ta, tp = self._seq(target)
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        m = int(round(len(mat) ** 0.5)) if m is None else m
        own = ez is None
        ez = KswExtz() if own else ez
        self.lib.ksw_extd2_sse(None, len(qa), qp, len(ta), tp, m, mat.ctypes.data_as(_i8p), q, e, q2, e2, w, zdrop, end_bonus, flag, ez)
        return ez_to_dict(ez, free_cigar=own)

    def exts2(self, query, target, mat, q, e, q2, noncan, zdrop=-1, junc_bonus=0, flag=0, junc=None, m=None):
        """ksw_exts2_sse(km=NULL, ...): splice-aware extension -> dict of ksw_extz_t fields (+ CIGAR list)."""
        qa, qp = self._seq(query)
        ta, tp = self._seq(target)
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        m = int(round(len(mat) ** 0.5)) if m is None else m
        ja, jp = (None, None) if junc is None else self._seq(junc)
        ez = KswExtz()
        self.lib.ksw_exts2_sse(None, len(qa), qp, len(ta), tp, m, mat.ctypes.data_as(_i8p), q, e, q2, noncan, zdrop, junc_bonus, flag, jp, ez)
        return ez_to_dict(ez, free_cigar=True)

    def extf2(self, query, target, mch, mis, e, w=-1, xdrop=-1):
        """ksw_extf2_sse(km=NULL, ...): gap-linear X-drop extension, score only -> dict of ksw_extz_t fields."""
        qa, qp = self._seq(query)
        ta, tp = self._seq(target)
        ez = KswExtz()
        self.lib.ksw_extf2_sse(None, len(qa), qp, len(ta), tp, mch, mis, e, w, xdrop, ez)
        return ez_to_dict(ez, free_cigar=True)

    def make_linear_batch(self, queries, targets, mch, mis, e, w=-1, xdrop=-1):
        return LinearBatch(self, queries, targets, mch, mis, e, w, xdrop)

    def extf_batch(self, queries, targets, mch, mis, e, **kw):
        """ksw2amd_extf_batch: n independent gap-linear X-drop extensions -> list of dicts."""
        return self.make_linear_batch(queries, targets, mch, mis, e, **kw).run_oneshot()

    def make_splice_batch(self, queries, targets, mat, q, e, q2, noncan, zdrop=-1, junc_bonus=0, flag=0, juncs=None, m=None):
        return SpliceBatch(self, queries, targets, mat, q, e, q2, noncan, zdrop, junc_bonus, flag, juncs, m)

    def exts_batch(self, queries, targets, mat, q, e, q2, noncan, **kw):
        """ksw2amd_exts_batch: n independent splice-aware extensions -> list of dicts."""
        return self.make_splice_batch(queries, targets, mat, q, e, q2, noncan, **kw).run_oneshot()

    def extz(self, query, target, mat, q, e, w=-1, zdrop=-1, flag=0, m=None):
        qa, qp = self._seq(query)
        ta, tp = self._seq(target)
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        m = int(round(len(mat) ** 0.5)) if m is None else m
        ez = KswExtz()
        self.lib.ksw_extz(None, len(qa), qp, len(ta), tp, m, mat.ctypes.data_as(_i8p), q, e, w, zdrop, flag, ez)
        return ez_to_dict(ez, free_cigar=True)

    def extd(self, query, target, mat, q, e, q2, e2, w=-1, zdrop=-1, flag=0, m=None):
        qa, qp = self._seq(query)
        ta, tp = self._seq(target)
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        m = int(round(len(mat) ** 0.5)) if m is None else m
        ez = KswExtz()
        self.lib.ksw_extd(None, len(qa), qp, len(ta), tp, m, mat.ctypes.data_as(_i8p), q, e, q2, e2, w, zdrop, flag, ez)
        return ez_to_dict(ez, free_cigar=True)

    def gg(self, func, query, target, mat, q, e, w=-1, with_cigar=True, m=None):
        """ksw_gg / ksw_gg2 / ksw_gg2_sse -> (score, CIGAR list)."""
        qa, qp = self._seq(query)
        ta, tp = self._seq(target)
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        m = int(round(len(mat) ** 0.5)) if m is None else m
        f = getattr(self.lib, "ksw_" + func)
        mc, nc = _int(0), _int(0)
        cig = ctypes.POINTER(ctypes.c_uint32)()
        args = (None, len(qa), qp, len(ta), tp, m, mat.ctypes.data_as(_i8p), q, e, w)
        if with_cigar:
            score = f(*args, ctypes.byref(mc), ctypes.byref(nc), ctypes.byref(cig))
        else:
            score = f(*args, None, None, None)
        out = [int(cig[i]) for i in range(nc.value)]
        if cig:
            _libc.free(ctypes.cast(cig, ctypes.c_void_p))
        return int(score), out

    # ---- batches
    def make_batch(self, queries, targets, mat, q, e, q2=0, e2=0, w=-1, zdrop=-1, end_bonus=0, flag=0, m=None):
        return Batch(self, queries, targets, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, m)

    def extz_batch(self, queries, targets, mat, q, e, **kw):
        b = self.make_batch(queries, targets, mat, q, e, 0, 0, **kw)
        return b.run_oneshot(dual=False)

    def make_flat_batch(self, queries, targets, mat, q, e, q2=0, e2=0, w=-1, zdrop=-1, end_bonus=0, flag=0, m=None, device_base=None):
        """ksw2amd_flat_t over one arena: the sequences concatenated in pair order (q0 t0 q1 t1 ...).  device_base: address of a
        device copy of that arena (then the plan / batch never touches the host copy)."""
        return FlatBatch(self, queries, targets, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, m, device_base)

    def extd_batch(self, queries, targets, mat, q, e, q2, e2, **kw):
        b = self.make_batch(queries, targets, mat, q, e, q2, e2, **kw)
        return b.run_oneshot(dual=True)


def _per_pair(v, n):
    a = np.asarray(v, dtype=np.int64)
    if a.ndim == 0:
        a = np.full(n, int(a), dtype=np.int64)
    assert len(a) == n
    return a


class Batch:
    """Host-side description of n pairs (the arguments of n ksw_ext?2_sse calls)."""

    def __init__(self, lib, queries, targets, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, m):
        self.L = lib
        n = len(queries)
        assert len(targets) == n
        self.n = n
        # keep the sequences alive: 2-D arrays are used row by row without copying
        if isinstance(queries, np.ndarray) and queries.ndim == 2:
            self.qa = np.ascontiguousarray(queries, dtype=np.uint8)
            qptr = self.qa.ctypes.data + np.arange(n, dtype=np.int64) * self.qa.shape[1]
            qlen = np.full(n, self.qa.shape[1], dtype=np.int64)
        else:
            self.qa = [np.ascontiguousarray(x, dtype=np.uint8) for x in queries]
            qptr = np.array([x.ctypes.data for x in self.qa], dtype=np.int64)
            qlen = np.array([len(x) for x in self.qa], dtype=np.int64)
        if isinstance(targets, np.ndarray) and targets.ndim == 2:
            self.ta = np.ascontiguousarray(targets, dtype=np.uint8)
            tptr = self.ta.ctypes.data + np.arange(n, dtype=np.int64) * self.ta.shape[1]
            tlen = np.full(n, self.ta.shape[1], dtype=np.int64)
        else:
            self.ta = [np.ascontiguousarray(x, dtype=np.uint8) for x in targets]
            tptr = np.array([x.ctypes.data for x in self.ta], dtype=np.int64)
            tlen = np.array([len(x) for x in self.ta], dtype=np.int64)
        rec = np.zeros(n, dtype=np.dtype([("query", "<u8"), ("target", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"),
                                          ("zdrop", "<i4"), ("end_bonus", "<i4"), ("flag", "<i4")]))
        assert rec.dtype.itemsize == ctypes.sizeof(Pair)
        rec["query"], rec["target"], rec["qlen"], rec["tlen"] = qptr, tptr, qlen, tlen
        rec["w"], rec["zdrop"] = _per_pair(w, n), _per_pair(zdrop, n)
        rec["end_bonus"], rec["flag"] = _per_pair(end_bonus, n), _per_pair(flag, n)
        self.rec = rec
        self.pairs = rec.ctypes.data_as(ctypes.POINTER(Pair))
        self.mat = np.ascontiguousarray(mat, dtype=np.int8)
        self.sc = Scoring(int(round(len(self.mat) ** 0.5)) if m is None else m, self.mat.ctypes.data_as(_i8p), q, e, q2, e2)
        self.qlen, self.tlen = qlen, tlen

    def run_oneshot(self, dual):
        """ksw2amd_ext?_batch: upload, run, download; list of result dicts."""
        ez = (KswExtz * max(self.n, 1))()
        f = self.L.lib.ksw2amd_extd_batch if dual else self.L.lib.ksw2amd_extz_batch
        self.L._check(f(None, ctypes.byref(self.sc), self.n, self.pairs, ez))
        return [ez_to_dict(ez[i], free_cigar=True) for i in range(self.n)]

    def plan(self, dual):
        return Plan(self, dual)

    def sse_plan(self, dual):
        """ksw2amd_sse_plan_create: every pair through the SSE-compatible kernels."""
        return Plan(self, dual, handle=self.L.lib.ksw2amd_sse_plan_create(1 if dual else 0, ctypes.byref(self.sc), self.n, self.pairs))


class FlatBatch:
    """Host-side description of n pairs as ONE arena + offsets (ksw2amd_flat_t); same results as Batch on the same pairs."""

    def __init__(self, lib, queries, targets, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, m, device_base=None):
        self.L = lib
        self.n = n = len(queries)
        if isinstance(queries, np.ndarray) and queries.ndim == 2 and isinstance(targets, np.ndarray) and targets.ndim == 2:
            ql, tl = queries.shape[1], targets.shape[1]
            self.arena = np.ascontiguousarray(np.concatenate([queries, targets], axis=1), dtype=np.uint8).reshape(-1)
            self.qoff = (np.arange(n, dtype=np.uint64) * np.uint64(ql + tl))
            self.toff = self.qoff + np.uint64(ql)
            self.qlen = np.full(n, ql, dtype=np.int32)
            self.tlen = np.full(n, tl, dtype=np.int32)
        else:
            qs = [np.ascontiguousarray(x, dtype=np.uint8) for x in queries]
            ts = [np.ascontiguousarray(x, dtype=np.uint8) for x in targets]
            self.qlen = np.array([len(x) for x in qs], dtype=np.int32)
            self.tlen = np.array([len(x) for x in ts], dtype=np.int32)
            lens = np.empty(2 * n, dtype=np.uint64)
            lens[0::2], lens[1::2] = self.qlen, self.tlen
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            self.qoff, self.toff = offs[0:2 * n:2].copy(), offs[1:2 * n:2].copy()
            parts = [None] * (2 * n)
            parts[0::2], parts[1::2] = qs, ts
            self.arena = np.concatenate(parts + [np.zeros(1, np.uint8)]).astype(np.uint8) if n else np.zeros(1, np.uint8)
        self.per = [np.ascontiguousarray(_per_pair(v, n), dtype=np.int32) for v in (w, zdrop, end_bonus, flag)]
        self.mat = np.ascontiguousarray(mat, dtype=np.int8)
        self.sc = Scoring(int(round(len(self.mat) ** 0.5)) if m is None else m, self.mat.ctypes.data_as(_i8p), q, e, q2, e2)
        self.flat = Flat(device_base if device_base is not None else self.arena.ctypes.data, self.qoff.ctypes.data, self.toff.ctypes.data,
                         self.qlen.ctypes.data, self.tlen.ctypes.data, *[a.ctypes.data for a in self.per], 0, 0, 0, 0, 1 if device_base is not None else 0)
        self.registered = False

    def register(self):
        """ksw2amd_host_register: page-lock the arena (a caller that reuses its buffer does this once)."""
        self.L._check(self.L.lib.ksw2amd_host_register(self.arena.ctypes.data, self.arena.nbytes))
        self.registered = True

    def unregister(self):
        if self.registered:
            self.L.lib.ksw2amd_host_unregister(self.arena.ctypes.data)
            self.registered = False

    def run_oneshot(self, dual):
        ez = (KswExtz * max(self.n, 1))()
        f = self.L.lib.ksw2amd_extd_batch_flat if dual else self.L.lib.ksw2amd_extz_batch_flat
        self.L._check(f(None, ctypes.byref(self.sc), self.n, ctypes.byref(self.flat), ez))
        return [ez_to_dict(ez[i], free_cigar=True) for i in range(self.n)]

    def plan(self, dual):
        return Plan(self, dual, handle=self.L.lib.ksw2amd_plan_create_flat(1 if dual else 0, ctypes.byref(self.sc), self.n, ctypes.byref(self.flat)))


class LinearBatch:
    """Arguments of n ksw_extf2_sse calls with shared scoring, kept alive for the C side."""

    def __init__(self, L, queries, targets, mch, mis, e, w, xdrop):
        self.L = L
        self.n = n = len(queries)
        self.par = (mch, mis, e)
        self.qs = [np.ascontiguousarray(x, dtype=np.uint8) for x in queries]
        self.ts = [np.ascontiguousarray(x, dtype=np.uint8) for x in targets]
        w, xdrop = _per_pair(w, n), _per_pair(xdrop, n)
        self.pairs = (LinearPair * max(n, 1))()
        for i in range(n):
            self.pairs[i] = LinearPair(self.qs[i].ctypes.data, self.ts[i].ctypes.data, len(self.qs[i]), len(self.ts[i]), int(w[i]), int(xdrop[i]))

    def run_oneshot(self):
        ez = (KswExtz * max(self.n, 1))()
        self.L._check(self.L.lib.ksw2amd_extf_batch(None, *self.par, self.n, self.pairs, ez))
        return [ez_to_dict(ez[i], free_cigar=True) for i in range(self.n)]

    def plan(self):
        return Plan(self, False, handle=self.L.lib.ksw2amd_extf_plan_create(*self.par, self.n, self.pairs))


class SpliceBatch:
    """Arguments of n ksw_exts2_sse calls with shared scoring, kept alive for the C side."""

    def __init__(self, L, queries, targets, mat, q, e, q2, noncan, zdrop, junc_bonus, flag, juncs, m):
        self.L = L
        self.n = n = len(queries)
        self.mat = np.ascontiguousarray(mat, dtype=np.int8)
        m = int(round(len(self.mat) ** 0.5)) if m is None else m
        self.qs = [np.ascontiguousarray(x, dtype=np.uint8) for x in queries]
        self.ts = [np.ascontiguousarray(x, dtype=np.uint8) for x in targets]
        self.js = [None if (juncs is None or juncs[i] is None) else np.ascontiguousarray(juncs[i], dtype=np.uint8) for i in range(n)]
        bc = lambda v: np.full(n, v) if np.ndim(v) == 0 else np.asarray(v)      # noqa: E731
        zdrop, flag = bc(zdrop), bc(flag)
        self.sc = SpliceScoring(m, self.mat.ctypes.data_as(_i8p), q, e, q2, noncan, junc_bonus)
        self.pairs = (SplicePair * max(n, 1))()
        for i in range(n):
            self.pairs[i] = SplicePair(self.qs[i].ctypes.data, self.ts[i].ctypes.data, None if self.js[i] is None else self.js[i].ctypes.data,
                                       len(self.qs[i]), len(self.ts[i]), int(zdrop[i]), int(flag[i]))

    def run_oneshot(self):
        ez = (KswExtz * max(self.n, 1))()
        self.L._check(self.L.lib.ksw2amd_exts_batch(None, ctypes.byref(self.sc), self.n, self.pairs, ez))
        return [ez_to_dict(ez[i], free_cigar=True) for i in range(self.n)]

    def plan(self):
        return Plan(self, False, handle=self.L.lib.ksw2amd_exts_plan_create(ctypes.byref(self.sc), self.n, self.pairs))


class Plan:
    """ksw2amd_plan_*: batch resident in HBM; run() enqueues kernels only."""

    def __init__(self, batch, dual, handle=None):
        self.b, self.L = batch, batch.L
        self.h = handle if handle is not None else self.L.lib.ksw2amd_plan_create(1 if dual else 0, ctypes.byref(batch.sc), batch.n, batch.pairs)
        if not self.h:
            raise Ksw2Error("plan_create failed: " + self.L.last_error())

    def run(self, stream=None):
        self.L._check(self.L.lib.ksw2amd_plan_run(self.h, stream))

    def timing(self):
        a, b = ctypes.c_float(0), ctypes.c_float(0)
        self.L._check(self.L.lib.ksw2amd_plan_timing(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def cells(self):
        return int(self.L.lib.ksw2amd_plan_cells(self.h))

    def packed_pairs(self):
        return int(self.L.lib.ksw2amd_plan_packed_pairs(self.h))

    def describe(self):
        """ksw2amd_plan_describe -> one dict per kernel class the next run() launches (kernel, G, C, gaps, mode, rebased, nomax,
        generic, form, tasks)."""
        buf = ctypes.create_string_buffer(16384)
        self.L.lib.ksw2amd_plan_describe(self.h, buf, len(buf))
        out = []
        for line in buf.value.decode().splitlines():
            d = dict(kv.split("=") for kv in line.split())
            out.append({k: (int(v) if v.lstrip("-").isdigit() else v) for k, v in d.items()})
        return out

    def device_bytes(self):
        return int(self.L.lib.ksw2amd_plan_device_bytes(self.h))

    def fetch(self):
        ez = (KswExtz * max(self.b.n, 1))()
        self.L._check(self.L.lib.ksw2amd_plan_fetch(self.h, None, ez))
        return [ez_to_dict(ez[i], free_cigar=True) for i in range(self.b.n)]

    def fetch_raw(self):
        """int32 [n, 16]: max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, n_cigar, rows_done, ti, tj, 0, 0"""
        out = np.zeros((max(self.b.n, 1), 16), dtype=np.int32)
        self.L._check(self.L.lib.ksw2amd_plan_fetch_raw(self.h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return out[:self.b.n]

    def close(self):
        if self.h:
            self.L.lib.ksw2amd_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default = None


def library():
    """The product library (HIP).  Raises if it has not been built: there is no fallback."""
    global _default
    if _default is None:
        # KSW2AMD_LIB: another build of the SAME library (A/B runs of kernel variants under build_ab/, tools/scripts/ab_libs.sh) -- so that
        # no script ever has to copy a variant over the product .so
        _default = Library(os.environ.get("KSW2AMD_LIB") or DEFAULT_SO)
    return _default
