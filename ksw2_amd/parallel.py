"""Sharding a batch of independent pairs over the GPUs of one node (one process per GPU).

Pairs are independent units and the algorithm has no exchange step, so the only
communication is the trivial scatter of inputs from one rank and the gather of ksw_extz_t records + CIGARs
back to it.  Both are point-to-point transfers of exactly the bytes each rank needs (`dist.batch_isend_irecv`:
one ncclGroupStart/End of sends and receives on backend "nccl" = RCCL over xGMI, "gloo" in the CPU test tier) --
no padding to the largest shard, no `world` full-size staging copies on the source GPU.  Partitioning is
longest-processing-time-first on exact band cells.

All four batched functions shard the same way: `kind` = "extz" | "extd" | "exts" | "extf"
(ksw_extz2_sse / ksw_extd2_sse / ksw_exts2_sse / ksw_extf2_sse semantics per pair).
"""
import ctypes
import heapq

import numpy as np
import torch
import torch.distributed as dist

from . import Flat, KswExtz, LinearPair, Pair, Scoring, SplicePair, SpliceScoring, _i8p, _libc

META = 8   # qlen, tlen, w, zdrop, end_bonus, flag, original index, has-junction-array
RES = 12   # score, max, max_t, max_q, mqe, mqe_t, mte, mte_q, zdropped, reach_end, n_cigar, original index
KEYS = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]
_EZ_DTYPE = np.dtype([("max_zd", "<u4"), ("max_q", "<i4"), ("max_t", "<i4"), ("mqe", "<i4"), ("mqe_t", "<i4"), ("mte", "<i4"),
                      ("mte_q", "<i4"), ("score", "<i4"), ("m_cigar", "<i4"), ("n_cigar", "<i4"), ("reach_end", "<i4"), ("pad", "<i4"),
                      ("cigar", "<u8")])
assert _EZ_DTYPE.itemsize == ctypes.sizeof(KswExtz)


def band_cells(qlen, tlen, w):
    """Exact-band cells per pair (arrays): the closed form of ksw2_host_plan.c::band_cells.  w < 0 = unbanded."""
    qlen, tlen, w = (np.asarray(x, dtype=np.int64) for x in (qlen, tlen, w))
    mx = np.maximum(qlen, tlen)
    w = np.where((w < 0) | (w > mx), mx, w)
    T = np.minimum(qlen + w, tlen)
    a = qlen - 1 - w
    na = np.where(a < 0, 0, np.minimum(a + 1, T))
    nb = np.minimum(w + 1, T)
    sum_en = na * (na - 1) // 2 + na * w + (T - na) * (qlen - 1)
    sum_st = (T - nb) * (T - 1 + nb) // 2 - (T - nb) * w
    return np.where(T <= 0, 0, sum_en - sum_st + T)


def lpt_partition(costs, world):
    """Greedy longest-first partition; returns a list of index arrays (ascending original order inside a shard)."""
    costs = np.asarray(costs, dtype=np.int64)
    order = np.argsort(-costs, kind="stable")
    heap = [(0, r) for r in range(world)]
    owner = np.empty(len(costs), dtype=np.int64)
    for i in order:
        load, r = heapq.heappop(heap)
        owner[i] = r
        heapq.heappush(heap, (load + int(costs[i]), r))
    return [np.flatnonzero(owner == r) for r in range(world)]


def _is_nccl(group):
    return dist.get_backend(group) == "nccl"


def _dev(group=None):
    return torch.device("cuda", torch.cuda.current_device()) if _is_nccl(group) else torch.device("cpu")


def _to_wire(a, group):
    """numpy array -> flat tensor the backend can send (device memory for RCCL)."""
    t = torch.from_numpy(np.ascontiguousarray(a).reshape(-1))
    return t.to(_dev(group), non_blocking=False) if _is_nccl(group) else t


def _from_wire(t):
    return t.cpu().numpy() if t.is_cuda else t.numpy()


def _peer(group, r):
    return r if group is None else dist.get_global_rank(group, r)


def _exchange(ops):
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def scatter_arrays(parts, dtypes, src=0, group=None, keep_device=()):
    """Rank `src` holds parts[r] = tuple of numpy arrays for rank r (one per entry of `dtypes`); every rank returns its own
    tuple.  Sizes travel in one broadcast, payloads in one group of point-to-point transfers of exactly their size.
    keep_device: entries that a receiving rank gets back as the device tensor RCCL delivered (no copy to host memory)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    k = len(dtypes)
    sizes = torch.zeros((world, k), dtype=torch.int64)
    if rank == src:
        sizes = torch.tensor([[int(np.asarray(p[x]).size) for x in range(k)] for p in parts], dtype=torch.int64)
    sizes = sizes.to(_dev(group))
    dist.broadcast(sizes, src=_peer(group, src), group=group)
    sizes = sizes.cpu().numpy()
    ops, keep, mine = [], [], None
    if rank == src:
        for r in range(world):
            if r == rank:
                mine = tuple(np.ascontiguousarray(parts[r][x], dtype=dtypes[x]).reshape(-1) for x in range(k))
                continue
            for x in range(k):
                if sizes[r, x]:
                    t = _to_wire(np.ascontiguousarray(parts[r][x], dtype=dtypes[x]), group)
                    keep.append(t)
                    ops.append(dist.P2POp(dist.isend, t, _peer(group, r), group))
        _exchange(ops)
        return mine
    bufs = [torch.empty(int(sizes[rank, x]), dtype=torch.from_numpy(np.zeros(0, dtype=dtypes[x])).dtype, device=_dev(group)) for x in range(k)]
    for x in range(k):
        if sizes[rank, x]:
            ops.append(dist.P2POp(dist.irecv, bufs[x], _peer(group, src), group))
    _exchange(ops)
    if any(x in keep_device and bufs[x].is_cuda for x in range(k)):
        # a device tensor goes to the library, which copies it on a stream of its own: the receive must be complete on the device
        # first (req.wait() orders torch's current stream only), whatever else this function downloads
        torch.cuda.current_stream(bufs[0].device).synchronize()
    return tuple(bufs[x] if (x in keep_device and bufs[x].is_cuda) else _from_wire(bufs[x]) for x in range(k))


def gather_arrays(local, dtypes, dst=0, group=None):
    """Every rank passes a tuple of numpy arrays; rank `dst` returns the list (by rank) of tuples, the others None."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    k = len(dtypes)
    mine = torch.tensor([int(np.asarray(a).size) for a in local], dtype=torch.int64, device=_dev(group))
    allsz = [torch.zeros(k, dtype=torch.int64, device=_dev(group)) for _ in range(world)]
    dist.all_gather(allsz, mine, group=group)
    sizes = np.stack([s.cpu().numpy() for s in allsz])
    ops, keep = [], []
    if rank != dst:
        for x in range(k):
            if sizes[rank, x]:
                t = _to_wire(np.ascontiguousarray(local[x], dtype=dtypes[x]), group)
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, _peer(group, dst), group))
        _exchange(ops)
        return None
    bufs = {}
    for r in range(world):
        if r == rank:
            continue
        for x in range(k):
            bufs[(r, x)] = torch.empty(int(sizes[r, x]), dtype=torch.from_numpy(np.zeros(0, dtype=dtypes[x])).dtype, device=_dev(group))
            if sizes[r, x]:
                ops.append(dist.P2POp(dist.irecv, bufs[(r, x)], _peer(group, r), group))
    _exchange(ops)
    return [tuple(np.ascontiguousarray(local[x], dtype=dtypes[x]).reshape(-1) for x in range(k)) if r == rank
            else tuple(_from_wire(bufs[(r, x)]) for x in range(k)) for r in range(world)]


def _flatten(seqs, idx):
    """Concatenate seqs[i] for i in idx -> (uint8 buffer, lengths).  2-D arrays (fixed length) take the vectorised path."""
    if isinstance(seqs, np.ndarray) and seqs.ndim == 2:
        sub = seqs[idx]
        return np.ascontiguousarray(sub, dtype=np.uint8).reshape(-1), np.full(len(idx), seqs.shape[1], dtype=np.int64)
    parts = [np.asarray(seqs[i], dtype=np.uint8) for i in idx]
    lens = np.array([len(p) for p in parts], dtype=np.int64)
    return (np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)), lens


def _lengths(seqs):
    if isinstance(seqs, np.ndarray) and seqs.ndim == 2:
        return np.full(seqs.shape[0], seqs.shape[1], dtype=np.int64)
    return np.array([len(x) for x in seqs], dtype=np.int64)


def _bc(v, n):
    a = np.asarray(v, dtype=np.int64)
    return np.full(n, int(a), dtype=np.int64) if a.ndim == 0 else a


def align_flat(lib, kind, seq, meta, scoring, junc=None, device_base=None):
    """One rank's shard through the C-ABI batch entry point.  `seq` = all queries then all targets of the shard (then all
    junction arrays, exts only), `meta` int32 [n, META].  Returns (records int32 [n, RES], CIGAR words int32 flat).
    A shard that RCCL delivered into device memory (`seq` a CUDA tensor, or `device_base` = its address) stays there: extz / extd
    shards go through ksw2amd_ext?_batch_flat with on_device = 1, exts / extf shards through ksw2amd_ext?_batch_device (device
    pointers per pair, gathered into the plan's arena by one kernel) -- scatter -> align -> gather never copies the sequences
    through host memory on a receiving rank."""
    n = len(meta)
    rec = np.zeros((n, RES), dtype=np.int32)
    if n == 0:
        return rec, np.zeros(0, dtype=np.int32)
    on_device = device_base is not None or (isinstance(seq, torch.Tensor) and seq.is_cuda)
    if on_device and device_base is None:
        device_base = seq.data_ptr()
    if not on_device:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
    ql, tl = meta[:, 0].astype(np.int64), meta[:, 1].astype(np.int64)
    base = int(device_base) if on_device else seq.ctypes.data
    qoff = np.concatenate([[0], np.cumsum(ql)[:-1]])
    toff = int(ql.sum()) + np.concatenate([[0], np.cumsum(tl)[:-1]])
    ez = np.zeros(n, dtype=_EZ_DTYPE)
    ezp = ez.ctypes.data_as(ctypes.POINTER(KswExtz))
    if on_device and kind in ("extz", "extd"):
        qo, to = np.ascontiguousarray(qoff, dtype=np.uint64), np.ascontiguousarray(toff, dtype=np.uint64)
        cols = [np.ascontiguousarray(meta[:, c], dtype=np.int32) for c in (0, 1, 2, 3, 4, 5)]
        flat = Flat(int(device_base), qo.ctypes.data, to.ctypes.data, *[c.ctypes.data for c in cols], 0, 0, 0, 0, 1)
        mat = np.ascontiguousarray(scoring["mat"], dtype=np.int8)
        sc = Scoring(int(scoring.get("m") or round(len(mat) ** 0.5)), mat.ctypes.data_as(_i8p), scoring["q"], scoring["e"],
                     scoring.get("q2", 0), scoring.get("e2", 0))
        f = lib.lib.ksw2amd_extd_batch_flat if kind == "extd" else lib.lib.ksw2amd_extz_batch_flat
        lib._check(f(None, ctypes.byref(sc), n, ctypes.byref(flat), ezp))
    elif kind in ("extz", "extd"):
        pr = np.zeros(n, dtype=np.dtype([("query", "<u8"), ("target", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"),
                                         ("zdrop", "<i4"), ("end_bonus", "<i4"), ("flag", "<i4")]))
        assert pr.dtype.itemsize == ctypes.sizeof(Pair)
        pr["query"], pr["target"], pr["qlen"], pr["tlen"] = base + qoff, base + toff, ql, tl
        pr["w"], pr["zdrop"], pr["end_bonus"], pr["flag"] = meta[:, 2], meta[:, 3], meta[:, 4], meta[:, 5]
        mat = np.ascontiguousarray(scoring["mat"], dtype=np.int8)
        sc = Scoring(int(scoring.get("m") or round(len(mat) ** 0.5)), mat.ctypes.data_as(_i8p), scoring["q"], scoring["e"],
                     scoring.get("q2", 0), scoring.get("e2", 0))
        f = lib.lib.ksw2amd_extd_batch if kind == "extd" else lib.lib.ksw2amd_extz_batch
        lib._check(f(None, ctypes.byref(sc), n, pr.ctypes.data_as(ctypes.POINTER(Pair)), ezp))
    elif kind == "exts":
        pr = np.zeros(n, dtype=np.dtype([("query", "<u8"), ("target", "<u8"), ("junc", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"),
                                         ("zdrop", "<i4"), ("flag", "<i4")]))
        assert pr.dtype.itemsize == ctypes.sizeof(SplicePair)
        pr["query"], pr["target"], pr["qlen"], pr["tlen"], pr["zdrop"], pr["flag"] = base + qoff, base + toff, ql, tl, meta[:, 3], meta[:, 5]
        has = meta[:, 7] != 0
        joff = int(ql.sum() + tl.sum()) + np.concatenate([[0], np.cumsum(np.where(has, tl, 0))[:-1]])
        pr["junc"] = np.where(has, base + joff, 0)
        mat = np.ascontiguousarray(scoring["mat"], dtype=np.int8)
        sc = SpliceScoring(int(scoring.get("m") or round(len(mat) ** 0.5)), mat.ctypes.data_as(_i8p), scoring["q"], scoring["e"],
                           scoring["q2"], scoring["noncan"], scoring.get("junc_bonus", 0))
        f = lib.lib.ksw2amd_exts_batch_device if on_device else lib.lib.ksw2amd_exts_batch
        lib._check(f(None, ctypes.byref(sc), n, pr.ctypes.data_as(ctypes.POINTER(SplicePair)), ezp))
    elif kind == "extf":
        pr = np.zeros(n, dtype=np.dtype([("query", "<u8"), ("target", "<u8"), ("qlen", "<i4"), ("tlen", "<i4"), ("w", "<i4"), ("xdrop", "<i4")]))
        assert pr.dtype.itemsize == ctypes.sizeof(LinearPair)
        pr["query"], pr["target"], pr["qlen"], pr["tlen"], pr["w"], pr["xdrop"] = base + qoff, base + toff, ql, tl, meta[:, 2], meta[:, 3]
        f = lib.lib.ksw2amd_extf_batch_device if on_device else lib.lib.ksw2amd_extf_batch
        lib._check(f(None, scoring["mch"], scoring["mis"], scoring["e"], n, pr.ctypes.data_as(ctypes.POINTER(LinearPair)), ezp))
    else:
        raise ValueError("kind must be extz, extd, exts or extf")
    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = ez["score"], ez["max_zd"] & 0x7fffffff, ez["max_t"], ez["max_q"]
    rec[:, 4], rec[:, 5], rec[:, 6], rec[:, 7] = ez["mqe"], ez["mqe_t"], ez["mte"], ez["mte_q"]
    rec[:, 8], rec[:, 9], rec[:, 10], rec[:, 11] = ez["max_zd"] >> 31, ez["reach_end"], ez["n_cigar"], meta[:, 6]
    nc = ez["n_cigar"].astype(np.int64)
    cig = np.empty(int(nc.sum()), dtype=np.int32)
    pos = 0
    for i in np.flatnonzero(ez["cigar"]):
        c = int(nc[i])
        if c:
            ctypes.memmove(cig.ctypes.data + 4 * pos, int(ez["cigar"][i]), 4 * c)
            pos += c
        _libc.free(ctypes.c_void_p(int(ez["cigar"][i])))
    return rec, cig


def sharded(lib, kind, queries, targets, scoring, w=-1, zdrop=-1, end_bonus=0, flag=0, juncs=None, group=None, src=0, raw=False):
    """Rank `src` passes the whole batch (the other ranks pass None for queries / targets); every rank aligns its shard on its
    own GPU; rank `src` returns the results in the original order, the others None.
    scoring: dict(mat, q, e[, q2, e2]) for extz / extd, dict(mat, q, e, q2, noncan[, junc_bonus]) for exts, dict(mch, mis, e) for
    extf (w = band, zdrop = X-drop).  raw=True returns (records int32 [n, RES - 1], CIGAR offsets, CIGAR words) instead of dicts."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = None
    n = 0
    if rank == src:
        n = len(queries)
        qlen, tlen = _lengths(queries), _lengths(targets)
        w_, zd_, eb_, fl_ = _bc(w, n), _bc(zdrop, n), _bc(end_bonus, n), _bc(flag, n)
        cost = qlen * tlen if kind == "exts" else band_cells(qlen, tlen, w_)
        parts = []
        for idx in lpt_partition(cost, world):
            m = np.zeros((len(idx), META), dtype=np.int32)
            qbuf, _ = _flatten(queries, idx)
            tbuf, _ = _flatten(targets, idx)
            bufs = [qbuf, tbuf]
            if len(idx):
                m[:, 0], m[:, 1], m[:, 2], m[:, 3], m[:, 4], m[:, 5], m[:, 6] = qlen[idx], tlen[idx], w_[idx], zd_[idx], eb_[idx], fl_[idx], idx
                if kind == "exts" and juncs is not None:
                    for row, i in enumerate(idx):
                        if juncs[i] is not None:
                            m[row, 7] = 1
                            bufs.append(np.asarray(juncs[i], dtype=np.uint8))
            parts.append((m.reshape(-1), np.concatenate(bufs)))
    meta, seq = scatter_arrays(parts, (np.int32, np.uint8), src, group, keep_device=(1,))      # the sequences stay where RCCL put them
    rec, cig = align_flat(lib, kind, seq, meta.reshape(-1, META), scoring)
    got = gather_arrays((rec.reshape(-1), cig), (np.int32, np.int32), src, group)
    if rank != src:
        return None
    recs = np.concatenate([g[0].reshape(-1, RES) for g in got])
    cigs = np.concatenate([g[1] for g in got])
    ends = np.cumsum(recs[:, 10].astype(np.int64))
    starts = ends - recs[:, 10]
    order = np.argsort(recs[:, 11], kind="stable")
    if raw:
        return recs[order, :RES - 1], starts[order], cigs
    out = []
    for r in order:
        d = dict(zip(KEYS, (int(x) for x in recs[r, :RES - 1])))
        d["cigar"] = [int(x) & 0xffffffff for x in cigs[starts[r]:ends[r]]]
        out.append(d)
    return out


def sharded_align(lib, dual, queries, targets, mat, q, e, q2=0, e2=0, w=-1, zdrop=-1, end_bonus=0, flag=0, group=None, src=0):
    """extz2 / extd2 form of `sharded` (kept for callers of the first version)."""
    return sharded(lib, "extd" if dual else "extz", queries, targets, dict(mat=mat, q=q, e=e, q2=q2, e2=e2), w=w, zdrop=zdrop,
                   end_bonus=end_bonus, flag=flag, group=group, src=src)
