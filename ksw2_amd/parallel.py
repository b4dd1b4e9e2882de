"""Sharding a batch of independent pairs over the GPUs of one node (one process per GPU).

Pairs are independent units and the algorithm has no exchange step, so the only
communication is the trivial scatter of inputs from rank 0 and the gather of ksw_extz_t records + CIGARs
back to rank 0, done with torch.distributed collectives (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU test tier).  Partitioning is longest-processing-time-first on exact band cells.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import synth

META = 8   # qlen, tlen, w, zdrop, end_bonus, flag, original index, reserved
RES = 11   # score, max, max_t, max_q, mqe, mqe_t, mte, mte_q, zdropped, reach_end, n_cigar


def lpt_partition(costs, world):
    """Greedy longest-first partition; returns a list of index arrays (ascending original order inside a shard)."""
    order = np.argsort(-np.asarray(costs, dtype=np.int64), kind="stable")
    load = np.zeros(world, dtype=np.int64)
    shards = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += int(costs[i])
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def _dev(group=None):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def _scatter_padded(parts, dtype, width, src, group):
    """rank src: list of 2-D arrays (one per rank, rows x width) -> every rank gets its own rows."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _dev(group)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    if rank == src:
        sizes = torch.tensor([len(p) for p in parts], dtype=torch.int64, device=dev)
    dist.broadcast(sizes, src=src, group=group)
    mx = int(sizes.max().item())
    out = torch.zeros((max(mx, 1), width), dtype=dtype, device=dev)
    if rank == src:
        lst = []
        for p in parts:
            buf = torch.zeros((max(mx, 1), width), dtype=dtype, device=dev)
            if len(p):
                buf[:len(p)] = torch.as_tensor(np.ascontiguousarray(p).reshape(len(p), width)).to(dev)
            lst.append(buf)
        dist.scatter(out, scatter_list=lst, src=src, group=group)
    else:
        dist.scatter(out, scatter_list=None, src=src, group=group)
    return out[:int(sizes[rank].item())].cpu().numpy()


def _gather_padded(local, dtype, width, dst, group):
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _dev(group)
    n = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros((mx, width), dtype=dtype, device=dev)
    if len(local):
        buf[:len(local)] = torch.as_tensor(np.ascontiguousarray(local).reshape(len(local), width)).to(dev)
    if rank == dst:
        lst = [torch.zeros((mx, width), dtype=dtype, device=dev) for _ in range(world)]
        dist.gather(buf, gather_list=lst, dst=dst, group=group)
        return [lst[r][:sizes[r]].cpu().numpy() for r in range(world)]
    dist.gather(buf, gather_list=None, dst=dst, group=group)
    return None


def sharded_align(lib, dual, queries, targets, mat, q, e, q2=0, e2=0, w=-1, zdrop=-1, end_bonus=0, flag=0, group=None, src=0):
    """Rank `src` passes the whole batch (other ranks may pass None for queries/targets); every rank aligns its
    shard on its own GPU; rank `src` returns the list of result dicts in the original order, others return None."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    meta_parts = seq_parts = None
    if rank == src:
        n = len(queries)
        bc = lambda v: np.full(n, v, dtype=np.int64) if np.ndim(v) == 0 else np.asarray(v, dtype=np.int64)
        w_, zd_, eb_, fl_ = bc(w), bc(zdrop), bc(end_bonus), bc(flag)
        qlen = np.array([len(x) for x in queries], dtype=np.int64)
        tlen = np.array([len(x) for x in targets], dtype=np.int64)
        cost = np.array([synth.band_cells(int(qlen[i]), int(tlen[i]), int(w_[i])) for i in range(n)], dtype=np.int64)
        shards = lpt_partition(cost, world)
        meta_parts, seq_parts = [], []
        for idx in shards:
            m = np.zeros((len(idx), META), dtype=np.int32)
            if len(idx):
                m[:, 0], m[:, 1], m[:, 2], m[:, 3], m[:, 4], m[:, 5], m[:, 6] = qlen[idx], tlen[idx], w_[idx], zd_[idx], eb_[idx], fl_[idx], idx
            meta_parts.append(m)
            seq = [np.asarray(queries[i], dtype=np.uint8) for i in idx] + [np.asarray(targets[i], dtype=np.uint8) for i in idx]
            seq_parts.append(np.concatenate(seq).reshape(-1, 1) if len(seq) else np.zeros((0, 1), dtype=np.uint8))
    meta = _scatter_padded(meta_parts, torch.int32, META, src, group)
    seq = _scatter_padded(seq_parts, torch.uint8, 1, src, group).reshape(-1)
    n_loc = len(meta)
    qs, ts, off = [], [], 0
    for i in range(n_loc):
        qs.append(seq[off:off + meta[i, 0]]); off += int(meta[i, 0])
    for i in range(n_loc):
        ts.append(seq[off:off + meta[i, 1]]); off += int(meta[i, 1])
    res = []
    if n_loc:
        kw = dict(w=meta[:, 2], zdrop=meta[:, 3], end_bonus=meta[:, 4], flag=meta[:, 5])
        res = lib.extd_batch(qs, ts, mat, q, e, q2, e2, **kw) if dual else lib.extz_batch(qs, ts, mat, q, e, **kw)
    keys = ["score", "max", "max_t", "max_q", "mqe", "mqe_t", "mte", "mte_q", "zdropped", "reach_end", "n_cigar"]
    rec = np.array([[r[k] for k in keys] + [int(meta[i, 6])] for i, r in enumerate(res)], dtype=np.int32).reshape(n_loc, RES + 1)
    cig = np.array([c for r in res for c in r["cigar"]], dtype=np.int64).reshape(-1, 1)
    recs = _gather_padded(rec, torch.int32, RES + 1, src, group)
    cigs = _gather_padded(cig, torch.int64, 1, src, group)
    if rank != src:
        return None
    out = [None] * len(queries)
    for r in range(world):
        pos = 0
        for row in recs[r]:
            d = dict(zip(keys, (int(x) for x in row[:RES])))
            d["cigar"] = [int(x) for x in cigs[r][pos:pos + d["n_cigar"], 0]]
            pos += d["n_cigar"]
            out[int(row[RES])] = d
    return out
