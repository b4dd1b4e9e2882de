"""Seeded synthetic read/reference pairs for parity tests and benches.

target: i.i.d. uniform over {0,1,2,3}; query: the target sent through a substitution / indel channel
(half insertions of uniform bases, half deletions, geometric lengths with mean `indel_mean`), then
trimmed or padded with random bases to the requested query length.  numpy PCG64 streams seeded with
20260001 + config index, so every process regenerates the same batch.
"""
import numpy as np

BASE_SEED = 20260001


def rng_for(config_index, stream=0):
    return np.random.Generator(np.random.PCG64([BASE_SEED + int(config_index), int(stream)]))


def mutate_fixed(target, qlen, rng, sub=0.05, ind=0.06, indel_mean=1.5, tail_random_frac=0.0, tail_pairs=0.0):
    """Vectorised channel for a [n, L] target array; returns a [n, qlen] uint8 query array.

    tail_pairs: fraction of pairs whose last `tail_random_frac` of the query is replaced by random bases
    (makes Z-drop fire, cfg3).
    """
    n, L = target.shape
    p_geo = 1.0 / indel_mean
    u = rng.random((n, L))
    base = target.copy()
    is_sub = u < sub
    base[is_sub] = (base[is_sub] + rng.integers(1, 4, size=int(is_sub.sum()), dtype=np.uint8)) & 3
    is_del = (u >= sub) & (u < sub + ind / 2)
    is_ins = (u >= sub + ind / 2) & (u < sub + ind)
    # deletions: mark [p, p+len) as dropped
    dl = np.zeros((n, L + 1), dtype=np.int32)
    rows, cols = np.nonzero(is_del)
    lens = rng.geometric(p_geo, size=len(rows))
    np.add.at(dl, (rows, cols), 1)
    np.add.at(dl, (rows, np.minimum(cols + lens, L)), -1)
    keep = (np.cumsum(dl[:, :L], axis=1) <= 0)
    ins = np.zeros((n, L), dtype=np.int64)
    rows, cols = np.nonzero(is_ins)
    ins[rows, cols] = rng.geometric(p_geo, size=len(rows))
    counts = keep.astype(np.int64) + ins                      # emitted query bases per target position
    row_len = counts.sum(axis=1)
    flat_counts = counts.reshape(-1)
    total = int(flat_counts.sum())
    src = np.repeat(np.arange(n * L, dtype=np.int64), flat_counts)       # source target position
    starts = np.cumsum(flat_counts) - flat_counts
    within = np.arange(total, dtype=np.int64) - np.repeat(starts, flat_counts)
    is_orig = (within == 0) & keep.reshape(-1)[src]
    vals = rng.integers(0, 4, size=total, dtype=np.uint8)
    vals[is_orig] = base.reshape(-1)[src[is_orig]]
    row_of = src // L
    row_start = np.cumsum(row_len) - row_len
    pos = np.arange(total, dtype=np.int64) - row_start[row_of]
    out = rng.integers(0, 4, size=(n, qlen), dtype=np.uint8)            # padding = random bases
    ok = pos < qlen
    out[row_of[ok], pos[ok]] = vals[ok]
    if tail_pairs > 0 and tail_random_frac > 0:
        sel = rng.random(n) < tail_pairs
        k = int(qlen * tail_random_frac)
        out[sel, qlen - k:] = rng.integers(0, 4, size=(int(sel.sum()), k), dtype=np.uint8)
    return out


def fixed_batch(config_index, n, qlen, tlen, sub=0.05, ind=0.06, tail_random_frac=0.0, tail_pairs=0.0, stream=0):
    """[n, qlen] queries and [n, tlen] targets (uint8 codes 0..3)."""
    rng = rng_for(config_index, stream)
    target = rng.integers(0, 4, size=(n, tlen), dtype=np.uint8)
    query = mutate_fixed(target, qlen, rng, sub=sub, ind=ind, tail_random_frac=tail_random_frac, tail_pairs=tail_pairs)
    return query, target


def mutate_one(target, rng, sub=0.05, ind=0.10, indel_mean=1.5, max_indel=None):
    """Channel for one sequence, natural output length."""
    out = []
    i, L = 0, len(target)
    u = rng.random(L)
    while i < L:
        x = u[i]
        if x < sub:
            out.append((int(target[i]) + int(rng.integers(1, 4))) & 3)
            i += 1
        elif x < sub + ind / 2:
            k = int(rng.geometric(1.0 / indel_mean))
            if max_indel:
                k = min(k, max_indel)
            i += k
        elif x < sub + ind:
            k = int(rng.geometric(1.0 / indel_mean))
            if max_indel:
                k = min(k, max_indel)
            out.append(int(target[i]))
            out.extend(int(v) for v in rng.integers(0, 4, size=k))
            i += 1
        else:
            out.append(int(target[i]))
            i += 1
    return np.array(out, dtype=np.uint8)


def ragged_pairs(rng, n, min_len, max_len, sub=0.05, ind=0.10, indel_mean=1.5, n_rate=0.0, m=5):
    """n (query, target) pairs with target length uniform in [min_len, max_len]; optional wildcard rate."""
    pairs = []
    for _ in range(n):
        tl = int(rng.integers(min_len, max_len + 1))
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        q = mutate_one(t, rng, sub=sub, ind=ind, indel_mean=indel_mean)
        if len(q) == 0:
            q = rng.integers(0, 4, size=1, dtype=np.uint8)
        if n_rate > 0:
            t = t.copy()
            t[rng.random(len(t)) < n_rate] = m - 1
            q[rng.random(len(q)) < n_rate] = m - 1
        pairs.append((q, t))
    return pairs


def simple_mat(m=5, a=2, b=4, sc_n=-1):
    """m x m scoring matrix: +a match, -b mismatch, wildcard row/column = sc_n (bench default -1)."""
    mat = np.full((m, m), -abs(b), dtype=np.int8)
    np.fill_diagonal(mat, abs(a))
    mat[m - 1, :] = sc_n
    mat[:, m - 1] = sc_n
    return mat.reshape(-1).copy()


def band_cells(qlen, tlen, w):
    """In-band DP cells of the exact band |i-j| <= w (metric definition)."""
    if qlen <= 0 or tlen <= 0:
        return 0
    if w < 0 or w > max(qlen, tlen):
        w = max(qlen, tlen)
    i = np.arange(tlen, dtype=np.int64)
    st = np.maximum(0, i - w)
    en = np.minimum(qlen - 1, i + w)
    return int(np.maximum(en - st + 1, 0).sum())


# ---------------------------------------------------------------------------------------------------------------------
# Fast generator (tools/synth/ksw2_synth.c -> tools/libksw2_synth.so): the same channel with one xorshift64* stream per
# pair, multi-threaded.  bench.py's batches (up to a GB of sequence) come from here; the numpy functions above stay for the
# tests whose inputs they have always produced.

_fast = None


def _fast_lib():
    global _fast
    if _fast is None:
        import ctypes
        import os
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        so = os.path.join(root, "tools", "libksw2_synth.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", os.path.join(root, "tools"), "libksw2_synth.so"], check=True, capture_output=True)
        L = ctypes.CDLL(so)
        vp, i64, dbl, c_int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_double, ctypes.c_int
        L.k2s_fixed.argtypes = [ctypes.c_uint64, i64, c_int, c_int, c_int, dbl, dbl, dbl, dbl, vp, vp, c_int]
        L.k2s_fixed.restype = None
        L.k2s_ragged_lengths.argtypes = [ctypes.c_uint64, i64, c_int, c_int, c_int, dbl, dbl, c_int, vp, vp, c_int]
        L.k2s_ragged_lengths.restype = None
        L.k2s_ragged_fill.argtypes = [ctypes.c_uint64, i64, c_int, c_int, c_int, dbl, dbl, c_int, vp, vp, vp, vp, c_int]
        L.k2s_ragged_fill.restype = None
        _fast = L
    return _fast


def _threads():
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(32, n))


def fast_fixed(config_index, n, qlen, tlen, sub=0.05, ind=0.06, tail_random_frac=0.0, tail_pairs=0.0, first=0):
    """[n, qlen] queries and [n, tlen] targets; pair i is a function of (config_index, first + i) only."""
    q = np.empty((n, qlen), dtype=np.uint8)
    t = np.empty((n, tlen), dtype=np.uint8)
    _fast_lib().k2s_fixed(BASE_SEED + int(config_index), int(first), n, qlen, tlen, sub, ind, tail_pairs, tail_random_frac,
                          q.ctypes.data, t.ctypes.data, _threads())
    return q, t


def ragged_lengths(config_index, n, lo, hi, sub=0.03, ind=0.15, maxdiff=450, first=0):
    """The (qlen, tlen) arrays fast_ragged would produce, without the sequences (partition studies on a million pairs)."""
    ql = np.empty(n, dtype=np.int32)
    tl = np.empty(n, dtype=np.int32)
    _fast_lib().k2s_ragged_lengths(BASE_SEED + int(config_index), int(first), n, lo, hi, sub, ind, maxdiff, ql.ctypes.data, tl.ctypes.data, _threads())
    return ql, tl


def fast_ragged(config_index, n, lo, hi, sub=0.03, ind=0.15, maxdiff=450, first=0):
    """Config 5's mix: query length uniform in [lo, hi], target = the query through the channel (its natural length; redrawn
    while |tlen - qlen| > maxdiff).  Returns two lists of uint8 views into two flat arrays (kept alive by the views)."""
    L = _fast_lib()
    ql = np.empty(n, dtype=np.int32)
    tl = np.empty(n, dtype=np.int32)
    seed = BASE_SEED + int(config_index)
    L.k2s_ragged_lengths(seed, int(first), n, lo, hi, sub, ind, maxdiff, ql.ctypes.data, tl.ctypes.data, _threads())
    qoff = np.zeros(n + 1, dtype=np.int64)
    toff = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(ql, out=qoff[1:])
    np.cumsum(tl, out=toff[1:])
    qbuf = np.empty(int(qoff[-1]), dtype=np.uint8)
    tbuf = np.empty(int(toff[-1]), dtype=np.uint8)
    L.k2s_ragged_fill(seed, int(first), n, lo, hi, sub, ind, maxdiff, qoff.ctypes.data, toff.ctypes.data, qbuf.ctypes.data, tbuf.ctypes.data, _threads())
    qs = [qbuf[qoff[i]:qoff[i + 1]] for i in range(n)]
    ts = [tbuf[toff[i]:toff[i + 1]] for i in range(n)]
    return qs, ts
