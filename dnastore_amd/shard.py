"""Read sharding across the GPUs of one node (one process per GPU, torch.distributed).

Reads are independent DPs (reference: the serial loop of viterbi.cpp:312-318), so the
path has no exchange step: rank 0 scatters the packed reads once (RCCL scatter over
xGMI on GPUs, gloo on CPU for tests), every rank decodes its own shard, and the decoded
symbol strings + log-likelihoods are gathered back.  No collective sits inside the DP.
"""
import numpy as np
import torch
import torch.distributed as dist


def partition(lengths, world):
    """Deal reads to ranks so that every rank gets the same count (+-1) and a similar length mix:
    sort by length (longest first) and deal in snake order (0..W-1, W-1..0, ...).  Returns index arrays."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    parts = [[] for _ in range(world)]
    for pos, i in enumerate(order):
        rnd, k = divmod(pos, world)
        parts[k if rnd % 2 == 0 else world - 1 - k].append(int(i))
    return [np.sort(np.array(p, dtype=np.int64)) for p in parts]


def _dist_ready(world):
    return world > 1 and dist.is_available() and dist.is_initialized()


def scatter_reads(read_offsets, bases, world, rank, device):
    """Rank 0 holds (read_offsets uint64[n+1], bases uint8[...]) for the whole job; every rank
    returns its shard as (indices int64[k], offsets uint64[k+1], bases uint8 tensor on `device`).
    With world == 1 nothing is communicated."""
    if not _dist_ready(world):
        t = torch.from_numpy(np.ascontiguousarray(bases)).to(device)
        return np.arange(len(read_offsets) - 1, dtype=np.int64), np.asarray(read_offsets, dtype=np.uint64), t
    meta = [None]
    if rank == 0:
        lengths = np.diff(read_offsets).astype(np.int64)
        parts = partition(lengths, world)
        shard_lens = [lengths[p] for p in parts]
        pad = int(max(int(sl.sum()) for sl in shard_lens))
        meta = [(parts, shard_lens, pad)]
    dist.broadcast_object_list(meta, src=0)
    parts, shard_lens, pad = meta[0]
    recv = torch.empty(max(pad, 1), dtype=torch.uint8, device=device)
    chunks = None
    if rank == 0:
        chunks = []
        for p in parts:
            buf = np.zeros(max(pad, 1), dtype=np.uint8)
            pos = 0
            for i in p:
                seg = bases[int(read_offsets[i]):int(read_offsets[i + 1])]
                buf[pos:pos + len(seg)] = seg
                pos += len(seg)
            chunks.append(torch.from_numpy(buf).to(device))
    dist.scatter(recv, scatter_list=chunks, src=0)
    off = np.zeros(len(parts[rank]) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(shard_lens[rank])
    return parts[rank].astype(np.int64), off, recv


def gather_results(sym, out_len, loglike, status, world, rank):
    """Per-rank result tensors (sym uint8[k*cap], out_len/loglike/status [k]) -> on rank 0 a list over ranks of
    the same tuples (None elsewhere).  The ranks may hold different numbers of reads k (a job of n reads dealt
    over W ranks gives n // W or n // W + 1 each): the buffers are padded to the largest shard for the gather and
    cut back on rank 0.  The output capacity per read (cap) must be the same on every rank."""
    if not _dist_ready(world):
        return [(sym, out_len, loglike, status)]
    k = int(out_len.numel())
    cap = sym.numel() // k if k else 0
    shape = torch.tensor([k, cap, -cap if k else -(1 << 40)], dtype=torch.int64, device=sym.device)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX)
    kmax, capmax, capmin = int(shape[0]), int(shape[1]), -int(shape[2])
    if capmin != capmax and kmax > 0:      # (global values: every rank raises, none is left waiting in a collective)
        raise ValueError("gather_results: the output capacity per read differs across ranks")
    cap = capmax

    def padded(t, n):
        if t.numel() == n:
            return t
        out = torch.zeros(n, dtype=t.dtype, device=t.device)
        out[:t.numel()] = t
        return out
    ks = [torch.zeros(1, dtype=torch.int64, device=sym.device) for _ in range(world)] if rank == 0 else None
    dist.gather(torch.tensor([k], dtype=torch.int64, device=sym.device), gather_list=ks, dst=0)
    outs = []
    for t, n in ((sym, kmax * cap), (out_len, kmax), (loglike, kmax), (status, kmax)):
        t = padded(t, n)
        bucket = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, gather_list=bucket, dst=0)
        outs.append(bucket)
    if rank != 0:
        return None
    res = []
    for r in range(world):
        kr = int(ks[r])
        res.append((outs[0][r][:kr * cap], outs[1][r][:kr], outs[2][r][:kr], outs[3][r][:kr]))
    return res


def shard_pairs(pk, world, rank):
    """Alignment pairs of a packed Stockholm set (api.StockholmDB.arrays() layout) for one rank: pairs
    are dealt like reads (by output length, snake order).  Returns a dict of the same layout."""
    n = int(pk["n"])
    if world <= 1:
        return pk
    lengths = np.diff(np.asarray(pk["out_off"], dtype=np.int64))
    mine = partition(lengths, world)[rank]

    def take(data, off):
        off = np.asarray(off, dtype=np.int64)
        segs = [np.asarray(data)[off[i]:off[i + 1]] for i in mine]
        new_off = np.zeros(len(mine) + 1, dtype=np.int64)
        if len(mine):
            new_off[1:] = np.cumsum([len(s) for s in segs])
        flat = np.concatenate(segs) if segs else np.asarray(data)[:0]
        return np.ascontiguousarray(flat), new_off

    out = {"n": len(mine)}
    for d, o in (("ins", "in_off"), ("outs", "out_off"), ("cm_in", "cm_in_off"), ("cm_out", "cm_out_off")):
        out[d], out[o] = take(pk[d], pk[o])
    assert n >= len(mine)
    return out


def allreduce_counts(counts, ll, world, device="cpu"):
    """The reduction of the E-step (fwdback.cpp:204-205: counts += stockCounts; ll += ...): one
    all-reduce(sum) of the 21+P expected counts and the log-likelihood over the ranks (RCCL over xGMI on
    GPUs, gloo in the CPU tests).  The fp64 summation order differs from the serial loop, so parity with
    the single-process result is to ~1e-12 relative, not bit for bit."""
    if not _dist_ready(world):
        return np.asarray(counts, dtype=np.float64), float(ll)
    t = torch.empty(len(counts) + 1, dtype=torch.float64, device=device)
    t[:-1] = torch.as_tensor(np.asarray(counts, dtype=np.float64), device=device)
    t[-1] = float(ll)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.cpu().numpy()
    return t[:-1].copy(), float(t[-1])


def expected_counts_sharded(params, pk, world, rank, estep, device="cpu"):
    """Expected counts of the whole set with the pairs sharded over the ranks.  estep(params, shard_dict)
    -> (counts, ll, ...) is the per-rank E-step (api.expectedCounts on a GPU)."""
    mine = shard_pairs(pk, world, rank)
    if int(mine["n"]) > 0:
        res = estep(params, mine)
        counts, ll = res[0], res[1]
    else:
        counts, ll = None, 0.0
    n_counts = torch.tensor([0 if counts is None else len(counts)], dtype=torch.int64, device=device)
    if _dist_ready(world):
        dist.all_reduce(n_counts, op=dist.ReduceOp.MAX)
    if counts is None:
        counts = np.zeros(int(n_counts.item()))
    return allreduce_counts(counts, ll, world, device)
