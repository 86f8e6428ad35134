"""Read sharding across ranks (dnastore_amd/shard.py) exercised with world_size 2 on CPU (gloo):
partition -> scatter -> per-rank "decode" (a stand-in transform, the DP itself needs a GPU) -> gather,
checked against the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reads(n, seed):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 40, size=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    bases = rng.integers(0, 4, size=int(off[-1]), dtype=np.uint8)
    return off, bases


def _fake_decode(off, bases, cap):
    """Stand-in for the GPU decode with the same result layout: symbol = base + 48, loglike = -sum(bases)."""
    k = len(off) - 1
    sym = np.zeros(k * cap, dtype=np.uint8)
    olen = np.zeros(k, dtype=np.int32)
    ll = np.zeros(k, dtype=np.float64)
    st = np.zeros(k, dtype=np.uint8)
    for i in range(k):
        seg = bases[int(off[i]):int(off[i + 1])]
        sym[i * cap:i * cap + len(seg)] = seg + 48
        olen[i] = len(seg)
        ll[i] = -float(seg.sum())
        st[i] = 1 if len(seg) == 0 else 0
    return sym, olen, ll, st


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    from dnastore_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off_all, bases_all = _reads(n, 7) if rank == 0 else (None, None)
    idx, off, d_bases = shard.scatter_reads(off_all, bases_all, world, rank, torch.device("cpu"))
    cap = 48
    sym, olen, ll, st = _fake_decode(off, d_bases.numpy(), cap)
    gathered = shard.gather_results(torch.from_numpy(sym), torch.from_numpy(olen), torch.from_numpy(ll), torch.from_numpy(st),
                                    world, rank)
    # every rank also tells rank 0 which reads it held
    idx_list = [None] * world
    dist.gather_object(idx.tolist(), idx_list if rank == 0 else None, dst=0)
    if rank == 0:
        q.put(([tuple(t.numpy() for t in g) for g in gathered], idx_list))
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_partition_balances_counts_and_lengths():
    sys.path.insert(0, ROOT)
    from dnastore_amd import shard
    lens = np.array([5, 100, 7, 90, 80, 3, 60, 61])
    parts = shard.partition(lens, 2)
    assert sorted(np.concatenate(parts).tolist()) == list(range(8))
    assert len(parts[0]) == len(parts[1]) == 4
    assert abs(int(lens[parts[0]].sum()) - int(lens[parts[1]].sum())) <= int(lens.max()) // 2


def test_bench_reads_are_made_per_rank_by_index(ref_data):
    """bench.py: every rank makes its own reads by index -- the shards of W ranks are the reads of one rank of a
    W-times-larger job, read for read (no scatter needed to agree on the workload)."""
    import bench
    import dnastore_amd as da
    m = da.Machine.fromFile(bench.MACHINE)
    whole = bench.make_reads(m, 0, 12)
    parts = [bench.make_reads(m, r * 3, 3) for r in range(4)]
    assert [x for p in parts for x in p] == whole
    assert len(set(whole)) == 12


@pytest.mark.parametrize("world,n", [(2, 24), (8, 45)])      # 45 reads over 8 ranks: shards of 5 and 6 reads
def test_scatter_decode_gather_matches_single_process(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, idx_list = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    off_all, bases_all = _reads(n, 7)
    cap = 48
    ref_sym, ref_len, ref_ll, ref_st = _fake_decode(off_all, bases_all, cap)
    seen = set()
    assert sorted(len(x) for x in idx_list) == sorted([n // world + (1 if r < n % world else 0) for r in range(world)])
    for r in range(world):
        sym, olen, ll, st = gathered[r]
        assert len(olen) == len(idx_list[r]) and len(sym) == len(idx_list[r]) * cap
        for j, i in enumerate(idx_list[r]):
            seen.add(i)
            assert olen[j] == ref_len[i] and ll[j] == ref_ll[i] and st[j] == ref_st[i]
            assert np.array_equal(sym[j * cap:(j + 1) * cap], ref_sym[i * cap:(i + 1) * cap])
    assert seen == set(range(n))


def _unpack(pk):
    """packed pair arrays -> list of (ins, outs, cmIn, cmOut)"""
    out = []
    for i in range(int(pk["n"])):
        out.append(tuple(np.asarray(pk[d])[int(pk[o][i]):int(pk[o][i + 1])]
                         for d, o in (("ins", "in_off"), ("outs", "out_off"), ("cm_in", "cm_in_off"), ("cm_out", "cm_out_off"))))
    return out


def _pairs(O):
    import random
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from synth import synthetic_alignment
    rng = random.Random(11)
    return [O.alignment_pair(synthetic_alignment(rng, rng.choice([3, 20, 64, 101]), sub=.03, dele=.02, dup=.02)) for _ in range(13)]


def _estep_worker(rank, world, port, stk, q):
    sys.path.insert(0, ROOT)
    from dnastore_amd import shard
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pk = O.pack_pairs(_pairs(O))
    params = O.MutatorParams.from_cli(length=6)
    # the per-rank E-step: the CPU oracle stands in for the GPU kernel (no GPU in this test)
    counts, ll = shard.expected_counts_sharded(params, pk, world, rank, lambda p, d: O.expected_counts(p, _unpack(d)))
    if rank == 0:
        q.put((counts, ll))
    dist.barrier()
    dist.destroy_process_group()


def test_estep_allreduce_matches_single_process():
    """fwd-back: pairs sharded over 2 ranks + all-reduce(sum) of counts and log-likelihood == one process."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    stk = None
    want = O.expected_counts(O.MutatorParams.from_cli(length=6), _pairs(O))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_estep_worker, args=(r, 2, port, stk, q)) for r in range(2)]
    for p in procs:
        p.start()
    counts, ll = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_allclose(counts, want[0], rtol=1e-12, atol=1e-300)
    assert abs(ll - want[1]) <= 1e-12 * max(1.0, abs(want[1]))
