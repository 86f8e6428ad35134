#!/usr/bin/env python3
"""Headline benchmark: decoded nt/s, Viterbi error decoding on the s16h74l4c4 composite.

Workload (BASELINE.json configs[2], SURVEY.md 8(d) "Config 3"): per GPU `--reads` synthetic
reads, each 29 random payload bytes (MT19937, seed 1000 + read index) encoded through
data/s16h74l4c4.json (479-505 nt), 1 % i.i.d. substitutions; error model --error-global with
the CLI defaults (sub .01, iv 10, dup .001, del-open .001, del-ext .01, P = 6).  A "step" is
one pass of the hot path (lattice fill + traceback for every read of the shard) with the
reads already resident in HBM; with N > 1 ranks the per-GPU work is fixed (weak scaling) and
the decoded strings are gathered to rank 0 inside the timed region.

Launch: `python bench.py` (1 GPU) or, for N > 1,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
   --master-port P bench.py --gpus N --steps K --warmup W`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MACHINE = os.path.join(ROOT, "tests", "golden", "ref_data", "s16h74l4c4.json")
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def make_reads(machine, first_index, count, sub_rate=0.01):
    """Config-3 generator: payload -> exact encoding -> i.i.d. substitutions.  Deterministic per read index."""
    reads = []
    for i in range(first_index, first_index + count):
        rng = random.Random(1000 + i)
        payload = bytes(rng.randrange(256) for _ in range(29))
        dna = np.frombuffer(machine.encodeBytes(payload).encode(), dtype=np.uint8).copy()
        nrng = np.random.default_rng(1000 + i)
        hit = nrng.random(len(dna)) < sub_rate
        if hit.any():
            code = {65: 0, 67: 1, 71: 2, 84: 3}
            cur = np.array([code[int(c)] for c in dna[hit]])
            new = (cur + nrng.integers(1, 4, size=cur.size)) % 4
            dna[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[new]
        reads.append(dna.tobytes().decode())
    return reads


def cpu_baseline(reads, max_seconds=20.0):
    """Time the CPU oracle (the port of the reference's algorithm) on a bounded sample of the
    same workload, single thread.  Checker code: used here only as the reported baseline."""
    from oracle import oracle as O
    O.build()
    orc = O.ViterbiOracle(O.Machine.from_file(MACHINE), O.MutatorParams.from_cli(global_=True))
    nt, n, results = 0, 0, []
    t0 = time.perf_counter()
    for r in reads:
        results.append(orc.decode(r))
        nt += len(r)
        n += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    dt = time.perf_counter() - t0
    return dict(value=nt / dt, unit="nt/s", cores=1, kind="port",
                sample="%d reads (%d nt) of the same batch, oracle/viterbi_oracle.c, 1 thread, %.1f s" % (n, nt, dt)), results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10000, help="reads per GPU (config 3: 10k)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline time budget (0 = skip)")
    ap.add_argument("--arena-gb", type=float, default=0.0, help="lattice arena per GPU (0 = the library's default, 60 %% of free HBM)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import dnastore_amd as da
    from dnastore_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # DNAS_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (the ranks
    # share the cards, the collectives run on host copies); the measured configuration is always nccl (= RCCL)
    backend = os.environ.get("DNAS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    machine = da.Machine.fromFile(MACHINE)
    params = da.MutatorParams.fromFlags(global_=True)
    dec = da.ViterbiDecoder(machine, params, device=local_rank, arena_bytes=int(args.arena_gb * 1e9))

    # ---- inputs: rank 0 makes the whole job's reads and scatters them (RCCL), untimed
    total_reads = args.reads * world
    all_reads = None
    if rank == 0:
        all_reads = make_reads(machine, 0, total_reads)
        off_all, bases_all = da.pack_reads(all_reads)
    else:
        off_all, bases_all = None, None
    idx, off, d_bases = shard.scatter_reads(off_all, bases_all, world, rank, coll_device)
    d_bases = d_bases.to(device)
    torch.cuda.synchronize()   # the library launches on its own streams: the scattered reads must have landed
    k = len(off) - 1
    lens = np.diff(off).astype(np.int64)
    cap = int(lens.max()) + 64 if k else 64
    if world > 1:   # the gathered symbol buffers must have one shape on every rank
        tcap = torch.tensor([cap], dtype=torch.int64, device=coll_device)
        dist.all_reduce(tcap, op=dist.ReduceOp.MAX)
        cap = int(tcap.item())
    out_off = (np.arange(k + 1, dtype=np.uint64) * np.uint64(cap))
    d_sym = torch.zeros(max(k * cap, 1), dtype=torch.uint8, device=device)
    d_len = torch.zeros(max(k, 1), dtype=torch.int32, device=device)
    d_ll = torch.zeros(max(k, 1), dtype=torch.float64, device=device)
    d_st = torch.zeros(max(k, 1), dtype=torch.uint8, device=device)
    shard_nt = int(lens.sum())

    def step():
        dec.decode_device(off, d_bases.data_ptr(), d_sym.data_ptr(), out_off, d_len.data_ptr(), d_ll.data_ptr(),
                          d_st.data_ptr())
        dec.sync()
        return shard.gather_results(d_sym.to(coll_device), d_len.to(coll_device), d_ll.to(coll_device), d_st.to(coll_device), world, rank)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()   # torch's fills of the output buffers are done before the library's streams write them
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    fill_ms = 0.0
    tb_ms = 0.0
    stats = None
    gathered = None
    for _ in range(args.steps):
        gathered = step()
        stats = dec.stats()
        fill_ms += stats["fill_ms"]
        tb_ms += stats["traceback_ms"]
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(shard_nt)], dtype=torch.float64, device=coll_device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, total_nt = float(tmax[0]), float(tsum[1])
    else:
        total_nt = float(shard_nt)

    if rank == 0:
        # ---- parity spot check + CPU baseline (rank 0, N = 1 only), outside the timed region
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu, results = cpu_baseline(all_reads, args.cpu_seconds)
            sym, olen, ll, st = [x.cpu().numpy() for x in gathered[0]]
            for i, (s_ref, ll_ref) in enumerate(results):
                got = sym[i * cap:i * cap + int(olen[i])].tobytes().decode()
                if got != s_ref or float(ll[i]) != ll_ref:
                    raise SystemExit("PARITY FAILURE on read %d: %r/%r vs oracle %r/%r" % (i, got, ll[i], s_ref, ll_ref))
            cpu["parity_checked_reads"] = len(results)
        value = total_nt * args.steps / elapsed
        launches = stats["fill_launches"] * args.steps
        achieved = stats["lattice_bytes"] * args.steps / (fill_ms / 1e3) / 1e9 if fill_ms > 0 else 0.0
        # HBM traffic of the fill kernel: PMC counters are collected in separate rocprofv3 passes of this same
        # command (profiles/r1_traffic.json: FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md, + WRITE_SIZE,
        # in KB), recorded per lattice column and scaled here to the columns of one launch
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
            if tj.get("kernel", "").endswith("tiera") == dec.tier.startswith("tier A"):
                traffic = tj["hbm_bytes_per_column_corrected"] * stats["columns"] / max(stats["fill_launches"], 1)
        except (OSError, ValueError, KeyError):
            pass
        line = {
            "metric": "decoded nt/sec (whole node), Viterbi on composite FST",
            "value": value, "unit": "nt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2]: %d x ~490-nt reads/GPU through s16h74l4c4.json (12361 states), "
                                   "--error-global, 1%% substitutions" % args.reads,
                       "reads_per_gpu": args.reads, "total_nt": int(total_nt), "parallelism": "read-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "viterbi_fill_tiera" if dec.tier.startswith("tier A") else "viterbi_fill_kernel",
                         "tier": dec.tier[:6], "avg_launch_ms": fill_ms / max(launches, 1),
                         "algorithmic_bytes_per_launch": stats["lattice_bytes"] / max(stats["fill_launches"], 1),
                         "rounds_per_column": stats["rounds"] / max(stats["columns"], 1),
                         "traceback_ms_per_step": tb_ms / args.steps},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    dec.close()


if __name__ == "__main__":
    main()
