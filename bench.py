#!/usr/bin/env python3
"""Benchmarks of the hot path on MI355X, one BASELINE.json configuration per run (`--config`, index into
BASELINE.json `configs`; the default, 2, is the configuration the metric is quoted on):

  1  Viterbi -V on the mixradar6 composite (flusher * mixradar6 * l4c4, 46 670 states, README.md:34-47), ~1 kb reads
     (128 random payload bytes), --error-global; the fill runs on clusters of work-groups (tier C).  BASELINE names
     a single read: its latency is reported as `latency_ms_single_read`; the throughput line uses `--reads` reads.
  2  10 000 x ~490-nt reads per GPU through s16h74l4c4.json (12 361 states), --error-global  [headline]
  3  ~1 kb reads through water64.1 * l4c4 (7 066 states: the composable reading of "water64.1 + hamming74", SURVEY
     8d 4a), 12 500 reads per GPU (100k over 8 GPUs); `--variant b`: hamming74 * dropdot * water64.1 * l4c4
     (258 538 states, tier C), 24 reads per GPU (two launches of 12 clusters)
  4  forward-backward E-step (expectedCounts, fwdback.cpp:190-209), 256-nt pairs with dup + sub + del errors,
     125 000 pairs per GPU (1M over 8 GPUs); unit nt/s over the read (output) lengths

Reads are synthetic and deterministic per read index: random payload (MT19937, seed 1000 + index) encoded through
the machine, 1 % i.i.d. substitutions; error model = CLI defaults, P = 6.  A "step" is one pass of the hot path
(lattice fill + traceback of every read of the shard) with the reads resident in HBM; with N > 1 ranks the per-GPU
work is fixed (weak scaling), every rank makes its own reads by index (`--scatter`: rank 0 makes all and scatters
them over RCCL instead) and the decoded strings are gathered to rank 0 inside the timed region.

Launch: `python bench.py` (1 GPU) or, for N > 1,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
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

REF_DATA = os.path.join(ROOT, "tests", "golden", "ref_data")
MACHINE = os.path.join(REF_DATA, "s16h74l4c4.json")
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# the 3-state adaptor that swallows hamming74's flush symbol '.' (not part of the reference)
DROPDOT = ('{"state":[{"n":0,"id":"S","trans":[{"in":"^","out":"^","to":1}]},{"n":1,"id":"T","trans":[{"in":"0","out":"0","to":1},'
           '{"in":"1","out":"1","to":1},{"in":".","to":1},{"in":"$","out":"$","to":2}]},{"n":2,"id":"U","trans":[]}]}')


def _compose(da, *parts):
    ms = [p if isinstance(p, da.Machine) else da.Machine.fromFile(os.path.join(REF_DATA, p)) for p in parts]
    m = ms[-1]
    for a in reversed(ms[:-1]):
        m = da.Machine.compose(a, m)
    return m


def workload(da, config, variant):
    """-> dict(machine, payload_bytes, default_reads, name)."""
    if config == 1:
        return dict(machine=_compose(da, "flusher.json", "mixradar6.json", "l4c4.json"), payload_bytes=128, default_reads=64,
                    name="configs[1]: ~980-nt reads through flusher*mixradar6*l4c4 (46670 states)")
    if config == 2:
        return dict(machine=da.Machine.fromFile(MACHINE), payload_bytes=29, default_reads=10000,
                    name="configs[2]: ~490-nt reads through s16h74l4c4.json (12361 states)")
    if config == 3 and variant == "b":
        return dict(machine=_compose(da, "hamming74.json", da.Machine.fromJSON(DROPDOT), "water64.1.json", "l4c4.json"), payload_bytes=32,
                    default_reads=24, name="configs[3] as written: ~1050-nt reads through hamming74*dropdot*water64.1*l4c4 (258538 states)")
    if config == 3:
        return dict(machine=_compose(da, "water64.1.json", "l4c4.json"), payload_bytes=56, default_reads=12500,
                    name="configs[3] (4a): ~1050-nt reads through water64.1*l4c4 (7066 states)")
    raise SystemExit("unknown --config %r" % config)


def make_reads(machine, first_index, count, sub_rate=0.01, payload_bytes=29):
    """Payload -> exact encoding -> i.i.d. substitutions.  Deterministic per read index."""
    reads = []
    for i in range(first_index, first_index + count):
        rng = random.Random(1000 + i)
        payload = bytes(rng.randrange(256) for _ in range(payload_bytes))
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


# ----------------------------------------------------------------------------- CPU baseline (the oracle = the port)
def host_cores():
    """Host cores this job may really use: the affinity mask, cut by the cgroup CPU quota, and by the 16-core share a
    one-GPU box of this pool gives a job (DNAS_BENCH_CORES overrides).  Oversubscribing would understate the CPU."""
    if os.environ.get("DNAS_BENCH_CORES"):
        return max(1, int(os.environ["DNAS_BENCH_CORES"]))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 16))


def _cpu_worker(job):
    """One host core: build the oracle for the machine, decode its reads, return (seconds, results)."""
    machine_json, global_, reads = job
    from oracle import oracle as O
    orc = O.ViterbiOracle(O.Machine.from_json(machine_json), O.MutatorParams.from_cli(global_=global_))
    t0 = time.perf_counter()
    res = [orc.decode(r) for r in reads]
    return time.perf_counter() - t0, res


def cpu_baseline(machine_json, reads, seconds, max_nt=None):
    """The CPU oracle (the port of the reference's algorithm) on a bounded sample of the same workload: first one
    thread for ~seconds/3 (the one-core rate, which also sizes the sample), then one process per host core over
    disjoint reads for ~seconds.  Checker code: used here only as the reported baseline and for the parity check.
    max_nt: decode only a prefix of each read (machines whose single read takes the CPU a minute)."""
    import multiprocessing as mp
    from oracle import oracle as O
    O.build()
    if max_nt:
        reads = [r[:max_nt] for r in reads]
    cores = host_cores()
    orc = O.ViterbiOracle(O.Machine.from_json(machine_json), O.MutatorParams.from_cli(global_=True))
    nt1, n1, results = 0, 0, {}
    t0 = time.perf_counter()
    for i, r in enumerate(reads):
        results[i] = orc.decode(r)
        nt1 += len(r)
        n1 += 1
        if time.perf_counter() - t0 > seconds / 3.0:
            break
    dt1 = time.perf_counter() - t0
    rate1 = nt1 / dt1
    mean_len = max(1.0, nt1 / n1)
    per_core = max(1, int(seconds * rate1 / mean_len))
    jobs, at = [], n1
    for _ in range(cores):
        chunk = reads[at:at + per_core]
        if not chunk:
            break
        jobs.append((machine_json, True, chunk))
        at += len(chunk)
    out = dict(value=rate1, unit="nt/s", cores=1, kind="port", value_one_core=rate1,
               sample="%d reads (%d nt%s) of the same batch, oracle/viterbi_oracle.c, 1 thread, %.1f s" % (
                   n1, nt1, ", first %d nt of each" % max_nt if max_nt else "", dt1))
    if len(jobs) >= 2:
        ctx = mp.get_context("spawn")
        with ctx.Pool(len(jobs)) as pool:
            pool.map(_cpu_worker, [(machine_json, True, j[2][:1]) for j in jobs])     # start-up (imports, machine parse) untimed
            t0 = time.perf_counter()
            parts = pool.map(_cpu_worker, jobs)
            wall = time.perf_counter() - t0
        nt = sum(len(r) for j in jobs for r in j[2])
        pos = n1
        for (_, res), j in zip(parts, jobs):
            for k, rr in enumerate(res):
                results[pos + k] = rr
            pos += len(j[2])
        out.update(value=nt / wall, cores=len(jobs),
                   sample="%d reads (%d nt%s) of the same batch over %d processes (one per host core), oracle/viterbi_oracle.c, %.1f s wall; "
                          "one core alone: %.0f nt/s" % (sum(len(j[2]) for j in jobs), nt, ", first %d nt of each" % max_nt if max_nt else "",
                                                          len(jobs), wall, rate1))
    return out, results


class Ctx:
    """Ranks, devices and the process group of this run (one process per GPU)."""

    def __init__(self, gpus):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (gpus, self.world))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool's driver
        # DNAS_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (the ranks
        # share the cards, the collectives run on host copies); the measured configuration is always nccl (= RCCL)
        self.backend = os.environ.get("DNAS_BENCH_BACKEND", "nccl")
        n_dev = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
        if n_dev <= 0:
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        if self.backend != "nccl":
            self.local_rank %= n_dev
        self.device = torch.device("cuda", self.local_rank)
        self.coll_device = self.device if self.backend == "nccl" else torch.device("cpu")
        if self.world > 1:
            # the process group first: nothing of this process has touched a GPU before RCCL binds the rank to its device
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        torch.cuda.set_device(self.local_rank)
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")

    def describe(self, per_rank):
        """What the process group really was (a scaling record must show that RCCL saw N ranks)."""
        d = {"world_size": self.world, "backend": None, "per_rank": per_rank, "devices_visible": self.torch.cuda.device_count()}
        if self.world > 1:
            d["backend"] = self.dist.get_backend()
            d["world_size"] = self.dist.get_world_size()
        return d

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def viterbi_line(ctx, config, variant, n_reads, steps, warmup, cpu_seconds, timed_only=False, scatter=False, arena_gb=0.0, options=None,
                 dump_decoded=None):
    """One BASELINE Viterbi configuration on this run's GPUs -> the JSON line as a dict (rank 0; None elsewhere)."""
    import dnastore_amd as da
    from dnastore_amd import shard
    torch, dist = ctx.torch, ctx.dist
    rank, world, device, coll_device = ctx.rank, ctx.world, ctx.device, ctx.coll_device

    wl = workload(da, config, variant)
    machine = wl["machine"]
    n_reads = n_reads or wl["default_reads"]
    params = da.MutatorParams.fromFlags(global_=True)
    dec = da.ViterbiDecoder(machine, params, device=ctx.local_rank, arena_bytes=int(arena_gb * 1e9), options=options)

    # ---- inputs (untimed): every rank makes its own reads by index; --scatter: rank 0 makes all, RCCL scatter
    my_reads = None
    if scatter:
        all_reads = make_reads(machine, 0, n_reads * world, payload_bytes=wl["payload_bytes"]) if rank == 0 else None
        off_all, bases_all = da.pack_reads(all_reads) if rank == 0 else (None, None)
        idx, off, d_bases = shard.scatter_reads(off_all, bases_all, world, rank, coll_device)
        d_bases = d_bases.to(device)
        if rank == 0:
            my_reads = [all_reads[int(i)] for i in idx]
    else:
        my_reads = make_reads(machine, rank * n_reads, n_reads, payload_bytes=wl["payload_bytes"])
        off, bases = da.pack_reads(my_reads)
        d_bases = torch.from_numpy(bases).to(device)
    torch.cuda.synchronize()   # the library launches on its own streams: the reads must have landed
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

    torch.cuda.synchronize()   # torch's fills of the output buffers are done before the library's streams write them
    for _ in range(warmup):
        step()
    ctx.fence()
    t0 = time.perf_counter()
    fill_ms = 0.0
    tb_ms = 0.0
    stats = None
    gathered = None
    for _ in range(steps):
        gathered = step()
        stats = dec.stats()
        fill_ms += stats["fill_ms"]
        tb_ms += stats["traceback_ms"]
    ctx.fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(shard_nt)], dtype=torch.float64, device=coll_device)
    per_rank = [{"rank": 0, "value": shard_nt * steps / elapsed, "seconds": elapsed, "nt": shard_nt, "fill_ms": fill_ms}]
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        # every rank's own clock and shard, so that a scaling record shows what each of the N ranks did (all_gather: RCCL)
        mine = torch.tensor([elapsed, float(shard_nt), fill_ms], dtype=torch.float64, device=coll_device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [{"rank": r, "value": float(e[1]) * steps / float(e[0]), "seconds": float(e[0]), "nt": int(e[1]), "fill_ms": float(e[2])}
                    for r, e in enumerate(every)]
        elapsed, total_nt = float(tmax[0]), float(tsum[1])
    else:
        total_nt = float(shard_nt)

    line = None
    if rank == 0:
        extra = {}
        cpu = None
        if dump_decoded and not scatter:
            # what came back from every rank, in read-index order (rank r made the reads r * n_reads ...): tests compare runs
            # with different numbers of ranks
            dump = []
            for sym_r, len_r, ll_r, st_r in gathered:
                sym_r, len_r, ll_r, st_r = (x.cpu().numpy() for x in (sym_r, len_r, ll_r, st_r))
                for i in range(len(len_r)):
                    dump.append([sym_r[i * cap:i * cap + int(len_r[i])].tobytes().decode(), float(ll_r[i]).hex(), int(st_r[i])])
            with open(dump_decoded, "w") as f:
                json.dump(dump, f)
        if world == 1 and not timed_only:
            # ---- the same shard through the host-pointer entry point (dnas_viterbi_batch): H2D of the reads, D2H of the
            # decoded strings, per-call device buffers -- SURVEY 8(d)'s PCIe-inclusive rate; never `value`
            # (packed host arrays in, host arrays out: what a C caller hands over -- Python's string handling is not part of it; cap per
            #  read as in the timed steps)
            h_off, h_bases = da.pack_reads(my_reads)
            dec.decode_packed(h_off[:min(k, 64) + 1], h_bases, out_cap=cap)
            tp = time.perf_counter()
            sym_h, _, len_h, ll_h, st_h = dec.decode_packed(h_off, h_bases, out_cap=cap)
            extra["value_pcie_inclusive"] = shard_nt / (time.perf_counter() - tp)
            # ... and must say what the resident-input steps said
            g_sym, g_len, g_ll, g_st = [x.cpu().numpy() for x in gathered[0]]
            if not (np.array_equal(len_h, g_len[:k]) and np.array_equal(ll_h.view(np.uint64), g_ll[:k].view(np.uint64)) and
                    np.array_equal(sym_h[:k * cap].reshape(k, cap)[np.arange(cap)[None, :] < len_h[:, None]],
                                   g_sym[:k * cap].reshape(k, cap)[np.arange(cap)[None, :] < len_h[:, None]])):
                raise SystemExit("PARITY FAILURE: dnas_viterbi_batch (host buffers) and dnas_viterbi_batch_device disagree")
            # ---- BASELINE configs[1] names ONE read: its latency
            if config == 1:
                # the default plan is the throughput one (512-thread work-groups: the machine on 4 CUs per read); one read alone is
                # decoded soonest on MORE, smaller members -- 14 work-groups of 1024 threads, 4 rows per thread (measured, fill of
                # one ~980-nt read with the final kernel: 28.5-28.7 ms; 16 x 1024: 29.0-29.6; 16 x 512: 33.9-34.7; 10 x 1024: 32.7-34.4;
                # 18 ... 32 x 1024: 33-40; profiles/experiments/r4_single_read_latency.txt.  Before the cluster's epoch was bumped once
                # per work-group instead of once per wave, wide members lost: 16 x 1024 took 71 ms)
                dec_lat = da.ViterbiDecoder(machine, params, device=ctx.local_rank, options="threads=1024,cluster=14")
                dec_lat.decode(my_reads[:1])
                walls, fills, split = [], [], []
                for _ in range(3):           # median of three (a cluster whose members land on more than one XCD is slower: the census says)
                    tp = time.perf_counter()
                    dec_lat.decode(my_reads[:1])
                    walls.append((time.perf_counter() - tp) * 1e3)
                    fills.append(dec_lat.stats()["fill_ms"])
                    split.append(dec_lat.cluster_census()[1])
                extra["latency_ms_single_read"] = sorted(walls)[1]
                extra["fill_ms_single_read"] = sorted(fills)[1]
                extra["single_read_runs"] = {"wall_ms": walls, "fill_ms": fills, "clusters_split_over_xcds": split}
                extra["single_read_plan"] = dec_lat.tier[:60]
                dec_lat.close()
            # ---- parity spot check + CPU baseline (rank 0, N = 1 only), outside the timed region
            if cpu_seconds > 0:
                max_nt = 256 if machine.nStates() > 100000 else None
                cpu, results = cpu_baseline(machine.toJSON(), my_reads, cpu_seconds, max_nt=max_nt)
                sym, olen, ll, st = [x.cpu().numpy() for x in gathered[0]]
                if max_nt:      # prefixes were decoded on the CPU: decode the same prefixes on the GPU
                    idxs = sorted(results)
                    pre_out, pre_ll, _ = dec.decode([my_reads[i][:max_nt] for i in idxs])
                    got = {i: (pre_out[j], float(pre_ll[j])) for j, i in enumerate(idxs)}
                else:
                    got = {i: (sym[i * cap:i * cap + int(olen[i])].tobytes().decode(), float(ll[i])) for i in results}
                for i, (s_ref, ll_ref) in results.items():
                    if got[i][0] != s_ref or got[i][1] != ll_ref:
                        raise SystemExit("PARITY FAILURE on read %d: %r/%r vs oracle %r/%r" % (i, got[i][0], got[i][1], s_ref, ll_ref))
                cpu["parity_checked_reads"] = len(results)
        value = total_nt * steps / elapsed
        launches = stats["fill_launches"] * steps
        achieved = stats["lattice_bytes"] * steps / (fill_ms / 1e3) / 1e9 if fill_ms > 0 else 0.0
        tier = dec.tier[:6]
        # HBM traffic of the fill kernel from PMC counters is collected in separate rocprofv3 passes (profiles/README.md)
        # and is NOT measured in this run: the field carries a recorded per-column figure only when a profile of this
        # configuration and kernel specialisation exists, with its source named; otherwise null.
        traffic, traffic_source = None, None
        for rnd in ("r4", "r3", "r2"):
            try:
                src = os.path.join("profiles", "%s_traffic_config%d%s.json" % (rnd, config, variant if config == 3 else ""))
                tj = json.load(open(os.path.join(ROOT, src)))
                if tj.get("tier") == tier:
                    traffic = tj["hbm_bytes_per_column_corrected"] * stats["columns"] / max(stats["fill_launches"], 1)
                    traffic_source = "%s (separate rocprofv3 --pmc passes of this command, per column, scaled to this run's columns per launch)" % src
                    break
            except (OSError, ValueError, KeyError):
                pass
        line = {
            "metric": "decoded nt/sec (whole node), Viterbi on composite FST",
            "value": value, "unit": "nt/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %d reads/GPU, --error-global, 1%% substitutions" % (wl["name"], n_reads),
                       "reads_per_gpu": n_reads, "total_nt": int(total_nt), "parallelism": "read-sharded x%d" % world,
                       "inputs": ("rank 0 scatter (%s)" % ("RCCL" if ctx.backend == "nccl" else ctx.backend)) if scatter else "generated per rank by read index",
                       # which fill kernel and row program served the run, and the tuning record that chose it
                       "program": dec.tier},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "frac_whole_step": stats["lattice_bytes"] * steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "kernel": "viterbi_fill_tiera" if tier in ("tier A", "tier C") else "viterbi_fill_kernel",
                         "tier": tier, "avg_launch_ms": fill_ms / max(launches, 1),
                         "algorithmic_bytes_per_launch": stats["lattice_bytes"] / max(stats["fill_launches"], 1),
                         "algorithmic_bytes_per_column": 8 * (dec.max_dup_len + 2) * dec.n_states,
                         "rounds_per_column": stats["rounds"] / max(stats["columns"], 1),
                         "traceback_ms_per_step": tb_ms / steps},
            "cpu_baseline": cpu,
            "distributed": ctx.describe(per_rank),
        }
        line.update(extra)
    dec.close()
    del d_sym, d_len, d_ll, d_st, d_bases
    torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[1, 2, 3, 4], help="index into BASELINE.json configs (default 2: the headline)")
    ap.add_argument("--variant", default="a", choices=["a", "b"], help="config 3: a = water64.1*l4c4, b = hamming74*dropdot*water64.1*l4c4")
    ap.add_argument("--reads", type=int, default=0, help="reads (config 4: pairs) per GPU; 0 = the configuration's default")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline time budget (0 = skip)")
    ap.add_argument("--arena-gb", type=float, default=0.0, help="lattice arena per GPU (0 = the library's default, 60 %% of free HBM)")
    ap.add_argument("--scatter", action="store_true", help="rank 0 makes every rank's reads and scatters them (RCCL) instead of per-rank generation")
    ap.add_argument("--timed-only", action="store_true", help="skip the passes outside the timed region (PCIe-inclusive, single-read latency, CPU baseline, "
                                                              "the other configurations): every launch of the run is then one of the timed steps (profiling)")
    ap.add_argument("--no-other-configs", action="store_true", help="the headline run alone: skip the short passes of the other BASELINE configurations")
    ap.add_argument("--options", default=None, help='dnas_model_create_ex options, e.g. "max_slots=690"')
    ap.add_argument("--dump-decoded", default=None, help="rank 0 writes the gathered results (decoded string, log-likelihood bits, status per read, "
                                                         "in read-index order) to this file as JSON (tests)")
    args = ap.parse_args()

    ctx = Ctx(args.gpus)
    failed = []
    if args.config == 4:
        import bench_fwdback
        line = bench_fwdback.fwdback_line(ctx, args.reads, args.steps, args.warmup, args.cpu_seconds, args.timed_only)
    else:
        line = viterbi_line(ctx, args.config, args.variant, args.reads, args.steps, args.warmup, args.cpu_seconds, args.timed_only,
                            args.scatter, args.arena_gb, args.options, args.dump_decoded)
    # ---- the default run (the one the driver makes) also carries a short pass of every other BASELINE configuration, so that
    # one driver-observed line holds them all: value, roofline, CPU baseline and parity count each.  One GPU only.
    if ctx.world == 1 and args.config == 2 and not args.timed_only and not args.no_other_configs and not args.reads:
        import bench_fwdback
        others = {}
        t_all = time.perf_counter()
        for name, fn in (
                ("configs[1]", lambda: viterbi_line(ctx, 1, "a", 64, 1, 1, 4.0)),
                ("configs[3] (water64.1*l4c4)", lambda: viterbi_line(ctx, 3, "a", 6255, 1, 1, 4.0)),
                ("configs[3] as written (hamming74*dropdot*water64.1*l4c4)", lambda: viterbi_line(ctx, 3, "b", 24, 1, 0, 4.0)),
                ("configs[4]", lambda: bench_fwdback.fwdback_line(ctx, 0, 2, 1, 4.0, False))):
            t1 = time.perf_counter()
            try:
                o = fn()
                o["bench_seconds"] = time.perf_counter() - t1
            except (Exception, SystemExit) as e:       # a failure here is reported, the headline line still goes out -- and the run fails
                o = {"error": "%s: %s" % (type(e).__name__, e)}
                failed.append(name)
            others[name] = o
        line["other_configs"] = others
        line["other_configs_seconds"] = time.perf_counter() - t_all
        line["other_configs_failed"] = failed
    if ctx.rank == 0:
        print(json.dumps(line), flush=True)
    ctx.close()
    if failed:      # a parity failure or a crash in any configuration must not look like a successful run
        raise SystemExit("bench.py: %s failed (see other_configs[...].error in the line above)" % ", ".join(failed))


if __name__ == "__main__":
    main()
