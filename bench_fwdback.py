"""bench.py --config 4: the forward-backward E-step (expectedCounts, reference src/fwdback.cpp:190-209) over synthetic
(original, read) pairs -- BASELINE configs[4], SURVEY 8(d) "Config 5": originals of 256 random nt, reads with tandem
duplications (length 1-3, rate .01), substitutions (.02) and deletions (.01), the true alignment as the guide, the CLI
default error model (P = 6).  125 000 pairs per GPU by default (1M over 8 GPUs); 16 000 distinct synthetic pairs
are tiled to that number (making a million alignments in Python would take longer than the measurement).

A "step" is one E-step over the shard with the database resident in HBM (dnas_fb handle: table and pairs uploaded
once); `value` = read (output) nt per second.  The roofline object reports the bound the kernel is actually under -- fp64
vector issue (`issue`) -- beside the HBM figure the contract asks for.
With N > 1 ranks every rank makes its own pairs; the 22+P counts are all-reduced (RCCL) inside the timed region.
"""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0


def make_pairs(O, first, count, unique):
    from synth import synthetic_alignment
    uniq = []
    for u in range(min(unique, count)):
        rng = random.Random(50000 + (first + u) % unique)
        uniq.append(O.alignment_pair(synthetic_alignment(rng, 256, sub=.02, dele=.01, dup=.01)))
    return [uniq[i % len(uniq)] for i in range(count)]


def _cpu_worker(job):
    pairs, = job
    from oracle import oracle as O
    t0 = time.perf_counter()
    res = O.expected_counts(O.MutatorParams.from_cli(), pairs)
    return time.perf_counter() - t0, res


def fwdback_line(ctx, n_pairs, steps, warmup, cpu_seconds, timed_only):
    """BASELINE configs[4] on this run's GPUs -> the JSON line as a dict (rank 0; None elsewhere).  ctx: bench.Ctx."""
    import dnastore_amd as da
    from dnastore_amd import shard
    from oracle import oracle as O          # pair packing + CPU baseline (checker code, outside the timed region)
    import bench
    torch, dist = ctx.torch, ctx.dist
    rank, world, local_rank, coll_device = ctx.rank, ctx.world, ctx.local_rank, ctx.coll_device

    n_pairs = n_pairs or 125000
    unique = 16000      # distinct synthetic pairs (42 MB of sequences and guide columns: more than the chip's L2), tiled to n_pairs
    pairs = make_pairs(O, rank * n_pairs, n_pairs, unique)
    pk = O.pack_pairs(pairs)
    nt = int(pk["out_off"][-1])
    params = da.MutatorParams.fromFlags()
    fb = da.ForwardBackward(pk, device=local_rank)

    def step():
        counts, ll, _ = fb.expectedCounts(params, want_pair_ll=False)
        return shard.allreduce_counts(counts, ll, world, coll_device)

    fence = ctx.fence
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms, lse_ops = 0.0, 0
    res = None
    for _ in range(steps):
        res = step()
        st = fb.stats()
        kernel_ms += st["kernel_ms"]
        lse_ops += st["lse_ops"]
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed, float(nt), float(n_pairs)], dtype=torch.float64, device=coll_device)
    elapsed_rank, per_rank = elapsed, None
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank = [{"rank": r, "value": float(e[1]) * steps / float(e[0]), "seconds": float(e[0]), "nt": int(e[1])} for r, e in enumerate(every)]
        elapsed, total_nt, total_pairs = float(tmax[0]), float(tsum[1]), float(tsum[2])
    else:
        total_nt, total_pairs = float(nt), float(n_pairs)

    line = None
    if rank == 0:
        st = fb.stats()
        cpu, extra = None, {}
        if world == 1 and not timed_only:
            # PCIe-inclusive: the database goes to the GPU (dnas_fb_load_pairs: upload + checks), one E-step runs on it (the first on a
            # database: the envelope census and the routing included) and the counts come back -- a fresh handle, whose creation (stream,
            # the 800 KB log-sum-exp table: once per process and device) is not timed.  The one-call convenience form
            # dnas_fwdback_estep (handle + all of the above + teardown per call) is reported beside it.
            da.ForwardBackward(pk, device=local_rank).close()          # (warm: the first upload of a process pays for the driver's staging buffers)
            fb2 = da.ForwardBackward(None, device=local_rank)
            tp = time.perf_counter()
            fb2.load(pk)
            fb2.expectedCounts(params, want_pair_ll=False)
            extra["value_pcie_inclusive"] = nt / (time.perf_counter() - tp)
            fb2.close()
            tp = time.perf_counter()
            c1, ll1, per1 = da.expectedCounts(params, pk, device=local_rank)
            extra["value_one_call_form"] = nt / (time.perf_counter() - tp)
            if cpu_seconds > 0:
                import multiprocessing as mp
                O.build()
                oparams = O.MutatorParams.from_cli()
                t1 = time.perf_counter()
                n1 = 0
                while time.perf_counter() - t1 < cpu_seconds / 3.0 and n1 < n_pairs:
                    O.expected_counts(oparams, pairs[n1:n1 + 50])
                    n1 += 50
                dt1 = time.perf_counter() - t1
                nt1 = int(sum(len(p[1]) for p in pairs[:n1]))
                rate1 = nt1 / dt1
                cores = bench.host_cores()
                per_core = max(50, int(cpu_seconds * (n1 / dt1)))
                jobs = [(pairs[(c * per_core) % max(n_pairs - per_core, 1):][:per_core],) for c in range(cores)]
                mpctx = mp.get_context("spawn")
                with mpctx.Pool(len(jobs)) as pool:
                    pool.map(_cpu_worker, [(j[0][:2],) for j in jobs])
                    t2 = time.perf_counter()
                    parts = pool.map(_cpu_worker, jobs)
                    wall = time.perf_counter() - t2
                ntc = int(sum(len(p[1]) for j in jobs for p in j[0]))
                cpu = dict(value=ntc / wall, unit="nt/s", cores=len(jobs), kind="port", value_one_core=rate1,
                           sample="%d pairs (%d nt) over %d processes (one per host core), oracle/fwdback_oracle.c, %.1f s wall; one core alone: %.0f nt/s"
                                  % (sum(len(j[0]) for j in jobs), ntc, len(jobs), wall, rate1))
                # parity: per-pair log-likelihoods bit for bit, counts to 1e-9, on a sample
                m = min(300, n_pairs)
                oc, oll, oper = O.expected_counts(oparams, pairs[:m])
                if not np.array_equal(per1[:m], oper):
                    raise SystemExit("PARITY FAILURE: per-pair log-likelihoods differ from the oracle")
                gc, gll, _ = da.expectedCounts(params, O.pack_pairs(pairs[:m]), device=local_rank)
                if not np.allclose(gc, oc, rtol=1e-9, atol=1e-300):
                    raise SystemExit("PARITY FAILURE: counts differ from the oracle")
                cpu["parity_checked_pairs"] = m
        value = total_nt * steps / elapsed
        # algorithmic HBM bytes per pair: the two sequences (1 B/nt), the guide columns (4 B per position) and
        # the 22+P doubles that come back
        P = len(params.pLen)
        in_nt = int(pk["in_off"][-1])
        alg_bytes = (in_nt + nt) + 4 * (in_nt + nt + 2 * n_pairs) + 8 * (22 + P) * n_pairs
        # What bounds the kernel: not HBM (the algorithmic bytes are the inputs and the counts; the Forward cells it parks in HBM
        # between its two passes are ~0.6 MB per pair, a tenth of the chip's bandwidth at this rate) but the instruction stream of
        # the cells -- fp64 vector issue.  The instruction count per pair is a recorded figure (separate rocprofv3 --pmc passes of
        # this command, profiles/r3_sq_counters_fwdback.json), like `traffic` of the Viterbi lines; peak = one vector instruction per
        # SIMD every four cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 4.
        issue = None
        try:
            sq_file = next(f for f in ("r4_sq_counters_fwdback.json", "r3_sq_counters_fwdback.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            sq = json.load(open(os.path.join(ROOT, "profiles", sq_file)))
            per_pair = sq["derived"]["valu_per_pair"]
            peak = 256 * 4 * 2.4e9 / 4
            ach = per_pair * total_pairs * steps / elapsed / max(world, 1)
            issue = {"bound": "fp64 vector issue", "achieved": ach, "peak": peak, "unit": "wave instructions/s per GPU", "frac": ach / peak,
                     "valu_wave_instructions_per_pair": per_pair, "source": "profiles/%s (SQ_INSTS_VALU of one E-step / pairs)" % sq_file,
                     "valu_active_share_of_wave_cycles": sq["derived"].get("valu_issue_share_of_wave_cycles")}
        except (OSError, ValueError, KeyError, StopIteration):
            pass
        # HBM traffic of the E-step kernels: a recorded per-pair figure of separate rocprofv3 --pmc passes of this command (never
        # measured in the bench run), newest round first; null when no profile exists
        traffic, traffic_source = None, None
        for rnd in ("r4", "r3"):
            try:
                src = os.path.join("profiles", "%s_traffic_config4.json" % rnd)
                tj = json.load(open(os.path.join(ROOT, src)))
                traffic = tj["hbm_bytes_per_pair_corrected"] * total_pairs / max(world, 1)
                traffic_source = "%s (FETCH_SIZE x 2 + WRITE_SIZE per pair, separate rocprofv3 --pmc passes, scaled to this run's pairs per launch)" % src
                break
            except (OSError, ValueError, KeyError):
                pass
        line = {
            "metric": "forward-backward E-step, read nt/sec (whole node)",
            "value": value, "unit": "nt/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[4]: %d pairs/GPU of a 256-nt original and its read (dup .01, sub .02, del .01), guide = true alignment, "
                                   "CLI default model (P = 6), %d distinct pairs tiled" % (n_pairs, min(unique, n_pairs)),
                       "pairs_per_gpu": n_pairs, "total_nt": int(total_nt), "parallelism": "pair-sharded x%d, counts all-reduced" % world},
            "pairs_per_s": total_pairs * steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": alg_bytes * steps / (kernel_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes * steps / (kernel_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "fwdback_onchip8x16p6_kernel" if st.get("pairs_narrow", 0) * 2 > st["pairs_onchip"] else "fwdback_onchip16x16p6_kernel",
                         "avg_launch_ms": kernel_ms / steps,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "the algorithmic bytes are inputs and outputs only, so this fraction says nothing: the kernel is bound by the "
                                 "instruction stream of a cell (log-sum-exp look-ups, fp64 exp) -- see `issue`; lse_ops_per_s is its rate of "
                                 "log-sum-exp operations",
                         "issue": issue,
                         "lse_ops_per_s": lse_ops / (kernel_ms / 1e3) if kernel_ms > 0 else 0.0,
                         "pairs_onchip": st["pairs_onchip"], "pairs_narrow": st.get("pairs_narrow", 0), "pairs_streaming": st["pairs_streaming"]},
            "cpu_baseline": cpu,
            "distributed": ctx.describe([{"rank": 0, "value": nt * steps / elapsed_rank, "seconds": elapsed_rank, "nt": nt}] if per_rank is None else per_rank),
        }
        line.update(extra)
    fb.close()
    return line
