"""The multi-rank path of bench.py as the driver launches it (python -m torch.distributed.run, one process per rank, rendezvous on
127.0.0.1), rehearsed with two ranks that share the one GPU of the test box: the collectives run over gloo on host copies
(DNAS_BENCH_BACKEND=gloo; with RCCL two ranks cannot share a device), everything else -- per-rank read generation by index, the
library on the rank's device, barrier + max-over-ranks timing, the gather of the decoded strings to rank 0, one JSON line from rank
0 -- is the code of the 8-GPU run.  What the two ranks decode together must equal what one rank decodes alone."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env, tmp_path, tag):
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, "%s failed:\n%s" % (tag, p.stderr.decode(errors="replace")[-3000:])
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "%s: rank 0 prints ONE JSON line, got %d" % (tag, len(lines))
    return json.loads(lines[0])


def test_two_ranks_on_one_card_equal_one_rank(tmp_path):
    env = dict(os.environ, DNAS_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    two, one = str(tmp_path / "two.json"), str(tmp_path / "one.json")
    common = ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--timed-only"]
    line2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--reads", "64", "--dump-decoded", two] + common, env, tmp_path, "2 ranks")
    line1 = _run([sys.executable, "bench.py", "--gpus", "1", "--reads", "128", "--dump-decoded", one] + common, env, tmp_path, "1 rank")
    assert line2["n_gpus"] == 2 and line2["scaling"] == "weak" and line2["steps"] == 1 and line2["unit"] == "nt/s"
    assert line2["config"]["reads_per_gpu"] == 64 and line2["config"]["total_nt"] == line1["config"]["total_nt"]
    assert line2["value"] > 0 and line2["roofline"]["frac"] > 0 and "program" in line2["config"]
    d2, d1 = json.load(open(two)), json.load(open(one))
    assert len(d2) == 128 and d2 == d1          # decoded strings, fp64 log-likelihood bits and status of every read


def test_two_ranks_over_rccl_when_two_gpus_are_there(tmp_path):
    """The measured configuration itself: two ranks, two GPUs, the real `nccl` backend (= RCCL over xGMI), reads scattered from rank
    0 and results gathered to it.  Needs two visible GPUs: skipped on a one-GPU box (the driver's scaling run is the first place
    where RCCL sees more than one rank; this test is collected there)."""
    import torch
    if torch.cuda.device_count() < 2:          # (counting devices does not initialise the GPU)
        pytest.skip("needs 2 GPUs (RCCL cannot put two ranks on one device)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("DNAS_BENCH_BACKEND", None)
    common = ["--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--timed-only"]
    for extra, tag in (([], "per-rank reads"), (["--scatter"], "scatter")):
        two = str(tmp_path / ("two_%s.json" % tag.split()[0]))
        line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--reads", "64", "--dump-decoded", two] + common + extra,
                    env, tmp_path, "2 ranks over nccl, " + tag)
        d = line["distributed"]
        assert d["backend"] == "nccl" and d["world_size"] == 2 and len(d["per_rank"]) == 2
        assert all(r["value"] > 0 and r["nt"] > 0 for r in d["per_rank"])
        assert line["n_gpus"] == 2 and line["value"] > 0
    one = str(tmp_path / "one.json")
    _run([sys.executable, "bench.py", "--gpus", "1", "--reads", "128", "--dump-decoded", one] + common, env, tmp_path, "1 rank")
    d2, d1 = json.load(open(str(tmp_path / "two_per-rank.json"))), json.load(open(one))
    assert len(d2) == 128 and d2 == d1          # what two GPUs decode together equals what one decodes alone


def test_line_says_what_the_process_group_was(tmp_path):
    line = _run([sys.executable, "bench.py", "--gpus", "1", "--reads", "64", "--steps", "1", "--warmup", "0", "--cpu-seconds", "0", "--timed-only"],
                dict(os.environ), tmp_path, "1 rank")
    d = line["distributed"]
    assert d["world_size"] == 1 and d["backend"] is None and len(d["per_rank"]) == 1 and d["per_rank"][0]["value"] > 0
