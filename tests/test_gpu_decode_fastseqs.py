"""decodeFastSeqs behind the C ABI beyond one device and beyond the decoded strings: device_id = -1 (the reads of a
file dealt over every GPU, one host thread and model per device -- on a one-GPU box several threads share the card,
DNAS_FAKE_DEVICES), the fill tier reported with the result, and the traceback's event log (the reference's level-3
messages, viterbi.cpp:266-293)."""
import os
import random
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dnastore_amd", "bin", "dnastore")


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def test_all_devices_matches_one_device(da, ref_data, tmp_path, monkeypatch):
    m = da.Machine.fromFile(os.path.join(ref_data, "h74l4c4.json"))
    params = da.MutatorParams.fromFlags(global_=True)
    rng = random.Random(3)
    fa = tmp_path / "reads.fa"
    with open(fa, "w") as f:
        for i in range(23):                                    # ragged lengths, not a multiple of the device count
            dna = m.encodeBytes(bytes(rng.randrange(256) for _ in range(1 + i % 7)))
            f.write(">read%d some comment\n%s\n" % (i, dna))
        f.write(">empty\n\n")
    one_info, all_info = {}, {}
    one = da.decode_fastseqs(fa, m, params, device=0, info=one_info)
    monkeypatch.setenv("DNAS_FAKE_DEVICES", "3")               # three host threads / models share the GPUs there are
    everywhere = da.decode_fastseqs(fa, m, params, device=-1, info=all_info)
    assert everywhere == one                                   # names, strings, fp64 log-likelihoods, file order
    assert [n for n, _, _ in one][:2] == ["read0", "read1"] and one_info["devices"] == 1 and all_info["devices"] == 3
    assert one_info["tier"].startswith("tier A") and all_info["tier"].startswith("tier A")
    assert da.lib.lib().dnas_device_count() >= 1


@pytest.mark.parametrize("kernel", ["wave", "thread"])
def test_traceback_event_log(da, ref_data, kernel, monkeypatch):
    """A substituted, a deleted and a duplicated base in the reference's own reads are found where they are -- by the
    wave-per-read traceback (small batches) and by the thread-per-read one (large batches) alike."""
    monkeypatch.setenv("DNAS_TRACEBACK", kernel)
    def events(mach, fa, **flags):
        m = da.Machine.fromFile(os.path.join(ref_data, mach))
        recs = da.decode_fastseqs(os.path.join(ref_data, fa), m, da.MutatorParams.fromFlags(**flags), events=True)
        return recs[0][3]
    # hello.h74.sub.fa is hello.h74.fa with base 25 changed A -> G
    assert events("h74l4c4.json", "hello.h74.sub.fa") == ["Substitution at 25: A -> G"]
    assert events("h74l4c4.json", "hello.h74.fa") == []
    # hello.s16h74.del.fa is hello.s16h74.fa without its base 29, a G
    assert events("s16h74l4c4.json", "hello.s16h74.del.fa") == ["Deletion between 28 and 29: G"]
    # hello.dup.fa is hello.fa with the two bases before position 10 written twice
    ev = events("l4c4.json", "hello.dup.fa", sub=0., del_open=0., global_=True)
    assert len(ev) == 1 and ev[0].startswith("Duplication at ") and ev[0].endswith(": GC")


def test_cli_verbose_3_prints_events_and_tier(ref_data):
    r = subprocess.run([BIN, "-v3", "--load-machine", os.path.join(ref_data, "h74l4c4.json"), "--decode-viterbi",
                        os.path.join(ref_data, "hello.h74.sub.fa"), "--raw", "--device", "-1"], capture_output=True, timeout=600)
    want = open(os.path.join(ref_data, "hello.exact.bits"), "rb").read()
    err = r.stderr.decode()
    assert r.returncode == 0 and r.stdout == want
    assert "Substitution at 25: A -> G" in err and "Viterbi fill: tier A" in err


def test_cli_device_failures_are_not_success(ref_data):
    """A failure that is not one of the reference's exceptions must not exit 0 with an empty stdout."""
    r = subprocess.run([BIN, "-v0", "--load-machine", os.path.join(ref_data, "l4c4.json"), "--decode-viterbi",
                        os.path.join(ref_data, "hello.fa"), "--device", "99"], capture_output=True, timeout=600)
    assert r.returncode != 0 and r.stdout == b""
    r = subprocess.run([BIN, "-v0", "--fit-error", os.path.join(ref_data, "no_such.stk")], capture_output=True, timeout=600)
    assert r.returncode == 1 and b"File not found" in r.stderr
