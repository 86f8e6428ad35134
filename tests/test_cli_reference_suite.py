"""The reference's own `make test` command lines (reference Makefile:116-186) run against this
repo's `dnastore` binary, compared with the reference's golden files the way t/testexpect.pl does
(stdout must equal the file).  CPU arms here; the -V / --error-counts / --fit-error arms need the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "dnastore_amd", "bin", "dnastore")
D = os.path.join(ROOT, "tests", "golden", "ref_data")

NOERRS = ["--error-sub-prob", "0", "--error-dup-prob", "0", "--error-del-open", "0", "--error-global"]   # Makefile:116-120
ONLYDUPS = ["--error-sub-prob", "0", "--error-del-open", "0", "--error-global"]                           # Makefile:121
TESTCOUNT = ["-l6", "--error-sub-prob", "1e-9", "--error-dup-prob", "1e-9", "--error-del-open", "1e-9"]


def d(name):
    return os.path.join(D, name)


def run(args, cwd=None):
    r = subprocess.run([BIN, "-v0"] + args, capture_output=True, cwd=cwd, timeout=600)
    return r.stdout, r.stderr.decode(errors="replace"), r.returncode


def expect(args, golden, fix_name=False):
    out, err, rc = run(args)
    want = open(d(golden), "rb").read()
    if fix_name:   # the golden FASTA headers carry the path the reference was given: data/hello.txt
        out = out.replace(D.encode(), b"data")
    assert out == want, err


def compose_args(*front):
    a = []
    for f in front:
        a += ["--compose-machine", d(f)]
    return a + ["--load-machine", d("l4c4.json")]


# ---------------------------------------------------------------- CPU arms
def test_testmachine_roundtrip():                                   # Makefile:135
    expect(["--load-machine", d("l4c4.json"), "--save-machine", "-"], "l4c4.json")


def test_testencode():                                              # Makefile:138-139
    expect(["--load-machine", d("l4c4.json"), "--encode-file", d("hello.txt")], "hello.fa", fix_name=True)
    expect(["--load-machine", d("l4c4.json"), "--raw", "--encode-string", "HELLO"], "hello.dna")


def test_testdecode():                                              # Makefile:142-144
    dna = open(d("hello.dna")).read().strip()
    expect(["--load-machine", d("l4c4.json"), "--decode-file", d("hello.fa")], "hello.txt")
    expect(["--load-machine", d("l4c4.json"), "--decode-string", dna], "hello.txt")
    expect(["--load-machine", d("l4c4.json"), "--decode-bits", dna], "hello.padded.bits")


@pytest.mark.parametrize("front,golden,fa", [
    (("mixradar2.json",), "mr2l4c4.json", "hello.mr2.fa"),                                   # testcompose, Makefile:151-153
    (("hamming74.json",), "h74l4c4.json", "hello.h74.fa"),                                   # testham, Makefile:166-168
    (("sync16.json", "flusher.json", "mixradar2.json"), "s16mr2l4c4.json", "hello.s16mr2.fa"),   # testsync, Makefile:174-176
    (("sync16.json", "flusher.json", "hamming74.json"), "s16h74l4c4.json", "hello.s16h74.fa"),   # testsyncham, Makefile:181-183
])
def test_compose_encode_decode(front, golden, fa):
    expect(compose_args(*front) + ["--save-machine", "-"], golden)
    expect(["--load-machine", d(golden), "--encode-file", d("hello.txt")], fa, fix_name=True)
    expect(["--load-machine", d(golden), "--decode-file", d(fa)], "hello.txt")


def test_config1_length4_decodes_hello():
    """BASELINE config 1: `dnastore -l 4 -d data/hello.fa` -> HELLO (the canonical -l 4 machine is data/l4c4.json)."""
    r = subprocess.run([BIN, "-l", "4", "-d", d("hello.fa")], capture_output=True, env=dict(os.environ, DNASTORE_L4C4=d("l4c4.json")))
    assert r.stdout == b"HELLO"


def test_error_conventions():
    out, err, rc = run(["--load-machine", d("nope.json")])
    assert rc == 1 and "File not found" in err                       # Fail -> exit(1)
    out, err, rc = run(["--length", "40", "--load-machine", d("l4c4.json")])
    assert rc == 1 and "Maximum context" in err


# ---------------------------------------------------------------- GPU arms
VITERBI = [
    ("l4c4.json", "hello.fa", NOERRS, "hello.padded.bits"),                    # testviterbi, Makefile:147-148
    ("l4c4.json", "hello.dup.fa", ONLYDUPS, "hello.padded.bits"),
    ("mr2l4c4.json", "hello.mr2.fa", NOERRS, "hello.exact.bits"),              # testcompose, Makefile:154
    ("h74l4c4.json", "hello.h74.fa", NOERRS, "hello.exact.bits"),              # testham, Makefile:169-171
    ("h74l4c4.json", "hello.h74.fa", [], "hello.exact.bits"),
    ("h74l4c4.json", "hello.h74.sub.fa", [], "hello.exact.bits"),
    ("s16mr2l4c4.json", "hello.s16mr2.fa", NOERRS, "hello.exact.bits"),        # testsync, Makefile:177-178
    ("s16mr2l4c4.json", "hello.s16mr2.fa", [], "hello.exact.bits"),
    ("s16h74l4c4.json", "hello.s16h74.fa", NOERRS, "hello.exact.bits"),        # testsyncham, Makefile:184-186
    ("s16h74l4c4.json", "hello.s16h74.fa", [], "hello.exact.bits"),
    ("s16h74l4c4.json", "hello.s16h74.del.fa", [], "hello.exact.bits"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("mach,fa,flags,golden", VITERBI)
def test_decode_viterbi(mach, fa, flags, golden):
    expect(["--load-machine", d(mach), "--decode-viterbi", d(fa)] + flags + ["--raw"], golden)


@pytest.mark.gpu
def test_decode_viterbi_fasta_output_and_no_path_warning():
    out, err, rc = run(["--load-machine", d("l4c4.json"), "--decode-viterbi", d("hello.fa")])
    assert out == b">data/hello.txt\n^00010010101000100011001000110010111100100$\n" and rc == 0
    # a read the no-error global model cannot explain: warning on stderr, empty sequence (viterbi.cpp:198-201)
    out, err, rc = run(["--load-machine", d("l4c4.json"), "--decode-viterbi", d("hello.dup.fa")] + NOERRS + ["--raw"])
    assert out == b"\n" and "No valid Viterbi decoding found" in err and rc == 0


@pytest.mark.gpu
@pytest.mark.parametrize("stk,golden", [("dup.stk", "dup.counts.json"), ("dup.sub.stk", "dup.sub.counts.json"),
                                        ("dup.sub.misaligned.stk", "dup.sub.counts.misaligned.json")])
def test_testcount(stk, golden):                                    # Makefile:157-159
    expect(TESTCOUNT + ["--error-counts", d(stk)], golden)


@pytest.mark.gpu
@pytest.mark.parametrize("stk,golden", [("tiny.stk", "tiny.params.json"), ("test.stk", "test.params.json")])
def test_testfit(stk, golden):                                      # Makefile:162-163
    expect(["--fit-error", d(stk), "--strict-guides"], golden)
