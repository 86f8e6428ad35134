"""CPU-side tests of the product: the C-ABI library loads and exports every declared symbol, and
its host logic (JSON loader, flattening, FASTA reader, encoder, error-model flags) agrees with the
oracle's independent Python/C restatement and with the reference's fixtures.  No GPU compute."""
import gzip
import math
import os

import numpy as np
import pytest

import dnastore_amd as da
from dnastore_amd import lib as L
from viterbi_cases import MACHINE_STATS


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    names = L.declared_symbols()
    assert len(names) >= 30
    assert [n for n in names if not hasattr(lib, n)] == []
    assert lib.dnas_has_device_code() == 1


@pytest.mark.parametrize("mach", sorted(MACHINE_STATS))
@pytest.mark.parametrize("flags", [dict(), dict(global_=True, sub=0., dup=0., del_open=0.), dict(length=6)])
def test_flatten_matches_oracle(oracle_mod, ref_data, mach, flags):
    """dnas_flatten (C++) vs orc_model_create (C) fed by two independent JSON readers."""
    O = oracle_mod
    path = os.path.join(ref_data, mach)
    a = da.FlatModel(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags)).arrays()
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    n, ne, nn, alph = MACHINE_STATS[mach]
    assert (a["n_states"], a["n_emit"], a["n_null"], a["alphabet"]) == (n, ne, nn, alph)
    assert a["max_dup_len"] == orc.D and a["local"] == int(orc.params.local)
    assert np.array_equal(a["scores"].view(np.uint64), orc.scores().view(np.uint64))     # incl. -inf entries
    for c in alph:
        assert a["sym_logp"][ord(c)] == orc.sym_logp(c)
    # CSR consistency: in- and out-edge views hold the same multiset of edges
    for kind in ("e", "n"):
        ip, isrc = a[kind + "in_ptr"], a[kind + "in_src"]
        op, odst = a[kind + "out_ptr"], a[kind + "out_dst"]
        ins = sorted((int(isrc[e]), d) for d in range(n) for e in range(ip[d], ip[d + 1]))
        outs = sorted((s, int(odst[e])) for s in range(n) for e in range(op[s], op[s + 1]))
        assert ins == outs
    # reference enumeration order: ascending source inside each destination row
    for d in range(n):
        row = a["ein_src"][a["ein_ptr"][d]:a["ein_ptr"][d + 1]]
        assert np.all(np.diff(row) >= 0)
    assert sorted(a["topo"].tolist()) == list(range(n))


def test_flags_to_params_match_reference_defaults():
    p = da.MutatorParams.fromFlags()                      # t/dnastore.cpp:41-82,119-129
    assert p.local and p.pLen == [1 / 6] * 6
    assert p.c.p_transition == .01 * 10 / 11 and p.c.p_transversion == .01 / 11
    assert (p.c.p_del_open, p.c.p_del_extend, p.c.p_tan_dup) == (.001, .01, .001)
    assert not da.MutatorParams.fromFlags(global_=True).local
    assert da.MutatorParams.fromFlags(length=6).pLen == [1 / 3] * 3


def test_error_file_json(ref_data):
    p = da.MutatorParams.fromFile(os.path.join(ref_data, "tiny.params.json"))   # mutator.cpp:18-30
    assert p.local and p.c.p_del_open == 0.25 and p.c.p_del_extend == 0.5 and len(p.pLen) == 6


def test_lenient_json_and_roundtrip(ref_data):
    # sync16.json has no commas between states, flusher.json a trailing comma (gason leniency)
    assert da.Machine.fromFile(os.path.join(ref_data, "sync16.json")).nStates() == 19
    assert da.Machine.fromFile(os.path.join(ref_data, "flusher.json")).nStates() == 4
    # testmachine (reference Makefile:135): load -> save reproduces data/l4c4.json byte for byte
    text = open(os.path.join(ref_data, "l4c4.json")).read()
    assert da.Machine.fromJSON(text).toJSON() == text


def test_error_conventions(tmp_path, ref_data):
    with pytest.raises(da.DnasError) as e:
        da.Machine.fromFile(str(tmp_path / "missing.json"))
    assert e.value.code == -2                                             # Fail -> exit(1)
    with pytest.raises(da.DnasError) as e:
        da.Machine.fromJSON('{"state":[{"n":1,"trans":[]}]}')            # "State n=1 out of sequence"
    assert e.value.code == -3 and "out of sequence" in str(e.value)
    bad_ctx = '{"state":[{"n":0,"r":"A","trans":[{"out":"C","to":1}]},{"n":1,"trans":[]}]}'
    with pytest.raises(da.DnasError):
        da.Machine.fromJSON(bad_ctx)                                      # verifyContexts
    cyc = '{"state":[{"n":0,"trans":[{"to":1}]},{"n":1,"trans":[{"to":0},{"in":"$","out":"A","to":2}]},{"n":2,"trans":[]}]}'
    with pytest.raises(da.DnasError) as e:
        da.FlatModel(da.Machine.fromJSON(cyc), da.MutatorParams.fromFlags())
    assert e.value.code == -4 and "cyclic" in str(e.value)
    notdna = '{"state":[{"n":0,"trans":[{"in":"0","out":"0","to":1}]},{"n":1,"trans":[]}]}'
    with pytest.raises(da.DnasError) as e:
        da.FlatModel(da.Machine.fromJSON(notdna), da.MutatorParams.fromFlags())
    assert e.value.code == -5
    with pytest.raises(da.DnasError) as e:                                # no GPU here: loud, no fallback
        import torch
        if torch.cuda.is_available():
            pytest.skip("GPU present")
        da.ViterbiDecoder(da.Machine.fromFile(os.path.join(ref_data, "l4c4.json")), da.MutatorParams.fromFlags())
    assert e.value.code == -7


def test_fasta_reader(tmp_path, ref_data, oracle_mod):
    # multi-line record with comment (hello.s16h74.del.fa), gz, FASTQ, several records
    for fa in ("hello.s16h74.del.fa", "hello.h74.sub.fa", "hello.dup.fa"):
        got = da.read_fastseqs(os.path.join(ref_data, fa))
        assert got == oracle_mod.read_fasta(os.path.join(ref_data, fa))
        assert got[0][0] == "data/hello.txt"
    p = tmp_path / "x.fa.gz"
    with gzip.open(p, "wt") as f:
        f.write(">r1 first\nACGT\nAC\n>r2\nGG\n\n>r3\n")
    assert da.read_fastseqs(str(p)) == [("r1", "ACGTAC"), ("r2", "GG"), ("r3", "")]
    q = tmp_path / "x.fq"
    q.write_text("@q1 c\nACGT\n+\n@@II\n@q2\nGG\n+q2\n!!\n")
    assert da.read_fastseqs(str(q)) == [("q1", "ACGT"), ("q2", "GG")]
    with pytest.raises(da.DnasError) as e:
        da.read_fastseqs(str(tmp_path / "nope.fa"))
    assert e.value.code == -2


@pytest.mark.parametrize("mach,fa", [("l4c4.json", "hello.fa"), ("mr2l4c4.json", "hello.mr2.fa"),
                                     ("h74l4c4.json", "hello.h74.fa"), ("s16mr2l4c4.json", "hello.s16mr2.fa"),
                                     ("s16h74l4c4.json", "hello.s16h74.fa")])
def test_encoder_goldens(ref_data, oracle_mod, mach, fa):
    """testencode-style goldens (reference Makefile:138,152,167,175,182): HELLO -> data/hello*.fa;
    the Python generator used by the parity tests must agree too."""
    from synth import bytes_to_symbols, encode
    want = da.read_fastseqs(os.path.join(ref_data, fa))[0][1]
    m = da.Machine.fromFile(os.path.join(ref_data, mach))
    assert m.encodeBytes(b"HELLO") == want
    assert m.encodeSymbols(bytes_to_symbols(b"HELLO")) == want
    assert encode(oracle_mod.Machine.from_file(os.path.join(ref_data, mach)), bytes_to_symbols(b"HELLO")) == want
    if mach == "l4c4.json":
        assert want == open(os.path.join(ref_data, "hello.dna")).read().strip()      # Makefile:139


def test_tokenize():
    assert da.tokenize("ACGTacgt").tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    with pytest.raises(ValueError):
        da.tokenize("ACGU")
