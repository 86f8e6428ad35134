"""Pin the CPU oracle against the reference's own golden vectors (CPU only)."""
import math
import os

import pytest

from viterbi_cases import MACHINE_STATS, VITERBI_GOLDENS


@pytest.mark.parametrize("mach,fa,flags,bits,loglike", VITERBI_GOLDENS)
def test_viterbi_goldens(oracle_mod, ref_data, mach, fa, flags, bits, loglike):
    O = oracle_mod
    m = O.Machine.from_file(os.path.join(ref_data, mach))
    v = O.ViterbiOracle(m, O.MutatorParams.from_cli(**flags))
    want = open(os.path.join(ref_data, bits)).read().strip()
    recs = O.read_fasta(os.path.join(ref_data, fa))
    assert len(recs) == 1
    got, ll = v.decode(recs[0][1])
    assert got == want                      # reference golden (testexpect.pl diff)
    assert ll == loglike                    # fp64-exact vs. the value captured from the reference


@pytest.mark.parametrize("mach", sorted(MACHINE_STATS))
def test_machine_scores_shape(oracle_mod, ref_data, mach):
    O = oracle_mod
    n, ne, nn, alph = MACHINE_STATS[mach]
    m = O.Machine.from_file(os.path.join(ref_data, mach))
    v = O.ViterbiOracle(m, O.MutatorParams.from_cli())
    assert m.n == n and v.edge_counts() == (ne, nn) and v.alphabet == alph and v.D == 4


def test_input_model_values(oracle_mod, ref_data):
    # s16h74l4c4 symbols each 0.25; l4c4 controls get 4^-24/norm -> log -34.657
    O = oracle_mod
    v = O.ViterbiOracle(O.Machine.from_file(os.path.join(ref_data, "s16h74l4c4.json")), O.MutatorParams.from_cli())
    for c in "$01^":
        assert v.sym_logp(c) == math.log(0.25)
    v = O.ViterbiOracle(O.Machine.from_file(os.path.join(ref_data, "l4c4.json")), O.MutatorParams.from_cli())
    cw = 4.0 ** -24
    norm = 4 + 2 * cw
    assert v.sym_logp("A") == math.log(cw / norm) and abs(v.sym_logp("A") + 34.657) < 1e-3
    assert v.sym_logp("0") == math.log(1 / norm)


def test_mutator_scores(oracle_mod, ref_data):
    # mutator.cpp:56-75 with CLI defaults (dnastore.cpp:119-129)
    O = oracle_mod
    v = O.ViterbiOracle(O.Machine.from_file(os.path.join(ref_data, "l4c4.json")), O.MutatorParams.from_cli())
    sc = v.scores()
    assert sc[0] == math.log(.001) and sc[1] == math.log(.001) and sc[2] == math.log(1 - .001 - .001)
    assert sc[3] == math.log(.01) and sc[4] == math.log(1 - .01)
    pi, pv = .01 * 10 / 11, .01 / 11
    sub = sc[5:21].reshape(4, 4)
    assert sub[0][0] == math.log(1 - pi - pv) - math.log(.25)
    assert sub[0][2] == math.log(pi) - math.log(.25)        # A<->G transition
    assert sub[0][1] == math.log(pv / 2) - math.log(.25)    # A<->C transversion
    assert all(x == math.log(1 / 6) for x in sc[21:27])


def test_no_valid_path_gives_empty_string(oracle_mod, ref_data):
    # viterbi.cpp:198-201: loglike == -inf -> "" (no-error global model cannot explain a corrupted read)
    O = oracle_mod
    v = O.ViterbiOracle(O.Machine.from_file(os.path.join(ref_data, "l4c4.json")),
                        O.MutatorParams.from_cli(sub=0., dup=0., del_open=0., global_=True))
    s, ll = v.decode("ACGTACGTACGT")
    assert s == "" and ll == -math.inf


def test_bad_base_is_an_error(oracle_mod, ref_data):
    O = oracle_mod
    v = O.ViterbiOracle(O.Machine.from_file(os.path.join(ref_data, "l4c4.json")), O.MutatorParams.from_cli())
    with pytest.raises(RuntimeError):
        v.decode("ACGTN")


def test_lowercase_reads(oracle_mod, ref_data):
    # fastseq.cpp:9-15: case-insensitive tokens (hello.dup.fa holds lowercase inserted bases)
    O = oracle_mod
    v = O.ViterbiOracle(O.Machine.from_file(os.path.join(ref_data, "l4c4.json")), O.MutatorParams.from_cli())
    seq = O.read_fasta(os.path.join(ref_data, "hello.fa"))[0][1]
    assert v.decode(seq.lower()) == v.decode(seq)


# ---------------------------------------------------------------- forward-backward (reference Makefile:156-163)
@pytest.mark.parametrize("stk,gold,ll", [("dup.stk", "dup.counts.json", 22.539541340221977),
                                         ("dup.sub.stk", "dup.sub.counts.json", 2.0692720187364166),
                                         ("dup.sub.misaligned.stk", "dup.sub.counts.misaligned.json", 2.0692720187364166)])
def test_error_counts_goldens(oracle_mod, ref_data, stk, gold, ll):
    O = oracle_mod
    p = O.MutatorParams.from_cli(sub=1e-9, dup=1e-9, del_open=1e-9, length=6)
    pairs = [O.alignment_pair(r) for r in O.read_stockholm(os.path.join(ref_data, stk))]
    counts, got_ll, per = O.expected_counts(p, pairs)
    assert O.counts_json(counts, 3) == open(os.path.join(ref_data, gold)).read()     # testexpect.pl diff
    assert got_ll == ll


@pytest.mark.parametrize("stk,gold", [("tiny.stk", "tiny.params.json"), ("test.stk", "test.params.json")])
def test_fit_error_goldens(oracle_mod, ref_data, stk, gold):
    O = oracle_mod
    pairs = [O.alignment_pair(r) for r in O.read_stockholm(os.path.join(ref_data, stk))]
    fit = O.baum_welch(O.MutatorParams.from_cli(), pairs, strict=True)
    assert O.params_json(fit) == open(os.path.join(ref_data, gold)).read()


def test_log_sum_exp_table_semantics(oracle_mod):
    # logsumexp.h:34-54: table step 1e-4, linear interpolation, exactly 0 from x = 10 on
    O = oracle_mod
    L = O.lib()
    import ctypes
    L.orc_log_sum_exp.restype = ctypes.c_double
    f = lambda a, b: L.orc_log_sum_exp(ctypes.c_double(a), ctypes.c_double(b))
    assert f(0.0, 0.0) == math.log(2.0)
    assert f(0.0, -10.0) == 0.0 and f(-3.0, -13.5) == -3.0
    assert f(-math.inf, -math.inf) == -math.inf and f(1.5, -math.inf) == 1.5
    assert abs(f(0.0, -1.00005) - math.log1p(math.exp(-1.00005))) < 1e-8
    assert f(0.0, -1.00005) != math.log1p(math.exp(-1.00005))       # interpolated, not exact
