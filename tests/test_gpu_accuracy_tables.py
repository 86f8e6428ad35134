"""Long reads pinned to numbers the reference PUBLISHES: rows of doc/len4.ham.subs.tab (Hamming(7,4) * DNASTORE(4), 8192-bit
payloads = reads of ~13.5 kb, 20 repetitions per substitution rate), reproduced with the method of doc/errdecode.pl through
the GPU decoder.  tests/accuracy_tables.py restates the method and cites it line by line; the table itself is a data fixture
(tests/golden/ref_doc/).  Two of the same reads also go through the oracle, full length, bit for bit."""
import os

import numpy as np
import pytest

import accuracy_tables as AT

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DATA = os.path.join(ROOT, "tests", "golden", "ref_data")
TABLE = AT.read_table(os.path.join(ROOT, "tests", "golden", "ref_doc", "len4.ham.subs.tab"))
REPS = 20


def _decode_row(da, machine, rate, reps=REPS):
    """-> (payloads, reads, decoded payload strings, log-likelihoods) of one table row"""
    cases = [AT.make_case(machine, rate, rep) for rep in range(reps)]
    params = da.MutatorParams.fromFlags(sub=rate, dup=0.0, del_open=0.0, del_ext=0.2, global_=True, length=4)
    dec = da.ViterbiDecoder(machine, params)
    out, ll, st = dec.decode([c[1] for c in cases])
    dec.close()
    assert not np.asarray(st).any(), "a read of the table experiment did not decode"
    return [c[0] for c in cases], [c[1] for c in cases], [s.replace("^", "").replace("$", "") for s in out], ll, out


def test_next_to_no_edits_up_to_a_substitution_rate_of_0_004():
    """Rows 1-6 of the table: MeanEditsPerBit 0 over 20 x 8192 bits.  Without substitutions every payload comes back exactly.  With
    them we do NOT find exactly zero: a single substitution sometimes has two equally likely explanations (two code words one
    transition away from the read) and the Viterbi path takes the wrong one -- 0 to 4 edited bits per row of 163 840
    (profiles/r4_len4_ham_subs.txt), where the table's own neighbouring rows (one edited bit at 0.0057, three at 0.008) suggest
    0.3 to 1.5.  The bound is therefore five bits per row, and every read that came back with edits is decoded again by the ORACLE
    and must come back the same: whatever is wrong with those payloads is the reference algorithm's own answer."""
    import dnastore_amd as da
    from oracle import oracle as O
    O.build()
    path = os.path.join(REF_DATA, "h74l4c4.json")
    machine, om = da.Machine.fromFile(path), O.Machine.from_file(path)
    for n, row in enumerate(TABLE[:6]):
        assert row["MeanEditsPerBit"] == 0 and row["StDevEditsPerBit"] == 0
        rate = row["SubProb"]
        payloads, reads, decoded, ll, raw = _decode_row(da, machine, rate)
        assert all(12000 < len(r) < 15000 for r in reads)           # 8192 bits are ~13.5 kb
        wrong = [i for i, (p, d) in enumerate(zip(payloads, decoded)) if p != d]
        if rate == 0:
            assert not wrong, "reads without a single error came back with edits"
        edits = sum(AT.edit_distance(payloads[i], decoded[i]) for i in wrong)
        print("substitution rate %g: %d edited bits of %d (repetitions %s)" % (rate, edits, REPS * AT.BITS, wrong))
        assert edits <= 5, "substitution rate %g: %d edited bits in repetitions %s" % (rate, edits, wrong)
        orc = O.ViterbiOracle(om, O.MutatorParams.from_cli(sub=rate, dup=0.0, del_open=0.0, del_ext=0.2, global_=True, length=4))
        for i in wrong[:2]:
            s_ref, ll_ref = orc.decode(reads[i])
            assert raw[i] == s_ref and float(ll[i]) == ll_ref, "substitution rate %g, repetition %d: the GPU's edits are not the oracle's" % (rate, i)


def test_about_one_edit_per_thousand_bits_at_0_128():
    """Row 16: MeanEditsPerBit 1.117e-3, StDev 5.3e-4 over 20 repetitions.  The mean of our 20 repetitions must lie within three
    of the table's standard deviations of the table's mean (and is reported with its distance in standard errors)."""
    import dnastore_amd as da
    machine = da.Machine.fromFile(os.path.join(REF_DATA, "h74l4c4.json"))
    row = TABLE[15]
    assert row["SubProb"] == 0.128
    payloads, reads, decoded, _, _ = _decode_row(da, machine, row["SubProb"])
    per_bit = np.array([AT.edit_distance(p, d) / AT.BITS for p, d in zip(payloads, decoded)])
    mean, sd = float(per_bit.mean()), float(per_bit.std())
    se = float(np.hypot(row["StDevEditsPerBit"], sd) / np.sqrt(REPS))
    print("substitution rate 0.128: %.4g edits per bit (sd %.3g) against the table's %.4g (sd %.3g): %.2f standard errors apart"
          % (mean, sd, row["MeanEditsPerBit"], row["StDevEditsPerBit"], (mean - row["MeanEditsPerBit"]) / se))
    assert abs(mean - row["MeanEditsPerBit"]) <= 3 * row["StDevEditsPerBit"]
    assert abs(mean - row["MeanEditsPerBit"]) <= 3 * se          # ... and, tighter, within three standard errors of the difference of the two means
    assert mean > 0          # at this rate the code does not correct everything


def test_the_same_long_reads_through_the_oracle():
    """One read of row 6 and one of row 16 (13.5 kb each), full length: decoded string and fp64 log-likelihood bit for bit."""
    import dnastore_amd as da
    from oracle import oracle as O
    O.build()
    path = os.path.join(REF_DATA, "h74l4c4.json")
    machine = da.Machine.fromFile(path)
    om = O.Machine.from_file(path)
    for rate in (0.004, 0.128):
        _, reads, _, ll, raw = _decode_row(da, machine, rate, reps=2)
        orc = O.ViterbiOracle(om, O.MutatorParams.from_cli(sub=rate, dup=0.0, del_open=0.0, del_ext=0.2, global_=True, length=4))
        s_ref, ll_ref = orc.decode(reads[1])
        assert raw[1] == s_ref and float(ll[1]) == ll_ref, "substitution rate %g: GPU and oracle differ on a 13.5 kb read" % rate
