"""GPU parity tests of the forward-backward E-step / Baum-Welch path (through the C ABI) against the
CPU oracle and the reference's five golden files (reference Makefile:156-163).

Per-pair log-likelihoods are bit-exact.  Expected counts are sums of exp() terms: the kernel adds a
pair's terms in reverse cell order and the database in a tree, the reference serially, and exp() comes
from different math libraries, so counts are held to 1e-9 relative -- and to the
reference's printed 6 significant digits exactly."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNT_FLAGS = dict(sub=1e-9, dup=1e-9, del_open=1e-9, length=6)      # testcount, Makefile:156-159


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def _close(a, b):
    return np.allclose(a, b, rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("stk,gold", [("dup.stk", "dup.counts.json"), ("dup.sub.stk", "dup.sub.counts.json"),
                                      ("dup.sub.misaligned.stk", "dup.sub.counts.misaligned.json")])
def test_error_counts_goldens(da, oracle_mod, ref_data, stk, gold):
    O = oracle_mod
    params = da.MutatorParams.fromFlags(**COUNT_FLAGS)
    counts, ll, per = da.expectedCounts(params, da.StockholmDB(os.path.join(ref_data, stk)))
    assert da.countsJSON(counts, 3) == open(os.path.join(ref_data, gold)).read()          # reference golden, byte for byte
    oc, oll, oper = O.expected_counts(O.MutatorParams.from_cli(**COUNT_FLAGS),
                                      [O.alignment_pair(r) for r in O.read_stockholm(os.path.join(ref_data, stk))])
    assert list(per) == list(oper) and ll == oll                                            # fp64-exact log-likelihoods
    assert _close(counts, oc)


@pytest.mark.parametrize("stk,gold", [("tiny.stk", "tiny.params.json"), ("test.stk", "test.params.json")])
def test_fit_error_goldens(da, ref_data, stk, gold):
    fit, iters = da.baumWelchParams(da.MutatorParams.fromFlags(), da.StockholmDB(os.path.join(ref_data, stk)), strict=True)
    assert da.paramsJSON(fit) == open(os.path.join(ref_data, gold)).read()                 # testfit, Makefile:161-163
    assert 1 <= iters <= 100


@pytest.mark.parametrize("flags,strict", [(dict(), False), (dict(), True), (dict(length=6, sub=.05, dup=.02, del_open=.02), False)])
def test_synthetic_database(da, oracle_mod, flags, strict):
    from synth import synthetic_alignment
    O = oracle_mod
    rng = random.Random(11)
    rows = [synthetic_alignment(rng, rng.choice([1, 5, 40, 97, 256])) for _ in range(150)]
    pairs = [O.alignment_pair(r) for r in rows]
    pk = O.pack_pairs(pairs)
    counts, ll, per = da.expectedCounts(da.MutatorParams.fromFlags(**flags), pk, strict=strict)
    oc, oll, oper = O.expected_counts(O.MutatorParams.from_cli(**flags), pairs, strict=strict)
    assert np.array_equal(per, oper)               # bit-exact per pair (incl. -inf for pairs the band cannot explain)
    assert _close(counts, oc)
    assert ll == pytest.approx(oll, rel=1e-12) or (np.isinf(ll) and ll == oll)


def test_baum_welch_matches_oracle(da, oracle_mod):
    from synth import synthetic_alignment
    O = oracle_mod
    rng = random.Random(3)
    pairs = [O.alignment_pair(synthetic_alignment(rng, 120, sub=.03, dele=.02, dup=.02)) for _ in range(40)]
    fit, iters = da.baumWelchParams(da.MutatorParams.fromFlags(), O.pack_pairs(pairs))
    ofit = O.baum_welch(O.MutatorParams.from_cli(), pairs)
    got = [fit.c.p_del_open, fit.c.p_del_extend, fit.c.p_tan_dup, fit.c.p_transition, fit.c.p_transversion]
    want = [ofit.pDelOpen, ofit.pDelExtend, ofit.pTanDup, ofit.pTransition, ofit.pTransversion]
    assert np.allclose(got, want, rtol=1e-9)
    assert fit.pLen == ofit.pLen and fit.local == ofit.local


def test_empty_database(da):
    counts, ll, per = da.expectedCounts(da.MutatorParams.fromFlags(), dict(
        ins=np.zeros(0, np.int8), in_off=np.zeros(1, np.int64), outs=np.zeros(0, np.int8), out_off=np.zeros(1, np.int64),
        cm_in=np.zeros(0, np.int32), cm_in_off=np.zeros(1, np.int64), cm_out=np.zeros(0, np.int32),
        cm_out_off=np.zeros(1, np.int64), n=0))
    assert not counts.any() and ll == 0 and len(per) == 0


def test_onchip_and_streaming_kernels_agree(oracle_mod, monkeypatch):
    """The persistent handle (dnas_fb): the on-chip kernels take the pairs whose envelope rows are at most 16 cells (16 lanes
    per pair) or 32 cells (32 lanes) wide, the streaming kernel the rest; all reproduce the oracle's per-pair
    log-likelihoods bit for bit, agree with each other on the counts, and a second E-step on the same handle (another
    model) reuses the database."""
    import random
    import dnastore_amd as da
    from synth import synthetic_alignment
    O = oracle_mod
    rng = random.Random(21)
    pairs = [O.alignment_pair(synthetic_alignment(rng, rng.choice([1, 7, 33, 100, 256]), sub=.03, dele=.02, dup=.02)) for _ in range(90)]
    # pairs with long runs of duplications: envelope rows wider than 16 cells -> the 32-lane kernel; wider than 32 -> streaming
    pairs.append(O.alignment_pair(synthetic_alignment(random.Random(5), 60, sub=.02, dele=.0, dup=.35)))
    pairs.append(O.alignment_pair(synthetic_alignment(random.Random(6), 90, sub=.02, dele=.0, dup=.8)))
    pk = O.pack_pairs(pairs)
    params = da.MutatorParams.fromFlags()
    fb = da.ForwardBackward(pk)
    counts, ll, per = fb.expectedCounts(params)
    st = fb.stats()
    assert st["pairs_onchip"] >= 61 and st["pairs_streaming"] >= 1 and st["pairs_onchip"] + st["pairs_streaming"] == len(pairs) and st["lse_ops"] > 0
    oc, oll, oper = O.expected_counts(O.MutatorParams.from_cli(), pairs)
    assert np.array_equal(per, oper)
    assert np.allclose(counts, oc, rtol=1e-9, atol=1e-300)
    # strict guides on the same handle: another envelope, another census
    c2, ll2, per2 = fb.expectedCounts(params, strict=True)
    oc2, oll2, oper2 = O.expected_counts(O.MutatorParams.from_cli(), pairs, strict=True)
    assert np.array_equal(per2, oper2) and np.allclose(c2, oc2, rtol=1e-9, atol=1e-300)
    # and the first model again: identical bits run to run
    c3, ll3, per3 = fb.expectedCounts(params)
    assert np.array_equal(c3, counts) and ll3 == ll and np.array_equal(per3, per)
    fb.close()
    monkeypatch.setenv("DNAS_FB_STREAMING", "1")
    fb = da.ForwardBackward(pk)
    cs, lls, pers = fb.expectedCounts(params)
    assert fb.stats()["pairs_onchip"] == 0
    assert np.array_equal(pers, per) and np.allclose(cs, counts, rtol=1e-11, atol=1e-300)
    fb.close()


def test_one_long_pair_does_not_fail_the_database(oracle_mod):
    """A database with one input far longer than the on-chip kernel's LDS can hold (its checkpoints grow with the input) beside
    ordinary pairs: the long pair goes to the streaming kernel, the call succeeds and matches the oracle pair by pair."""
    import dnastore_amd as da
    from synth import synthetic_alignment
    O = oracle_mod
    rng = random.Random(8)
    pairs = [O.alignment_pair(synthetic_alignment(rng, n, sub=.02, dele=.01, dup=.01)) for n in (40, 256, 4000, 12000, 130)]
    fb = da.ForwardBackward(O.pack_pairs(pairs))
    counts, ll, per = fb.expectedCounts(da.MutatorParams.fromFlags())
    st = fb.stats()
    assert st["pairs_streaming"] >= 1 and st["pairs_onchip"] >= 1
    oc, oll, oper = O.expected_counts(O.MutatorParams.from_cli(), pairs)
    assert np.array_equal(per, oper) and np.allclose(counts, oc, rtol=1e-9, atol=1e-300)
    fb.close()


def test_half_width_wavefront_is_chosen_per_pair(da, oracle_mod, monkeypatch):
    """The wavefront kernels run a pair on HALF as many lanes as its widest envelope row has cells when every lane has left its row
    before its next one comes up (hi(ip) - lo(ip + W) < W: alignments that run down a diagonal).  A block of deleted input bases
    breaks that -- the rows below it all start at the same column --: such pairs must take the full-width kernel, and both kinds
    give the oracle's numbers, whatever the routing (DNAS_FB_NO_NARROW: nobody takes the half-width kernels)."""
    from synth import synthetic_alignment
    O = oracle_mod
    rng = random.Random(77)
    rows = [synthetic_alignment(rng, 256) for _ in range(120)]
    for i in range(40):                          # 24 input bases in a row deleted somewhere in the middle
        src = "".join(rng.choice("ACGT") for _ in range(200))
        cut = rng.randrange(40, 120)
        rows.append([("in", src), ("out", src[:cut] + "-" * 24 + src[cut + 24:])])
    pairs = [O.alignment_pair(r) for r in rows]
    pk = O.pack_pairs(pairs)
    flags = dict(sub=.02, dup=.01, del_open=.02, length=12)
    oc, oll, oper = O.expected_counts(O.MutatorParams.from_cli(**flags), pairs)
    stats = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("DNAS_FB_NO_NARROW", env)
        fb = da.ForwardBackward(pk)
        counts, ll, per = fb.expectedCounts(da.MutatorParams.fromFlags(**flags))
        stats.append(fb.stats())
        fb.close()
        assert np.array_equal(per, oper) and _close(counts, oc)
    assert stats[0]["pairs_narrow"] >= 100 and stats[0]["pairs_narrow"] <= stats[0]["pairs_onchip"] - 30, stats[0]   # the 40 gapped pairs: full width
    assert stats[1]["pairs_narrow"] == 0 and stats[1]["pairs_onchip"] == stats[0]["pairs_onchip"]


def test_database_checks_run_on_the_gpu(da, oracle_mod):
    """dnas_fb_load_pairs checks every base and guide column on the device: a bad base and a decreasing guide column are refused."""
    from synth import synthetic_alignment
    O = oracle_mod
    rng = random.Random(5)
    pairs = [O.alignment_pair(synthetic_alignment(rng, 60)) for _ in range(20)]
    pk = O.pack_pairs(pairs)
    bad = dict(pk); bad["outs"] = pk["outs"].copy(); bad["outs"][7] = 4
    with pytest.raises(da.DnasError, match="DNAS_E_BAD_BASE"):
        da.ForwardBackward(bad)
    bad = dict(pk); bad["cm_in"] = pk["cm_in"].copy(); bad["cm_in"][5] = bad["cm_in"][4] - 1
    with pytest.raises(da.DnasError, match="non-decreasing"):
        da.ForwardBackward(bad)
    fb = da.ForwardBackward(pk)                  # ... and the handle API still loads a good database afterwards
    assert np.array_equal(fb.expectedCounts(da.MutatorParams.fromFlags())[2], O.expected_counts(O.MutatorParams.from_cli(), pairs)[2])
    fb.close()
