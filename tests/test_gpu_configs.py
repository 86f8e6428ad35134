"""BASELINE.json configurations other than the bench line, as parity / property tests.

config 2: flusher*mixradar6*l4c4 (46 670 states), one ~1 kb read, --error-global, here under the general kernel
          (tier B); the cluster kernel that serves it by default (tier C) is held to the oracle in test_gpu_tier_c.py
config 4: water64.1*l4c4 (7 066 states; the literal "water64.1 + hamming74" product is empty), ~1 kb reads
config 3 at scale: round-trip property on a few hundred reads of the bench workload (bench.py itself
          compares a timed sample with the oracle bit for bit).
Oracle comparisons use sizes the CPU finishes in seconds; full sizes use size-independent properties:
a noise-free read decodes to exactly the symbols that were encoded, and with 1 % substitutions the
decoder still returns them (the codes' design distance), log-likelihoods finite."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def _compose(da, ref_data, *names):
    m = da.Machine.fromFile(os.path.join(ref_data, names[-1]))
    for n in reversed(names[:-1]):
        m = da.Machine.compose(da.Machine.fromFile(os.path.join(ref_data, n)), m)
    return m


def _oracle_machine(O, da_machine):
    return O.Machine.from_json(da_machine.toJSON())


def _bits(payload):
    return "^" + "".join(str((b >> n) & 1) for b in payload for n in range(8)) + "$"


def _substitute(rng, dna, rate):
    out = list(dna)
    for i, c in enumerate(out):
        if rng.random() < rate:
            out[i] = rng.choice([b for b in "ACGT" if b != c])
    return "".join(out)


def test_config2_mixradar6_composite_tier_b(da, oracle_mod, ref_data):
    O = oracle_mod
    m = _compose(da, ref_data, "flusher.json", "mixradar6.json", "l4c4.json")
    assert m.nStates() == 46670
    params = da.MutatorParams.fromFlags(global_=True)
    dec = da.ViterbiDecoder(m, params, options="tier=B")       # the general kernel (the default is tier C: test_gpu_tier_c.py)
    assert dec.tier.startswith("tier B")
    rng = random.Random(7)
    # oracle-sized case: 12 payload bytes (~100 nt) with 1 % substitutions
    small = _substitute(rng, m.encodeBytes(bytes(rng.randrange(256) for _ in range(12))), 0.01)
    out, ll, st = dec.decode([small])
    orc = O.ViterbiOracle(_oracle_machine(O, m), O.MutatorParams.from_cli(global_=True))
    s_ref, ll_ref = orc.decode(small)
    assert out[0] == s_ref and ll[0] == ll_ref and st[0] == 0
    # BASELINE size: one ~1 kb read (128 payload bytes); property: the payload comes back
    payload = bytes(rng.randrange(256) for _ in range(128))
    clean = m.encodeBytes(payload)
    assert 900 < len(clean) < 1100
    out, ll, st = dec.decode([clean, _substitute(rng, clean, 0.01)])
    assert list(st) == [0, 0] and np.isfinite(ll).all()
    assert da.symbolsToBytes(out[0]) == payload
    assert da.symbolsToBytes(out[1]) == payload
    assert ll[1] < ll[0]
    dec.close()


def test_config4_water64_composite(da, oracle_mod, ref_data):
    O = oracle_mod
    m = _compose(da, ref_data, "water64.1.json", "l4c4.json")
    assert m.nStates() == 7066
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
    assert dec.tier.startswith("tier A")
    orc = O.ViterbiOracle(_oracle_machine(O, m), O.MutatorParams.from_cli(global_=True))
    rng = random.Random(11)
    # payloads must be a multiple of 64 bits (reference README.md:49-53)
    reads, payloads = [], []
    for k in range(6):
        payload = bytes(rng.randrange(256) for _ in range(8 * (1 + k % 2)))
        payloads.append(payload)
        reads.append(_substitute(rng, m.encodeBytes(payload), 0.01))
    out, ll, st = dec.decode(reads)
    for i, r in enumerate(reads):
        s_ref, ll_ref = orc.decode(r)
        assert out[i] == s_ref and ll[i] == ll_ref
    # ~1 kb reads (64 payload bytes): round trip at the BASELINE read length
    big = [bytes(rng.randrange(256) for _ in range(64)) for _ in range(24)]
    dna = [m.encodeBytes(p) for p in big]
    assert all(900 < len(d) < 1400 for d in dna)
    out, ll, st = dec.decode(dna)
    assert not st.any()
    assert [da.symbolsToBytes(s) for s in out] == big
    dec.close()


def test_config3_roundtrip_at_scale(da, ref_data):
    import bench
    m = da.Machine.fromFile(bench.MACHINE)
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
    n = 800                                                     # more than one launch (720 reads per launch on MI355X)
    reads = bench.make_reads(m, 0, n)
    out, ll, st = dec.decode(reads)
    assert not st.any() and np.isfinite(ll).all()
    good = 0
    for i in range(n):
        rng = random.Random(1000 + i)                           # bench.make_reads' payload of read i
        payload = bytes(rng.randrange(256) for _ in range(29))
        good += da.symbolsToBytes(out[i]) == payload
    # 1 % substitutions over ~490 nt occasionally put two errors into one Hamming(7,4) block, which no
    # decoder can undo (doc/len4.ham.subs.tab: ~1e-3 edits/bit at p = .0128); nearly all reads come back
    assert good >= 0.9 * n, good
    # noise-free reads come back exactly
    clean = [m.encodeBytes(bytes(random.Random(77 + i).randrange(256) for _ in range(29))) for i in range(64)]
    outc, llc, stc = dec.decode(clean)
    assert [da.symbolsToBytes(s) for s in outc] == [bytes(random.Random(77 + i).randrange(256) for _ in range(29)) for i in range(64)]
    out, ll, st = dec.decode(reads)
    # idempotence: a second pass over the same batch gives identical bits
    out2, ll2, st2 = dec.decode(reads)
    assert out2 == out and np.array_equal(ll2, ll)
    assert dec.stats()["fill_launches"] == 2
    dec.close()
