"""Tier C -- the register/LDS fill kernel run by a CLUSTER of work-groups per read, for machines beyond one CU --
held to the same bit-exact parity as tiers A and B.  Small fixture machines are forced onto clusters of 2..4
work-groups (options "tier=C,cluster=G") so that every cell of the lattice can be compared with the oracle;
the machines that need tier C (BASELINE configs[1], the 46 670-state mixradar6 composite, and the literal
configs[3] reading, hamming74 * dropdot * water64.1 * l4c4, 258 538 states) are decoded against the oracle on
short reads and through round-trip properties at the BASELINE read length."""
import os
import random
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# the 3-state adaptor that swallows hamming74's flush symbol '.' (not part of the reference)
DROPDOT = ('{"state":[{"n":0,"id":"S","trans":[{"in":"^","out":"^","to":1}]},{"n":1,"id":"T","trans":[{"in":"0","out":"0","to":1},'
           '{"in":"1","out":"1","to":1},{"in":".","to":1},{"in":"$","out":"$","to":2}]},{"n":2,"id":"U","trans":[]}]}')


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def _compose(da, ref_data, *parts):
    ms = [p if isinstance(p, da.Machine) else da.Machine.fromFile(os.path.join(ref_data, p)) for p in parts]
    m = ms[-1]
    for a in reversed(ms[:-1]):
        m = da.Machine.compose(a, m)
    return m


def _substitute(rng, dna, rate):
    out = list(dna)
    for i, c in enumerate(out):
        if rng.random() < rate:
            out[i] = rng.choice([b for b in "ACGT" if b != c])
    return "".join(out)


@pytest.mark.parametrize("mach,fa,flags,members,threads", [
    ("l4c4.json", "hello.dup.fa", dict(sub=0., del_open=0., global_=True), 2, 512),
    ("l4c4.json", "hello.fa", dict(), 3, 1024),
    ("h74l4c4.json", "hello.h74.sub.fa", dict(), 2, 1024),
    ("h74l4c4.json", "hello.h74.sub.fa", dict(global_=True), 4, 512),
    ("s16mr2l4c4.json", "hello.s16mr2.fa", dict(global_=True), 3, 512),
    ("s16h74l4c4.json", "hello.s16h74.del.fa", dict(), 2, 1024),
    ("s16h74l4c4.json", "hello.s16h74.del.fa", dict(global_=True), 4, 512),
    ("s16h74l4c4.json", "hello.s16h74.del.fa", dict(), 3, 512),
    # the shape of the single-read plan (bench.py: one read alone on 14 work-groups of 1024 threads), on a fixture machine
    ("s16h74l4c4.json", "hello.s16h74.del.fa", dict(global_=True), 14, 1024),
    ("s16h74l4c4.json", "hello.s16h74.fa", dict(), 16, 1024),
])
def test_tier_c_full_lattice_bit_exact(da, oracle_mod, ref_data, mach, fa, flags, members, threads):
    """Both work-group shapes: 512 threads (8 waves, 256 registers each, the tier-C default) and 1024 (tier A's)."""
    O = oracle_mod
    path = os.path.join(ref_data, mach)
    dec = da.ViterbiDecoder(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags),
                            options="tier=C,cluster=%d,threads=%d" % (members, threads))
    assert dec.tier.startswith("tier C: %d work-groups" % members) and ("T%dK" % threads) in dec.tier
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    read = da.read_fastseqs(os.path.join(ref_data, fa))[0][1]
    out, ll, st = dec.decode([read])
    s, oll, olat = orc.decode(read, want_lattice=True)      # [L+1][N][lanes]
    lat = dec.lattice(0, len(read))                          # [L+1][lanes][N]
    assert out[0] == s and ll[0] == oll and st[0] == 0
    a = np.ascontiguousarray(lat.transpose(0, 2, 1))
    assert np.array_equal(a.view(np.uint64), olat.view(np.uint64))
    dec.close()


@pytest.mark.parametrize("flags", [dict(global_=True), dict()])
def test_tier_c_batch_matches_oracle_and_tier_a(da, oracle_mod, ref_data, flags):
    """More reads than clusters (persistent clusters walk several reads each), ragged lengths, an empty read and a
    read with no valid path: strings, fp64 log-likelihoods and status equal the oracle's and tier A's."""
    O = oracle_mod
    path = os.path.join(ref_data, "s16h74l4c4.json")
    m = da.Machine.fromFile(path)
    params = da.MutatorParams.fromFlags(**flags)
    rng = random.Random(5)
    reads = []
    for i in range(21):
        dna = m.encodeBytes(bytes(rng.randrange(256) for _ in range(1 + i % 5)))
        dna = _substitute(rng, dna, 0.02)
        if i % 4 == 1:
            k = rng.randrange(len(dna)); dna = dna[:k] + dna[k + 1:]            # a deletion
        if i % 4 == 2:
            k = rng.randrange(4, len(dna)); dna = dna[:k] + dna[k - 3:k] + dna[k:]  # a tandem duplication
        reads.append(dna)
    reads.append("")
    reads.append("ACGT")
    dec_c = da.ViterbiDecoder(m, params, options="tier=C,cluster=2,max_clusters=5")
    dec_a = da.ViterbiDecoder(m, params)
    assert dec_c.tier.startswith("tier C") and dec_a.tier.startswith("tier A")
    out_c, ll_c, st_c = dec_c.decode(reads)
    out_a, ll_a, st_a = dec_a.decode(reads)
    assert out_c == out_a and np.array_equal(ll_c.view(np.uint64), ll_a.view(np.uint64)) and list(st_c) == list(st_a)
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    for i in (0, 1, 2, 7, 21, 22):
        s_ref, ll_ref = orc.decode(reads[i])
        assert out_c[i] == s_ref and (ll_c[i] == ll_ref or (np.isinf(ll_ref) and np.isinf(ll_c[i])))
    clusters, split = dec_c.cluster_census()
    assert clusters >= 1
    # idempotence: a second pass over the same batch gives identical bits
    out2, ll2, st2 = dec_c.decode(reads)
    assert out2 == out_c and np.array_equal(ll2.view(np.uint64), ll_c.view(np.uint64))
    dec_c.close(); dec_a.close()


def test_config2_mixradar6_composite_tier_c(da, oracle_mod, ref_data):
    """BASELINE configs[1]: flusher * mixradar6 * l4c4 (46 670 states, reference README.md:34-47), --error-global."""
    O = oracle_mod
    m = _compose(da, ref_data, "flusher.json", "mixradar6.json", "l4c4.json")
    assert m.nStates() == 46670
    params = da.MutatorParams.fromFlags(global_=True)
    dec = da.ViterbiDecoder(m, params)
    assert dec.tier.startswith("tier C")                         # does not fit one CU
    rng = random.Random(7)
    # oracle-sized cases: ~100 nt with 1 % substitutions, one with a deletion, every lattice cell of the first
    orc = O.ViterbiOracle(O.Machine.from_json(m.toJSON()), O.MutatorParams.from_cli(global_=True))
    small = [_substitute(rng, m.encodeBytes(bytes(rng.randrange(256) for _ in range(12))), 0.01) for _ in range(3)]
    small[1] = small[1][:40] + small[1][41:]
    out, ll, st = dec.decode(small)
    for i, r in enumerate(small):
        s_ref, ll_ref, olat = orc.decode(r, want_lattice=True)
        assert out[i] == s_ref and ll[i] == ll_ref and st[i] == 0
        if i == 0:
            dec1 = da.ViterbiDecoder(m, params)
            dec1.decode([r])
            lat = np.ascontiguousarray(dec1.lattice(0, len(r)).transpose(0, 2, 1))
            assert np.array_equal(lat.view(np.uint64), olat.view(np.uint64))
            dec1.close()
    # BASELINE size: one ~1 kb read (128 payload bytes).  The noise-free read gives the payload back; the read with
    # 1 % substitutions (mixradar6 corrects nothing, so the payload need not survive) must decode exactly as under
    # the general kernel (tier B), string and fp64 log-likelihood
    payload = bytes(rng.randrange(256) for _ in range(128))
    clean = m.encodeBytes(payload)
    assert 900 < len(clean) < 1100
    noisy = _substitute(rng, clean, 0.01)
    dec.decode([noisy])                                          # warm-up (arena allocation)
    t0 = time.time()
    out, ll, st = dec.decode([noisy])
    ms = (time.time() - t0) * 1e3
    fill_ms = dec.stats()["fill_ms"]
    print("config 2: %d nt decoded in %.1f ms (fill %.1f ms, %s)" % (len(noisy), ms, fill_ms, dec.tier[:60]))
    assert st[0] == 0 and np.isfinite(ll[0])
    dec_b = da.ViterbiDecoder(m, params, options="tier=B")
    assert dec_b.tier.startswith("tier B")
    out_b, ll_b, st_b = dec_b.decode([noisy])
    dec_b.close()
    assert out[0] == out_b[0] and ll[0] == ll_b[0] and st_b[0] == 0
    out2, ll2, st2 = dec.decode([clean, noisy])
    assert list(st2) == [0, 0] and da.symbolsToBytes(out2[0]) == payload and out2[1] == out[0] and ll2[1] == ll[0]
    assert ll2[1] < ll2[0]
    assert fill_ms < 100.0                                       # one CU (tier B) took 1 050 ms
    dec.close()


def test_config4b_hamming74_water64_composite_tier_c(da, oracle_mod, ref_data):
    """BASELINE configs[3] as written: hamming74 * dropdot * water64.1 * l4c4, 258 538 states."""
    O = oracle_mod
    m = _compose(da, ref_data, "hamming74.json", da.Machine.fromJSON(DROPDOT), "water64.1.json", "l4c4.json")
    assert m.nStates() == 258538
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
    assert dec.tier.startswith("tier C")
    rng = random.Random(13)
    # hamming74 takes 4 payload bits per block, water64.1 wants its input in blocks of 64 bits: 32 payload bytes give
    # 256 * 7/4 = 448 = 7 * 64 code bits and a read of ~1 kb, the BASELINE length
    payload = bytes(rng.randrange(256) for _ in range(32))
    clean = m.encodeBytes(payload)
    assert 1000 < len(clean) < 1100
    short = _substitute(rng, clean[:60], 0.02)                   # a prefix of the read, oracle-sized
    om = O.Machine.from_json(m.toJSON())
    orc = O.ViterbiOracle(om, O.MutatorParams.from_cli(global_=True))
    out, ll, st = dec.decode([short])
    s_ref, ll_ref = orc.decode(short)
    assert out[0] == s_ref and ll[0] == ll_ref and st[0] == 0
    # local mode on the same prefix
    dec_l = da.ViterbiDecoder(m, da.MutatorParams.fromFlags())
    orc_l = O.ViterbiOracle(om, O.MutatorParams.from_cli())
    out_l, ll_l, st_l = dec_l.decode([short])
    s_ref, ll_ref = orc_l.decode(short)
    assert out_l[0] == s_ref and ll_l[0] == ll_ref and st_l[0] == 0
    dec_l.close()
    # whole reads at the BASELINE length: the payload comes back, with and without substitutions
    noisy = _substitute(rng, clean, 0.01)
    t0 = time.time()
    out, ll, st = dec.decode([clean, noisy])
    print("config 4b: 2 reads of %d nt in %.1f ms (fill %.1f ms, %s)" % (len(clean), (time.time() - t0) * 1e3, dec.stats()["fill_ms"], dec.tier[:60]))
    assert list(st) == [0, 0] and np.isfinite(ll).all()
    assert da.symbolsToBytes(out[0]) == payload and da.symbolsToBytes(out[1]) == payload
    assert ll[1] < ll[0]
    dec.close()


def test_config4b_full_lattice_on_clusters_dealt_over_the_xcds(da, oracle_mod, ref_data):
    """The 258 538-state machine as it ships: 21 work-groups per read, the members of a cluster DEALT OVER THE XCDs (the default for
    clusters this large: exchange through memory, write-through stores), 20 proxies.  Every cell of the lattice of a short read --
    6 lanes x 258 538 states x 25 columns -- against the oracle, and the census says the cluster really was split."""
    O = oracle_mod
    m = _compose(da, ref_data, "hamming74.json", da.Machine.fromJSON(DROPDOT), "water64.1.json", "l4c4.json")
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
    try:
        assert dec.tier.startswith("tier C: 21 work-groups") and "dealt over the XCDs" in dec.tier, dec.tier
        rng = random.Random(29)
        read = _substitute(rng, m.encodeBytes(bytes(rng.randrange(256) for _ in range(32)))[:24], 0.05)
        out, ll, st = dec.decode([read])
        clusters, split = dec.cluster_census()
        assert (clusters, split) == (1, 1), (clusters, split)
        orc = O.ViterbiOracle(O.Machine.from_json(m.toJSON()), O.MutatorParams.from_cli(global_=True))
        s, oll, olat = orc.decode(read, want_lattice=True)
        assert out[0] == s and ll[0] == oll and st[0] == 0
        lat = dec.lattice(0, len(read))
        for p in range(len(read) + 1):          # column by column: the transposed copy of the whole lattice would be another 300 MB
            assert np.array_equal(np.ascontiguousarray(lat[p].T).view(np.uint64), olat[p].view(np.uint64)), "column %d" % p
    finally:
        dec.close()


@pytest.mark.parametrize("mach,fa,flags,members", [("s16mr2l4c4.json", "hello.s16mr2.fa", dict(global_=True), 3), ("s16h74l4c4.json", "hello.s16h74.del.fa", dict(), 4),
                                                   ("s16h74l4c4.json", "hello.s16h74.del.fa", dict(global_=True), 2)])
def test_tier_c_proxies_full_lattice_bit_exact(da, oracle_mod, ref_data, monkeypatch, mach, fa, flags, members):
    """Where several null edges of one member end in the same state of another, the planner ends them at a PROXY inside the source
    member and forwards their maximum over one edge (the 258 538-state composite: 2 245 edges into state 0 become 20).  Here from
    two edges on (DNAS_PLAN_PROXY_MIN=2), so that the fixture machines have many: every lattice cell is still the oracle's."""
    O = oracle_mod
    monkeypatch.setenv("DNAS_PLAN_PROXY_MIN", "2")
    path = os.path.join(ref_data, mach)
    m = da.Machine.fromFile(path)
    params = da.MutatorParams.fromFlags(**flags)
    n_prox = len(da.FlatModel(m, params).cluster_plan(members)["proxy_member"])
    assert n_prox > 0
    dec = da.ViterbiDecoder(m, params, options="tier=C,cluster=%d" % members)
    try:
        orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
        read = da.read_fastseqs(os.path.join(ref_data, fa))[0][1]
        out, ll, st = dec.decode([read])
        s, oll, olat = orc.decode(read, want_lattice=True)
        lat = dec.lattice(0, len(read))
        assert out[0] == s and ll[0] == oll and st[0] == 0
        assert np.array_equal(np.ascontiguousarray(lat.transpose(0, 2, 1)).view(np.uint64), olat.view(np.uint64))
        print("%s, %d members: %d proxies" % (mach, members, n_prox))
    finally:
        dec.close()


@pytest.mark.parametrize("mach,fa,flags,members", [("s16h74l4c4.json", "hello.s16h74.del.fa", dict(global_=True), 4),
                                                   ("s16mr2l4c4.json", "hello.s16mr2.fa", dict(), 3)])
def test_tier_c_clusters_split_over_xcds(da, oracle_mod, ref_data, mach, fa, flags, members):
    """cluster_spread=1 puts the members of a cluster into neighbouring blocks, i.e. onto DIFFERENT XCDs (private L2s): the
    exchange is 8-byte agent-scope atomics and sc1 accesses on both sides, so the placement is a matter of speed only -- the
    census says the clusters were split, and every cell is still the oracle's."""
    O = oracle_mod
    path = os.path.join(ref_data, mach)
    m = da.Machine.fromFile(path)
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(**flags), options="tier=C,cluster=%d,cluster_spread=1" % members)
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    reads = [seq for _, seq in da.read_fastseqs(os.path.join(ref_data, fa))] + [m.encodeBytes(bytes([i, 255 - i, 7 * i % 256])) for i in range(11)]
    out, ll, st = dec.decode(reads)
    clusters, split = dec.cluster_census()
    assert clusters == len(reads) and split == clusters, (clusters, split)
    for i, r in enumerate(reads):
        res = orc.decode(r, want_lattice=(i == 0))
        assert out[i] == res[0] and ll[i] == res[1]
        if i == 0:
            lat = np.ascontiguousarray(dec.lattice(i, len(r)).transpose(0, 2, 1))
            assert np.array_equal(lat.view(np.uint64), res[2].view(np.uint64))
    dec.close()


def test_sync_words_are_placed_by_measured_latency(da, oracle_mod, ref_data, monkeypatch):
    """Where a cluster's sync words sit is a matter of speed only (option sync_place, DNAS_SYNC_OFFSET): the place the library
    measures, the start of the window and two forced offsets give the same strings and log-likelihood bits, which are the
    oracle's; the chosen place is named in the tier note."""
    O = oracle_mod
    path = os.path.join(ref_data, "s16h74l4c4.json")
    flags = dict(global_=True)
    m = da.Machine.fromFile(path)
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    reads = [m.encodeBytes(bytes([3 * i % 256, 255 - i, 11 * i % 256, i])) for i in range(9)]
    want = [orc.decode(r)[:2] for r in reads]
    for opts, env in (("tier=C,cluster=3", None), ("tier=C,cluster=3,sync_place=0", None), ("tier=C,cluster=3", "2048"), ("tier=C,cluster=3", "65536")):
        if env is None:
            monkeypatch.delenv("DNAS_SYNC_OFFSET", raising=False)
        else:
            monkeypatch.setenv("DNAS_SYNC_OFFSET", env)
        dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(**flags), options=opts)
        out, ll, st = dec.decode(reads)
        assert ("sync words at" in dec.tier) == (env is None and "sync_place=0" not in opts), dec.tier
        assert [(o, float(x)) for o, x in zip(out, ll)] == [(w[0], float(w[1])) for w in want]
        dec.close()
