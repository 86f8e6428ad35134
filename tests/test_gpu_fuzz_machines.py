"""Random machines (tests/random_machines.py) through the real HIP path: decoded strings, fp64
log-likelihoods and every lattice cell bit-identical to the oracle, under all three fill tiers."""
import numpy as np
import pytest

from random_machines import random_machine, random_read

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,n_states,global_", [(1, 40, True), (2, 90, False), (3, 150, True), (6, 2300, True), (7, 5000, False)])
@pytest.mark.parametrize("tier", ["A", "B", "C"])
def test_random_machine_gpu_matches_oracle(oracle_mod, seed, n_states, global_, tier, monkeypatch):
    import dnastore_amd as da
    O = oracle_mod
    if tier == "B":
        monkeypatch.setenv("DNAS_TIER", "B")
    text = random_machine(seed, n_states)
    flags = dict(global_=global_, sub=.02, dup=.01, del_open=.02, del_ext=.1)
    # tier C: the same machine cut over a cluster of 2 or 3 work-groups
    options = "tier=C,cluster=%d" % (2 + seed % 2) if tier == "C" else None
    dec = da.ViterbiDecoder(da.Machine.fromJSON(text), da.MutatorParams.fromFlags(**flags), options=options)
    assert dec.tier.startswith("tier " + tier)
    orc = O.ViterbiOracle(O.Machine.from_json(text), O.MutatorParams.from_cli(**flags))
    reads = [random_read(100 * seed + r, text, max_len=30) for r in range(4)]
    out, ll, st = dec.decode(reads)
    for i, r in enumerate(reads):
        s, oll, olat = orc.decode(r, want_lattice=True)
        assert out[i] == s and (ll[i] == oll or (np.isinf(ll[i]) and np.isinf(oll)))
        lat = np.ascontiguousarray(dec.lattice(i, len(r)).transpose(0, 2, 1))
        assert np.array_equal(lat.view(np.uint64), olat.view(np.uint64))
    dec.close()


@pytest.mark.parametrize("seed,n_states,global_", [(11, 60, True), (12, 400, False), (13, 2300, True)])
@pytest.mark.parametrize("tier", ["A", "B", "C"])
def test_random_machine_in_segments_matches_oracle(oracle_mod, seed, n_states, global_, tier):
    """The bounded-memory decode (lattice segments from checkpoints) on random machines: strings, status and
    fp64 log-likelihoods against the oracle, at the shortest segment the error model allows (D + 2 columns) and a longer one."""
    import dnastore_amd as da
    O = oracle_mod
    text = random_machine(seed, n_states)
    flags = dict(global_=global_, sub=.02, dup=.01, del_open=.02, del_ext=.1)
    base = {"A": "tier=A", "B": "tier=B", "C": "tier=C,cluster=%d" % (2 + seed % 2)}[tier]
    orc = O.ViterbiOracle(O.Machine.from_json(text), O.MutatorParams.from_cli(**flags))
    reads = [random_read(100 * seed + r, text, max_len=45) for r in range(6)] + [""]
    want = [orc.decode(r) for r in reads]
    for seg in (6, 17):
        dec = da.ViterbiDecoder(da.Machine.fromJSON(text), da.MutatorParams.fromFlags(**flags),
                                options=base + ",checkpoint=always,segment=%d" % seg)
        out, ll, st = dec.decode(reads)
        assert dec.stats()["checkpointed_reads"] == len(reads)
        for i, (s, oll) in enumerate(want):
            assert out[i] == s and (ll[i] == oll or (np.isinf(ll[i]) and np.isinf(oll))), (seg, i)
        dec.close()
