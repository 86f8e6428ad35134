"""What ships is what is tested: every model here is created the way bench.py and smoke() create it -- no options, so the row program
is the one the machine's tuning record names (dnastore_amd/tune/) -- for the two cluster machines of BASELINE.json (46 670 and
258 538 states, tier C) and water64.1*l4c4 (tier A), and one FULL-LENGTH (~1 kb) read of each cluster machine is held to the
oracle, decoded string and fp64 log-likelihood bit for bit (1.5 s and 17 s of one host core)."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def _oracle(O, machine, **flags):
    return O.ViterbiOracle(O.Machine.from_json(machine.toJSON()), O.MutatorParams.from_cli(**flags))


@pytest.mark.parametrize("config,variant,states,n_reads,tier", [(1, "a", 46670, 3, "tier C: 4 work-groups"), (3, "b", 258538, 2, "tier C: 21 work-groups"),
                                                                (3, "a", 7066, 4, "tier A")])
def test_bench_machine_under_its_record_full_length_read(da, oracle_mod, config, variant, states, n_reads, tier):
    import bench
    wl = bench.workload(da, config, variant)
    m = wl["machine"]
    assert m.nStates() == states
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))          # exactly as bench.py does
    try:
        assert dec.tier.startswith(tier) and "record tune_" in dec.tier, dec.tier
        reads = bench.make_reads(m, 0, n_reads, payload_bytes=wl["payload_bytes"])     # bench reads 0 .. n-1: ~1 kb, 1 % substitutions
        assert all(900 < len(r) < 1200 for r in reads)
        out, ll, st = dec.decode(reads)
        assert not st.any() and np.isfinite(ll).all()
        orc = _oracle(oracle_mod, m, global_=True)
        s_ref, ll_ref = orc.decode(reads[0])                                            # one full-length read against the oracle
        assert out[0] == s_ref and ll[0] == ll_ref
        # ... and every read decodes alone to what it decoded to in the batch (other clusters, other launches)
        alone, ll1, _ = dec.decode(reads[-1:])
        assert alone[0] == out[-1] and ll1[0] == ll[-1]
    finally:
        dec.close()


def test_two_cluster_models_decode_at_once(da, oracle_mod, ref_data):
    """Two tier-C models launch their clusters on ONE card at the same time (two host threads; each launch alone would fill every
    CU): work-groups of the second launch start only as CUs come free, clusters wait for their late members (arrival time, not
    the per-column watchdog) and both calls return the results of a lone run."""
    from test_gpu_checkpoint import _reads
    import random
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    params = da.MutatorParams.fromFlags(global_=True)
    rng = random.Random(4)
    reads = _reads(da, m, rng, [29] * 200, rate=0.01)
    lone = da.ViterbiDecoder(m, params, options="tier=C,cluster=2")
    want = lone.decode(reads)
    lone.close()
    decs = [da.ViterbiDecoder(m, params, options="tier=C,cluster=2,arena_fraction=0.25") for _ in range(2)]
    got, errs = [None, None], [None, None]

    def work(i):
        try:
            got[i] = decs[i].decode(reads)
        except Exception as e:           # noqa: BLE001 -- reported below
            errs[i] = e
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errs == [None, None], [str(e) for e in errs]
    for g in got:
        assert g[0] == want[0] and np.array_equal(g[1].view(np.uint64), want[1].view(np.uint64)) and np.array_equal(g[2], want[2])
    for d in decs:
        d.close()
