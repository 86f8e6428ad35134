"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
reference's golden vectors.  Bit-exact: decoded strings, fp64 log-likelihoods and, on the
small cases, every cell of the lattice."""
import math
import os
import random

import numpy as np
import pytest

from viterbi_cases import VITERBI_GOLDENS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def _pair(da, O, ref_data, mach, flags):
    path = os.path.join(ref_data, mach)
    dec = da.ViterbiDecoder(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags))
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    return dec, orc


@pytest.mark.parametrize("mach,fa,flags,bits,loglike", VITERBI_GOLDENS)
def test_reference_goldens(da, oracle_mod, ref_data, mach, fa, flags, bits, loglike):
    dec = da.ViterbiDecoder(da.Machine.fromFile(os.path.join(ref_data, mach)), da.MutatorParams.fromFlags(**flags))
    want = open(os.path.join(ref_data, bits)).read().strip()
    reads = [s for _, s in da.read_fastseqs(os.path.join(ref_data, fa))]
    out, ll, st = dec.decode(reads)
    assert out == [want] and st[0] == 0
    assert ll[0] == loglike            # bit-exact fp64
    dec.close()


@pytest.mark.parametrize("mach,fa,flags", [
    ("l4c4.json", "hello.dup.fa", dict(sub=0., del_open=0., global_=True)),
    ("l4c4.json", "hello.fa", dict()),
    ("h74l4c4.json", "hello.h74.sub.fa", dict()),
    ("s16mr2l4c4.json", "hello.s16mr2.fa", dict(global_=True)),
    ("s16h74l4c4.json", "hello.s16h74.del.fa", dict()),
])
def test_full_lattice_bit_exact(da, oracle_mod, ref_data, mach, fa, flags):
    dec, orc = _pair(da, oracle_mod, ref_data, mach, flags)
    read = da.read_fastseqs(os.path.join(ref_data, fa))[0][1]
    out, ll, st = dec.decode([read])
    s, oll, olat = orc.decode(read, want_lattice=True)      # [L+1][N][lanes]
    lat = dec.lattice(0, len(read))                          # [L+1][lanes][N]
    assert out[0] == s and ll[0] == oll
    a = np.ascontiguousarray(lat.transpose(0, 2, 1))
    assert a.shape == olat.shape
    assert np.array_equal(a.view(np.uint64), olat.view(np.uint64))
    dec.close()


@pytest.mark.parametrize("flags,noise", [
    (dict(global_=True), dict(sub=0.01)),
    (dict(), dict(sub=0.02, dele=0.01, dup=0.01)),
])
def test_synthetic_batch_s16h74(da, oracle_mod, ref_data, flags, noise):
    from synth import synthetic_reads
    dec, orc = _pair(da, oracle_mod, ref_data, "s16h74l4c4.json", flags)
    reads = synthetic_reads(orc.machine, 6, 6, seed=1000, **noise)
    reads.append(reads[0][:37])      # ragged: a truncated read
    out, ll, st = dec.decode(reads)
    for i, r in enumerate(reads):
        s, oll = orc.decode(r)
        assert out[i] == s, i
        assert ll[i] == oll or (math.isinf(oll) and ll[i] == oll), i
        assert st[i] == (1 if s == "" and math.isinf(oll) else 0)
    dec.close()


def test_ragged_and_degenerate_reads(da, oracle_mod, ref_data):
    """empty read, 1-nt read, unexplainable read (no valid path -> empty string, viterbi.cpp:198-201)."""
    flags = dict(sub=0., dup=0., del_open=0., global_=True)
    dec, orc = _pair(da, oracle_mod, ref_data, "l4c4.json", flags)
    good = da.read_fastseqs(os.path.join(ref_data, "hello.fa"))[0][1]
    reads = ["", "A", "ACGTACGTACGT", good, good.lower()]
    out, ll, st = dec.decode(reads)
    for i, r in enumerate(reads):
        s, oll = orc.decode(r)
        assert out[i] == s and ll[i] == oll
    assert list(st) == [1, 1, 1, 0, 0]
    dec.close()


def test_local_mode_many_reads_order_preserved(da, oracle_mod, ref_data):
    from synth import synthetic_reads
    dec, orc = _pair(da, oracle_mod, ref_data, "mr2l4c4.json", dict())
    rng = random.Random(5)
    reads = []
    for k in range(40):
        reads.extend(synthetic_reads(orc.machine, 1, rng.randint(1, 6), seed=300 + k, sub=0.02, dele=0.005, dup=0.005))
    out, ll, st = dec.decode(reads)
    for i, r in enumerate(reads):
        s, oll = orc.decode(r)
        assert out[i] == s and ll[i] == oll, i
    dec.close()


def test_decode_fastseqs_drop_in(da, ref_data):
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    res = da.decode_fastseqs(os.path.join(ref_data, "hello.s16h74.del.fa"), m, da.MutatorParams.fromFlags())
    want = open(os.path.join(ref_data, "hello.exact.bits")).read().strip()
    assert res == [("data/hello.txt", want, 54.416887183956284)]


def test_output_overflow_is_reported(da, ref_data):
    dec = da.ViterbiDecoder(da.Machine.fromFile(os.path.join(ref_data, "l4c4.json")), da.MutatorParams.fromFlags())
    good = da.read_fastseqs(os.path.join(ref_data, "hello.fa"))[0][1]
    out, ll, st = dec.decode([good], out_cap=8)
    assert st[0] == 2 and out[0] == ""
    dec.close()


def test_bad_base_rejected(da, ref_data):
    dec = da.ViterbiDecoder(da.Machine.fromFile(os.path.join(ref_data, "l4c4.json")), da.MutatorParams.fromFlags())
    with pytest.raises(ValueError):
        dec.decode(["ACGTN"])
    dec.close()


def test_many_small_batches_equal_one_batch(da, ref_data, monkeypatch):
    """Ping-pong arenas and batch scheduling: the same reads cut into many launches (3 slots each, and again
    with an arena that holds only two lattices) give exactly the bits of a single launch."""
    import random
    path = os.path.join(ref_data, "h74l4c4.json")
    m = da.Machine.fromFile(path)
    params = da.MutatorParams.fromFlags()
    rng = random.Random(5)
    reads = [m.encodeBytes(bytes(rng.randrange(256) for _ in range(rng.choice([1, 3, 6])))) for _ in range(23)]
    reads[4] = reads[4][:9]
    reads[11] = ""
    dec = da.ViterbiDecoder(m, params)
    want = dec.decode(reads)
    assert dec.stats()["fill_launches"] == 1
    dec.close()
    monkeypatch.setenv("DNAS_MAX_SLOTS", "3")
    dec = da.ViterbiDecoder(m, params)
    got = dec.decode(reads)
    assert dec.stats()["fill_launches"] == 8
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    dec.close()
    monkeypatch.delenv("DNAS_MAX_SLOTS")
    longest = max(len(r) for r in reads)
    n_pad = 6 * 1024                                            # tier A row stride of this machine (5 242 states)
    dec = da.ViterbiDecoder(m, params, arena_bytes=2 * 2 * (2 * n_pad * (longest + 1) + 8) * 8 + 4096)
    got = dec.decode(reads)
    assert dec.stats()["fill_launches"] > 4
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    dec.close()


def test_device_pointer_entry_point_matches_host_entry_point(da, ref_data):
    """dnas_viterbi_batch_device (inputs and outputs resident in HBM, what bench.py times) against
    dnas_viterbi_batch (host buffers)."""
    import random
    import torch
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
    rng = random.Random(21)
    reads = [m.encodeBytes(bytes(rng.randrange(256) for _ in range(rng.choice([2, 5, 9])))) for _ in range(9)]
    want_sym, want_ll, want_st = dec.decode(reads)
    off, bases = da.pack_reads(reads)
    dev = torch.device("cuda", 0)
    d_bases = torch.from_numpy(np.ascontiguousarray(bases)).to(dev)
    k = len(reads)
    cap = int(np.diff(off).max()) + 64
    out_off = np.arange(k + 1, dtype=np.uint64) * np.uint64(cap)
    d_sym = torch.zeros(k * cap, dtype=torch.uint8, device=dev)
    d_len = torch.zeros(k, dtype=torch.int32, device=dev)
    d_ll = torch.zeros(k, dtype=torch.float64, device=dev)
    d_st = torch.zeros(k, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()    # torch's copies / fills are done before the library's own streams touch the buffers
    dec.decode_device(off, d_bases.data_ptr(), d_sym.data_ptr(), out_off, d_len.data_ptr(), d_ll.data_ptr(), d_st.data_ptr())
    dec.sync()
    sym, olen = d_sym.cpu().numpy(), d_len.cpu().numpy()
    got = [sym[i * cap:i * cap + int(olen[i])].tobytes().decode() for i in range(k)]
    assert got == want_sym
    assert np.array_equal(d_ll.cpu().numpy(), want_ll) and np.array_equal(d_st.cpu().numpy(), want_st)
    dec.close()


def test_device_entry_point_rejects_bad_bases_and_stale_lattices(da, ref_data):
    """dnas_viterbi_batch_device checks its inputs too (a base code > 3 would index past the kernels' tables), and
    dnas_model_read_lattice refuses a read whose arena half a later batch of the call has reused."""
    import torch
    from dnastore_amd import lib as L
    m = da.Machine.fromFile(os.path.join(ref_data, "l4c4.json"))
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(), options="max_slots=2")
    reads = [m.encodeBytes(bytes([i])) for i in range(7)]           # 4 batches of at most 2 reads
    out, ll, st = dec.decode(reads)
    assert dec.stats()["fill_launches"] == 4
    alive = 0                                                       # batches of 2, 2, 2, 1 reads (longest first): the arena's two
    for i, r in enumerate(reads):                                   # halves still hold the last two batches, 3 lattices
        try:
            dec.lattice(i, len(r))
            alive += 1
        except L.DnasError as e:
            assert "overwritten" in str(e)
    assert alive == 3
    off, bases = da.pack_reads(reads[:2])
    bad = torch.from_numpy(bases.copy()).cuda()
    bad[3] = 7
    n = 2
    cap = 256
    ooff = np.arange(n + 1, dtype=np.uint64) * np.uint64(cap)
    sym = torch.zeros(n * cap, dtype=torch.uint8, device="cuda")
    olen = torch.zeros(n, dtype=torch.int32, device="cuda")
    llt = torch.zeros(n, dtype=torch.float64, device="cuda")
    stt = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    with pytest.raises(L.DnasError) as e:
        dec.decode_device(off, bad.data_ptr(), sym.data_ptr(), ooff, olen.data_ptr(), llt.data_ptr(), stt.data_ptr())
    assert "DNAS_E_BAD_BASE" in str(e.value)
    dec.close()


def test_both_traceback_kernels_agree(da, ref_data):
    """The wave-per-read traceback (batches of up to 256 reads) and the thread-per-read one (larger batches, or
    traceback=thread) walk the same lattice to the same strings: noisy reads, local and global."""
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    rng = random.Random(17)
    reads = []
    for i in range(40):
        dna = list(m.encodeBytes(bytes(rng.randrange(256) for _ in range(2 + i % 6))))
        for j in range(len(dna)):
            if rng.random() < 0.03:
                dna[j] = rng.choice("ACGT")
        if i % 3 == 0:
            del dna[rng.randrange(len(dna))]
        reads.append("".join(dna))
    for flags in (dict(global_=True), dict()):
        params = da.MutatorParams.fromFlags(**flags)
        a = da.ViterbiDecoder(m, params)
        b = da.ViterbiDecoder(m, params, options="traceback=thread")
        out_a, ll_a, st_a = a.decode(reads)
        out_b, ll_b, st_b = b.decode(reads)
        assert out_a == out_b and np.array_equal(ll_a.view(np.uint64), ll_b.view(np.uint64)) and list(st_a) == list(st_b)
        a.close(); b.close()


@pytest.mark.parametrize("machine,options", [("s16h74l4c4.json", "traceback=thread"), ("s16mr2l4c4.json", "traceback=thread"),
                                             ("mr2l4c4.json", "traceback=thread,tier=B"), ("l4c4.json", "traceback=thread")])
def test_tracebacks_with_and_without_node_records(da, ref_data, machine, options, monkeypatch):
    """Both traceback kernels step from one 64-byte record per state where the state's in-edges fit one (three emitting,
    one null: device_model.h) and through the CSR arrays where they do not -- s16mr2l4c4 has states with 35 / 73 in-edges, the
    composites a start state with 98 -- or when DNAS_NO_NODE_RECORDS says so: the same strings either way, and the wave-per-read
    kernel's; substitutions, deletions and duplications in the reads, global and local, tiers A and B."""
    m = da.Machine.fromFile(os.path.join(ref_data, machine))
    rng = random.Random(23)
    reads = []
    for i in range(48):
        dna = list(m.encodeBytes(bytes(rng.randrange(256) for _ in range(1 + i % 9))))
        for j in range(len(dna)):
            if rng.random() < 0.04:
                dna[j] = rng.choice("ACGT")
        if i % 3 == 0 and len(dna) > 4:
            del dna[rng.randrange(len(dna))]
        if i % 4 == 1 and len(dna) > 6:
            at = rng.randrange(3, len(dna))
            dna[at:at] = dna[at - 3:at]          # a tandem duplication
        reads.append("".join(dna))
    for flags in (dict(global_=True), dict()):
        params = da.MutatorParams.fromFlags(**flags)
        wave_options = options.replace("traceback=thread,", "").replace("traceback=thread", "") or None
        wave = da.ViterbiDecoder(m, params, options=wave_options)
        want = wave.decode(reads)
        wave.close()
        monkeypatch.setenv("DNAS_NO_NODE_RECORDS", "1")      # the wave-per-read kernel through the CSR arrays
        wave = da.ViterbiDecoder(m, params, options=wave_options)
        got = wave.decode(reads)
        wave.close()
        assert got[0] == want[0] and np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64)) and list(got[2]) == list(want[2]), (machine, flags)
        for records in (True, False):
            if records:
                monkeypatch.delenv("DNAS_NO_NODE_RECORDS", raising=False)
            else:
                monkeypatch.setenv("DNAS_NO_NODE_RECORDS", "1")
            dec = da.ViterbiDecoder(m, params, options=options)
            got = dec.decode(reads)
            dec.close()
            assert got[0] == want[0] and np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64)) and list(got[2]) == list(want[2]), (machine, flags, records)


def test_arena_replanned_when_the_device_has_less_memory_than_at_creation(da, ref_data):
    """The arena cap is taken from the free memory when the model is created.  If the device cannot give that much any more
    (here: a tensor takes all but 20 GB afterwards), the call is planned again against what is free -- more, smaller batches --
    and returns the same results."""
    import torch
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    params = da.MutatorParams.fromFlags(global_=True)
    import random
    reads = []
    for i in range(300):                                        # one batch of 34 GB of lattice
        rng = random.Random(4000 + i)
        reads.append(m.encodeBytes(bytes(rng.randrange(256) for _ in range(29))))
    ref_dec = da.ViterbiDecoder(m, params)
    want = ref_dec.decode(reads)
    assert ref_dec.stats()["fill_launches"] == 1
    ref_dec.close()
    dec = da.ViterbiDecoder(m, params)
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(free - (20 << 30), dtype=torch.uint8, device="cuda")
    try:
        got = dec.decode(reads)
        st = dec.stats()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert st["fill_launches"] > 1 and st["checkpointed_reads"] == 0
    assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
    dec.close()


@pytest.mark.parametrize("options", ["plan_order=0", "plan_order=1", "plan_order=2", "plan_order=2,plan_slack=8", "plan_order=2,plan_slack=3"])
def test_full_lattice_under_every_dealing_order(da, oracle_mod, ref_data, options):
    """The row program may be dealt depth first, breadth first (the default without a tuning record) or by longest-path level
    (the tuning records pick per machine; the other tests run what the records say): the cells are the oracle's every time."""
    O = oracle_mod
    path = os.path.join(ref_data, "s16h74l4c4.json")
    flags = dict(global_=True)
    dec = da.ViterbiDecoder(da.Machine.fromFile(path), da.MutatorParams.fromFlags(**flags), options=options)
    orc = O.ViterbiOracle(O.Machine.from_file(path), O.MutatorParams.from_cli(**flags))
    reads = [seq for _, seq in da.read_fastseqs(os.path.join(ref_data, "hello.s16h74.del.fa"))]
    out, ll, st = dec.decode(reads)
    for i, r in enumerate(reads):
        s, oll, olat = orc.decode(r, want_lattice=True)
        assert out[i] == s and ll[i] == oll
        lat = np.ascontiguousarray(dec.lattice(i, len(r)).transpose(0, 2, 1))
        assert np.array_equal(lat.view(np.uint64), olat.view(np.uint64))
    dec.close()


def test_row_program_autotune(da, oracle_mod, ref_data, tmp_path, monkeypatch):
    """autotune=1 (opt-in): the first model of a tier-A machine WITHOUT a tuning record times the candidate row programs on
    synthetic reads, keeps the verdict in the kernel cache, and later models read it; for the fixture and bench machines the
    verdict ships with the library (dnastore_amd/tune/) and nothing is timed.  Whatever is picked, the results are the default
    program's, bit for bit."""
    from random_machines import random_machine, random_read
    monkeypatch.setenv("DNAS_KCACHE_DIR", str(tmp_path))
    params = da.MutatorParams.fromFlags(global_=True)
    # a machine nobody has seen: without autotune nothing is timed and nothing is written
    text = random_machine(77, 5000)
    m = da.Machine.fromJSON(text)
    reads = [random_read(500 + r, text, max_len=40) for r in range(5)]
    plain = da.ViterbiDecoder(m, params)
    assert "no tuning record" in plain.tier and not [f for f in os.listdir(tmp_path) if f.startswith("tune_")]
    want = plain.decode(reads)
    # with it: timed, one record (the arena of the timing models is the caller's)
    tuned = da.ViterbiDecoder(m, params, arena_bytes=2 << 30, options="autotune=1")
    notes = [f for f in os.listdir(tmp_path) if f.startswith("tune_")]
    assert len(notes) == 1
    verdict = open(os.path.join(tmp_path, notes[0])).read()
    assert verdict.startswith("order=") and " kernel=" + da.FlatModel.kernel_source_hash() in verdict
    got = tuned.decode(reads)
    again = da.ViterbiDecoder(m, params)                           # reads the record: no second one appears
    assert [f for f in os.listdir(tmp_path) if f.startswith("tune_")] == notes and "record " + notes[0] in again.tier
    assert got[0] == want[0] and np.array_equal(got[1].view(np.uint64), want[1].view(np.uint64)) and np.array_equal(got[2], want[2])
    for d in (tuned, again, plain):
        d.close()
    # a machine with a shipped verdict (dnastore_amd/tune/): nothing is timed, nothing is written
    water = da.Machine.compose(da.Machine.fromFile(os.path.join(ref_data, "water64.1.json")), da.Machine.fromFile(os.path.join(ref_data, "l4c4.json")))
    shipped = da.ViterbiDecoder(water, params, options="autotune=1")
    assert "record tune_" in shipped.tier          # (whether the record is of this kernel source: tests/test_tune_records.py)
    forced = [da.ViterbiDecoder(water, params, options="plan_order=%d" % v) for v in (1, 2)]
    assert [f for f in os.listdir(tmp_path) if f.startswith("tune_")] == notes
    assert forced[0].tier != forced[1].tier
    wreads = [water.encodeBytes(bytes(range(8 * i, 8 * i + 8))) for i in range(3)]
    a, b, c = shipped.decode(wreads), forced[0].decode(wreads), forced[1].decode(wreads)
    assert a[0] == b[0] == c[0] and np.array_equal(a[1].view(np.uint64), b[1].view(np.uint64)) and np.array_equal(a[1].view(np.uint64), c[1].view(np.uint64))
    for d in [shipped] + forced:
        d.close()


def test_packed_host_arrays_entry_point(da, oracle_mod, ref_data):
    """decode_packed = dnas_viterbi_batch on packed host arrays, results as arrays (what a C caller sees, and what bench.py times as
    the PCIe-inclusive rate): the same strings, log-likelihood bits and status as decode()."""
    m = da.Machine.fromFile(os.path.join(ref_data, "h74l4c4.json"))
    dec = da.ViterbiDecoder(m, da.MutatorParams.fromFlags(global_=True))
    reads = [m.encodeBytes(bytes([i, 3 * i % 256, 255 - i])) for i in range(9)] + [""]
    out, ll, st = dec.decode(reads)
    off, bases = da.pack_reads(reads)
    sym, ooff, olen, ll2, st2 = dec.decode_packed(off, bases)
    assert [sym[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes().decode() for i in range(len(reads))] == out
    assert np.array_equal(ll2.view(np.uint64), ll.view(np.uint64)) and np.array_equal(st2, st)
    dec.close()
