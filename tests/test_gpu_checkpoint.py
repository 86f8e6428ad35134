"""Bounded-memory decode: reads whose lattice does not fit the arena are filled in segments from
checkpoints and traced back segment by segment.  The reference keeps every read's whole lattice (viterbi.h:48-50) and
therefore has no such path; what it must reproduce is the whole-lattice result, bit for bit: decoded symbols,
log-likelihood, status and the traceback's event log, for every tier, both modes (local / --error-global), ragged
and empty reads, segment lengths from the minimum (D + 2 columns) up, and for a call that mixes reads that fit with
reads that do not."""
import os
import random
import re
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def da():
    import dnastore_amd
    return dnastore_amd


def _reads(da, m, rng, payload_sizes, rate=0.02):
    out = []
    for nbytes in payload_sizes:
        dna = list(m.encodeBytes(bytes(rng.randrange(256) for _ in range(nbytes))))
        i = 0
        noisy = []
        while i < len(dna):            # substitutions, deletions and tandem duplications
            u = rng.random()
            if u < rate:
                noisy.append(rng.choice([b for b in "ACGT" if b != dna[i]]))
            elif u < 1.5 * rate:
                pass
            elif u < 2.0 * rate and i + 3 < len(dna):
                noisy.extend(dna[i:i + 3])
                noisy.append(dna[i])
            else:
                noisy.append(dna[i])
            i += 1
        out.append("".join(noisy))
    return out


def _same(a, b):
    (oa, la, sa), (ob, lb, sb) = a, b
    assert oa == ob
    assert np.array_equal(la.view(np.uint64), lb.view(np.uint64))
    assert np.array_equal(sa, sb)


@pytest.mark.parametrize("mach,flags,options,segments", [
    ("s16h74l4c4.json", dict(global_=True), "", (6, 7, 16, 100)),                 # tier A
    ("s16h74l4c4.json", dict(), "", (6, 33)),                                     # tier A, local alignment
    ("h74l4c4.json", dict(global_=True), "tier=B", (6, 19)),                      # the general kernel: every lane stored
    ("h74l4c4.json", dict(), "tier=B", (8,)),
    ("s16h74l4c4.json", dict(global_=True), "tier=C,cluster=3,threads=512", (6, 40)),
    ("s16h74l4c4.json", dict(), "tier=C,cluster=2,threads=1024", (11,)),
    ("l4c4.json", dict(sub=0., del_open=0., global_=True), "", (6,)),             # zero-probability edits (-inf scores)
])
def test_segments_match_whole_lattice(da, ref_data, mach, flags, options, segments):
    m = da.Machine.fromFile(os.path.join(ref_data, mach))
    params = da.MutatorParams.fromFlags(**flags)
    rng = random.Random(zlib.crc32((mach + options).encode()))
    noise = 0.0 if "sub" in flags else 0.02
    reads = _reads(da, m, rng, [1, 2, 3, 5, 8, 13, 21, 29, 29, 16, 4], rate=noise) + ["", "A", "ACGTA", "ACGTAC", "ACGTACG"]
    whole = da.ViterbiDecoder(m, params, options=options or None)
    ref = whole.decode(reads)
    assert whole.stats()["checkpointed_reads"] == 0
    whole.close()
    for seg in segments:
        opt = ",".join(x for x in (options, "checkpoint=always,segment=%d" % seg) if x)
        dec = da.ViterbiDecoder(m, params, options=opt)
        got = dec.decode(reads)
        st = dec.stats()
        assert st["checkpointed_reads"] == len(reads)
        longest = max(len(r) for r in reads)
        assert st["fill_launches"] >= 2 * (longest // seg + 1) - 1      # every segment but the last is filled twice
        _same(got, ref)
        # a second call over the same handle (walks and tables are rebuilt) and a different batch composition
        _same(dec.decode(reads[3:9]), tuple(x[3:9] for x in ref))
        with pytest.raises(da.DnasError):
            dec.lattice(0, len(reads[0]))                               # no whole lattice exists
        dec.close()


def test_event_log_across_segments(da, ref_data):
    """The level-3 traceback messages (viterbi.cpp:266-293) of a walk that is parked and resumed are those of the whole walk."""
    path = os.path.join(ref_data, "s16h74l4c4.json")
    m = da.Machine.fromFile(path)
    params = da.MutatorParams.fromFlags(global_=True)
    rng = random.Random(5)
    reads = _reads(da, m, rng, [29, 20, 11], rate=0.03)
    names = ["r%d" % i for i in range(len(reads))]
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".fa", delete=False) as f:
        for n, r in zip(names, reads):
            f.write(">%s\n%s\n" % (n, r))
        fa = f.name
    try:
        whole = da.decode_fastseqs(fa, m, params, events=True)
        os.environ["DNAS_CHECKPOINT"] = "always"
        os.environ["DNAS_SEGMENT"] = "9"
        info = {}
        seg = da.decode_fastseqs(fa, m, params, events=True, info=info)
    finally:
        os.environ.pop("DNAS_CHECKPOINT", None)
        os.environ.pop("DNAS_SEGMENT", None)
        os.unlink(fa)
    assert seg == whole
    assert any(len(rec[3]) > 0 for rec in whole)           # the noisy reads did produce events


def test_reads_beyond_the_arena(da, ref_data):
    """checkpoint=auto: with an arena that holds only the short reads' lattices, the long reads go through the segments
    and the short ones through the usual batches, in one call; checkpoint=never restores the old refusal."""
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    params = da.MutatorParams.fromFlags(global_=True)
    rng = random.Random(99)
    sizes = [29, 2, 29, 3, 1, 29, 5, 2, 29, 4]
    reads = _reads(da, m, rng, sizes, rate=0.01)
    whole = da.ViterbiDecoder(m, params)
    ref = whole.decode(reads)
    shape = re.search(r"T(\d+)K(\d+)", whole.tier)                      # tier A: K rows x T threads of lattice slots, S and D lanes
    n_slots = int(shape.group(1)) * int(shape.group(2))
    whole.close()
    longest, shortest_long = max(len(r) for r in reads), min(len(r) for r, s in zip(reads, sizes) if s == 29)
    col = 2 * n_slots * 8
    # half the arena holds a lattice of 120 columns: the 29-byte reads (~490 nt) do not fit, the others do
    arena = 2 * 120 * col
    assert shortest_long > 130 and max(len(r) for r, s in zip(reads, sizes) if s != 29) < 110
    dec = da.ViterbiDecoder(m, params, arena_bytes=arena)
    got = dec.decode(reads)
    assert dec.stats()["checkpointed_reads"] == sizes.count(29)
    _same(got, ref)
    short_index = sizes.index(2)
    lat = dec.lattice(short_index, len(reads[short_index]))            # a read of the usual batches still has its lattice
    assert np.isfinite(lat[-1, 0, -1])
    with pytest.raises(da.DnasError):
        dec.lattice(0, len(reads[0]))
    dec.close()
    never = da.ViterbiDecoder(m, params, arena_bytes=arena, options="checkpoint=never")
    with pytest.raises(da.DnasError) as e:
        never.decode(reads)
    assert "exceeds the lattice arena" in str(e.value)
    never.close()
    # an arena too small for even one read's segments is still an error, and says what it would take
    tiny = da.ViterbiDecoder(m, params, arena_bytes=8 * col)
    with pytest.raises(da.DnasError) as e:
        tiny.decode(reads[:1])
    assert "checkpoints" in str(e.value)
    tiny.close()
    assert longest > 0


def test_long_read_roundtrip(da, ref_data):
    """A 10 kb read (20x the bench read) through a 1 GB arena: the whole lattice would take 2.3 GB.  Property: the
    payload comes back, and the log-likelihood equals the whole-lattice decoder's bit for bit."""
    m = da.Machine.fromFile(os.path.join(ref_data, "s16h74l4c4.json"))
    params = da.MutatorParams.fromFlags(global_=True)
    rng = random.Random(2026)
    payload = bytes(rng.randrange(256) for _ in range(600))
    dna = m.encodeBytes(payload)
    assert len(dna) > 9000
    dec = da.ViterbiDecoder(m, params, arena_bytes=1 << 30)
    out, ll, st = dec.decode([dna])
    assert dec.stats()["checkpointed_reads"] == 1 and st[0] == 0
    assert da.symbolsToBytes(out[0]) == payload
    whole = da.ViterbiDecoder(m, params, arena_bytes=6 << 30)
    out2, ll2, st2 = whole.decode([dna])
    assert whole.stats()["checkpointed_reads"] == 0
    assert out2 == out and ll2.view(np.uint64)[0] == ll.view(np.uint64)[0]
    dec.close()
    whole.close()
