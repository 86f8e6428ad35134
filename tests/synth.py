"""Synthetic-read generator for the parity tests (test infrastructure).

encode(): a state-set tracking encoder with the semantics of the reference's Encoder
(src/encoder.h:35-186: epsilon-closure `expand`, one transition per input symbol, FLUSH
inserted when a symbol cannot be taken), checked in test_synth_encoder_goldens against the
reference's own encoded fixtures data/hello*.fa.  mutate(): i.i.d. substitutions, deletions
and tandem duplications with a seeded RNG.
"""
import random


def _expand(machine, cur):
    while True:
        nxt, found = {}, False
        for s, q in cur.items():
            tr = machine.states[s]["trans"]
            if not tr or any(i for (i, o, d) in tr):
                nxt[s] = q
        for s, q in cur.items():
            for (i, o, d) in machine.states[s]["trans"]:
                if not i:
                    nq = q + o
                    if d in cur:
                        assert cur[d] == nq, "two possible output queues"
                    if d not in nxt:
                        found = found or d not in cur
                    nxt[d] = nq
        if not found:
            return nxt
        cur = nxt


def encode(machine, symbols):
    """machine: oracle.Machine; symbols: e.g. '^0110$' -> emitted DNA string."""
    cur = _expand(machine, {0: ""})

    def step(cur, c):
        nxt = {}
        for s, q in cur.items():
            for (i, o, d) in machine.states[s]["trans"]:
                if i == c:
                    nxt[d] = q + o
        return nxt
    for c in symbols:
        if c != "." and not any(i == c for s in cur for (i, o, d) in machine.states[s]["trans"]):
            cur = _expand(machine, step(cur, "."))
        nxt = step(cur, c)
        assert nxt, "can't encode symbol %r" % c
        cur = _expand(machine, nxt)
    ends = [q for s, q in cur.items() if not machine.states[s]["trans"]]
    assert len(ends) == 1, "encoder unresolved"
    return ends[0]


def bytes_to_symbols(payload):
    """^ + bits LSB-first (encoder.h:222-231) + $"""
    return "^" + "".join(str((b >> n) & 1) for b in payload for n in range(8)) + "$"


def mutate(seq, rng, sub=0.0, dele=0.0, dup=0.0):
    out = []
    i = 0
    while i < len(seq):
        c = seq[i]
        r = rng.random()
        if r < dele:
            i += 1 + (1 if rng.random() < 0.3 else 0)
            continue
        if r < dele + dup and i >= 3:
            k = rng.randint(1, 3)
            out.append(seq[i])
            out.extend(seq[i - k + 1:i + 1])
            i += 1
            continue
        if rng.random() < sub:
            c = rng.choice([b for b in "ACGT" if b != c])
        out.append(c)
        i += 1
    return "".join(out)


def synthetic_reads(machine, n, nbytes, seed, sub=0.01, dele=0.0, dup=0.0):
    reads = []
    for r in range(n):
        rng = random.Random(seed + r)
        payload = bytes(rng.randrange(256) for _ in range(nbytes))
        reads.append(mutate(encode(machine, bytes_to_symbols(payload)), rng, sub, dele, dup))
    return reads


def synthetic_alignment(rng, length, sub=0.02, dele=0.01, dup=0.01):
    """A random (original, read) pair with its true alignment as two gapped rows (the guide
    the reference's --error-counts / --fit-error take): tandem duplications of length 1-3,
    deletions, substitutions."""
    src = "".join(rng.choice("ACGT") for _ in range(length))
    r1, r2 = [], []
    i = 0
    while i < len(src):
        r = rng.random()
        if r < dele:
            r1.append(src[i]); r2.append("-"); i += 1
            continue
        c = src[i]
        if rng.random() < sub:
            c = rng.choice([b for b in "ACGT" if b != c])
        r1.append(src[i]); r2.append(c); i += 1
        if r < dele + dup and i >= 3:
            k = rng.randint(1, 3)
            for b in src[i - k:i]:
                r1.append("-"); r2.append(b)
    return [("in", "".join(r1)), ("out", "".join(r2))]
