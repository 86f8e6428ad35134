"""The reference's accuracy experiment (doc/errdecode.pl) restated: random payload bits -> exact encoding through a machine ->
point substitutions -> Viterbi error decoding -> Levenshtein edits per payload bit.  The reference publishes the outcome for
Hamming(7,4) * DNASTORE(4) in doc/len4.ham.subs.tab (kept as a data fixture under tests/golden/ref_doc/); the tests reproduce
rows of it on the GPU path.  What is restated, with the lines it follows:

  * 8192 random bits per repetition, 20 repetitions per row (doc/Makefile:127 `-bits 8192 ... -reps 20`, errdecode.pl:205-207)
  * machine `--length 4 --controls 4 --compose-machine hamming74.json` = data/h74l4c4.json (errdecode.pl:151-176, doc/Makefile:150)
  * substitutions: round(rate * len) draws of a random position (a position may be hit twice), each a transversion with
    probability 1 / (1 + ivratio) (one of the two at random) and the transition otherwise, ivratio 10 (errdecode.pl:265-272,
    287-293, 308-318, :33)
  * decoding `-V --error-global --error-sub-prob <rate> --error-dup-prob 0 --error-del-open 0 --error-del-ext 0.2 --length 4`
    (errdecode.pl:229-231 with -exacterrs; delext = 2 / maxdelsize = 0.2, :109; `--length 4` -> P = 2), control symbols
    stripped (:233), Levenshtein distance to the payload / 8192 (:242-244)

The published tables trained their error model per row (`--fit-error --error-global` on ten simulated alignments of 8192 random
bases, the fitted model written as JSON and read back with `--error-file`: errdecode.pl:183-203, doc/Makefile:127 `-trainalign 10`)
instead of being told the rates.  fit_model() below does exactly that through the GPU Baum-Welch path; make_case() / the
`exact` model is the script's `-exacterrs` variant.  Deletions (doc/len4*.dels.tab): round(rate * len) draws of a segment of
1..10 bases at a random place, removed (errdecode.pl:265-293, 301-305; -maxdelsize 10, :32); tandem duplications
(doc/len4.mix2.dups.tab): segments of 1..4 bases copied in place, no overlaps (:31, 295-299).  The machines: `--length 4
--controls 4` alone (data/l4c4.json: len4.*.tab), composed with mixradar2.json (mr2l4c4.json: len4.mix2.*.tab) or with
hamming74.json (h74l4c4.json: len4.ham.*.tab) -- errdecode.pl:151-176, doc/Makefile:129-153."""
import random

import numpy as np

TRANSITION = {"A": "G", "C": "T", "G": "A", "T": "C"}
TRANSVERSION = {"A": "CT", "C": "AG", "G": "CT", "T": "AG"}
BITS = 8192


def read_table(path):
    """rows of a doc/*.tab file: dicts with SubProb, MeanEditsPerBit, StDevEditsPerBit (the first column is the row number)"""
    lines = [l.split() for l in open(path) if l.strip()]
    head = lines[0]
    return [dict(zip(head, map(float, l[1:]))) for l in lines[1:]]


def substitute(rng, dna, rate, ivratio=10.0):
    seq = list(dna)
    for _ in range(int(round(rate * len(seq)))):
        pos = int(rng.random() * len(seq))
        base = seq[pos]
        seq[pos] = TRANSVERSION[base][int(rng.random() * 2)] if rng.random() < 1.0 / (1.0 + ivratio) else TRANSITION[base]
    return "".join(seq)


def make_case(machine, rate, rep, bits=BITS):
    """-> (payload bit string, mutated DNA read); deterministic per (rate, repetition)"""
    rng = random.Random("len4.ham.subs %r %d" % (rate, rep))
    payload = "".join(rng.choice("01") for _ in range(bits))
    return payload, substitute(rng, machine.encodeSymbols(payload), rate)


def edit_distance(a, b):
    """Levenshtein distance of two strings: a banded dynamic programme (band doubled until the answer lies inside it), rows as
    numpy vectors; the in-row recurrence D[j] = min(t[j], D[j-1] + 1) is a running minimum of t[j] - j."""
    if a == b:
        return 0
    if not a or not b:
        return len(a) + len(b)
    x = np.frombuffer(a.encode(), dtype=np.uint8)
    y = np.frombuffer(b.encode(), dtype=np.uint8)
    n, m = len(x), len(y)
    band = max(64, abs(n - m) + 8)
    big = n + m + 1
    while True:
        # row i covers columns j = i - band .. i + band (clipped to 0 .. m); cell (i, j) lives at index j - i + band
        width = 2 * band + 1
        off = np.arange(width) - band
        prev = np.where((off >= 0) & (off <= m), off, big).astype(np.int64)      # row 0: D[0][j] = j
        idx = np.arange(width, dtype=np.int64)
        for i in range(1, n + 1):
            j = off + i
            ok = (j >= 0) & (j <= m)
            # diagonal (i-1, j-1) sits at the same index in prev; up (i-1, j) at index + 1
            up = np.concatenate([prev[1:], [big]]) + 1
            jj = np.clip(j - 1, 0, m - 1)
            diag = prev + (y[jj] != x[i - 1])
            diag = np.where(j >= 1, diag, big)
            t = np.minimum(up, diag)
            t = np.where(j == 0, i, t)
            t = np.where(ok, t, big)
            cur = np.minimum.accumulate(t - idx) + idx
            prev = np.where(ok, np.minimum(cur, big), big)
        d = int(prev[m - n + band]) if 0 <= m - n + band < width else big
        if d <= band:
            return d
        band *= 2


# ---- deletions, training alignments, the fitted model -----------------------------------------------------------------------

def delete(rng, seq_cols, rate, maxsize=10):
    """seq_cols: list of (original index or None, base) of the current sequence; round(rate * len) draws of a segment of 1..maxsize
    bases at a random place, removed (errdecode.pl evolve / randcoords / del)."""
    n = int(round(rate * len(seq_cols)))
    for _ in range(n):
        ln = len(seq_cols)
        if ln == 0:
            break
        size = int(rng.random() * (min(ln, maxsize) + 1 - 1)) + 1
        pos = int(rng.random() * (ln + 1 - size))
        del seq_cols[pos:pos + size]
    return seq_cols


def duplicate(rng, cols, rate, maxsize=4):
    """cols: list of [original index or None, base, mutated]; round(rate * len) draws of a segment of 1..maxsize bases; a segment
    that touches an earlier duplication's copy is skipped (no overlaps: errdecode.pl:270-273, :31), else its copy is inserted
    right behind it (errdecode.pl dup, :295-299)"""
    n = int(round(rate * len(cols)))
    for _ in range(n):
        ln = len(cols)
        size = int(rng.random() * (min(ln, maxsize) + 1 - 1)) + 1
        pos = int(rng.random() * (ln + 1 - size))
        seg = cols[pos:pos + size]
        if any(c[2] for c in seg):
            continue
        cols[pos + size:pos + size] = [[None, c[1], True] for c in seg]
    return cols


def evolve(rng, dna, sub_rate, del_rate, dup_rate=0.0, ivratio=10.0):
    """duplications, then substitutions, then deletions, as errdecode.pl applies them (:214-216)
    -> (read, alignment rows (original, read) with gaps)"""
    cols = [[i, b, False] for i, b in enumerate(dna)]
    if dup_rate > 0:
        cols = duplicate(rng, cols, dup_rate)
    if sub_rate > 0:
        for _ in range(int(round(sub_rate * len(cols)))):
            pos = int(rng.random() * len(cols))
            base = cols[pos][1]
            cols[pos][1] = TRANSVERSION[base][int(rng.random() * 2)] if rng.random() < 1.0 / (1.0 + ivratio) else TRANSITION[base]
            cols[pos][2] = True
    if del_rate > 0:
        cols = delete(rng, cols, del_rate)
    read = "".join(c[1] for c in cols)
    # alignment: walk the original; read-only columns (copies) go where they stand
    row_old, row_new, nxt = [], [], 0
    for c in cols:
        if c[0] is None:
            row_old.append("-"); row_new.append(c[1])
            continue
        while nxt < c[0]:
            row_old.append(dna[nxt]); row_new.append("-"); nxt += 1
        row_old.append(dna[c[0]]); row_new.append(c[1]); nxt = c[0] + 1
    while nxt < len(dna):
        row_old.append(dna[nxt]); row_new.append("-"); nxt += 1
    return read, ("".join(row_old), "".join(row_new))


def make_case_general(machine, sub_rate, del_rate, rep, tag, bits=BITS, dup_rate=0.0):
    rng = random.Random("%s %r %r %r %d" % (tag, sub_rate, del_rate, dup_rate, rep) if dup_rate else "%s %r %r %d" % (tag, sub_rate, del_rate, rep))
    payload = "".join(rng.choice("01") for _ in range(bits))
    read, _ = evolve(rng, machine.encodeSymbols(payload), sub_rate, del_rate, dup_rate)
    return payload, read


def training_stockholm(sub_rate, del_rate, tag, n=10, length=BITS, dup_rate=0.0):
    """ten alignments of a random sequence of 8192 bases and its mutated copy (errdecode.pl:187-199), as Stockholm text"""
    out = []
    for k in range(n):
        rng = random.Random("train %s %r %r %r %d" % (tag, sub_rate, del_rate, dup_rate, k) if dup_rate else "train %s %r %r %d" % (tag, sub_rate, del_rate, k))
        orig = "".join(rng.choice("ACGT") for _ in range(length))
        _, (row_old, row_new) = evolve(rng, orig, sub_rate, del_rate, dup_rate)
        out.append("# STOCKHOLM 1.0\nold %s\nnew %s\n//\n" % (row_old, row_new))
    return "".join(out)


def fit_model(da, sub_rate, del_rate, tag, workdir, dup_rate=0.0):
    """`dnastore --length 4 --fit-error train.stk --error-global` (errdecode.pl:201), the printed JSON read back as --error-file:
    -> (MutatorParams as the decoder loads them, the JSON text, Baum-Welch iterations)"""
    import os
    stk = os.path.join(workdir, "train.%s.stk" % tag)
    open(stk, "w").write(training_stockholm(sub_rate, del_rate, tag, dup_rate=dup_rate))
    init = da.MutatorParams.fromFlags(global_=True, length=4)          # the CLI's defaults otherwise (t/dnastore.cpp:115-130)
    fit, iters = da.baumWelchParams(init, da.StockholmDB(stk))
    text = da.paramsJSON(fit)
    err = os.path.join(workdir, "fit.%s.err.json" % tag)
    open(err, "w").write(text)
    return da.MutatorParams.fromFile(err), text, iters


MACHINE_OF_TABLE = {"len4": "l4c4.json", "len4.mix2": "mr2l4c4.json", "len4.ham": "h74l4c4.json"}


def machine_file(table_stem):
    """'len4.mix2.dups' -> 'mr2l4c4.json'"""
    return MACHINE_OF_TABLE[table_stem.rsplit(".", 1)[0]]
