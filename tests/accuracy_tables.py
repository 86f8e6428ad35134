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

The published table trained its error model per row (`--fit-error` on ten simulated alignments, errdecode.pl:183-203) instead of
being told the rates; the two agree where the table is decisive: no edits at all up to a substitution rate of 0.004, and
about one edit per thousand bits at 0.128."""
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
