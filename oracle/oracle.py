"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; dnastore_amd/ never does.  It holds an independent (Python) reader for the
reference's Machine / error-model JSON, Stockholm and FASTA formats so that the oracle
and the product's C++ host code do not share a loader.

Reference formats: Machine JSON  src/trans.cpp:402-469 (lenient commas: gason.cpp:55-56,
295-299); MutatorParams JSON src/mutator.cpp:6-30; CLI error-model flags
t/dnastore.cpp:115-130.
"""
import ctypes
import gzip
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERRORS = {1: "context mismatch", 2: "not a DNA-outputting machine", 3: "transducer is cyclic",
          4: "unknown base in read", 5: "traceback failure", 6: "allocation failure", 7: "output overflow"}


def build(force=False):
    """Compile oracle/liboracle.so with gcc (oracle/Makefile)."""
    if os.environ.get("DNAS_ORACLE_LIBRARY"):      # another build of the oracle (tools/run_asan.sh: the sanitizer build)
        return os.environ["DNAS_ORACLE_LIBRARY"]
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("viterbi_oracle.c", "fwdback_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("DNAS_ORACLE_LIBRARY") or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_model_sym_logp.restype = ctypes.c_double
        _LIB.orc_model_alphabet.restype = ctypes.c_char_p
        _LIB.orc_model_pops.restype = ctypes.c_longlong
    return _LIB


# ----------------------------------------------------------------------------- lenient JSON
_TOK = re.compile(r'\s*(?:([{}\[\]:,])|"((?:[^"\\]|\\.)*)"|([^\s{}\[\]:,"]+))')


def parse_lenient_json(text):
    """JSON where commas are optional/trailing (what gason accepts, gason.cpp:55-56,295-299)."""
    toks = []
    for m in _TOK.finditer(text):
        if m.group(1):
            if m.group(1) != ",":
                toks.append(m.group(1))
        elif m.group(2) is not None:
            toks.append(("s", m.group(2)))
        elif m.group(3):
            toks.append(("a", m.group(3)))
    pos = 0

    def value():
        nonlocal pos
        t = toks[pos]
        pos += 1
        if t == "{":
            d = {}
            while toks[pos] != "}":
                k = toks[pos]
                assert isinstance(k, tuple) and k[0] == "s", "unquoted key"
                assert toks[pos + 1] == ":"
                pos += 2
                d[k[1]] = value()
            pos += 1
            return d
        if t == "[":
            a = []
            while toks[pos] != "]":
                a.append(value())
            pos += 1
            return a
        if isinstance(t, tuple):
            if t[0] == "s":
                return t[1].encode().decode("unicode_escape") if "\\" in t[1] else t[1]
            w = t[1]
            if w == "true":
                return True
            if w == "false":
                return False
            if w == "null":
                return None
            return float(w) if any(c in w for c in ".eE") else int(w)
        raise ValueError("unexpected token %r" % (t,))

    return value()


class Machine:
    """Flattened Machine (src/trans.h:82-126): per-state transition lists in file order."""

    def __init__(self, states):
        self.states = states  # list of dict(name, l, r, trans=[(in, out, to)])
        n = len(states)
        self.n = n
        tp = [0]
        tin, tout, tdest = [], [], []
        lp, lc, rp, rc = [0], [], [0], []
        for st in states:
            for (i, o, d) in st["trans"]:
                tin.append(ord(i) if i else 0)
                tout.append(ord(o) if o else 0)
                tdest.append(d)
            tp.append(len(tin))
            lc.extend(st["l"].encode())
            lp.append(len(lc))
            rc.extend(st["r"].encode())
            rp.append(len(rc))
        self.trans_ptr = np.array(tp, dtype=np.int32)
        self.trans_in = np.array(tin, dtype=np.int8)
        self.trans_out = np.array(tout, dtype=np.int8)
        self.trans_dest = np.array(tdest, dtype=np.int32)
        self.lctx_ptr = np.array(lp, dtype=np.int32)
        self.lctx = np.array(lc if lc else [0], dtype=np.int8)
        self.rctx_ptr = np.array(rp, dtype=np.int32)
        self.rctx = np.array(rc if rc else [0], dtype=np.int8)

    @staticmethod
    def from_json(text):
        # the reference concatenates lines without newlines before parsing (jsonutil.cpp:159-169)
        j = parse_lenient_json(text.replace("\n", ""))
        states = []
        for idx, js in enumerate(j["state"]):
            if "n" in js:
                assert int(js["n"]) == idx, "State n=%d out of sequence" % js["n"]  # trans.cpp:438-441
            trans = []
            for jt in js["trans"]:
                i = jt.get("in", "")
                o = jt.get("out", "")
                assert len(i) <= 1 and len(o) <= 1
                trans.append((i, o, int(jt["to"])))
            states.append(dict(name=js.get("id", ""), l=js.get("l", ""), r=js.get("r", ""), trans=trans))
        return Machine(states)

    @staticmethod
    def from_file(path):
        with open(path) as f:
            return Machine.from_json(f.read())


class MutatorParams:
    """src/mutator.h:9-31; CLI construction t/dnastore.cpp:119-129."""

    def __init__(self, pDelOpen=.001, pDelExtend=.01, pTanDup=.001, pTransition=None, pTransversion=None,
                 pLen=None, local=True, sub=.01, iv=10., length=12):
        self.pDelOpen, self.pDelExtend, self.pTanDup = pDelOpen, pDelExtend, pTanDup
        self.pTransition = sub * iv / (1 + iv) if pTransition is None else pTransition
        self.pTransversion = sub / (1 + iv) if pTransversion is None else pTransversion
        if pLen is None:
            n = length // 2
            pLen = [1. / n] * n  # initMaxDupLen, mutator.cpp:51-54
        self.pLen = list(pLen)
        self.local = bool(local)

    @staticmethod
    def from_cli(sub=.01, iv=10., dup=.001, del_open=.001, del_ext=.01, global_=False, length=12):
        return MutatorParams(pDelOpen=del_open, pDelExtend=del_ext, pTanDup=dup, sub=sub, iv=iv,
                             local=not global_, length=length)

    @staticmethod
    def from_json(text):
        j = parse_lenient_json(text)
        return MutatorParams(j["pDelOpen"], j["pDelExtend"], j["pTanDup"], j["pTransition"], j["pTransversion"],
                             j["pLen"], j["local"])


class _CParams(ctypes.Structure):
    _fields_ = [("pDelOpen", ctypes.c_double), ("pDelExtend", ctypes.c_double), ("pTanDup", ctypes.c_double),
                ("pTransition", ctypes.c_double), ("pTransversion", ctypes.c_double),
                ("nLen", ctypes.c_int), ("local", ctypes.c_int), ("pLen", ctypes.POINTER(ctypes.c_double))]


def _cparams(p):
    arr = (ctypes.c_double * max(1, len(p.pLen)))(*p.pLen)
    cp = _CParams(p.pDelOpen, p.pDelExtend, p.pTanDup, p.pTransition, p.pTransversion, len(p.pLen), int(p.local), arr)
    cp._keep = arr
    return cp


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


class ViterbiOracle:
    """(machine, params) -> per-read decode; mirrors decodeFastSeqs (viterbi.cpp:306-320)."""

    def __init__(self, machine, params):
        self.machine, self.params = machine, params
        self._h = ctypes.c_void_p()
        cp = _cparams(params)
        m = machine
        rc = lib().orc_model_create(
            ctypes.c_int(m.n), _ptr(m.trans_ptr, ctypes.c_int), _ptr(m.trans_in, ctypes.c_char),
            _ptr(m.trans_out, ctypes.c_char), _ptr(m.trans_dest, ctypes.c_int),
            _ptr(m.lctx_ptr, ctypes.c_int), _ptr(m.lctx, ctypes.c_char),
            _ptr(m.rctx_ptr, ctypes.c_int), _ptr(m.rctx, ctypes.c_char),
            ctypes.byref(cp), ctypes.byref(self._h))
        if rc != 0:
            raise ValueError("oracle model: " + ERRORS.get(rc, str(rc)))
        self.D = lib().orc_model_D(self._h)
        self.n = m.n

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().orc_model_free(self._h)
            self._h = ctypes.c_void_p()

    @property
    def alphabet(self):
        return lib().orc_model_alphabet(self._h).decode()

    def sym_logp(self, c):
        return lib().orc_model_sym_logp(self._h, ctypes.c_int(ord(c)))

    def edge_counts(self):
        a, b = ctypes.c_int(), ctypes.c_int()
        lib().orc_model_edge_counts(self._h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def scores(self):
        out = np.zeros(21 + len(self.params.pLen))
        lib().orc_model_scores(self._h, _ptr(out, ctypes.c_double))
        return out

    def decode(self, read, want_lattice=False):
        """-> (decoded symbol string, loglike[, lattice [L+1][N][D+2]]); '' when no valid path."""
        rb = read.encode() if isinstance(read, str) else bytes(read)
        L = len(rb)
        cap = 4 * (L + 16) + 4 * self.n
        out = ctypes.create_string_buffer(cap)
        olen, ll, steps = ctypes.c_int(), ctypes.c_double(), ctypes.c_int()
        lat = None
        latp = None
        if want_lattice:
            lat = np.empty((L + 1, self.n, self.D + 2), dtype=np.float64)
            latp = _ptr(lat, ctypes.c_double)
        rc = lib().orc_viterbi_read(self._h, rb, ctypes.c_int(L), out, ctypes.c_int(cap), ctypes.byref(olen),
                                    ctypes.byref(ll), latp, ctypes.byref(steps))
        if rc != 0:
            raise RuntimeError("oracle viterbi: " + ERRORS.get(rc, str(rc)))
        self.last_steps = steps.value
        self.last_pops = lib().orc_model_pops(self._h)
        s = out.raw[:olen.value].decode()
        return (s, ll.value, lat) if want_lattice else (s, ll.value)


# ----------------------------------------------------------------------------- FASTA
def read_fasta(path):
    """[(name, seq)] -- multi-line, optionally gzipped FASTA (fastseq.cpp:123-148); name = first word."""
    opener = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    recs = []
    with opener(path, "rt") as f:
        name, chunks = None, []
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(chunks)))
                name = line[1:].split()[0] if line[1:].split() else ""
                chunks = []
            elif name is not None:
                chunks.append("".join(line.split()))
        if name is not None:
            recs.append((name, "".join(chunks)))
    return recs


# ----------------------------------------------------------------------------- Stockholm / forward-backward
def read_stockholm(path):
    """readStockholmDatabase (stockholm.cpp:30-68,154-167): list of alignments, each [(name, gapped row)]."""
    db, rows, order = [], {}, []
    with open(path) as f:
        for line in f:
            s = line.strip()
            if not s:
                continue
            if s.startswith("//"):
                if order:
                    db.append([(n, rows[n]) for n in order])
                rows, order = {}, []
                continue
            if s.startswith("#"):
                continue
            parts = s.split()
            if len(parts) == 2:
                if parts[0] not in rows:
                    order.append(parts[0])
                    rows[parts[0]] = ""
                rows[parts[0]] += parts[1]
    if order:
        db.append([(n, rows[n]) for n in order])
    return db


def alignment_pair(rows):
    """2-row alignment -> (in tokens, out tokens, cmIn, cmOut) as GuideAlignmentEnvelope sees it
    (alignpath.cpp:189-204,237-265): cm*[pos] = cumulative matches at the alignment column of `pos`."""
    assert len(rows) == 2, "Training mutator model requires a 2-row alignment"   # fwdback.cpp:25
    g1, g2 = rows[0][1], rows[1][1]
    assert len(g1) == len(g2)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    is_gap = lambda c: c in "-."
    cum, matches = [0], 0
    p1, p2 = [0], [0]
    for col, (a, b) in enumerate(zip(g1, g2)):
        if not is_gap(a):
            p1.append(col + 1)
        if not is_gap(b):
            p2.append(col + 1)
        if not is_gap(a) and not is_gap(b):
            matches += 1
        cum.append(matches)
    ins = np.array([code[c.upper()] for c in g1 if not is_gap(c)], dtype=np.int8)
    outs = np.array([code[c.upper()] for c in g2 if not is_gap(c)], dtype=np.int8)
    return ins, outs, np.array([cum[c] for c in p1], dtype=np.int32), np.array([cum[c] for c in p2], dtype=np.int32)


def pack_pairs(pairs):
    """[(ins, outs, cmIn, cmOut)] -> concatenated arrays + offsets (what orc_expected_counts / the C ABI take)."""
    def cat(idx, dt):
        arrs = [p[idx] for p in pairs]
        off = np.zeros(len(arrs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(a) for a in arrs])
        data = np.concatenate(arrs).astype(dt) if arrs and off[-1] else np.zeros(1, dt)
        return data, off
    ins, in_off = cat(0, np.int8)
    outs, out_off = cat(1, np.int8)
    cmi, cmi_off = cat(2, np.int32)
    cmo, cmo_off = cat(3, np.int32)
    return dict(ins=ins, in_off=in_off, outs=outs, out_off=out_off, cm_in=cmi, cm_in_off=cmi_off, cm_out=cmo,
                cm_out_off=cmo_off, n=len(pairs))


def _fb_args(pk):
    i64 = ctypes.c_int64
    return [_ptr(pk["ins"], ctypes.c_int8), _ptr(pk["in_off"], i64), _ptr(pk["outs"], ctypes.c_int8), _ptr(pk["out_off"], i64),
            _ptr(pk["cm_in"], ctypes.c_int), _ptr(pk["cm_in_off"], i64), _ptr(pk["cm_out"], ctypes.c_int), _ptr(pk["cm_out_off"], i64)]


def expected_counts(params, pairs, strict=False):
    """expectedCounts (fwdback.cpp:190-209) -> (counts float64[21+P], total ll, per-pair ll)."""
    L = lib()
    L.orc_expected_counts.restype = ctypes.c_double
    pk = pack_pairs(pairs)
    cp = _cparams(params)
    counts = np.zeros(21 + len(params.pLen))
    per = np.zeros(max(pk["n"], 1))
    ll = L.orc_expected_counts(ctypes.byref(cp), ctypes.c_int(int(strict)), ctypes.c_int(pk["n"]), *_fb_args(pk),
                               _ptr(counts, ctypes.c_double), _ptr(per, ctypes.c_double))
    return counts, ll, per[:pk["n"]]


def baum_welch(params, pairs, strict=False):
    """baumWelchParams with the Laplace prior (fwdback.cpp:211-230, dnastore.cpp:137-138) -> fitted MutatorParams."""
    L = lib()
    pk = pack_pairs(pairs)
    p5 = np.array([params.pDelOpen, params.pDelExtend, params.pTanDup, params.pTransition, params.pTransversion])
    plen = np.array(params.pLen, dtype=np.float64)
    L.orc_baum_welch(_ptr(p5, ctypes.c_double), ctypes.c_int(len(plen)), ctypes.c_int(int(params.local)),
                     ctypes.c_int(int(strict)), ctypes.c_int(pk["n"]), *_fb_args(pk), _ptr(plen, ctypes.c_double))
    return MutatorParams(p5[0], p5[1], p5[2], p5[3], p5[4], list(plen), params.local)


def _g(x):
    return "%g" % x          # default ostream formatting: 6 significant digits


def counts_json(counts, P):
    """MutatorCounts::writeJSON (mutator.cpp:108-124), byte for byte."""
    sub = counts[5:21].reshape(4, 4)
    trans = lambda i, j: i != j and (i & 1) == (j & 1)
    n_match = sum(sub[i][i] for i in range(4))
    n_ti = sum(sub[i][j] for i in range(4) for j in range(4) if trans(i, j))
    n_tv = sum(sub[i][j] for i in range(4) for j in range(4) if i != j and not trans(i, j))
    s = "{\n"
    s += ' "nDelOpen": %s,\n "nTanDup": %s,\n "nNoGap": %s,\n "nDelExtend": %s,\n "nDelEnd": %s,\n' % tuple(
        _g(counts[i]) for i in (0, 1, 2, 3, 4))
    s += ' "nLen": [ %s ],\n' % ", ".join(_g(x) for x in counts[21:21 + P])
    s += ' "nSub": [ %s ],\n' % ", ".join("[" + ",".join(_g(x) for x in row) + "]" for row in sub)
    s += ' "nMatch": %s,\n "nTransition": %s,\n "nTransversion": %s\n}\n' % (_g(n_match), _g(n_ti), _g(n_tv))
    return s


def params_json(p):
    """MutatorParams::writeJSON (mutator.cpp:6-16), byte for byte."""
    return ("{\n \"pDelOpen\": %s,\n \"pDelExtend\": %s,\n \"pTanDup\": %s,\n \"pTransition\": %s,\n \"pTransversion\": %s,\n"
            " \"pLen\": [ %s ],\n \"local\": %s\n}\n") % (_g(p.pDelOpen), _g(p.pDelExtend), _g(p.pTanDup), _g(p.pTransition),
                                                        _g(p.pTransversion), ", ".join(_g(x) for x in p.pLen),
                                                        "true" if p.local else "false")
