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
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("viterbi_oracle.c", "fwdback_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
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
