"""Python mirror of the reference's interface for the Viterbi path, over the C ABI.

Same names and argument meaning as the reference: Machine.fromFile (trans.cpp:477-482),
MutatorParams from CLI flags (t/dnastore.cpp:119-129) or --error-file JSON
(mutator.cpp:18-49), decodeFastSeqs(filename, machine, params) (viterbi.cpp:306-320),
ViterbiMatrix.traceback()/loglike() (viterbi.h:94-102) batched as ViterbiDecoder.decode.
All computation happens in libdnastore_amd.so; nothing here touches the DP.
"""
import ctypes

import numpy as np

from . import lib as _l


class Machine:
    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def fromFile(path):
        h = ctypes.c_void_p()
        _l.check(_l.lib().dnas_machine_load_json(str(path).encode(), ctypes.byref(h)))
        return Machine(h)

    @staticmethod
    def fromJSON(text):
        b = text.encode() if isinstance(text, str) else text
        h = ctypes.c_void_p()
        _l.check(_l.lib().dnas_machine_parse_json(b, len(b), ctypes.byref(h)))
        return Machine(h)

    def nStates(self):
        return _l.lib().dnas_machine_n_states(self._h)

    def toJSON(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _l.check(_l.lib().dnas_machine_write_json(self._h, ctypes.byref(p), ctypes.byref(n)))
        s = ctypes.string_at(p, n.value).decode()
        _l.lib().dnas_free(p)
        return s

    def _encode(self, fn, data):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _l.check(fn(self._h, data, len(data), ctypes.byref(p), ctypes.byref(n)))
        s = ctypes.string_at(p, n.value).decode()
        _l.lib().dnas_free(p)
        return s

    @staticmethod
    def compose(first, second):
        """Machine::compose(first, second) (trans.cpp:505-602)."""
        h = ctypes.c_void_p()
        _l.check(_l.lib().dnas_machine_compose(first._h, second._h, ctypes.byref(h)))
        return Machine(h)

    def decodeExact(self, dna):
        """Decoder<ostream>::decodeString + close (decoder.h:123-190): DNA -> symbol string."""
        return self._encode(_l.lib().dnas_decode_exact, dna.encode() if isinstance(dna, str) else dna)

    def encodeSymbols(self, symbols):
        """Encoder::encodeSymbolString + close (encoder.h:33-57,238-241) -> DNA string."""
        return self._encode(_l.lib().dnas_encode_symbols, symbols.encode() if isinstance(symbols, str) else symbols)

    def encodeBytes(self, payload):
        """Encoder::encodeString/encodeStream (encoder.h:222-237): bits LSB first -> DNA string."""
        return self._encode(_l.lib().dnas_encode_bytes, bytes(payload))

    def __del__(self):
        try:      # (at interpreter shutdown the module globals may be gone already)
            if getattr(self, "_h", None) is not None and self._h.value:
                _l.lib().dnas_machine_free(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


class MutatorParams:
    def __init__(self, c):
        self.c = c

    @staticmethod
    def fromFlags(sub=.01, iv=10., dup=.001, del_open=.001, del_ext=.01, global_=False, length=12):
        """--error-sub-prob/--error-iv-ratio/--error-dup-prob/--error-del-open/--error-del-ext/--error-global/--length."""
        c = _l.MutatorParamsC()
        _l.check(_l.lib().dnas_mutator_params_from_flags(sub, iv, dup, del_open, del_ext, int(bool(global_)), int(length),
                                                         ctypes.byref(c)))
        return MutatorParams(c)

    @staticmethod
    def fromFile(path):
        c = _l.MutatorParamsC()
        _l.check(_l.lib().dnas_mutator_params_load_json(str(path).encode(), ctypes.byref(c)))
        return MutatorParams(c)

    @property
    def local(self):
        return bool(self.c.local)

    @property
    def pLen(self):
        return [self.c.p_len[i] for i in range(self.c.n_len)]


_BASE = np.full(256, 255, dtype=np.uint8)
for _i, _ch in enumerate("ACGT"):
    _BASE[ord(_ch)] = _i
    _BASE[ord(_ch.lower())] = _i


def tokenize(seq):
    """FastSeq::tokens over ACGT, case-insensitive (fastseq.cpp:9-39); raises on any other character."""
    b = np.frombuffer(seq.encode() if isinstance(seq, str) else bytes(seq), dtype=np.uint8)
    t = _BASE[b]
    if (t == 255).any():
        bad = chr(int(b[np.argmax(t == 255)]))
        raise ValueError("Unknown symbol %s in sequence (alphabet is ACGT)" % bad)
    return t


class FlatModel:
    """MachineScores + InputModel + MutatorScores as flat arrays (dnas_flatten)."""

    def __init__(self, machine, params):
        self._h = ctypes.c_void_p()
        _l.check(_l.lib().dnas_flatten(machine._h, ctypes.byref(params.c), ctypes.byref(self._h)))
        self.view = _l.lib().dnas_flat_view(self._h)

    def arrays(self):
        v = self.view.contents
        n, ne, nn, d = v.n_states, v.n_emit, v.n_null, max(1, v.max_dup_len)

        def arr(p, k, dt):
            return np.ctypeslib.as_array(p, shape=(max(k, 1),))[:k].astype(dt).copy()
        return dict(
            n_states=n, max_dup_len=v.max_dup_len, n_len=v.n_len, local=v.local, n_emit=ne, n_null=nn,
            ein_ptr=arr(v.ein_ptr, n + 1, np.int32), ein_src=arr(v.ein_src, ne, np.int32),
            ein_score=arr(v.ein_score, ne, np.float64), ein_in=arr(v.ein_in, ne, np.uint8),
            ein_base=arr(v.ein_base, ne, np.uint8),
            nin_ptr=arr(v.nin_ptr, n + 1, np.int32), nin_src=arr(v.nin_src, nn, np.int32),
            nin_score=arr(v.nin_score, nn, np.float64), nin_in=arr(v.nin_in, nn, np.uint8),
            eout_ptr=arr(v.eout_ptr, n + 1, np.int32), eout_dst=arr(v.eout_dst, ne, np.int32),
            nout_ptr=arr(v.nout_ptr, n + 1, np.int32), nout_dst=arr(v.nout_dst, nn, np.int32),
            mdl=arr(v.mdl, n, np.uint8), ctx=arr(v.ctx, n * d, np.uint8).reshape(n, d), topo=arr(v.topo, n, np.int32),
            scores=np.array([v.del_open, v.tan_dup, v.no_gap, v.del_extend, v.del_end] + list(v.sub)
                            + [v.len[i] for i in range(v.n_len)]),
            alphabet=v.alphabet.decode(), sym_logp=np.array(list(v.sym_logp)))

    def plan_slots(self):
        """Tier-A placement: (lds_index int32[N] = row*T + lane, lattice_slot int32[N], T, K)."""
        n = self.view.contents.n_states
        lds = np.full(n, -1, dtype=np.int32)
        lat = np.full(n, -1, dtype=np.int32)
        t, k = ctypes.c_int32(), ctypes.c_int32()
        _l.check(_l.lib().dnas_tiera_plan_slots(self.view, lds.ctypes.data, lat.ctypes.data, ctypes.addressof(t), ctypes.addressof(k)))
        return lds, lat, t.value, k.value

    def plan_tables(self):
        """Tier-A tables as the fill kernel receives them: (row_shapes int32[K][2], entries uint32[nEntries][T],
        meta uint32[K][T], n_s_rows)."""
        _, _, t, k = self.plan_slots()
        ne, ns = ctypes.c_int32(), ctypes.c_int32()
        _l.check(_l.lib().dnas_tiera_plan_tables(self.view, None, None, 0, None, ctypes.addressof(ne), ctypes.addressof(ns)))
        shapes = np.zeros((k, 2), dtype=np.int32)
        ent = np.zeros((ne.value, t), dtype=np.uint32)
        meta = np.zeros((k, t), dtype=np.uint32)
        _l.check(_l.lib().dnas_tiera_plan_tables(self.view, shapes.ctypes.data, ent.ctypes.data, ent.size, meta.ctypes.data,
                                                 ctypes.addressof(ne), ctypes.addressof(ns)))
        return shapes, ent, meta, ns.value

    def cluster_plan(self, members=0):
        """Tier-C tables (members = 1: the tier-A plan): dict(G, K, T, n_entries, n_s_rows, n_inbox_rows, shapes int32[K][6],
        entries uint32[G][n_entries][T], meta uint32[G][K][T], member_of, lds_index, lattice_slot, fold uint32[G][n_inbox_rows][T],
        proxy_member, proxy_lds_index: the places of the plan's proxies)."""
        n = self.view.contents.n_states
        info = np.zeros(8, dtype=np.int32)
        _l.check(_l.lib().dnas_tierc_plan(self.view, int(members), info.ctypes.data, None, None, 0, None, None, None, None, None))
        G, K, T, ne = (int(v) for v in info[:4])
        shapes = np.zeros((K, 6), dtype=np.int32)
        fold = np.zeros((G, max(int(info[5]), 1), T), dtype=np.uint32)
        ent = np.zeros((G, ne, T), dtype=np.uint32)
        meta = np.zeros((G, K, T), dtype=np.uint32)
        member_of = np.full(n, -1, dtype=np.int32)
        lds = np.full(n, -1, dtype=np.int32)
        lat = np.full(n, -1, dtype=np.int32)
        _l.check(_l.lib().dnas_tierc_plan(self.view, G, info.ctypes.data, shapes.ctypes.data, ent.ctypes.data, ent.size, meta.ctypes.data,
                                          member_of.ctypes.data, lds.ctypes.data, lat.ctypes.data, fold.ctypes.data))
        n_prox = int(info[6])
        pm = np.zeros(max(n_prox, 1), dtype=np.int32)
        pl = np.zeros(max(n_prox, 1), dtype=np.int32)
        if n_prox:
            _l.check(_l.lib().dnas_tierc_plan_proxies(self.view, G, pm.ctypes.data, pl.ctypes.data, n_prox))
        out = dict(G=G, K=K, T=T, n_entries=ne, n_s_rows=int(info[4]), n_inbox_rows=int(info[5]), shapes=shapes, entries=ent, meta=meta,
                   member_of=member_of, lds_index=lds, lattice_slot=lat, fold=fold[:, :int(info[5])],
                   proxy_member=pm[:n_prox], proxy_lds_index=pl[:n_prox])
        return out

    def tune_record_name(self, members=1, threads=0):
        """File name of this machine's row-program tuning record (kernel cache / dnastore_amd/tune/): members = 1 as tier A,
        0 / >= 2 as tier C with the smallest / that cluster."""
        buf = ctypes.create_string_buffer(64)
        _l.check(_l.lib().dnas_tune_record_name(self.view, int(members), int(threads), buf, 64))
        return buf.value.decode()

    @staticmethod
    def kernel_source_hash():
        """Hash of the fill kernel's source inside the library (what tuning records name as kernel=)."""
        buf = ctypes.create_string_buffer(32)
        _l.check(_l.lib().dnas_kernel_source_hash(buf, 32))
        return buf.value.decode()

    def precompile_cluster(self, members=0):
        """JIT-specialise the cluster (tier C) fill kernel for this machine into the kernel cache (no GPU needed)."""
        buf = ctypes.create_string_buffer(4096)
        _l.check(_l.lib().dnas_tierc_precompile(self.view, int(members), buf, 4096))
        return buf.value.decode()

    def precompile(self):
        """JIT-specialise the tier-A fill kernel for this machine into dnastore_amd/kcache (no GPU needed)."""
        buf = ctypes.create_string_buffer(1024)
        _l.check(_l.lib().dnas_tiera_precompile(self.view, buf, 1024))
        return buf.value.decode()

    def __del__(self):
        try:      # (at interpreter shutdown the module globals may be gone already)
            if getattr(self, "_h", None) is not None and self._h.value:
                _l.lib().dnas_flat_free(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


def pack_reads(reads):
    """list of str -> (read_offsets uint64[n+1], bases uint8[total]) as the C ABI wants them."""
    toks = [tokenize(r) for r in reads]
    off = np.zeros(len(toks) + 1, dtype=np.uint64)
    if toks:
        off[1:] = np.cumsum([len(t) for t in toks])
    bases = np.concatenate(toks).astype(np.uint8) if toks and off[-1] else np.zeros(1, np.uint8)
    return off, bases


class ViterbiDecoder:
    """A (machine, params) pair resident on one GPU; decode() is the batched ViterbiMatrix + traceback."""

    def __init__(self, machine, params, device=0, arena_bytes=0, options=None):
        """options: dnas_model_create_ex's "key=value,..." string, e.g. "tier=C,cluster=2"."""
        self.flat = FlatModel(machine, params)
        v = self.flat.view.contents
        self.n_states, self.max_dup_len = v.n_states, v.max_dup_len
        self._h = ctypes.c_void_p()
        _l.check(_l.lib().dnas_model_create_ex(self.flat.view, int(device), int(arena_bytes),
                                               options.encode() if options else None, ctypes.byref(self._h)))

    def decode(self, reads, out_cap=None):
        """reads: list of str (ACGT, any case) -> (decoded symbol strings, loglike float64[n], status uint8[n])."""
        n = len(reads)
        off, bases = pack_reads(reads)
        lens = np.diff(off).astype(np.int64)
        caps = (4 * lens + 64) if out_cap is None else np.full(n, int(out_cap), dtype=np.int64)
        ooff = np.zeros(n + 1, dtype=np.uint64)
        if n:
            ooff[1:] = np.cumsum(caps)
        sym = np.zeros(max(int(ooff[-1]), 1), dtype=np.uint8)
        olen = np.zeros(max(n, 1), dtype=np.uint32)
        ll = np.zeros(max(n, 1), dtype=np.float64)
        st = np.zeros(max(n, 1), dtype=np.uint8)
        _l.check(_l.lib().dnas_viterbi_batch(self._h, n, off.ctypes.data, bases.ctypes.data, sym.ctypes.data,
                                             ooff.ctypes.data, olen.ctypes.data, ll.ctypes.data, st.ctypes.data))
        out = [sym[int(ooff[i]):int(ooff[i]) + int(olen[i])].tobytes().decode() for i in range(n)]
        return out, ll[:n], st[:n]

    def decode_packed(self, read_offsets, bases, out_cap=None):
        """dnas_viterbi_batch on packed HOST arrays (pack_reads' layout), results as arrays: (sym uint8[...], out_offsets uint64[n+1],
        out_len uint32[n], loglike float64[n], status uint8[n]) -- the call a C caller makes: bases in over PCIe, strings out."""
        n = len(read_offsets) - 1
        lens = np.diff(read_offsets).astype(np.int64)
        caps = (4 * lens + 64) if out_cap is None else np.full(n, int(out_cap), dtype=np.int64)
        ooff = np.zeros(n + 1, dtype=np.uint64)
        if n:
            ooff[1:] = np.cumsum(caps)
        sym = np.empty(max(int(ooff[-1]), 1), dtype=np.uint8)
        olen = np.zeros(max(n, 1), dtype=np.uint32)
        ll = np.zeros(max(n, 1), dtype=np.float64)
        st = np.zeros(max(n, 1), dtype=np.uint8)
        _l.check(_l.lib().dnas_viterbi_batch(self._h, n, read_offsets.ctypes.data, bases.ctypes.data, sym.ctypes.data,
                                             ooff.ctypes.data, olen.ctypes.data, ll.ctypes.data, st.ctypes.data))
        return sym, ooff, olen[:n], ll[:n], st[:n]

    def decode_device(self, read_offsets, d_bases_ptr, d_sym_ptr, out_offsets, d_len_ptr, d_ll_ptr, d_status_ptr):
        """dnas_viterbi_batch_device: raw device pointers (ints), host offset arrays; asynchronous."""
        n = len(read_offsets) - 1
        _l.check(_l.lib().dnas_viterbi_batch_device(self._h, n, read_offsets.ctypes.data, d_bases_ptr, d_sym_ptr,
                                                    out_offsets.ctypes.data, d_len_ptr, d_ll_ptr, d_status_ptr))

    def sync(self):
        _l.check(_l.lib().dnas_model_sync(self._h))

    @property
    def tier(self):
        """'tier A: <shape>' or 'tier B: <reason>' -- which fill kernel serves this machine."""
        return _l.lib().dnas_model_tier(self._h).decode()

    def cluster_census(self):
        """Tier C, last call: (clusters that ran, clusters whose members sat on more than one XCD)."""
        c, s = ctypes.c_int32(), ctypes.c_int32()
        _l.check(_l.lib().dnas_model_cluster_census(self._h, ctypes.addressof(c), ctypes.addressof(s)))
        return c.value, s.value

    def stats(self):
        s = _l.BatchStatsC()
        _l.check(_l.lib().dnas_model_last_stats(self._h, ctypes.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def lattice(self, read_index, length):
        """Lattice of one read of the last decode() call: float64 [L+1][D+2][N] (lanes S, D, T1..TD)."""
        out = np.empty((length + 1, self.max_dup_len + 2, self.n_states), dtype=np.float64)
        _l.check(_l.lib().dnas_model_read_lattice(self._h, int(read_index), int(length), out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _l.lib().dnas_model_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_fastseqs(path):
    """readFastSeqs (fastseq.cpp:123-148) -> [(name, seq)]."""
    h = ctypes.c_void_p()
    _l.check(_l.lib().dnas_fastseqs_read(str(path).encode(), ctypes.byref(h)))
    L = _l.lib()
    out = [(L.dnas_fastseqs_name(h, i).decode(), L.dnas_fastseqs_seq(h, i).decode())
           for i in range(L.dnas_fastseqs_count(h))]
    L.dnas_fastseqs_free(h)
    return out


def format_event(ev):
    """One traceback event as the reference's level-3 log line (viterbi.cpp:266-293)."""
    kind, pos, pay = ev >> 62, (ev >> 32) & 0x3fffffff, ev & 0xffffffff
    if kind == 1:
        return "Substitution at %d: %s -> %s" % (pos, "ACGT"[(pay >> 2) & 3], "ACGT"[pay & 3])
    if kind == 2:
        return "Deletion between %d and %d: %s" % (pos - 1, pos, "ACGT"[pay & 3])
    n = pay >> 26
    return "Duplication at %d: %s" % (pos, "".join("ACGT"[(pay >> (2 * (n - 1 - i))) & 3] for i in range(n)))


def decode_fastseqs(filename, machine, params, device=0, events=False, info=None):
    """decodeFastSeqs(filename, machine, params) (viterbi.cpp:306-320) -> [(name, decoded symbols, loglike)]
    (with events=True: [(name, symbols, loglike, [event lines])]).  device=-1: every GPU of the node.
    info: an optional dict that receives the fill tier and the number of devices used."""
    h = ctypes.c_void_p()
    _l.check(_l.lib().dnas_decode_fastseqs_ex(str(filename).encode(), machine._h, ctypes.byref(params.c), int(device),
                                              int(bool(events)), ctypes.byref(h)))
    L = _l.lib()
    out = []
    for i in range(L.dnas_decoded_count(h)):
        rec = (L.dnas_decoded_name(h, i).decode(), L.dnas_decoded_seq(h, i).decode(), L.dnas_decoded_loglike(h, i))
        if events:
            p = ctypes.c_void_p()
            n = L.dnas_decoded_events(h, i, ctypes.byref(p))
            evs = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint64)), shape=(n,)).copy() if n else []
            rec = rec + ([format_event(int(e)) for e in evs],)
        out.append(rec)
    if info is not None:
        info["tier"] = L.dnas_decoded_tier(h).decode()
        info["devices"] = L.dnas_decoded_devices(h)
    L.dnas_decoded_free(h)
    return out


class StockholmDB:
    """readStockholmDatabase (stockholm.cpp:154-167) of two-row alignments, flattened for the E-step."""

    def __init__(self, path):
        self._h = ctypes.c_void_p()
        _l.check(_l.lib().dnas_stockholm_read(str(path).encode(), ctypes.byref(self._h)))
        self.view = _l.lib().dnas_pairs_get(self._h).contents
        self.n = self.view.n_pairs

    def arrays(self):
        """numpy copies: ins, in_off, outs, out_off, cm_in, cm_in_off, cm_out, cm_out_off."""
        v, n = self.view, self.n

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(count,)).copy()
        in_off = arr(v.in_off, n + 1, np.int64)
        out_off = arr(v.out_off, n + 1, np.int64)
        ci_off = arr(v.cm_in_off, n + 1, np.int64)
        co_off = arr(v.cm_out_off, n + 1, np.int64)
        return dict(ins=arr(v.in_seqs, int(in_off[-1]), np.int8), in_off=in_off, outs=arr(v.out_seqs, int(out_off[-1]), np.int8),
                    out_off=out_off, cm_in=arr(v.cm_in, int(ci_off[-1]), np.int32), cm_in_off=ci_off,
                    cm_out=arr(v.cm_out, int(co_off[-1]), np.int32), cm_out_off=co_off, n=n)

    def __del__(self):
        try:      # (at interpreter shutdown the module globals may be gone already)
            if getattr(self, "_h", None) is not None and self._h.value:
                _l.lib().dnas_pairs_free(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass


def _pair_ptrs(pk):
    keys = ("ins", "in_off", "outs", "out_off", "cm_in", "cm_in_off", "cm_out", "cm_out_off")
    dts = (np.int8, np.int64, np.int8, np.int64, np.int32, np.int64, np.int32, np.int64)
    keep = [np.ascontiguousarray(pk[k], dtype=d) if len(pk[k]) else np.zeros(1, d) for k, d in zip(keys, dts)]
    return keep, [a.ctypes.data for a in keep]


class ForwardBackward:
    """A database of alignment pairs resident on one GPU (dnas_fb): load once, run the E-step many times."""

    def __init__(self, pairs, device=0):
        """pairs: a StockholmDB, the packed dict of its arrays(), or None (an empty handle: load() later)."""
        self._h = ctypes.c_void_p()
        self.n = 0
        _l.check(_l.lib().dnas_fb_create(int(device), ctypes.byref(self._h)))
        if pairs is not None:
            self.load(pairs)

    def load(self, pairs):
        """dnas_fb_load_pairs: the database goes to the GPU (replacing the one that was there)."""
        pk = pairs.arrays() if isinstance(pairs, StockholmDB) else pairs
        keep, ptrs = _pair_ptrs(pk)          # (keep: the arrays must outlive the call)
        self.n = int(pk["n"])
        _l.check(_l.lib().dnas_fb_load_pairs(self._h, self.n, *ptrs))

    def expectedCounts(self, params, strict=False, want_pair_ll=True):
        """-> (counts float64[21+P], ll, per-pair ll float64[n] or None)."""
        counts = np.zeros(21 + params.c.n_len)
        ll = ctypes.c_double()
        per = np.zeros(max(self.n, 1)) if want_pair_ll else None
        _l.check(_l.lib().dnas_fb_estep(self._h, ctypes.byref(params.c), int(bool(strict)), counts.ctypes.data, ctypes.addressof(ll),
                                        per.ctypes.data if want_pair_ll else None))
        return counts, ll.value, (per[:self.n] if want_pair_ll else None)

    def stats(self):
        s = _l.FbStatsC()
        _l.check(_l.lib().dnas_fb_last_stats(self._h, ctypes.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _l.lib().dnas_fb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def expectedCounts(params, pairs, strict=False, device=0):
    """expectedCounts(params, db, ll, strict) (fwdback.cpp:190-209) on the GPU.
    pairs: StockholmDB or a dict of packed arrays (ins, in_off, outs, out_off, cm_in, cm_in_off, cm_out, cm_out_off, n).
    -> (counts float64[21+P], ll, per-pair ll float64[n])."""
    pk = pairs.arrays() if isinstance(pairs, StockholmDB) else pairs
    keep, ptrs = _pair_ptrs(pk)
    n = int(pk["n"])
    counts = np.zeros(21 + params.c.n_len)
    ll = ctypes.c_double()
    per = np.zeros(max(n, 1))
    _l.check(_l.lib().dnas_fwdback_estep(ctypes.byref(params.c), int(bool(strict)), n, *ptrs, int(device), counts.ctypes.data,
                                         ctypes.addressof(ll), per.ctypes.data))
    return counts, ll.value, per[:n]


def baumWelchParams(init, pairs, strict=False, device=0):
    """baumWelchParams(init, Laplace prior, db, strict) (fwdback.cpp:211-230) -> (fitted MutatorParams, iterations)."""
    pk = pairs.arrays() if isinstance(pairs, StockholmDB) else pairs
    keep, ptrs = _pair_ptrs(pk)
    out = _l.MutatorParamsC()
    it = ctypes.c_int32()
    _l.check(_l.lib().dnas_baum_welch(ctypes.byref(init.c), int(bool(strict)), int(pk["n"]), *ptrs, int(device), ctypes.byref(out),
                                      ctypes.addressof(it)))
    return MutatorParams(out), it.value


def paramsJSON(params):
    buf = ctypes.create_string_buffer(4096)
    _l.check(_l.lib().dnas_mutator_params_json(ctypes.byref(params.c), buf, 4096))
    return buf.value.decode()


def countsJSON(counts, n_len):
    c = np.ascontiguousarray(counts, dtype=np.float64)
    buf = ctypes.create_string_buffer(8192)
    _l.check(_l.lib().dnas_mutator_counts_json(c.ctypes.data, int(n_len), buf, 8192))
    return buf.value.decode()


def symbolsToBytes(symbols):
    """BinaryWriter (decoder.h:193-240): '0'/'1' symbols -> bytes, LSB first; other symbols ignored."""
    b = symbols.encode() if isinstance(symbols, str) else symbols
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    _l.check(_l.lib().dnas_symbols_to_bytes(b, len(b), ctypes.byref(p), ctypes.byref(n)))
    out = ctypes.string_at(p, n.value)
    _l.lib().dnas_free(p)
    return out
