"""ctypes binding of libdnastore_amd.so (the C ABI of include/dnastore_amd.h).

There is no CPU fallback: if the shared library (host C++ + gfx950 HIP kernels) has not
been built, importing it raises, and device entry points fail with DNAS_E_DEVICE when no
GPU is present.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# DNAS_LIBRARY: another build of the same library (tools/run_asan.sh: the sanitizer build under dnastore_amd/asan/)
LIB_PATH = os.environ.get("DNAS_LIBRARY") or os.path.join(_HERE, "libdnastore_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dnastore_amd.h")

DNAS_OK = 0
READ_OK, READ_NO_PATH, READ_OUT_OVERFLOW, READ_TRACEBACK_FAIL = 0, 1, 2, 3
ERROR_NAMES = {-1: "DNAS_E_INVALID", -2: "DNAS_E_IO", -3: "DNAS_E_PARSE", -4: "DNAS_E_CYCLIC", -5: "DNAS_E_NOT_DNA",
               -6: "DNAS_E_BAD_BASE", -7: "DNAS_E_DEVICE", -8: "DNAS_E_NOMEM", -9: "DNAS_E_UNSUPPORTED"}


class DnasError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERROR_NAMES.get(code, code), msg))
        self.code = code


class MutatorParamsC(ctypes.Structure):
    _fields_ = [("p_del_open", ctypes.c_double), ("p_del_extend", ctypes.c_double), ("p_tan_dup", ctypes.c_double),
                ("p_transition", ctypes.c_double), ("p_transversion", ctypes.c_double),
                ("n_len", ctypes.c_int32), ("local", ctypes.c_int32), ("p_len", ctypes.c_double * 32)]


_I32P, _F64P, _U8P = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint8)


class FlatModelC(ctypes.Structure):
    _fields_ = [("n_states", ctypes.c_int32), ("max_dup_len", ctypes.c_int32), ("n_len", ctypes.c_int32),
                ("local", ctypes.c_int32), ("n_emit", ctypes.c_int32), ("n_null", ctypes.c_int32),
                ("ein_ptr", _I32P), ("ein_src", _I32P), ("ein_score", _F64P), ("ein_in", _U8P), ("ein_base", _U8P),
                ("nin_ptr", _I32P), ("nin_src", _I32P), ("nin_score", _F64P), ("nin_in", _U8P),
                ("eout_ptr", _I32P), ("eout_dst", _I32P), ("eout_score", _F64P),
                ("nout_ptr", _I32P), ("nout_dst", _I32P), ("nout_score", _F64P),
                ("mdl", _U8P), ("ctx", _U8P), ("topo", _I32P),
                ("no_gap", ctypes.c_double), ("del_open", ctypes.c_double), ("del_extend", ctypes.c_double),
                ("del_end", ctypes.c_double), ("tan_dup", ctypes.c_double), ("sub", ctypes.c_double * 16),
                ("len", _F64P), ("alphabet", ctypes.c_char * 64), ("sym_logp", ctypes.c_double * 128)]


class PairsViewC(ctypes.Structure):
    _fields_ = [("n_pairs", ctypes.c_int64), ("in_seqs", ctypes.c_void_p), ("in_off", ctypes.c_void_p),
                ("out_seqs", ctypes.c_void_p), ("out_off", ctypes.c_void_p), ("cm_in", ctypes.c_void_p),
                ("cm_in_off", ctypes.c_void_p), ("cm_out", ctypes.c_void_p), ("cm_out_off", ctypes.c_void_p)]


class FbStatsC(ctypes.Structure):
    _fields_ = [("kernel_ms", ctypes.c_double), ("pairs_onchip", ctypes.c_int64), ("pairs_streaming", ctypes.c_int64),
                ("lse_ops", ctypes.c_int64), ("out_nt", ctypes.c_int64), ("pairs_narrow", ctypes.c_int64)]


class BatchStatsC(ctypes.Structure):
    _fields_ = [("fill_ms", ctypes.c_double), ("traceback_ms", ctypes.c_double), ("fill_launches", ctypes.c_int64),
                ("columns", ctypes.c_int64), ("lattice_bytes", ctypes.c_int64), ("rounds", ctypes.c_int64),
                ("checkpointed_reads", ctypes.c_int64)]


def declared_symbols():
    """Every function name include/dnastore_amd.h declares."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dnas_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C dnastore_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    vp, cp, i64, sz = ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_size_t
    P = ctypes.POINTER
    sigs = {
        "dnas_machine_load_json": (ctypes.c_int, [cp, P(vp)]),
        "dnas_machine_parse_json": (ctypes.c_int, [cp, sz, P(vp)]),
        "dnas_machine_free": (None, [vp]),
        "dnas_machine_n_states": (ctypes.c_int32, [vp]),
        "dnas_machine_write_json": (ctypes.c_int, [vp, P(vp), P(sz)]),
        "dnas_machine_compose": (ctypes.c_int, [vp, vp, P(vp)]),
        "dnas_decode_exact": (ctypes.c_int, [vp, cp, sz, P(vp), P(sz)]),
        "dnas_symbols_to_bytes": (ctypes.c_int, [cp, sz, P(vp), P(sz)]),
        "dnas_encode_symbols": (ctypes.c_int, [vp, cp, sz, P(vp), P(sz)]),
        "dnas_encode_bytes": (ctypes.c_int, [vp, cp, sz, P(vp), P(sz)]),
        "dnas_mutator_params_from_flags": (ctypes.c_int, [ctypes.c_double] * 5 + [ctypes.c_int, ctypes.c_int, P(MutatorParamsC)]),
        "dnas_mutator_params_load_json": (ctypes.c_int, [cp, P(MutatorParamsC)]),
        "dnas_flatten": (ctypes.c_int, [vp, P(MutatorParamsC), P(vp)]),
        "dnas_flat_view": (P(FlatModelC), [vp]),
        "dnas_flat_free": (None, [vp]),
        "dnas_model_create": (ctypes.c_int, [P(FlatModelC), ctypes.c_int, sz, P(vp)]),
        "dnas_model_create_ex": (ctypes.c_int, [P(FlatModelC), ctypes.c_int, sz, cp, P(vp)]),
        "dnas_model_destroy": (None, [vp]),
        "dnas_viterbi_batch": (ctypes.c_int, [vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        "dnas_viterbi_batch_device": (ctypes.c_int, [vp, i64, vp, vp, vp, vp, vp, vp, vp]),
        "dnas_model_sync": (ctypes.c_int, [vp]),
        "dnas_model_tier": (cp, [vp]),
        "dnas_tiera_plan_slots": (ctypes.c_int, [P(FlatModelC), vp, vp, vp, vp]),
        "dnas_tiera_plan_tables": (ctypes.c_int, [P(FlatModelC), vp, vp, ctypes.c_size_t, vp, vp, vp]),
        "dnas_model_debug_words": (ctypes.c_int, [vp, vp]),
        "dnas_tiera_precompile": (ctypes.c_int, [P(FlatModelC), ctypes.c_char_p, sz]),
        "dnas_tierc_precompile": (ctypes.c_int, [P(FlatModelC), ctypes.c_int32, ctypes.c_char_p, sz]),
        "dnas_tierc_plan": (ctypes.c_int, [P(FlatModelC), ctypes.c_int32, vp, vp, vp, sz, vp, vp, vp, vp, vp]),
        "dnas_tierc_plan_proxies": (ctypes.c_int, [P(FlatModelC), ctypes.c_int32, vp, vp, sz]),
        "dnas_model_cluster_census": (ctypes.c_int, [vp, vp, vp]),
        "dnas_tune_record_name": (ctypes.c_int, [vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p, ctypes.c_size_t]),
        "dnas_kernel_source_hash": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_size_t]),
        "dnas_model_last_stats": (ctypes.c_int, [vp, P(BatchStatsC)]),
        "dnas_model_read_lattice": (ctypes.c_int, [vp, i64, i64, vp]),
        "dnas_fwdback_estep": (ctypes.c_int, [P(MutatorParamsC), ctypes.c_int, i64] + [vp] * 8 + [ctypes.c_int, vp, vp, vp]),
        "dnas_fb_create": (ctypes.c_int, [ctypes.c_int, P(vp)]),
        "dnas_fb_load_pairs": (ctypes.c_int, [vp, i64] + [vp] * 8),
        "dnas_fb_estep": (ctypes.c_int, [vp, P(MutatorParamsC), ctypes.c_int, vp, vp, vp]),
        "dnas_fb_last_stats": (ctypes.c_int, [vp, P(FbStatsC)]),
        "dnas_fb_destroy": (None, [vp]),
        "dnas_baum_welch": (ctypes.c_int, [P(MutatorParamsC), ctypes.c_int, i64] + [vp] * 8 + [ctypes.c_int, P(MutatorParamsC), vp]),
        "dnas_stockholm_read": (ctypes.c_int, [cp, P(vp)]),
        "dnas_pairs_get": (P(PairsViewC), [vp]),
        "dnas_pairs_free": (None, [vp]),
        "dnas_mutator_params_json": (ctypes.c_int, [P(MutatorParamsC), ctypes.c_char_p, sz]),
        "dnas_mutator_counts_json": (ctypes.c_int, [vp, ctypes.c_int32, ctypes.c_char_p, sz]),
        "dnas_decode_fastseqs": (ctypes.c_int, [cp, vp, P(MutatorParamsC), ctypes.c_int, P(vp)]),
        "dnas_decode_fastseqs_ex": (ctypes.c_int, [cp, vp, P(MutatorParamsC), ctypes.c_int, ctypes.c_int, P(vp)]),
        "dnas_decoded_tier": (cp, [vp]),
        "dnas_decoded_devices": (ctypes.c_int, [vp]),
        "dnas_decoded_events": (i64, [vp, i64, P(vp)]),
        "dnas_device_count": (ctypes.c_int, []),
        "dnas_model_set_event_log": (ctypes.c_int, [vp, ctypes.c_int]),
        "dnas_model_read_events": (ctypes.c_int, [vp, i64, vp, i64, vp]),
        "dnas_decoded_count": (i64, [vp]),
        "dnas_decoded_name": (cp, [vp, i64]),
        "dnas_decoded_seq": (cp, [vp, i64]),
        "dnas_decoded_loglike": (ctypes.c_double, [vp, i64]),
        "dnas_decoded_free": (None, [vp]),
        "dnas_fastseqs_read": (ctypes.c_int, [cp, P(vp)]),
        "dnas_fastseqs_count": (i64, [vp]),
        "dnas_fastseqs_name": (cp, [vp, i64]),
        "dnas_fastseqs_seq": (cp, [vp, i64]),
        "dnas_fastseqs_free": (None, [vp]),
        "dnas_last_error": (cp, []),
        "dnas_free": (None, [vp]),
        "dnas_has_device_code": (ctypes.c_int, []),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != DNAS_OK:
        raise DnasError(rc, lib().dnas_last_error().decode(errors="replace"))
