"""dnastore_amd: MI355X-native Viterbi error decoder behind ihh/dnastore's interface.

The product is the C-ABI shared library built from dnastore_amd/csrc (host C++ + hand
written gfx950 HIP kernels); this package is the thin Python mirror of the reference's
operator surface used by tests and bench.py: Machine (src/trans.h), MutatorParams
(src/mutator.h) and decodeFastSeqs / ViterbiMatrix (src/viterbi.h:94-108).
"""
from .api import (FlatModel, ForwardBackward, Machine, MutatorParams, StockholmDB, ViterbiDecoder, baumWelchParams, countsJSON,  # noqa: F401
                  decode_fastseqs, expectedCounts, paramsJSON, symbolsToBytes,
                  pack_reads, read_fastseqs, tokenize)
from . import lib  # noqa: F401
from .lib import DnasError, LIB_PATH  # noqa: F401
