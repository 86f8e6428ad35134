// Host-only half of the C ABI (include/dnastore_amd.h): file formats, flattening and the
// decodeFastSeqs convenience call.  The device half lives in runtime.hip.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dnastore_amd.h"
#include "errors.hpp"
#include "host/decoder.hpp"
#include "host/encoder.hpp"
#include "host/fastseq.hpp"
#include "host/machine.hpp"
#include "host/model.hpp"
#include "host/stockholm.hpp"

struct dnas_machine { dnas::Machine machine; };
struct dnas_flat { dnas::FlatModel flat; };
struct dnas_fastseqs { std::vector<dnas::FastSeq> seqs; };
struct dnas_pairs { dnas::AlignmentPairs db; dnas_pairs_view view; };
struct dnas_decoded {
  std::vector<dnas::FastSeq> seqs;
  std::vector<double> loglike;
  std::vector<std::vector<uint64_t>> events;   // per read, when asked for
  std::string tier;                            // which fill kernel served the machine
  int devices = 1;
};

namespace dnas {
std::string& lastErrorSlot() {
  thread_local std::string slot;
  return slot;
}
}  // namespace dnas

namespace {

// Map the host layer's exceptions to ABI status codes (reference behaviour in comments).
template <class F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const std::domain_error& e) {  // cyclic null graph, trans.cpp:631-632
    return dnas::fail(DNAS_E_CYCLIC, e.what());
  } catch (const std::bad_alloc&) {
    return dnas::fail(DNAS_E_NOMEM, "out of memory");
  } catch (const std::exception& e) {
    const std::string w = e.what();
    int code = DNAS_E_PARSE;
    if (w.rfind("File not found", 0) == 0 || w.rfind("Couldn't open", 0) == 0) code = DNAS_E_IO;  // Fail -> exit(1)
    else if (w.rfind("Not a DNA-outputting machine", 0) == 0) code = DNAS_E_NOT_DNA;
    else if (w.rfind("Unknown symbol", 0) == 0) code = DNAS_E_BAD_BASE;
    return dnas::fail(code, w);
  }
}

}  // namespace

extern "C" {

const char* dnas_last_error(void) { return dnas::lastErrorSlot().c_str(); }
void dnas_free(void* p) { free(p); }

int dnas_machine_load_json(const char* path, dnas_machine** out) {
  if (!path || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = nullptr;
  return guarded([&] {
    *out = new dnas_machine{dnas::Machine::fromFile(path)};
    return DNAS_OK;
  });
}

int dnas_machine_parse_json(const char* text, size_t len, dnas_machine** out) {
  if (!text || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = nullptr;
  return guarded([&] {
    *out = new dnas_machine{dnas::Machine::fromJSON(std::string(text, len))};
    return DNAS_OK;
  });
}

void dnas_machine_free(dnas_machine* m) { delete m; }
int32_t dnas_machine_n_states(const dnas_machine* m) { return m ? (int32_t)m->machine.nStates() : 0; }

int dnas_machine_write_json(const dnas_machine* m, char** out_text, size_t* out_len) {
  if (!m || !out_text || !out_len) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    std::ostringstream ss;
    m->machine.writeJSON(ss);
    const std::string s = ss.str();
    char* buf = (char*)malloc(s.size() + 1);
    if (!buf) throw std::bad_alloc();
    memcpy(buf, s.c_str(), s.size() + 1);
    *out_text = buf;
    *out_len = s.size();
    return DNAS_OK;
  });
}

static int encoded_to_c(dnas::Encoder& enc, char** out_dna, size_t* out_len) {
  enc.close();
  const std::string& s = enc.output();
  char* buf = (char*)malloc(s.size() + 1);
  if (!buf) throw std::bad_alloc();
  memcpy(buf, s.c_str(), s.size() + 1);
  *out_dna = buf;
  *out_len = s.size();
  return DNAS_OK;
}

int dnas_encode_symbols(const dnas_machine* m, const char* symbols, size_t n, char** out_dna, size_t* out_len) {
  if (!m || (!symbols && n) || !out_dna || !out_len) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    dnas::Encoder enc(m->machine);
    enc.encodeSymbolString(std::string(symbols, n));
    return encoded_to_c(enc, out_dna, out_len);
  });
}

int dnas_encode_bytes(const dnas_machine* m, const uint8_t* bytes, size_t n, char** out_dna, size_t* out_len) {
  if (!m || (!bytes && n) || !out_dna || !out_len) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    dnas::Encoder enc(m->machine);
    enc.encodeBytes(std::string((const char*)bytes, n));
    return encoded_to_c(enc, out_dna, out_len);
  });
}

int dnas_machine_compose(const dnas_machine* first, const dnas_machine* second, dnas_machine** out) {
  if (!first || !second || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = nullptr;
  return guarded([&] {
    *out = new dnas_machine{dnas::Machine::compose(first->machine, second->machine)};
    return DNAS_OK;
  });
}

int dnas_decode_exact(const dnas_machine* m, const char* dna, size_t n, char** out_symbols, size_t* out_len) {
  if (!m || (!dna && n) || !out_symbols || !out_len) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    dnas::Decoder dec(m->machine);
    dec.decodeString(std::string(dna, n));
    dec.close();
    const std::string& s = dec.symbols();
    char* buf = (char*)malloc(s.size() + 1);
    if (!buf) throw std::bad_alloc();
    memcpy(buf, s.c_str(), s.size() + 1);
    *out_symbols = buf;
    *out_len = s.size();
    return DNAS_OK;
  });
}

int dnas_symbols_to_bytes(const char* symbols, size_t n, uint8_t** out_bytes, size_t* out_len) {
  if ((!symbols && n) || !out_bytes || !out_len) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    const std::string b = dnas::symbolsToBytes(std::string(symbols, n));
    uint8_t* buf = (uint8_t*)malloc(b.size() + 1);
    if (!buf) throw std::bad_alloc();
    memcpy(buf, b.data(), b.size());
    *out_bytes = buf;
    *out_len = b.size();
    return DNAS_OK;
  });
}

int dnas_mutator_params_from_flags(double sub_prob, double iv_ratio, double dup_prob, double del_open, double del_ext,
                                   int global, int length, dnas_mutator_params* out) {
  if (!out) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    dnas::MutatorParams::fromFlags(sub_prob, iv_ratio, dup_prob, del_open, del_ext, global != 0, length).toC(out);
    return DNAS_OK;
  });
}

int dnas_mutator_params_load_json(const char* path, dnas_mutator_params* out) {
  if (!path || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    dnas::MutatorParams::fromFile(path).toC(out);
    return DNAS_OK;
  });
}

int dnas_flatten(const dnas_machine* m, const dnas_mutator_params* p, dnas_flat** out) {
  if (!m || !p || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = nullptr;
  return guarded([&] {
    dnas_flat* f = new dnas_flat{dnas::FlatModel::build(m->machine, dnas::MutatorParams::fromC(*p))};
    f->flat.bind();
    *out = f;
    return DNAS_OK;
  });
}

const dnas_flat_model* dnas_flat_view(const dnas_flat* f) { return f ? &f->flat.view : nullptr; }
void dnas_flat_free(dnas_flat* f) { delete f; }

int dnas_fastseqs_read(const char* path, dnas_fastseqs** out) {
  if (!path || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = nullptr;
  return guarded([&] {
    *out = new dnas_fastseqs{dnas::readFastSeqs(path)};
    return DNAS_OK;
  });
}
int64_t dnas_fastseqs_count(const dnas_fastseqs* f) { return f ? (int64_t)f->seqs.size() : 0; }
const char* dnas_fastseqs_name(const dnas_fastseqs* f, int64_t i) { return f->seqs[(size_t)i].name.c_str(); }
const char* dnas_fastseqs_seq(const dnas_fastseqs* f, int64_t i) { return f->seqs[(size_t)i].seq.c_str(); }
void dnas_fastseqs_free(dnas_fastseqs* f) { delete f; }

int dnas_stockholm_read(const char* path, dnas_pairs** out) {
  if (!path || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = nullptr;
  return guarded([&] {
    dnas_pairs* p = new dnas_pairs{dnas::readStockholmPairs(path), {}};
    const dnas::AlignmentPairs& d = p->db;
    p->view = dnas_pairs_view{d.n, d.inSeqs.data(), d.inOff.data(), d.outSeqs.data(), d.outOff.data(),
                              d.cmIn.data(), d.cmInOff.data(), d.cmOut.data(), d.cmOutOff.data()};
    *out = p;
    return DNAS_OK;
  });
}
const dnas_pairs_view* dnas_pairs_get(const dnas_pairs* p) { return p ? &p->view : nullptr; }
void dnas_pairs_free(dnas_pairs* p) { delete p; }

// MutatorParams::writeJSON / MutatorCounts::writeJSON (mutator.cpp:6-16,108-124): the text the
// reference prints for --fit-error / --error-counts, default 6-digit ostream formatting.
int dnas_mutator_params_json(const dnas_mutator_params* p, char* buf, size_t cap) {
  if (!p || !buf || !cap) return dnas::fail(DNAS_E_INVALID, "null argument");
  return guarded([&] {
    const std::string s = dnas::MutatorParams::fromC(*p).toJSON();
    if (s.size() + 1 > cap) return dnas::fail(DNAS_E_INVALID, "buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return DNAS_OK;
  });
}
int dnas_mutator_counts_json(const double* counts, int32_t n_len, char* buf, size_t cap) {
  if (!counts || !buf || !cap || n_len < 0) return dnas::fail(DNAS_E_INVALID, "bad argument");
  return guarded([&] {
    auto trans = [](int i, int j) { return i != j && (i & 1) == (j & 1); };
    double nm = 0, ni = 0, nv = 0;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        const double c = counts[5 + i * 4 + j];
        if (i == j) nm += c; else if (trans(i, j)) ni += c; else nv += c;
      }
    // nMatch / nTransition / nTransversion are summed in the reference's loop order (mutator.cpp:180-196)
    nm = 0; for (int i = 0; i < 4; ++i) nm += counts[5 + i * 5];
    ni = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (trans(i, j)) ni += counts[5 + i * 4 + j];
    nv = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (i != j && !trans(i, j)) nv += counts[5 + i * 4 + j];
    std::ostringstream o;
    o << "{\n";
    o << " \"nDelOpen\": " << counts[0] << ",\n";
    o << " \"nTanDup\": " << counts[1] << ",\n";
    o << " \"nNoGap\": " << counts[2] << ",\n";
    o << " \"nDelExtend\": " << counts[3] << ",\n";
    o << " \"nDelEnd\": " << counts[4] << ",\n";
    o << " \"nLen\": [ ";
    for (int k = 0; k < n_len; ++k) o << (k ? ", " : "") << counts[21 + k];
    o << " ],\n \"nSub\": [ ";
    for (int i = 0; i < 4; ++i) {
      o << (i ? ", " : "") << "[";
      for (int j = 0; j < 4; ++j) o << (j ? "," : "") << counts[5 + i * 4 + j];
      o << "]";
    }
    o << " ],\n";
    o << " \"nMatch\": " << nm << ",\n";
    o << " \"nTransition\": " << ni << ",\n";
    o << " \"nTransversion\": " << nv << "\n";
    o << "}\n";
    const std::string s = o.str();
    if (s.size() + 1 > cap) return dnas::fail(DNAS_E_INVALID, "buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return DNAS_OK;
  });
}

// decodeFastSeqs (viterbi.cpp:306-320): read the FASTA, build the input model once, decode every read on the GPU,
// keep names, drop comments.  device_id >= 0: that GPU.  device_id = -1: every GPU of the node -- the reads are dealt
// over the devices by length in snake order (longest first: 0..n-1, n-1..0, ...; what shard.partition does for the
// one-process-per-GPU bench), one host thread and one model per device, results back in file order.  The loop the
// reference runs serially (viterbi.cpp:312-318) has no dependence between reads, so nothing is exchanged.
static int decode_shard(const dnas_flat_model* fm, int device, const std::vector<dnas::FastSeq>& reads, const std::vector<int64_t>& mine,
                        bool events, std::vector<std::string>* seqs, std::vector<double>* lls, std::vector<std::vector<uint64_t>>* evs,
                        std::string* tier, std::string* err) {
  dnas_model* model = nullptr;
  auto fail = [&](int rc) {
    *err = dnas_last_error();
    if (model) dnas_model_destroy(model);
    return rc;
  };
  try {
    std::vector<uint64_t> off{0}, outOff{0};
    std::vector<uint8_t> bases;
    for (int64_t i : mine) {
      const std::vector<uint8_t> tok = dnas::tokenizeDNA(reads[(size_t)i].seq, reads[(size_t)i].name);
      bases.insert(bases.end(), tok.begin(), tok.end());
      off.push_back(bases.size());
      outOff.push_back(outOff.back() + 4 * tok.size() + 64);
    }
    const int64_t n = (int64_t)mine.size();
    if (n == 0) return DNAS_OK;
    int rc = dnas_model_create(fm, device, 0, &model);
    if (rc != DNAS_OK) return fail(rc);
    *tier = dnas_model_tier(model);
    if (events && (rc = dnas_model_set_event_log(model, 1)) != DNAS_OK) return fail(rc);
    std::vector<char> sym(outOff.back());
    std::vector<uint32_t> len((size_t)n);
    std::vector<double> ll((size_t)n);
    std::vector<uint8_t> st((size_t)n);
    if (bases.empty()) bases.push_back(0);
    rc = dnas_viterbi_batch(model, n, off.data(), bases.data(), sym.data(), outOff.data(), len.data(), ll.data(), st.data());
    if (rc != DNAS_OK) return fail(rc);
    for (int64_t k = 0; k < n; ++k) {
      if (st[(size_t)k] == DNAS_READ_OUT_OVERFLOW || st[(size_t)k] == DNAS_READ_TRACEBACK_FAIL) {
        dnas::lastErrorSlot() = st[(size_t)k] == DNAS_READ_OUT_OVERFLOW ? "decoded string overflowed its slot" : "Traceback failure";
        return fail(DNAS_E_DEVICE);
      }
      (*seqs)[(size_t)mine[(size_t)k]].assign(sym.data() + outOff[(size_t)k], len[(size_t)k]);
      (*lls)[(size_t)mine[(size_t)k]] = ll[(size_t)k];
      if (events) {
        int64_t ne = 0;
        rc = dnas_model_read_events(model, k, nullptr, 0, &ne);
        if (rc != DNAS_OK) return fail(rc);
        std::vector<uint64_t>& e = (*evs)[(size_t)mine[(size_t)k]];
        e.resize((size_t)ne);
        if (ne && (rc = dnas_model_read_events(model, k, e.data(), ne, &ne)) != DNAS_OK) return fail(rc);
      }
    }
    dnas_model_destroy(model);
    return DNAS_OK;
  } catch (const std::exception& e) {
    *err = e.what();
    if (model) dnas_model_destroy(model);
    return DNAS_E_DEVICE;
  }
}

int dnas_decode_fastseqs(const char* fasta_path, const dnas_machine* m, const dnas_mutator_params* p, int device_id,
                         dnas_decoded** out) {
  return dnas_decode_fastseqs_ex(fasta_path, m, p, device_id, 0, out);
}

int dnas_decode_fastseqs_ex(const char* fasta_path, const dnas_machine* m, const dnas_mutator_params* p, int device_id,
                            int want_events, dnas_decoded** out) {
  if (!fasta_path || !m || !p || !out) return dnas::fail(DNAS_E_INVALID, "null argument");
  *out = nullptr;
  dnas_flat* flat = nullptr;
  int rc = guarded([&] {
    const std::vector<dnas::FastSeq> reads = dnas::readFastSeqs(fasta_path);
    int r = dnas_flatten(m, p, &flat);
    if (r != DNAS_OK) return r;
    const int64_t n = (int64_t)reads.size();
    // which devices
    std::vector<int> devices;
    if (device_id >= 0) {
      devices.push_back(device_id);
    } else {
      int have = dnas_device_count();
      if (have <= 0) return dnas::fail(DNAS_E_DEVICE, "no HIP device available");
      int use = have;
      if (const char* s = getenv("DNAS_FAKE_DEVICES")) use = std::max(1, atoi(s));   // tests: several host threads share the GPUs there are
      for (int d = 0; d < use && d < std::max<int64_t>(n, 1); ++d) devices.push_back(d % have);
    }
    // deal the reads: by length, longest first, in snake order
    std::vector<int64_t> order((size_t)n);
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return reads[(size_t)a].seq.size() > reads[(size_t)b].seq.size(); });
    const size_t W = devices.size();
    std::vector<std::vector<int64_t>> shard(W);
    for (size_t pos = 0; pos < order.size(); ++pos) {
      const size_t round = pos / W, k = pos % W;
      shard[round % 2 == 0 ? k : W - 1 - k].push_back(order[pos]);
    }
    for (auto& sh : shard) std::sort(sh.begin(), sh.end());
    std::unique_ptr<dnas_decoded> d(new dnas_decoded());
    std::vector<std::string> seqs((size_t)n);
    std::vector<double> lls((size_t)n, 0.);
    d->events.resize((size_t)n);
    std::vector<int> rcs(W, DNAS_OK);
    std::vector<std::string> errs(W), tiers(W);
    if (W == 1) {
      rcs[0] = decode_shard(dnas_flat_view(flat), devices[0], reads, shard[0], want_events != 0, &seqs, &lls, &d->events, &tiers[0], &errs[0]);
    } else {
      std::vector<std::thread> workers;
      for (size_t w = 0; w < W; ++w)
        workers.emplace_back([&, w] {
          rcs[w] = decode_shard(dnas_flat_view(flat), devices[w], reads, shard[w], want_events != 0, &seqs, &lls, &d->events, &tiers[w], &errs[w]);
        });
      for (auto& t : workers) t.join();
    }
    for (size_t w = 0; w < W; ++w)
      if (rcs[w] != DNAS_OK) return dnas::fail(rcs[w], "device " + std::to_string(devices[w]) + ": " + errs[w]);
    for (int64_t i = 0; i < n; ++i) {
      dnas::FastSeq fs;
      fs.name = reads[(size_t)i].name;  // viterbi.cpp:315: name kept, comment dropped
      fs.seq = std::move(seqs[(size_t)i]);
      d->seqs.push_back(std::move(fs));
      d->loglike.push_back(lls[(size_t)i]);
    }
    for (const auto& t : tiers) if (!t.empty()) { d->tier = t; break; }
    d->devices = (int)W;
    *out = d.release();
    return DNAS_OK;
  });
  if (flat) dnas_flat_free(flat);
  return rc;
}

int64_t dnas_decoded_count(const dnas_decoded* d) { return d ? (int64_t)d->seqs.size() : 0; }
const char* dnas_decoded_name(const dnas_decoded* d, int64_t i) { return d->seqs[(size_t)i].name.c_str(); }
const char* dnas_decoded_seq(const dnas_decoded* d, int64_t i) { return d->seqs[(size_t)i].seq.c_str(); }
double dnas_decoded_loglike(const dnas_decoded* d, int64_t i) { return d->loglike[(size_t)i]; }
const char* dnas_decoded_tier(const dnas_decoded* d) { return d ? d->tier.c_str() : ""; }
int dnas_decoded_devices(const dnas_decoded* d) { return d ? d->devices : 0; }
int64_t dnas_decoded_events(const dnas_decoded* d, int64_t i, const uint64_t** events) {
  if (!d || i < 0 || (size_t)i >= d->events.size()) return 0;
  if (events) *events = d->events[(size_t)i].data();
  return (int64_t)d->events[(size_t)i].size();
}
void dnas_decoded_free(dnas_decoded* d) { delete d; }

}  // extern "C"
