// dnastore -- command-line driver with the reference's flags and output formats
// (reference t/dnastore.cpp:34-251) for everything on and around the error-decoding path:
// --load-machine / --compose-machine / --save-machine, the exact --encode-* / --decode-* arms,
// -V/--decode-viterbi with the --error-* model (GPU), --error-counts and --fit-error (GPU).
// It is a client of the C ABI in include/dnastore_amd.h only.
//
// Not provided: the `-l k` de Bruijn code builder (reference src/builder.cpp; its output is
// platform dependent).  Without --load-machine, `-l 4 -c 4` resolves to the
// canonical machine data/l4c4.json when DNASTORE_L4C4 names it; other lengths are refused.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/dnastore_amd.h"

namespace {

struct Options {
  int length = 12, controls = 4, verbose = 2, device = 0;
  std::string loadMachine, saveMachine, encodeFile, decodeFile, encodeString, decodeString, encodeBits, decodeBits,
      decodeViterbi, errorFile, fitError, errorCounts;
  std::vector<std::string> compose;
  bool raw = false, errorGlobal = false, strictGuides = false, help = false;
  double subProb = .01, ivRatio = 10, dupProb = .001, delOpen = .001, delExt = .01;
};

const char* kHelp =
    "Allowed options:\n"
    "  -h [ --help ]                 display this help message\n"
    "  -l [ --length ] arg (=12)     length of k-mers in de Bruijn graph (sets the error model's pLen to length/2)\n"
    "  -c [ --controls ] arg (=4)    number of control words\n"
    "  -L [ --load-machine ] arg     load machine from JSON file\n"
    "  -S [ --save-machine ] arg     save machine to JSON file\n"
    "  -C [ --compose-machine ] arg  load machine from JSON file and compose in front of primary machine\n"
    "  -e [ --encode-file ] arg      encode binary file to FASTA on stdout\n"
    "  -d [ --decode-file ] arg      decode FASTA file to binary on stdout\n"
    "  -E [ --encode-string ] arg    encode ASCII string to FASTA on stdout\n"
    "  -D [ --decode-string ] arg    decode DNA sequence to binary on stdout\n"
    "  -b [ --encode-bits ] arg      encode string of bits and control symbols to FASTA on stdout\n"
    "  -B [ --decode-bits ] arg      decode DNA sequence to string of bits and control symbols on stdout\n"
    "  -V [ --decode-viterbi ] arg   decode FASTA file using Viterbi algorithm (MI355X)\n"
    "  -r [ --raw ]                  strip headers from FASTA output; just print raw sequence\n"
    "  --error-sub-prob arg (=0.01)  substitution probability for error model\n"
    "  --error-iv-ratio arg (=10)    transition/transversion ratio for error model\n"
    "  --error-dup-prob arg (=0.001) tandem duplication probability for error model\n"
    "  --error-del-open arg (=0.001) deletion opening probability for error model\n"
    "  --error-del-ext arg (=0.01)   deletion extension probability for error model\n"
    "  --error-global                force global alignment in error model (disallow partial reads)\n"
    "  -F [ --error-file ] arg       load error model from file\n"
    "  -f [ --fit-error ] arg        train error model on Stockholm database of pairwise alignments and print to stdout\n"
    "  --error-counts arg            estimate posterior expected counts of various different types of error from Stockholm database\n"
    "  --strict-guides               treat alignments in Stockholm database as strict truth, not just hints\n"
    "  -v [ --verbose ] arg (=2)     verbosity level\n"
    "  --device arg (=0)             GPU to use; -1 = every GPU of the node, reads dealt over them\n";

[[noreturn]] void die(const std::string& msg) {
  std::cerr << msg << std::endl;
  exit(1);
}

void check(int rc) {
  if (rc != DNAS_OK) {
    // the reference prints e.what() and still exits 0 for the exceptions it catches in main
    // (dnastore.cpp:245-250: parse errors, cyclic machines, ...); missing files exit 1 (Fail, util.cpp:47-54).
    // Failures that are not reference exceptions -- no GPU, out of device memory, an unsupported or invalid
    // request -- must not look like success: an empty stdout with exit 0 would feed empty decodes downstream.
    std::cerr << dnas_last_error() << std::endl;
    const bool referenceException = rc == DNAS_E_PARSE || rc == DNAS_E_CYCLIC || rc == DNAS_E_NOT_DNA || rc == DNAS_E_BAD_BASE;
    exit(referenceException ? 0 : (rc == DNAS_E_IO ? 1 : 2));
  }
}

void writeFasta(std::ostream& out, const char* name, const std::string& seq, bool raw) {
  if (raw) { out << seq << "\n"; return; }
  out << ">" << name << "\n";
  for (size_t i = 0; i < seq.size(); i += 50) out << seq.substr(i, 50) << "\n";   // fastseq.h:14
}

Options parse(int argc, char** argv) {
  Options o;
  auto need = [&](int& i, const std::string& flag) -> std::string {
    if (i + 1 >= argc) die("the required argument for option '" + flag + "' is missing");
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], val;
    bool hasVal = false;
    if (a.rfind("--", 0) == 0) {
      const size_t eq = a.find('=');
      if (eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); hasVal = true; }
    } else if (a.size() > 2 && a[0] == '-' && a[1] != '-') {   // -l4, -v0
      val = a.substr(2); a = a.substr(0, 2); hasVal = true;
    }
    auto arg = [&]() { return hasVal ? val : need(i, a); };
    if (a == "-h" || a == "--help") o.help = true;
    else if (a == "-l" || a == "--length") o.length = atoi(arg().c_str());
    else if (a == "-c" || a == "--controls") o.controls = atoi(arg().c_str());
    else if (a == "-L" || a == "--load-machine") o.loadMachine = arg();
    else if (a == "-S" || a == "--save-machine") o.saveMachine = arg();
    else if (a == "-C" || a == "--compose-machine") o.compose.push_back(arg());
    else if (a == "-e" || a == "--encode-file") o.encodeFile = arg();
    else if (a == "-d" || a == "--decode-file") o.decodeFile = arg();
    else if (a == "-E" || a == "--encode-string") o.encodeString = arg();
    else if (a == "-D" || a == "--decode-string") o.decodeString = arg();
    else if (a == "-b" || a == "--encode-bits") o.encodeBits = arg();
    else if (a == "-B" || a == "--decode-bits") o.decodeBits = arg();
    else if (a == "-V" || a == "--decode-viterbi") o.decodeViterbi = arg();
    else if (a == "-r" || a == "--raw") o.raw = true;
    else if (a == "--error-sub-prob") o.subProb = atof(arg().c_str());
    else if (a == "--error-iv-ratio") o.ivRatio = atof(arg().c_str());
    else if (a == "--error-dup-prob") o.dupProb = atof(arg().c_str());
    else if (a == "--error-del-open") o.delOpen = atof(arg().c_str());
    else if (a == "--error-del-ext") o.delExt = atof(arg().c_str());
    else if (a == "--error-global") o.errorGlobal = true;
    else if (a == "-F" || a == "--error-file") o.errorFile = arg();
    else if (a == "-f" || a == "--fit-error") o.fitError = arg();
    else if (a == "--error-counts") o.errorCounts = arg();
    else if (a == "--strict-guides") o.strictGuides = true;
    else if (a == "-v" || a == "--verbose") o.verbose = atoi(arg().c_str());
    else if (a == "--device") o.device = atoi(arg().c_str());
    else if (a == "--nocolor") {}
    else if (a == "--log") (void)arg();
    else die("unrecognised option '" + a + "'");
  }
  return o;
}

std::string slurp(const std::string& path, const char* what) {
  std::ifstream in(path, std::ios::binary);
  if (!in) die(std::string(what) + " not found");
  std::stringstream ss;
  ss << in.rdbuf();
  return ss.str();
}

}  // namespace

int main(int argc, char** argv) {
  const Options o = parse(argc, argv);
  if (o.help) { std::cout << kHelp << "\n"; return 1; }
  if (o.length > 31) die("Maximum context is 31 bases");

  // error model: --error-file wins over the flags, `local` included (dnastore.cpp:115-130)
  dnas_mutator_params mut;
  if (!o.errorFile.empty()) check(dnas_mutator_params_load_json(o.errorFile.c_str(), &mut));
  else check(dnas_mutator_params_from_flags(o.subProb, o.ivRatio, o.dupProb, o.delOpen, o.delExt, o.errorGlobal ? 1 : 0, o.length, &mut));

  if (!o.fitError.empty() || !o.errorCounts.empty()) {
    dnas_pairs* db = nullptr;
    check(dnas_stockholm_read((!o.fitError.empty() ? o.fitError : o.errorCounts).c_str(), &db));
    const dnas_pairs_view* v = dnas_pairs_get(db);
    char buf[16384];
    if (!o.fitError.empty()) {                                     // dnastore.cpp:135-140
      dnas_mutator_params fit;
      check(dnas_baum_welch(&mut, o.strictGuides, v->n_pairs, v->in_seqs, v->in_off, v->out_seqs, v->out_off, v->cm_in, v->cm_in_off,
                            v->cm_out, v->cm_out_off, o.device, &fit, nullptr));
      check(dnas_mutator_params_json(&fit, buf, sizeof buf));
    } else {                                                       // dnastore.cpp:142-146
      std::vector<double> counts(21 + mut.n_len);
      double ll = 0;
      check(dnas_fwdback_estep(&mut, o.strictGuides, v->n_pairs, v->in_seqs, v->in_off, v->out_seqs, v->out_off, v->cm_in, v->cm_in_off,
                               v->cm_out, v->cm_out_off, o.device, counts.data(), &ll, nullptr));
      check(dnas_mutator_counts_json(counts.data(), mut.n_len, buf, sizeof buf));
    }
    std::cout << buf;
    dnas_pairs_free(db);
    return 0;
  }

  // primary machine
  dnas_machine* machine = nullptr;
  if (!o.loadMachine.empty()) {
    check(dnas_machine_load_json(o.loadMachine.c_str(), &machine));
  } else {
    const char* canon = getenv("DNASTORE_L4C4");
    if (o.length == 4 && o.controls == 4 && canon) check(dnas_machine_load_json(canon, &machine));
    else die("this build does not contain the de Bruijn code builder: pass --load-machine (for -l 4 -c 4, set DNASTORE_L4C4 to data/l4c4.json)");
  }
  for (auto it = o.compose.rbegin(); it != o.compose.rend(); ++it) {   // right to left, dnastore.cpp:159-165
    dnas_machine *front = nullptr, *prod = nullptr;
    check(dnas_machine_load_json(it->c_str(), &front));
    check(dnas_machine_compose(front, machine, &prod));
    dnas_machine_free(front);
    dnas_machine_free(machine);
    machine = prod;
  }
  if (!o.saveMachine.empty()) {
    char* text = nullptr;
    size_t n = 0;
    check(dnas_machine_write_json(machine, &text, &n));
    if (o.saveMachine == "-") std::cout << text;
    else { std::ofstream out(o.saveMachine); out << text; }
    dnas_free(text);
  }

  char* text = nullptr;
  size_t n = 0;
  auto encodeOut = [&](int rc, const char* name) {
    check(rc);
    if (o.raw) std::cout << text << "\n";                           // FastaWriter with no header: one line
    else writeFasta(std::cout, name, text, false);
    dnas_free(text);
  };
  if (!o.encodeFile.empty()) {
    const std::string data = slurp(o.encodeFile, "Binary file");
    encodeOut(dnas_encode_bytes(machine, (const uint8_t*)data.data(), data.size(), &text, &n), o.encodeFile.c_str());
  } else if (!o.decodeFile.empty()) {                               // dnastore.cpp:188-193
    dnas_fastseqs* fs = nullptr;
    check(dnas_fastseqs_read(o.decodeFile.c_str(), &fs));
    for (int64_t i = 0; i < dnas_fastseqs_count(fs); ++i) {
      const char* seq = dnas_fastseqs_seq(fs, i);
      check(dnas_decode_exact(machine, seq, strlen(seq), &text, &n));
      uint8_t* bytes = nullptr;
      size_t nb = 0;
      check(dnas_symbols_to_bytes(text, n, &bytes, &nb));
      std::cout.write((const char*)bytes, (std::streamsize)nb);
      dnas_free(bytes);
      dnas_free(text);
    }
    dnas_fastseqs_free(fs);
  } else if (!o.encodeString.empty()) {
    encodeOut(dnas_encode_bytes(machine, (const uint8_t*)o.encodeString.data(), o.encodeString.size(), &text, &n), "ASCII_string");
  } else if (!o.decodeString.empty()) {
    check(dnas_decode_exact(machine, o.decodeString.data(), o.decodeString.size(), &text, &n));
    uint8_t* bytes = nullptr;
    size_t nb = 0;
    check(dnas_symbols_to_bytes(text, n, &bytes, &nb));
    std::cout.write((const char*)bytes, (std::streamsize)nb);
    dnas_free(bytes);
    dnas_free(text);
  } else if (!o.encodeBits.empty()) {
    encodeOut(dnas_encode_symbols(machine, o.encodeBits.data(), o.encodeBits.size(), &text, &n), "bit_string");
  } else if (!o.decodeBits.empty()) {
    check(dnas_decode_exact(machine, o.decodeBits.data(), o.decodeBits.size(), &text, &n));
    std::cout << text << "\n";
    dnas_free(text);
  } else if (!o.decodeViterbi.empty()) {                            // dnastore.cpp:217-223
    dnas_decoded* dec = nullptr;
    check(dnas_decode_fastseqs_ex(o.decodeViterbi.c_str(), machine, &mut, o.device, o.verbose >= 3, &dec));
    if (o.verbose >= 3) std::cerr << "Viterbi fill: " << dnas_decoded_tier(dec) << "; devices: " << dnas_decoded_devices(dec) << std::endl;
    for (int64_t i = 0; i < dnas_decoded_count(dec); ++i) {
      const std::string seq = dnas_decoded_seq(dec, i);
      if (o.verbose >= 3) {                                          // what the traceback found (viterbi.cpp:266-293)
        const uint64_t* ev = nullptr;
        const int64_t ne = dnas_decoded_events(dec, i, &ev);
        static const char base[] = "ACGT";
        for (int64_t k = 0; k < ne; ++k) {
          const unsigned kind = (unsigned)(ev[k] >> 62), pos = (unsigned)((ev[k] >> 32) & 0x3fffffffu), pay = (unsigned)(ev[k] & 0xffffffffu);
          if (kind == 1) std::cerr << "Substitution at " << pos << ": " << base[(pay >> 2) & 3] << " -> " << base[pay & 3] << std::endl;
          else if (kind == 2) std::cerr << "Deletion between " << (long)pos - 1 << " and " << pos << ": " << base[pay & 3] << std::endl;
          else if (kind == 3) {
            std::string dup;
            const unsigned cnt = pay >> 26;
            for (unsigned q = 0; q < cnt; ++q) dup.push_back(base[(pay >> (2 * (cnt - 1 - q))) & 3]);
            std::cerr << "Duplication at " << pos << ": " << dup << std::endl;
          }
        }
      }
      if (seq.empty()) std::cerr << "No valid Viterbi decoding found" << std::endl;   // viterbi.cpp:198-201
      writeFasta(std::cout, dnas_decoded_name(dec, i), seq, o.raw);
    }
    dnas_decoded_free(dec);
  }
  dnas_machine_free(machine);
  return 0;
}
