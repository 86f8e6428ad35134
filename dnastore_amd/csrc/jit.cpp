#include "jit.hpp"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <sstream>
#include <stdexcept>

namespace dnas {
namespace {

uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

// a name no other writer of the same file uses: several host threads of one process (dnas_decode_fastseqs with device -1) and
// several processes (one per GPU) may compile or record the same thing at the same time; each writes its own temporary and
// renames it into place
std::string tempName(const std::string& path) {
  static std::atomic<unsigned long> counter{0};
  return path + "." + std::to_string((long)getpid()) + "." + std::to_string(counter.fetch_add(1)) + ".tmp";
}

// hiprtc through a handle of its own: the library does not link it (a machine with a warm kernel cache needs no compiler), and
// DNAS_HIPRTC_LIBRARY names another one.  NOTE what this does NOT cure: a process that has loaded PyTorch already holds PyTorch's
// bundled libhiprtc.so.7 / libamd_comgr.so.3 (another ROCm release's compiler under the SAME sonames), and whatever asks for those
// sonames afterwards -- the system's hiprtc asking for comgr included -- gets PyTorch's copies.  The same source then compiles to
// other code than build() makes with the system's compiler (measured: water64.1*l4c4's 64-register kernel spills 31 registers
// instead of 5 and runs 15 % slower; the 128-register kernels differ by 1 %).  The code objects of the fixture and bench machines
// therefore ship precompiled (dnastore_amd/kcache/, __graft_entry__.build()); INTEGRATION.md section 6.
struct Hiprtc {
  void* lib = nullptr;
  hiprtcResult (*version)(int*, int*) = nullptr;
  hiprtcResult (*create)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  hiprtcResult (*compile)(hiprtcProgram, int, const char**) = nullptr;
  hiprtcResult (*logSize)(hiprtcProgram, size_t*) = nullptr;
  hiprtcResult (*log)(hiprtcProgram, char*) = nullptr;
  hiprtcResult (*codeSize)(hiprtcProgram, size_t*) = nullptr;
  hiprtcResult (*code)(hiprtcProgram, char*) = nullptr;
  hiprtcResult (*destroy)(hiprtcProgram*) = nullptr;
  const char* (*errorString)(hiprtcResult) = nullptr;
  std::string where;
  Hiprtc() {
    const char* names[] = {getenv("DNAS_HIPRTC_LIBRARY"), "/opt/rocm/lib/libhiprtc.so.7", "libhiprtc.so.7", "libhiprtc.so"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) { where = n; break; }
    }
    if (!lib) return;
#define DNAS_RTC_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(lib, name))
    DNAS_RTC_SYM(version, "hiprtcVersion"); DNAS_RTC_SYM(create, "hiprtcCreateProgram"); DNAS_RTC_SYM(compile, "hiprtcCompileProgram");
    DNAS_RTC_SYM(logSize, "hiprtcGetProgramLogSize"); DNAS_RTC_SYM(log, "hiprtcGetProgramLog"); DNAS_RTC_SYM(codeSize, "hiprtcGetCodeSize");
    DNAS_RTC_SYM(code, "hiprtcGetCode"); DNAS_RTC_SYM(destroy, "hiprtcDestroyProgram"); DNAS_RTC_SYM(errorString, "hiprtcGetErrorString");
#undef DNAS_RTC_SYM
  }
  bool ok() const { return lib && version && create && compile && logSize && log && codeSize && code && destroy && errorString; }
};
const Hiprtc& hiprtc() { static Hiprtc h; return h; }

std::string slurp(const std::string& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot read " + path);
  std::stringstream ss;
  ss << in.rdbuf();
  return ss.str();
}

}  // namespace

std::string libraryDir() {
  Dl_info info;
  if (dladdr((void*)&libraryDir, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t s = p.rfind('/');
    return s == std::string::npos ? "." : p.substr(0, s);
  }
  return ".";
}

// csrc/viterbi_tiera.hip travels INSIDE the library (the Makefile assembles it in with .incbin, see
// tiera_embed.S): a maintainer who installs only libdnastore_amd.so still gets the tier-A/C kernel.
extern "C" const char dnas_tiera_source[];
extern "C" const char dnas_tiera_source_end[];

std::string tieraSource() {
  if (const char* path = getenv("DNAS_TIERA_SRC")) return slurp(path);   // kernel development: compile this file instead
  return std::string(dnas_tiera_source, (size_t)(dnas_tiera_source_end - dnas_tiera_source));
}

std::string kernelCacheDir() {
  if (const char* dir = getenv("DNAS_KCACHE_DIR")) return dir;
  return libraryDir() + "/kcache";
}

#ifndef DNAS_ARCH
#define DNAS_ARCH "gfx950"
#endif

unsigned long long textHash(const std::string& text) { return (unsigned long long)fnv1a(text + "|" DNAS_ARCH); }
// What a tuning record names as "kernel=": the hash of the kernel's CODE -- comments and white space taken out, so that a
// reworded comment does not make the measured records stale (the code-object cache above keys on the full text).
unsigned long long kernelSourceHash() {
  const std::string src = tieraSource();
  std::string code;
  code.reserve(src.size());
  for (size_t i = 0; i < src.size();) {
    if (src.compare(i, 2, "//") == 0) { while (i < src.size() && src[i] != '\n') ++i; continue; }
    if (src.compare(i, 2, "/*") == 0) { const size_t e = src.find("*/", i + 2); i = e == std::string::npos ? src.size() : e + 2; continue; }
    if (src[i] == '"') {                      // (string literals stay as they are)
      const size_t e = src.find('"', i + 1);
      const size_t stop = e == std::string::npos ? src.size() : e + 1;
      code.append(src, i, stop - i);
      i = stop;
      continue;
    }
    if (src[i] == ' ' || src[i] == '\t' || src[i] == '\n' || src[i] == '\r' || src[i] == '\\') { ++i; continue; }
    code.push_back(src[i++]);
  }
  return (unsigned long long)fnv1a(code);
}

// looked for in the kernel cache first, then among the records shipped with the library (<library dir>/tune/: the verdicts
// for the fixture and bench machines, regenerated with tools/make_tune_records.sh whenever the kernel source changes)
std::string cacheNoteRead(const std::string& name) {
  for (const std::string& dir : {kernelCacheDir(), libraryDir() + "/tune"}) {
    std::ifstream in(dir + "/" + name);
    if (in) return std::string((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
  }
  return "";
}

void cacheNoteWrite(const std::string& name, const std::string& text) {
  const std::string dir = kernelCacheDir(), path = dir + "/" + name, tmp = tempName(path);
  mkdir(dir.c_str(), 0755);
  std::ofstream out(tmp);
  if (!out) return;
  out << text;
  out.close();
  if (rename(tmp.c_str(), path.c_str()) != 0) remove(tmp.c_str());
}

std::vector<char> jitCompile(const std::string& defines, const std::string& key) {
  // one compilation at a time per process: the threads of a multi-device call ask for the same code object, and the second
  // one finds it in the cache
  static std::mutex jitMutex;
  std::lock_guard<std::mutex> lock(jitMutex);
  const std::string src = tieraSource();
  const Hiprtc& rtc = hiprtc();
  int rtcMajor = 0, rtcMinor = 0;
  if (rtc.ok()) (void)rtc.version(&rtcMajor, &rtcMinor);
  char name[64];
  snprintf(name, sizeof name, "%016llx",
           (unsigned long long)fnv1a(src, fnv1a(defines + "|" + key + "|" DNAS_ARCH "|hiprtc" + std::to_string(rtcMajor) + "." + std::to_string(rtcMinor))));
  const std::string cacheDir = kernelCacheDir();
  const std::string cachePath = cacheDir + "/tiera_" + name + ".hsaco";
  {
    std::ifstream in(cachePath, std::ios::binary);
    if (in) {
      std::vector<char> code((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
      if (!code.empty()) return code;
    }
  }
  std::vector<std::string> opts{"--offload-arch=" DNAS_ARCH, "-O3", "-ffp-contract=off", "-std=c++17"};
  {
    std::stringstream ss(defines);
    std::string line;
    while (std::getline(ss, line))
      if (!line.empty()) opts.push_back(line);
  }
  std::vector<const char*> copts;
  for (const auto& o : opts) copts.push_back(o.c_str());
  if (!rtc.ok()) throw std::runtime_error("the run-time compiler (libhiprtc.so.7) could not be loaded: no code object for this machine in " + cacheDir);
  hiprtcProgram prog;
  if (rtc.create(&prog, src.c_str(), "viterbi_tiera.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    throw std::runtime_error("hiprtcCreateProgram failed");
  const hiprtcResult rc = rtc.compile(prog, (int)copts.size(), copts.data());
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    rtc.logSize(prog, &n);
    std::string log(n, '\0');
    if (n) rtc.log(prog, &log[0]);
    rtc.destroy(&prog);
    throw std::runtime_error(std::string("hiprtc: ") + rtc.errorString(rc) + "\n" + log);
  }
  size_t n = 0;
  rtc.codeSize(prog, &n);
  std::vector<char> code(n);
  rtc.code(prog, code.data());
  rtc.destroy(&prog);
  // best-effort cache write (atomic rename; a read-only tree just skips it)
  mkdir(cacheDir.c_str(), 0755);
  const std::string tmp = tempName(cachePath);
  {
    std::ofstream out(tmp, std::ios::binary);
    if (out) {
      out.write(code.data(), (std::streamsize)code.size());
      out.close();
      if (!out || rename(tmp.c_str(), cachePath.c_str()) != 0) remove(tmp.c_str());
    }
  }
  if (getenv("DNAS_JIT_DUMP")) {          // kernel development: the options this code object was compiled with, next to it
    std::ofstream note(cachePath + ".defs");
    if (note) note << defines << "\n";
  }
  return code;
}

}  // namespace dnas
