// JIT specialisation of csrc/viterbi_tiera.hip for one machine (hiprtc), with an on-disk
// cache of code objects (next to the library in kcache/, or $DNAS_KCACHE_DIR).
#pragma once
#include <string>
#include <vector>

namespace dnas {

// Directory holding libdnastore_amd.so (found with dladdr).
std::string libraryDir();

// Compile the tier-A/C kernel source (csrc/viterbi_tiera.hip, embedded in the library; DNAS_TIERA_SRC=<file>
// compiles that file instead) for gfx950 with the given -D options (one per line in `defines`).
// Returns the code object bytes; throws std::runtime_error with the compiler log on failure.
// `key` names the cache entry; a cached object compiled from the same source, options, architecture and
// hiprtc version is reused.  The cache directory is <library dir>/kcache, or $DNAS_KCACHE_DIR.
std::vector<char> jitCompile(const std::string& defines, const std::string& key);
std::string kernelCacheDir();
// A small text record next to the cached code objects (plan autotuning decisions): "" when absent; writes are best effort.
std::string cacheNoteRead(const std::string& name);
void cacheNoteWrite(const std::string& name, const std::string& text);
// names of tuning records hash the machine (textHash); the records name the kernel source they were measured with (kernelSourceHash)
unsigned long long textHash(const std::string& text);
unsigned long long kernelSourceHash();

}  // namespace dnas
