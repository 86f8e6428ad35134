// JIT specialisation of csrc/viterbi_tiera.hip for one machine (hiprtc), with an on-disk
// cache of code objects next to the library (dnastore_amd/kcache/*.hsaco).
#pragma once
#include <string>
#include <vector>

namespace dnas {

// Directory holding libdnastore_amd.so (found with dladdr).
std::string libraryDir();

// Compile `sourcePath` for gfx950 with the given -D options (one per line in `defines`).
// Returns the code object bytes; throws std::runtime_error with the compiler log on failure.
// `key` names the cache entry; a cached object compiled from the same source is reused.
std::vector<char> jitCompile(const std::string& sourcePath, const std::string& defines, const std::string& key);

}  // namespace dnas
