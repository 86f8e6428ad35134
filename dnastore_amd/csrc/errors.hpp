// Thread-local "last error" of the C ABI (dnas_last_error) and the exception -> status
// mapping used at every extern "C" entry point: no exception crosses the ABI.
#pragma once
#include <string>

namespace dnas {

std::string& lastErrorSlot();

inline int fail(int code, const std::string& msg) {
  lastErrorSlot() = msg;
  return code;
}

}  // namespace dnas
