// Minimal JSON DOM with the leniency the reference's files rely on: commas between
// elements are optional and a trailing comma is allowed (reference data/sync16.json has
// no commas between states, data/flusher.json ends its array with one; the reference
// parser tolerates both, src/gason.cpp:55-56,295-299).
#pragma once
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace dnas {

struct JsonValue {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<JsonValue> arr;
  std::vector<std::pair<std::string, JsonValue>> obj;  // insertion order kept

  const JsonValue* find(const std::string& key) const {
    for (const auto& kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  const JsonValue& at(const std::string& key) const {
    const JsonValue* v = find(key);
    if (!v) throw std::runtime_error("JSON: missing key \"" + key + "\"");
    return *v;
  }
  double number(const std::string& key) const {
    const JsonValue& v = at(key);
    if (v.kind != Number) throw std::runtime_error("JSON: \"" + key + "\" is not a number");
    return v.num;
  }
  bool boolean(const std::string& key) const {
    const JsonValue& v = at(key);
    if (v.kind != Bool) throw std::runtime_error("JSON: \"" + key + "\" is not a boolean");
    return v.b;
  }
  const std::string& string(const std::string& key) const {
    const JsonValue& v = at(key);
    if (v.kind != String) throw std::runtime_error("JSON: \"" + key + "\" is not a string");
    return v.str;
  }
  const std::vector<JsonValue>& array(const std::string& key) const {
    const JsonValue& v = at(key);
    if (v.kind != Array) throw std::runtime_error("JSON: \"" + key + "\" is not an array");
    return v.arr;
  }
};

// Throws std::runtime_error on malformed input.
JsonValue parseJson(const std::string& text);

}  // namespace dnas
