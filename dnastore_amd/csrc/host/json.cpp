#include "json.hpp"

#include <cstdlib>
#include <cstring>

namespace dnas {
namespace {

struct Parser {
  const char* p;
  const char* end;

  [[noreturn]] void fail(const char* what) const {
    throw std::runtime_error(std::string("JSON parse error: ") + what);
  }
  // commas count as white space: optional between elements, harmless when trailing
  void skip() {
    while (p < end && (*p == ' ' || (*p >= '\t' && *p <= '\r') || *p == ',')) ++p;
  }
  JsonValue value() {
    skip();
    if (p >= end) fail("unexpected end of input");
    JsonValue v;
    switch (*p) {
      case '{': {
        ++p;
        v.kind = JsonValue::Object;
        for (;;) {
          skip();
          if (p >= end) fail("unterminated object");
          if (*p == '}') { ++p; break; }
          if (*p != '"') fail("unquoted key");
          std::string key = stringLiteral();
          skip();
          if (p >= end || *p != ':') fail("expected ':'");
          ++p;
          v.obj.emplace_back(std::move(key), value());
        }
        return v;
      }
      case '[': {
        ++p;
        v.kind = JsonValue::Array;
        for (;;) {
          skip();
          if (p >= end) fail("unterminated array");
          if (*p == ']') { ++p; break; }
          v.arr.push_back(value());
        }
        return v;
      }
      case '"':
        v.kind = JsonValue::String;
        v.str = stringLiteral();
        return v;
      default:
        break;
    }
    if (!strncmp(p, "true", 4) && end - p >= 4) { p += 4; v.kind = JsonValue::Bool; v.b = true; return v; }
    if (!strncmp(p, "false", 5) && end - p >= 5) { p += 5; v.kind = JsonValue::Bool; v.b = false; return v; }
    if (!strncmp(p, "null", 4) && end - p >= 4) { p += 4; return v; }
    char* q = nullptr;
    v.num = strtod(p, &q);
    if (q == p) fail("unexpected character");
    p = q;
    v.kind = JsonValue::Number;
    return v;
  }
  std::string stringLiteral() {
    std::string s;
    ++p;  // opening quote
    while (p < end && *p != '"') {
      if (*p == '\\' && p + 1 < end) {
        ++p;
        switch (*p) {
          case 'n': s += '\n'; break;
          case 't': s += '\t'; break;
          case 'r': s += '\r'; break;
          case 'b': s += '\b'; break;
          case 'f': s += '\f'; break;
          case 'u': {
            if (end - p < 5) fail("bad \\u escape");
            unsigned cp = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) s += (char)cp;
            else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
            else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: s += *p;
        }
        ++p;
      } else {
        s += *p++;
      }
    }
    if (p >= end) fail("unterminated string");
    ++p;  // closing quote
    return s;
  }
};

}  // namespace

JsonValue parseJson(const std::string& text) {
  Parser ps{text.data(), text.data() + text.size()};
  JsonValue v = ps.value();
  ps.skip();
  return v;
}

}  // namespace dnas
