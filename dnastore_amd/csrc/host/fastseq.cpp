#include "fastseq.hpp"

#include <zlib.h>

#include <cctype>
#include <ostream>
#include <stdexcept>

#include "machine.hpp"

namespace dnas {
namespace {

// whole (possibly gzipped) file -> memory; gzread passes plain files through unchanged
std::string slurp(const std::string& path) {
  gzFile fp = gzopen(path.c_str(), "r");
  if (!fp) throw std::runtime_error("Couldn't open " + path);
  std::string data;
  char buf[1 << 16];
  int n;
  while ((n = gzread(fp, buf, sizeof buf)) > 0) data.append(buf, (size_t)n);
  gzclose(fp);
  return data;
}

}  // namespace

std::vector<FastSeq> readFastSeqs(const std::string& path) {
  const std::string data = slurp(path);
  std::vector<FastSeq> seqs;
  size_t i = 0;
  const size_t n = data.size();
  // skip to the first header
  while (i < n && data[i] != '>' && data[i] != '@') ++i;
  while (i < n) {
    FastSeq fs;
    ++i;  // header marker
    while (i < n && !isspace((unsigned char)data[i])) fs.name += data[i++];
    if (i < n && data[i] != '\n') {
      ++i;  // the separator after the name
      while (i < n && data[i] != '\n') fs.comment += data[i++];
      while (!fs.comment.empty() && (fs.comment.back() == '\r')) fs.comment.pop_back();
    }
    if (i < n) ++i;  // newline
    // sequence lines until the next record marker at a line start, or '+'
    bool lineStart = true;
    while (i < n) {
      const char c = data[i];
      if (lineStart && (c == '>' || c == '@' || c == '+')) break;
      if (c == '\n') lineStart = true;
      else {
        lineStart = false;
        if (isgraph((unsigned char)c)) fs.seq += c;
      }
      ++i;
    }
    if (i < n && data[i] == '+') {
      while (i < n && data[i] != '\n') ++i;  // rest of the '+' line
      if (i < n) ++i;
      while (i < n && fs.qual.size() < fs.seq.size()) {
        if (isgraph((unsigned char)data[i])) fs.qual += data[i];
        ++i;
      }
      while (i < n && data[i] != '>' && data[i] != '@') ++i;
      if (fs.qual.size() != fs.seq.size()) fs.qual.clear();
    }
    seqs.push_back(std::move(fs));
  }
  return seqs;
}

void writeFastaSeqs(std::ostream& out, const std::vector<FastSeq>& seqs, size_t width) {
  for (const auto& s : seqs) {
    out << '>' << s.name;
    if (!s.comment.empty()) out << ' ' << s.comment;
    out << '\n';
    for (size_t i = 0; i < s.seq.size(); i += width) out << s.seq.substr(i, width) << '\n';
  }
}

std::vector<uint8_t> tokenizeDNA(const std::string& seq, const std::string& name) {
  std::vector<uint8_t> tok;
  tok.reserve(seq.size());
  for (char c : seq) {
    const int b = charToBase(c);
    if (b < 0)
      throw std::runtime_error(std::string("Unknown symbol ") + c + " in sequence " + name + " (alphabet is ACGT)");
    tok.push_back((uint8_t)b);
  }
  return tok;
}

}  // namespace dnas
