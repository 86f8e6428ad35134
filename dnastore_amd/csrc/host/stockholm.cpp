#include "stockholm.hpp"

#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "machine.hpp"

namespace dnas {
namespace {

bool isGap(char c) { return c == '-' || c == '.'; }   // Alignment::isGap, alignpath.h:31

void addAlignment(AlignmentPairs& db, const std::vector<std::pair<std::string, std::string>>& rows) {
  if (rows.size() != 2)
    throw std::runtime_error("Training mutator model requires a 2-row alignment; this alignment has " +
                             std::to_string(rows.size()) + " rows");
  const std::string& g1 = rows[0].second;
  const std::string& g2 = rows[1].second;
  if (g1.size() != g2.size()) throw std::runtime_error("Alignment rows " + rows[0].first + " and " + rows[1].first + " differ in length");
  // GuideAlignmentEnvelope (alignpath.cpp:237-265): cumulativeMatches per column, position -> column
  std::vector<int32_t> cum{0};
  std::vector<size_t> p1{0}, p2{0};
  int32_t matches = 0;
  for (size_t col = 0; col < g1.size(); ++col) {
    const bool a = !isGap(g1[col]), b = !isGap(g2[col]);
    if (a) p1.push_back(col + 1);
    if (b) p2.push_back(col + 1);
    if (a && b) ++matches;
    cum.push_back(matches);
  }
  auto tokens = [&](const std::string& g, const std::string& name, std::vector<int8_t>& dst) {
    for (char c : g) {
      if (isGap(c)) continue;
      const int t = charToBase(c);
      if (t < 0) throw std::runtime_error(std::string("Unknown symbol ") + c + " in sequence " + name + " (alphabet is ACGT)");
      dst.push_back((int8_t)t);
    }
  };
  tokens(g1, rows[0].first, db.inSeqs);
  tokens(g2, rows[1].first, db.outSeqs);
  for (size_t c : p1) db.cmIn.push_back(cum[c]);
  for (size_t c : p2) db.cmOut.push_back(cum[c]);
  db.inOff.push_back((int64_t)db.inSeqs.size());
  db.outOff.push_back((int64_t)db.outSeqs.size());
  db.cmInOff.push_back((int64_t)db.cmIn.size());
  db.cmOutOff.push_back((int64_t)db.cmOut.size());
  db.inName.push_back(rows[0].first);
  db.outName.push_back(rows[1].first);
  ++db.n;
}

}  // namespace

AlignmentPairs readStockholmPairs(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("File not found: " + path);   // -> DNAS_E_IO (the reference: Fail, exit 1)
  AlignmentPairs db;
  db.inOff.push_back(0); db.outOff.push_back(0); db.cmInOff.push_back(0); db.cmOutOff.push_back(0);
  std::vector<std::pair<std::string, std::string>> rows;
  std::map<std::string, size_t> index;
  std::string line;
  auto flush = [&] {
    if (!rows.empty()) addAlignment(db, rows);
    rows.clear();
    index.clear();
  };
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    std::string a, b, c;
    ss >> a >> b >> c;
    if (a.empty()) continue;
    if (a.rfind("//", 0) == 0) { flush(); continue; }
    if (a[0] == '#') continue;                       // header and #=G? mark-up
    if (b.empty() || !c.empty()) continue;           // not a "name sequence" line (the reference warns)
    auto it = index.find(a);
    if (it == index.end()) { index[a] = rows.size(); rows.emplace_back(a, b); }
    else rows[it->second].second += b;               // interleaved blocks
  }
  flush();
  return db;
}

}  // namespace dnas
