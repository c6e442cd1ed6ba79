// Test infrastructure (not product): holds csrc/ref_field.hpp's hand-written heap against libstdc++'s std::priority_queue —
// (1) the order in which nodes of EQUAL distance leave the heap, on random push / pop / "top, push, push, pop" sequences full of
//     ties (what the reference's result depends on: grid_mapper.cpp:399-433 over grid_mapper.hpp:104-110);
// (2) whole distance fields: RefField::step against a brushfire written with std::priority_queue itself (the round-3/4
//     implementation of this class, which the GPU suite held against the oracle's — pinned to the compiled reference — bit for bit),
//     random maps, several scans of insertions and erasures, small and large radii.
// Prints one line per part; exit code 0 = identical.
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <random>
#include <unordered_set>
#include <vector>

#include "ref_field.hpp"

using tbnav::RefField;

namespace {
struct StdNode { uint32_t d2; uint16_t i, j, si, sj; };
struct StdFarther { bool operator()(const StdNode& a, const StdNode& b) const { return a.d2 > b.d2; } };
using StdHeap = std::priority_queue<StdNode, std::vector<StdNode>, StdFarther>;

bool same(const RefField::Node& a, const StdNode& b) { return a.d2 == b.d2 && a.i == b.i && a.j == b.j && a.si == b.si && a.sj == b.sj; }

int heap_order(unsigned seed, int ops, int key_range) {
  std::mt19937 rng(seed);
  std::vector<RefField::Node> store;
  RefField::Heap H(store);
  StdHeap S;
  uint32_t serial = 0;
  auto push = [&](uint32_t d2) {
    ++serial;
    const uint16_t a = (uint16_t)(serial & 0xFFFF), b = (uint16_t)(serial >> 16);
    H.push(RefField::Node{d2, a, b, (uint16_t)(a ^ 0x5555), (uint16_t)(b + 1), 0});
    S.push(StdNode{d2, a, b, (uint16_t)(a ^ 0x5555), (uint16_t)(b + 1)});
  };
  for (int q = 0; q < ops; ++q) {
    const unsigned what = rng() % 8;
    if (H.empty() != S.empty()) return 1;
    if (what < 3 || H.empty()) push(rng() % key_range);
    else if (what < 5) { if (!same(H.top(), S.top())) return 2; H.pop(); S.pop(); }
    else {  // the brushfire's step: read the top, push up to four (some nearer than the top), THEN pop
      if (!same(H.top(), S.top())) return 3;
      const uint32_t base = H.top().d2;
      const int n = rng() % 5;
      for (int t = 0; t < n; ++t) push(base + (rng() % 7) - (rng() % 3 == 0 ? 2 : 0) < 0x7FFFFFFFu ? base + (rng() % 7) : 0);
      H.pop(); S.pop();
    }
  }
  while (!S.empty()) { if (H.empty() || !same(H.top(), S.top())) return 4; H.pop(); S.pop(); }
  return H.empty() ? 0 : 5;
}

// the brushfire over std::priority_queue (euclideanSignedDistanceField, grid_mapper.cpp:333-435; enqueueCell :272-329)
void brushfire_std(int xs, int radius, const std::unordered_set<int>& occ, std::vector<uint16_t>& code) {
  if (occ.empty()) return;
  std::vector<uint8_t> marked((size_t)xs * xs, 0);
  StdHeap Q;
  for (int key : occ) {
    code[key] = 0; marked[key] = 1;
    const uint16_t ki = (uint16_t)(key / xs), kj = (uint16_t)(key % xs);
    Q.push(StdNode{0, ki, kj, ki, kj});
  }
  auto enqueue = [&](int i, int j, int si, int sj) {
    const int idx = i * xs + j;
    if (marked[idx]) return;
    const int di = std::abs(i - si), dj = std::abs(j - sj);
    if (di >= radius || dj >= radius) return;
    const int d2 = di * di + dj * dj;
    if (d2 > radius * radius) return;
    code[idx] = (uint16_t)d2;
    Q.push(StdNode{(uint32_t)d2, (uint16_t)i, (uint16_t)j, (uint16_t)si, (uint16_t)sj});
    marked[idx] = 1;
  };
  while (!Q.empty()) {
    const StdNode c = Q.top();
    if (c.i > 0) enqueue(c.i - 1, c.j, c.si, c.sj);
    if (c.j > 0) enqueue(c.i, c.j - 1, c.si, c.sj);
    if (c.i < xs - 1) enqueue(c.i + 1, c.j, c.si, c.sj);
    if (c.j < xs - 1) enqueue(c.i, c.j + 1, c.si, c.sj);
    Q.pop();
  }
}

int fields(unsigned seed, int xs, int radius, int particles, int scans, int threads) {
  std::mt19937 rng(seed);
  RefField rf(particles, xs, radius);
  std::vector<std::unordered_set<int>> occ(particles);
  std::vector<std::vector<uint16_t>> code(particles, std::vector<uint16_t>((size_t)xs * xs, 0xFFFF));
  for (int s = 0; s < scans; ++s) {
    std::vector<int> all; std::vector<size_t> off{0};
    std::vector<int> shared;   // most particles see the same changes (the sharing path), some their own
    const int n_sh = 5 + rng() % 40;
    for (int q = 0; q < n_sh; ++q) shared.push_back((int)(rng() % (unsigned)(xs * xs)));
    for (int p = 0; p < particles; ++p) {
      std::vector<int> ev = shared;
      if (rng() % 3 == 0) for (int q = 0; q < 3; ++q) ev.push_back((int)(rng() % (unsigned)(xs * xs)));
      if (s > 0 && !occ[p].empty() && rng() % 2) {   // erase a few cells that are in the set (bit 31)
        int k = 0;
        for (int c : occ[p]) { if (k++ % 7 == 0) ev.push_back((int)(0x80000000u | (unsigned)c)); if (k > 40) break; }
      }
      if (rng() % 11 == 0) ev.clear();   // a scan that changes nothing
      for (int e : ev) {   // updateCellHash, as RefField::apply does
        const int idx = e & 0x7FFFFFFF;
        if (e < 0) { if (occ[p].find(idx) != occ[p].end()) occ[p].erase(idx); }
        else if (occ[p].find(idx) == occ[p].end()) occ[p].insert(idx);
      }
      brushfire_std(xs, radius, occ[p], code[p]);
      all.insert(all.end(), ev.begin(), ev.end());
      off.push_back(all.size());
    }
    rf.step(0, particles, threads, all.data(), off.data());
    for (int p = 0; p < particles; ++p)
      if (std::memcmp(rf.codes(p), code[p].data(), sizeof(uint16_t) * code[p].size()) != 0) return 10 + s;
    if (s == scans / 2) {   // a resampling: every slot a copy of its parent (particle_filter.cpp:495-499)
      std::vector<int> parent(particles);
      for (int m = 0; m < particles; ++m) parent[m] = (int)(rng() % (unsigned)particles);
      rf.resample(parent.data());
      std::vector<std::unordered_set<int>> o2(particles); std::vector<std::vector<uint16_t>> c2(particles);
      for (int m = 0; m < particles; ++m) { o2[m] = occ[parent[m]]; c2[m] = code[parent[m]]; }
      occ.swap(o2); code.swap(c2);
    }
  }
  return 0;
}
}  // namespace

int main() {
  int bad = 0;
  for (unsigned seed = 1; seed <= 40; ++seed) {
    const int rc = heap_order(seed, 20000, seed % 4 == 0 ? 3 : (seed % 4 == 1 ? 50 : 4000));
    if (rc) { std::printf("heap order: seed %u differs from std::priority_queue (%d)\n", seed, rc); bad = 1; }
  }
  std::printf("heap order: 40 sequences of 20000 operations %s\n", bad ? "DIFFER" : "identical to std::priority_queue");
  int badf = 0;
  const int cases[][5] = {{40, 200, 6, 5, 1}, {80, 200, 5, 4, 3}, {64, 9, 6, 5, 2}, {120, 30, 4, 4, 4}, {200, 200, 3, 3, 2}};
  unsigned seed = 100;
  for (const auto& c : cases) {
    const int rc = fields(++seed, c[0], c[1], c[2], c[3], c[4]);
    if (rc) { std::printf("fields: %d x %d radius %d differs (%d)\n", c[0], c[0], c[1], rc); badf = 1; }
  }
  std::printf("fields: 5 maps x several scans %s\n", badf ? "DIFFER" : "identical to the std::priority_queue brushfire");
  return bad | badf;
}
