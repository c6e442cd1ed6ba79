// Test infrastructure (not product): holds csrc/ref_field.hpp's hand-written heap against libstdc++'s std::priority_queue —
// (1) the order in which nodes of EQUAL distance leave the heap, on random push / pop / "top, push, push, pop" sequences full of
//     ties (what the reference's result depends on: grid_mapper.cpp:399-433 over grid_mapper.hpp:104-110);
// (2) whole distance fields: RefField::step against a brushfire written with std::priority_queue itself (the round-3/4
//     implementation of this class, which the GPU suite held against the oracle's — pinned to the compiled reference — bit for bit),
//     random maps, several scans of insertions and erasures, small and large radii.
// (3) (round 6) the LAZY brushfire: passes stopped `reach` cells out, resumed on demand (RefField::ensure) and replayed where stale
//     cells are wanted, against the eager std::priority_queue brushfire — every written cell equal at every moment, whole fields
//     equal after codes() (completion + replay), and a simulated device slot driven by plan_flush() equal to the state's image
//     after every flush.  Random occupancy; sorted / reverse / history insert orders; erase events; resamplings; small radii
//     (cells out of reach keep stale values) and the 200-cell radius.
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

bool same(const RefField::Node& a, const StdNode& b) { return a.d2() == b.d2 && a.i() == b.i && a.j() == b.j && a.si() == b.si && a.sj() == b.sj; }

int heap_order(unsigned seed, int ops, int key_range) {
  std::mt19937 rng(seed);
  std::vector<RefField::Node> store;
  RefField::Heap H(store);
  StdHeap S;
  uint32_t serial = 0;
  auto push = [&](uint32_t d2) {
    ++serial;
    const uint16_t a = (uint16_t)(serial & 0xFFF), b = (uint16_t)((serial >> 12) & 0xFFF);   // (a node carries 12 bits per coordinate)
    H.push(RefField::Node::make(d2, a, b, (a ^ 0x555) & 0xFFF, (b + 1) & 0xFFF));
    S.push(StdNode{d2, a, b, (uint16_t)((a ^ 0x555) & 0xFFF), (uint16_t)((b + 1) & 0xFFF)});
  };
  for (int q = 0; q < ops; ++q) {
    const unsigned what = rng() % 8;
    if (H.empty() != S.empty()) return 1;
    if (what < 3 || H.empty()) push(rng() % key_range);
    else if (what < 5) { if (!same(H.top(), S.top())) return 2; H.pop(); S.pop(); }
    else {  // the brushfire's step: read the top, push up to four (some nearer than the top), THEN pop
      if (!same(H.top(), S.top())) return 3;
      const uint32_t base = H.top().d2();
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
  rf.set_reach(0);   // eager: every pass to the end, as up to round 5
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
    for (int p = 0; p < particles; ++p) {
      const uint16_t* got = rf.codes(p);
      if (!got || std::memcmp(got, code[p].data(), sizeof(uint16_t) * code[p].size()) != 0) return 10 + s;
    }
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

// a device slot as the handle keeps it: whole images, journal ranges (optionally from "everything pending"), copies between slots
struct SimDevice {
  std::vector<std::vector<uint16_t>> slot;
  SimDevice(int n, size_t G) : slot((size_t)n, std::vector<uint16_t>(G, RefField::kUnreached)) {}
  void flush(RefField& rf) {
    RefField::Flush f;
    rf.plan_flush(f);
    for (size_t q = 0; q < f.dense_slot.size(); ++q) if (f.dense_img[q] >= 0) slot[f.dense_slot[q]] = f.images[f.dense_img[q]];
    for (size_t q = 0; q < f.dense_slot.size(); ++q) if (f.dense_img[q] < 0) slot[f.dense_slot[q]] = slot[f.dense_src[q]];
    for (size_t p = 0; p < f.jobs.size(); ++p) {
      const auto& j = f.jobs[p];
      if (j.reset) std::fill(slot[p].begin(), slot[p].end(), RefField::kPending);
      for (uint32_t e = 0; e < j.count; ++e) slot[p][f.entries[j.off + e].cell] = (uint16_t)f.entries[j.off + e].code;
    }
  }
  void gather(const std::vector<int>& parent) {
    std::vector<std::vector<uint16_t>> t(slot.size());
    for (size_t m = 0; m < slot.size(); ++m) t[m] = slot[parent[m]];
    slot.swap(t);
  }
  // the slot equals the state's image; every non-pending cell of it equals the eager field
  int check(RefField& rf, const std::vector<std::vector<uint16_t>>& eager, bool occ_nonempty_only = false) {
    std::vector<uint16_t> img;
    for (size_t p = 0; p < slot.size(); ++p) {
      rf.image((int)p, img);
      if (img != slot[p]) return 1;
      for (size_t c = 0; c < img.size(); ++c) if (img[c] != RefField::kPending && img[c] != eager[p][c]) return 2;
    }
    return 0;
  }
};

long long g_pending = 0, g_resumes = 0, g_replays = 0, g_passes = 0, g_completions = 0, g_pops = 0;
// order: 0 random cells, 1 ascending, 2 descending, 3 a wall drawn cell by cell then partly erased ("history")
int lazy_fields(unsigned seed, int xs, int radius, int particles, int scans, int threads, int reach, int order) {
  std::mt19937 rng(seed);
  const size_t G = (size_t)xs * xs;
  RefField rf(particles, xs, radius);
  rf.set_reach(reach);
  SimDevice dev(particles, G);
  rf.slots_hold_initial_image();
  std::vector<std::unordered_set<int>> occ(particles);
  std::vector<std::vector<uint16_t>> code(particles, std::vector<uint16_t>(G, 0xFFFF));
  for (int s = 0; s < scans; ++s) {
    std::vector<int> all; std::vector<size_t> off{0};
    std::vector<int> shared;
    const int n_sh = 5 + rng() % 40;
    if (order == 3) {   // two walls of a room, extended every scan; from the third scan on the oldest stretch is erased again
      const int r0 = xs / 4 + (s % 3), c0 = xs / 4;
      for (int q = 0; q < n_sh; ++q) shared.push_back(r0 * xs + std::min(xs - 1, c0 + s * 3 + q));
      for (int q = 0; q < n_sh / 2; ++q) shared.push_back(std::min(xs - 1, r0 + q) * xs + c0);
      if (s >= 2) for (int q = 0; q < 6; ++q) shared.push_back((int)(0x80000000u | (unsigned)((xs / 4 + ((s - 2) % 3)) * xs + c0 + (s - 2) * 3 + q)));
    } else {
      for (int q = 0; q < n_sh; ++q) shared.push_back((int)(rng() % (unsigned)G));
      if (order == 1) std::sort(shared.begin(), shared.end());
      if (order == 2) std::sort(shared.rbegin(), shared.rend());
    }
    for (int p = 0; p < particles; ++p) {
      std::vector<int> ev = shared;
      if (rng() % 3 == 0) for (int q = 0; q < 3; ++q) ev.push_back((int)(rng() % (unsigned)G));
      if (s > 0 && !occ[p].empty() && rng() % 2) {
        int k = 0;
        for (int c : occ[p]) { if (k++ % 5 == 0) ev.push_back((int)(0x80000000u | (unsigned)c)); if (k > 40) break; }
      }
      if (s == scans - 2 && p % 4 == 1) {   // everything occupied leaves the set: the pass has nothing to do, every cell keeps its value
        ev.clear();
        for (int c : occ[p]) ev.push_back((int)(0x80000000u | (unsigned)c));
      }
      if (rng() % 11 == 0) ev.clear();
      for (int e : ev) {
        const int idx = e & 0x7FFFFFFF;
        if (e < 0) { if (occ[p].find(idx) != occ[p].end()) occ[p].erase(idx); }
        else if (occ[p].find(idx) == occ[p].end()) occ[p].insert(idx);
      }
      brushfire_std(xs, radius, occ[p], code[p]);
      all.insert(all.end(), ev.begin(), ev.end());
      off.push_back(all.size());
    }
    rf.step(0, particles, threads, all.data(), off.data());
    dev.flush(rf);
    if (int rc = dev.check(rf, code)) return 100 * (s + 1) + rc;
    // lookups: a few cells per particle; the pending ones are reported back until none is left (what the handle's settle loop does)
    for (int round = 0; round < 400; ++round) {
      std::vector<int> ps, cs;
      for (int p = 0; p < particles; ++p) {
        std::mt19937 lr(seed * 7919u + (unsigned)s * 131u + (unsigned)p);   // the same lookups every round
        for (int q = 0; q < 6; ++q) {
          int c;
          if (q < 4 && !occ[p].empty()) {   // next to an obstacle (what a beam's end point is) ...
            auto it = occ[p].begin(); std::advance(it, lr() % occ[p].size());
            const int i = std::min(xs - 1, std::max(0, *it / xs + (int)(lr() % 5) - 2)), j = std::min(xs - 1, std::max(0, *it % xs + (int)(lr() % 5) - 2));
            c = i * xs + j;
          } else c = (int)(lr() % (unsigned)G);   // ... and anywhere (an unmapped area: the pass runs far, or ends and the stale value is wanted)
          if (occ[p].empty()) continue;           // likelihoodFieldModel returns before any lookup (grid_mapper.cpp:94-98)
          if (dev.slot[p][c] == RefField::kPending) { ps.push_back(p); cs.push_back(c); ++g_pending; break; }
          if (dev.slot[p][c] != code[p][c]) return 100 * (s + 1) + 3;
        }
      }
      if (ps.empty()) break;
      if (round == 399) return 100 * (s + 1) + 4;
      if (int rc = rf.ensure(ps.data(), cs.data(), (int)ps.size(), threads)) return 100 * (s + 1) + 10 + (-rc);
      dev.flush(rf);
      if (int rc = dev.check(rf, code)) return 100 * (s + 1) + 20 + rc;
    }
    if (s == scans / 2) {
      std::vector<int> parent(particles);
      for (int m = 0; m < particles; ++m) parent[m] = (int)(rng() % (unsigned)particles);
      rf.resample(parent.data());
      dev.gather(parent);
      std::vector<std::unordered_set<int>> o2(particles); std::vector<std::vector<uint16_t>> c2(particles);
      for (int m = 0; m < particles; ++m) { o2[m] = occ[parent[m]]; c2[m] = code[parent[m]]; }
      occ.swap(o2); code.swap(c2);
      if (int rc = dev.check(rf, code)) return 100 * (s + 1) + 30 + rc;
    }
    if (s % 3 == 2) {   // a whole-field export of one particle in the middle of the run (completion + replay), then on with the scans
      const int p = (int)(rng() % (unsigned)particles);
      const uint16_t* got = rf.codes(p);
      if (!got || std::memcmp(got, code[p].data(), sizeof(uint16_t) * G) != 0) return 100 * (s + 1) + 40;
      dev.flush(rf);
      if (int rc = dev.check(rf, code)) return 100 * (s + 1) + 50 + rc;
    }
  }
  for (int p = 0; p < particles; ++p) {
    const uint16_t* got = rf.codes(p);
    if (!got || std::memcmp(got, code[p].data(), sizeof(uint16_t) * G) != 0) return 9000 + p;
  }
  dev.flush(rf);
  if (int rc = dev.check(rf, code)) return 9900 + rc;
  const auto& k = rf.counters();
  g_resumes += k.resumes; g_replays += k.replays; g_passes += k.passes; g_completions += k.completions; g_pops += k.pops;
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
  int badl = 0, nl = 0;
  // xs, radius, particles, scans, threads, reach
  const int lcases[][6] = {{40, 200, 6, 7, 1, 2}, {80, 200, 5, 6, 3, 3}, {64, 9, 6, 7, 2, 2}, {120, 30, 4, 6, 4, 4}, {200, 200, 3, 5, 2, 6}, {96, 20, 5, 8, 2, 1},
                           {400, 200, 2, 4, 2, 6}};
  for (const auto& c : lcases)
    for (int order = 0; order < 4; ++order) {
      const int rc = lazy_fields(++seed, c[0], c[1], c[2], c[3], c[4], c[5], order);
      ++nl;
      if (rc) { std::printf("lazy fields: %d x %d radius %d reach %d order %d differs (%d)\n", c[0], c[0], c[1], c[5], order, rc); badl = 1; }
    }
  std::printf("lazy fields: %lld passes (%lld ran to the end), %lld pending lookups, %lld states resumed, %lld lineages replayed, %lld iterations\n", g_passes, g_completions, g_pending, g_resumes, g_replays, g_pops);
  std::printf("lazy fields: %d runs (truncated, resumed, replayed; journal-driven device slots) %s\n", nl, badl ? "DIFFER" : "identical to the eager std::priority_queue brushfire");
  return bad | badf | badl;
}
