// ref_field.hpp — the reference's distance field, reproduced on the host for small ensembles.
//
// bmapping::GridMapper::euclideanSignedDistanceField (bmapping/src/bmapping/grid_mapper.cpp:333-435, enqueueCell
// :272-329) is a multi-source brushfire over a std::priority_queue, seeded by iterating the std::unordered_set<int>
// of occupied cells.  Its result is NOT a function of the occupied set: it depends on the hash set's iteration
// order (i.e. its whole insert / erase history) and on the heap's handling of equal distances, it pops the wrong
// element when a neighbour pushed during an expansion is nearer than the current top (:401-431 read top, push,
// THEN pop), and cells that are no longer reached keep their old value (hard part 1).
//
// The device's default lookups compute the exact nearest-obstacle distance instead (rbpf.hip, DistSrc).  This class
// is the third leg of SURVEY's contract: a mode in which the product reproduces the reference's field bit for bit
// — same libstdc++ containers, fed the same insert / erase sequence (the beam-ordered raycast kernel logs it),
// copied the way ParticleFilter::lowVarianceResampling copies particles (particle_filter.cpp:468-500) — so that
// an un-injected run matches the reference end to end.  It is serial host work per particle (~0.5 ms at 80 x 80,
// ~16 ms at 400 x 400) — particles are independent, so the handle spreads them over host threads (for_each_particle;
// round 3): the order of operations inside one particle's set and heap, which is what the result depends on, is untouched.
//
// Distances are kept as u16 codes = squared distance in cells (0xFFFF = never reached = max_occ_dist_):
// sqrt((double)code) * resolution is the reference's distances_[di][dj] * resolution_ bit for bit
// (grid_mapper.cpp:263,318), and comparing codes orders the heap exactly as comparing occ_dist does (x -> sqrt(x) *
// res is strictly increasing on the integers that occur), ties included.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <atomic>
#include <queue>
#include <thread>
#include <unordered_set>
#include <vector>

namespace tbnav {

class RefField {
 public:
  RefField(int n_particles, int xsize, int radius)
      : xs_(xsize), radius_(radius), sets_(n_particles), codes_(n_particles, std::vector<uint16_t>((size_t)xsize * xsize, 0xFFFF)) {}

  // fn(p, scratch) for p in [first, first + count) on up to `threads` host threads (particles are independent; each thread has
  // its own brushfire scratch).  threads <= 1: in the calling thread.
  struct Scratch { std::vector<uint8_t> marked; };
  template <class Fn>
  void for_each_particle(int first, int count, int threads, Fn fn) {
    const size_t G = (size_t)xs_ * xs_;
    if (threads > count) threads = count;
    if (threads <= 1) {
      Scratch sc; sc.marked.resize(G);
      for (int p = first; p < first + count; ++p) fn(p, sc);
      return;
    }
    std::atomic<int> next{first};
    auto work = [&] {
      Scratch sc; sc.marked.resize(G);
      for (int p = next.fetch_add(1); p < first + count; p = next.fetch_add(1)) fn(p, sc);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
  }

  int particles() const { return (int)sets_.size(); }
  const uint16_t* codes(int p) const { return codes_[p].data(); }
  size_t occupied(int p) const { return sets_[p].size(); }

  // updateCellHash (grid_mapper.cpp:480-546) for the logged changes of one scan, in the reference's call order.
  // ev: cell index, bit 31 set = the cell left the occupied state.
  void apply(int p, const int* ev, int n) {
    std::unordered_set<int>& occ = sets_[p];
    for (int q = 0; q < n; ++q) {
      const int idx = ev[q] & 0x7FFFFFFF;
      if (ev[q] < 0) { if (occ.find(idx) != occ.end()) occ.erase(idx); }
      else { if (occ.find(idx) == occ.end()) occ.insert(idx); }
    }
  }

  // The occupied set of a particle whose map was written from outside (tbnav_rbpf_set_log_odds): the history is
  // unknown, the cells go in in ascending order.
  void reset(int p, const std::vector<int>& cells_ascending) {
    sets_[p] = std::unordered_set<int>();
    for (int c : cells_ascending) sets_[p].insert(c);
  }
  // One particle of another filter, copied the way a GridMapper is copied (std::unordered_set's copy constructor keeps
  // the iteration order and bucket layout, so the copy behaves like the original from here on).
  void copy_slot(int p, const RefField& from, int q) {
    sets_[p] = std::unordered_set<int>(from.sets_[q]);  // copy-construct (as GridMapper's copy does), then move in
    codes_[p] = from.codes_[q];
  }
  void set_codes(int p, const uint16_t* codes) { codes_[p].assign(codes, codes + codes_[p].size()); }

  // euclideanSignedDistanceField, grid_mapper.cpp:333-435
  void brushfire(int p, Scratch& sc) {
    const std::unordered_set<int>& occ = sets_[p];
    if (occ.empty()) return;
    std::vector<uint16_t>& code = codes_[p];
    std::vector<uint8_t>& marked = sc.marked;
    std::fill(marked.begin(), marked.end(), 0);  // "std::vector<int> marked(xsize_ * ysize_)", :342
    std::vector<Node> store;
    store.reserve((size_t)xs_ * 8);
    std::priority_queue<Node, std::vector<Node>, Farther> Q(Farther(), std::move(store));
    for (int key : occ) {  // :348-362
      code[key] = 0;
      marked[key] = 1;
      const uint16_t ki = (uint16_t)(key / xs_), kj = (uint16_t)(key % xs_);
      Q.push(Node{0, ki, kj, ki, kj});
    }
    while (!Q.empty()) {  // :399-433: top, push the four neighbours, THEN pop
      const Node c = Q.top();
      if (c.i > 0) enqueue(c.i - 1, c.j, c.si, c.sj, Q, code, marked);
      if (c.j > 0) enqueue(c.i, c.j - 1, c.si, c.sj, Q, code, marked);
      if (c.i < xs_ - 1) enqueue(c.i + 1, c.j, c.si, c.sj, Q, code, marked);
      if (c.j < xs_ - 1) enqueue(c.i, c.j + 1, c.si, c.sj, Q, code, marked);
      Q.pop();
    }
  }
  void brushfire(int p) { Scratch sc; sc.marked.resize((size_t)xs_ * xs_); brushfire(p, sc); }

  // lowVarianceResampling's copies (particle_filter.cpp:495-499): push_back(copy) per slot, clear, copy-assign.
  void resample(const int* parent) {
    const int n = particles();
    std::vector<std::unordered_set<int>> tmp_sets;
    std::vector<std::vector<uint16_t>> tmp_codes;
    for (int m = 0; m < n; ++m) { tmp_sets.push_back(sets_[parent[m]]); tmp_codes.push_back(codes_[parent[m]]); }
    sets_.clear();
    sets_ = tmp_sets;
    codes_.clear();
    codes_ = tmp_codes;
  }

 private:
  // (12 bytes instead of the reference's 48-byte Cell: the heap is std::priority_queue — the same std::push_heap / std::pop_heap
  //  sequence over the same comparison results, hence the same order among equal distances — it just moves a quarter of the bytes)
  struct Node { uint32_t d2; uint16_t i, j, si, sj; };
  struct Farther { bool operator()(const Node& a, const Node& b) const { return a.d2 > b.d2; } };  // CompareDistance, grid_mapper.hpp:104-110

  // enqueueCell, grid_mapper.cpp:272-329
  void enqueue(int i, int j, int si, int sj, std::priority_queue<Node, std::vector<Node>, Farther>& Q, std::vector<uint16_t>& code,
               std::vector<uint8_t>& marked) {
    const int idx = i * xs_ + j;
    if (marked[idx]) return;
    const int di = std::abs(i - si), dj = std::abs(j - sj);
    if (di >= radius_ || dj >= radius_) return;  // distances_ is cell_radius_ x cell_radius_: .at() throws, caught, return (:300-308)
    const int d2 = di * di + dj * dj;
    if (d2 > radius_ * radius_) return;          // dist > cell_radius_ (:311-314); sqrt(d2) > r <=> d2 > r^2 exactly
    code[idx] = (uint16_t)d2;
    Q.push(Node{(uint32_t)d2, (uint16_t)i, (uint16_t)j, (uint16_t)si, (uint16_t)sj});
    marked[idx] = 1;
  }

  int xs_, radius_;
  std::vector<std::unordered_set<int>> sets_;
  std::vector<std::vector<uint16_t>> codes_;
};

}  // namespace tbnav
