// ref_field.hpp — the reference's distance field, reproduced on the host.
//
// bmapping::GridMapper::euclideanSignedDistanceField (bmapping/src/bmapping/grid_mapper.cpp:333-435, enqueueCell
// :272-329) is a multi-source brushfire over a std::priority_queue, seeded by iterating the std::unordered_set<int>
// of occupied cells.  Its result is NOT a function of the occupied set: it depends on the hash set's iteration
// order (i.e. its whole insert / erase history) and on the heap's handling of equal distances, it pops the wrong
// element when a neighbour pushed during an expansion is nearer than the current top (:401-431 read top, push,
// THEN pop), and cells that are no longer reached keep their old value (hard part 1).
//
// The device's default lookups compute the exact nearest-obstacle distance instead (rbpf_device.hpp, DistSrc).  This class
// is the third leg of SURVEY's contract: a mode in which the product reproduces the reference's field bit for bit
// — same libstdc++ containers, fed the same insert / erase sequence (the beam-ordered raycast kernel logs it),
// copied the way ParticleFilter::lowVarianceResampling copies particles (particle_filter.cpp:468-500) — so that
// an un-injected run matches the reference end to end.  One brushfire is serial host work (~0.5 ms at 80 x 80,
// ~16 ms at 400 x 400).  Two things keep that affordable (round 3):
//   * particles are independent: distinct work is spread over host threads (the order of operations inside one set and
//     heap, which is what the result depends on, is untouched);
//   * the result is a FUNCTION of (set with its history, stale field, the scan's insert / erase sequence), and particles share
//     those far more often than not: every particle starts from the same empty map, the shipped sampling spread is
//     1e-8 m (slam.launch:24-26) — the same beams end in the same cells — and a resample makes copies.  A particle's state is
//     therefore an immutable, shared object; a scan groups the particles by (state, event sequence — compared in full, not
//     by hash) and runs ONE copy + replay + brushfire per group, exactly what each member's own would have been.  Particles
//     that have diverged (another cell somewhere) are their own group and pay for themselves as before.
//
// Distances are kept as u16 codes = squared distance in cells (0xFFFF = never reached = max_occ_dist_):
// sqrt((double)code) * resolution is the reference's distances_[di][dj] * resolution_ bit for bit
// (grid_mapper.cpp:263,318), and comparing codes orders the heap exactly as comparing occ_dist does (x -> sqrt(x) *
// res is strictly increasing on the integers that occur), ties included.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <queue>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace tbnav {

class RefField {
 public:
  // One particle's occupied set (with its iteration order: its whole history) and field (with its stale cells).  Never
  // modified once a particle points at it.
  struct State {
    std::unordered_set<int> occ;
    std::vector<uint16_t> code;
    bool fresh = false;  // code is what the brushfire left for exactly this set (false once either was written from outside)
  };
  using StatePtr = std::shared_ptr<const State>;

  RefField(int n_particles, int xsize, int radius) : xs_(xsize), radius_(radius) {
    auto s = std::make_shared<State>();
    s->code.assign((size_t)xsize * xsize, 0xFFFF);
    st_.assign((size_t)n_particles, s);  // "N deep copies of the prototype" (particle_filter.cpp:125-138): equal, hence shared
  }

  int particles() const { return (int)st_.size(); }
  const uint16_t* codes(int p) const { return st_[p]->code.data(); }
  size_t occupied(int p) const { return st_[p]->occ.size(); }
  const State* state(int p) const { return st_[p].get(); }
  StatePtr state_ptr(int p) const { return st_[p]; }
  // distinct states among the particles / brushfires run by the last step() (what the sharing saved: tests, bench)
  int distinct_states() const { std::unordered_set<const State*> u; for (auto& s : st_) u.insert(s.get()); return (int)u.size(); }
  int last_step_brushfires() const { return last_brushfires_; }
  long long total_brushfires() const { return total_brushfires_; }

  // One scan for particles [first, first + count): evs[i] = the logged set changes of particle first + i in the reference's
  // call order (cell index, bit 31 set = the cell left the occupied state) -> updateCellHash (grid_mapper.cpp:480-546) for each,
  // then euclideanSignedDistanceField.  One replay + brushfire per distinct (state, sequence), on up to `threads` host threads.
  // (all / off: the particles' sequences back to back, particle first + i's at all[off[i]] .. all[off[i + 1]])
  void step(int first, int count, int threads, const int* all, const size_t* off) {
    struct Group { StatePtr from; const int* ev; size_t n; std::vector<int> members; StatePtr to; };
    std::vector<Group> groups;
    std::unordered_map<uint64_t, std::vector<int>> by_hash;  // hash -> indices into groups
    for (int i = 0; i < count; ++i) {
      const int p = first + i;
      const int* ev = all + off[i];
      const size_t n = off[i + 1] - off[i];
      uint64_t hsh = 1469598103934665603ull ^ (uint64_t)(uintptr_t)st_[p].get();
      for (size_t q = 0; q < n; ++q) { hsh ^= (uint32_t)ev[q]; hsh *= 1099511628211ull; }
      std::vector<int>& cand = by_hash[hsh];
      int g = -1;
      for (int c : cand)
        if (groups[c].from.get() == st_[p].get() && groups[c].n == n && (n == 0 || std::memcmp(groups[c].ev, ev, sizeof(int) * n) == 0)) { g = c; break; }
      if (g < 0) { g = (int)groups.size(); groups.push_back(Group{st_[p], ev, n, {}, nullptr}); cand.push_back(g); }
      groups[g].members.push_back(p);
    }
    std::atomic<int> next{0}, fires{0};
    const size_t G = (size_t)xs_ * xs_;
    auto work = [&] {
      std::vector<uint8_t> marked(G);
      std::vector<Node> store;   // the heap's storage, kept from one brushfire of this thread to the next
      for (int g = next.fetch_add(1); g < (int)groups.size(); g = next.fetch_add(1)) {
        Group& gr = groups[g];
        // no set operation at all: nothing occupied (the reference returns at :338), or the same set in the same order whose
        // brushfire the field already is — every reached cell would get the value it has, every other keeps it
        if (gr.n == 0 && (gr.from->occ.empty() || gr.from->fresh)) { gr.to = gr.from; continue; }
        // (copy-construct, as GridMapper's copy does: std::unordered_set's copy keeps the iteration order, the bucket layout and
        //  the rehash policy's state, so the copy behaves like the original from here on)
        auto s = std::make_shared<State>(*gr.from);
        apply(s->occ, gr.ev, (int)gr.n);
        brushfire(*s, marked, store);
        s->fresh = true;
        fires.fetch_add(1);
        gr.to = s;
      }
    };
    int nt = threads < 1 ? 1 : threads;
    if (nt > (int)groups.size()) nt = (int)groups.size();
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    for (Group& gr : groups) for (int p : gr.members) st_[p] = gr.to;
    last_brushfires_ = fires.load();
    total_brushfires_ += last_brushfires_;
  }

  // The occupied set of a particle whose map was written from outside (tbnav_rbpf_set_log_odds): the history is
  // unknown, the cells go in in ascending order.
  void reset(int p, const std::vector<int>& cells_ascending) {
    auto s = std::make_shared<State>();
    for (int c : cells_ascending) s->occ.insert(c);
    s->code = st_[p]->code;
    s->fresh = false;
    st_[p] = s;
  }
  // One particle of another filter, copied the way a GridMapper is copied: the set with its history, the field with its stale cells.
  void copy_slot(int p, const RefField& from, int q) { st_[p] = from.st_[q]; }
  void set_codes(int p, const uint16_t* codes) {
    auto s = std::make_shared<State>(*st_[p]);
    s->code.assign(codes, codes + s->code.size());
    s->fresh = false;
    st_[p] = s;
  }

  // lowVarianceResampling's copies (particle_filter.cpp:495-499): every slot becomes a copy of its parent.
  void resample(const int* parent) {
    const int n = particles();
    std::vector<StatePtr> tmp((size_t)n);
    for (int m = 0; m < n; ++m) tmp[m] = st_[parent[m]];
    st_.swap(tmp);
  }

  // (public: tests/ref_field_check.cpp drives Heap against std::priority_queue)
  // The reference's queue is std::priority_queue<Cell, std::vector<Cell>, CompareDistance> (grid_mapper.hpp:104-110, 48-byte cells).
  // What its result depends on is the ORDER in which equal distances leave the heap, i.e. libstdc++'s std::push_heap / std::pop_heap
  // (bits/stl_heap.h: __push_heap, __adjust_heap) applied to the same comparison results.  Heap below is those two algorithms written
  // out over 16-byte nodes — the hole walks down to a leaf taking, at every level, the child the library takes (the right one unless
  // it is FARTHER than the left), then the displaced last element is pushed up from there — with the child chosen by arithmetic
  // instead of a data-dependent branch (a 400 x 400 brushfire is 157 k pops through a heap of ~800 nodes, ten levels each: 8.0 -> 7.1 ms
  // on one core; the state's copy, 0.05 ms, and the cleared marks, 0.005 ms, are not where the time is).  tests/test_ref_field_heap.py holds it against std::priority_queue itself (pop order of equal
  // keys, whole fields) on random sequences; the GPU suite holds the fields against the oracle's, which is pinned to the compiled reference.
  struct Node { uint32_t d2; uint16_t i, j, si, sj; uint32_t pad; };
  static_assert(sizeof(Node) == 16, "one 16-byte move per level");
  class Heap {
   public:
    explicit Heap(std::vector<Node>& store) : v_(store) { v_.clear(); }
    bool empty() const { return v_.empty(); }
    const Node& top() const { return v_.front(); }
    // priority_queue::push = push_back + std::push_heap: __push_heap(first, len - 1, 0, value) with comp(parent, value) = parent.d2 > value.d2
    void push(const Node& value) {
      v_.push_back(value);
      Node* const a = v_.data();
      size_t hole = v_.size() - 1;
      while (hole > 0) {
        const size_t parent = (hole - 1) / 2;
        if (!(a[parent].d2 > value.d2)) break;
        a[hole] = a[parent];
        hole = parent;
      }
      a[hole] = value;
    }
    // priority_queue::pop = std::pop_heap + pop_back: value = last; last = first; __adjust_heap(first, 0, len - 1, value)
    void pop() {
      Node* const a = v_.data();
      const size_t len = v_.size() - 1;   // the heap that remains
      if (len == 0) { v_.pop_back(); return; }
      const Node value = a[len];
      size_t hole = 0, child = 0;
      const size_t inner = (len - 1) / 2;
      while (child < inner) {
        child = 2 * (child + 1);
        child -= (size_t)(a[child].d2 > a[child - 1].d2);   // comp(first + secondChild, first + (secondChild - 1)): --secondChild
        a[hole] = a[child];
        hole = child;
      }
      if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1];
        hole = child - 1;
      }
      while (hole > 0) {   // __push_heap(first, hole, 0, value)
        const size_t parent = (hole - 1) / 2;
        if (!(a[parent].d2 > value.d2)) break;
        a[hole] = a[parent];
        hole = parent;
      }
      a[hole] = value;
      v_.pop_back();
    }
   private:
    std::vector<Node>& v_;
  };

 private:
  // updateCellHash (grid_mapper.cpp:480-546) for the logged changes of one scan, in the reference's call order
  static void apply(std::unordered_set<int>& occ, const int* ev, int n) {
    for (int q = 0; q < n; ++q) {
      const int idx = ev[q] & 0x7FFFFFFF;
      if (ev[q] < 0) { if (occ.find(idx) != occ.end()) occ.erase(idx); }
      else { if (occ.find(idx) == occ.end()) occ.insert(idx); }
    }
  }

  // euclideanSignedDistanceField, grid_mapper.cpp:333-435
  void brushfire(State& st, std::vector<uint8_t>& marked, std::vector<Node>& store) const {
    const std::unordered_set<int>& occ = st.occ;
    if (occ.empty()) return;
    uint16_t* const code = st.code.data();
    std::fill(marked.begin(), marked.end(), 0);  // "std::vector<int> marked(xsize_ * ysize_)", :342
    uint8_t* const mk = marked.data();
    if (store.capacity() < (size_t)xs_ * 8) store.reserve((size_t)xs_ * 8);
    Heap Q(store);   // (the thread's storage: its capacity is kept from one brushfire to the next)
    for (int key : occ) {  // :348-362
      code[key] = 0;
      mk[key] = 1;
      const uint16_t ki = (uint16_t)(key / xs_), kj = (uint16_t)(key % xs_);
      Q.push(Node{0, ki, kj, ki, kj, 0});
    }
    const int xs = xs_, r = radius_, r2 = radius_ * radius_;
    // enqueueCell, grid_mapper.cpp:272-329
    auto enqueue = [&](int i, int j, int si, int sj) {
      const int idx = i * xs + j;
      if (mk[idx]) return;
      const int di = std::abs(i - si), dj = std::abs(j - sj);
      if (di >= r || dj >= r) return;  // distances_ is cell_radius_ x cell_radius_: .at() throws, caught, return (:300-308)
      const int d2 = di * di + dj * dj;
      if (d2 > r2) return;             // dist > cell_radius_ (:311-314); sqrt(d2) > r <=> d2 > r^2 exactly
      code[idx] = (uint16_t)d2;
      Q.push(Node{(uint32_t)d2, (uint16_t)i, (uint16_t)j, (uint16_t)si, (uint16_t)sj, 0});
      mk[idx] = 1;
    };
    while (!Q.empty()) {  // :399-433: top, push the four neighbours, THEN pop
      const Node c = Q.top();
      if (c.i > 0) enqueue(c.i - 1, c.j, c.si, c.sj);
      if (c.j > 0) enqueue(c.i, c.j - 1, c.si, c.sj);
      if (c.i < xs - 1) enqueue(c.i + 1, c.j, c.si, c.sj);
      if (c.j < xs - 1) enqueue(c.i, c.j + 1, c.si, c.sj);
      Q.pop();
    }
  }

  int xs_, radius_;
  int last_brushfires_ = 0;
  long long total_brushfires_ = 0;
  std::vector<StatePtr> st_;
};

}  // namespace tbnav
