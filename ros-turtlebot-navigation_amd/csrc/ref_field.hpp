// ref_field.hpp — the reference's distance field, reproduced on the host.
//
// bmapping::GridMapper::euclideanSignedDistanceField (bmapping/src/bmapping/grid_mapper.cpp:333-435, enqueueCell
// :272-329) is a multi-source brushfire over a std::priority_queue, seeded by iterating the std::unordered_set<int>
// of occupied cells.  Its result is NOT a function of the occupied set: it depends on the hash set's iteration
// order (i.e. its whole insert / erase history) and on the heap's handling of equal distances, it pops the wrong
// element when a neighbour pushed during an expansion is nearer than the current top (:401-431 read top, push,
// THEN pop), and cells that are no longer reached keep their old value (hard part 1).
//
// The device's default lookups compute the exact nearest-obstacle distance instead (rbpf_propose.hip, DistSrc).  This class
// is the third leg of SURVEY's contract: a mode in which the product reproduces the reference's field bit for bit
// — same libstdc++ containers, fed the same insert / erase sequence (the beam-ordered raycast kernel logs it),
// copied the way ParticleFilter::lowVarianceResampling copies particles (particle_filter.cpp:468-500) — so that
// an un-injected run matches the reference end to end.
//
// What keeps that affordable:
//   * particles are independent: distinct work is spread over host threads (the order of operations inside one set and
//     heap, which is what the result depends on, is untouched);
//   * the result is a FUNCTION of (set with its history, stale field, the scan's insert / erase sequence), and particles share
//     those far more often than not; a scan groups the particles by (state, event sequence — compared in full, not by hash)
//     and runs ONE copy + replay + brushfire per group (round 3);
//   * (round 6) the brushfire is LAZY.  enqueueCell returns at once for a marked cell and marks at the write
//     (grid_mapper.cpp:285-288,318,328): every cell is written exactly once per pass, and the sequence of heap operations is
//     deterministic.  A pass stopped between two iterations of :399-433 is therefore bit-identical to the completed pass on every
//     cell it has marked, and it can be resumed from its saved queue at any later time.  likelihoodFieldModel reads beam-endpoint
//     cells only (grid_mapper.cpp:101-128), which lie next to obstacles — among the first few thousand of a 400 x 400 pass's
//     157 k pops.  A state therefore keeps its queue, its marks and the list of cells it has written; a scan's pass stops once the
//     top of the queue is farther than `reach` cells; the device sees "not computed" (kPending) in every other cell and reports
//     a lookup that lands on one, upon which exactly that state is resumed (ensure()) and the proposal is run again.
//
// Stale cells.  A cell the completed pass does not reach keeps what an EARLIER pass left there (:311-314).  With truncated
// passes in the lineage that value is only known if the remainder of every earlier pass is known not to have written the cell.
// A state is `exact` when every cell outside its marks provably holds the reference's value: the pass wrote every cell of the
// grid, or the state descends from an exact complete state by complete passes.  Everything else is recovered on demand by
// REPLAYING the lineage's history: every state remembers the event lists since its last exact ancestor (Hist: a tree shared
// between lineages; a resampling prunes it), and replay() runs those generations' brushfires to completion.  That is the old
// eager cost, paid only when a lookup (or a whole-field export) really needs such a cell — never on a closed room.
//
// Distances are kept as u16 codes = squared distance in cells (0xFFFF = never reached = max_occ_dist_):
// sqrt((double)code) * resolution is the reference's distances_[di][dj] * resolution_ bit for bit
// (grid_mapper.cpp:263,318), and comparing codes orders the heap exactly as comparing occ_dist does (x -> sqrt(x) *
// res is strictly increasing on the integers that occur), ties included.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace tbnav {

// Host threads that stay: a scan hands them its groups (std::thread creation is ~25 us apiece, in sequence — 16 of them twice per scan
// was a tenth of a 10 ms scan).  run(n, f) runs f on n - 1 of them and on the caller, and returns when all are through.
class Workers {
 public:
  ~Workers() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  template <class F>
  void run(int n, F& f) {
    if (n <= 1) { f(); return; }
    {
      std::unique_lock<std::mutex> lk(mu_);
      while ((int)th_.size() < n - 1) th_.emplace_back([this, idx = (int)th_.size()] { loop(idx); });
      job_ = [&f] { f(); };
      want_ = n - 1; left_ = n - 1; ++gen_;
    }
    cv_.notify_all();
    f();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return left_ == 0; });
  }
 private:
  void loop(int idx) {
    unsigned long seen = 0;
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || (gen_ != seen && idx < want_); });
        if (stop_) return;
        seen = gen_;
        job = job_;
      }
      job();
      std::lock_guard<std::mutex> lk(mu_);
      if (--left_ == 0) done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> th_;
  std::function<void()> job_;
  int want_ = 0, left_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
};

class RefField {
 public:
  static constexpr uint16_t kUnreached = 0xFFFF;  // max_occ_dist_ (grid_mapper.cpp:49,58)
  static constexpr uint16_t kPending = 0xFFFE;    // device image only: "this pass has not got here" (the largest real code is radius^2 <= 40 000)
  static constexpr size_t kDenseLen = ~(size_t)0; // a slot that holds a state's whole field (exact complete states)

  // (public: tests/ref_field_check.cpp drives Heap against std::priority_queue)
  // The reference's queue is std::priority_queue<Cell, std::vector<Cell>, CompareDistance> (grid_mapper.hpp:104-110, 48-byte cells).
  // What its result depends on is the ORDER in which equal distances leave the heap, i.e. libstdc++'s std::push_heap / std::pop_heap
  // (bits/stl_heap.h: __push_heap, __adjust_heap) applied to the same comparison results.  Heap below is those two algorithms written
  // out over 16-byte nodes — the hole walks down to a leaf taking, at every level, the child the library takes (the right one unless
  // it is FARTHER than the left), then the displaced last element is pushed up from there — with the child chosen by arithmetic
  // instead of a data-dependent branch.  tests/test_ref_field_heap.py holds it against std::priority_queue itself (pop order of equal
  // keys, whole fields) on random sequences; the GPU suite holds the fields against the oracle's, which is pinned to the compiled reference.
  // One 8-byte word per node (round 6; 16 bytes before): squared distance in the top 16 bits — the ONLY part the heap compares, so
  // equal distances stay "not farther" as in CompareDistance — then the cell and its source, 12 bits each (grids up to 4096 x 4096;
  // the codes are u16, so a squared distance fits).
  struct Node {
    uint64_t v;
    static Node make(uint32_t d2, int i, int j, int si, int sj) {
      return Node{((uint64_t)d2 << 48) | ((uint64_t)(i & 0xFFF) << 36) | ((uint64_t)(j & 0xFFF) << 24) | ((uint64_t)(si & 0xFFF) << 12) | (uint64_t)(sj & 0xFFF)};
    }
    uint32_t d2() const { return (uint32_t)(v >> 48); }
    int i() const { return (int)((v >> 36) & 0xFFF); }
    int j() const { return (int)((v >> 24) & 0xFFF); }
    int si() const { return (int)((v >> 12) & 0xFFF); }
    int sj() const { return (int)(v & 0xFFF); }
  };
  static_assert(sizeof(Node) == 8, "one 8-byte move per level");
  static constexpr int kMaxSide = 4096;
  class Heap {
   public:
    // keep = false: a new queue in `store` (its capacity is reused); keep = true: the queue `store` already holds (a resumed pass)
    explicit Heap(std::vector<Node>& store, bool keep = false) : v_(store) { if (!keep) v_.clear(); }
    bool empty() const { return v_.empty(); }
    size_t size() const { return v_.size(); }
    const Node& top() const { return v_.front(); }
    // priority_queue::push = push_back + std::push_heap: __push_heap(first, len - 1, 0, value) with comp(parent, value) = parent.d2 > value.d2
    void push(const Node& value) {
      v_.push_back(value);
      Node* const a = v_.data();
      size_t hole = v_.size() - 1;
      while (hole > 0) {
        const size_t parent = (hole - 1) / 2;
        if (!(a[parent].d2() > value.d2())) break;
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
        child -= (size_t)(a[child].d2() > a[child - 1].d2());   // comp(first + secondChild, first + (secondChild - 1)): --secondChild
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
        if (!(a[parent].d2() > value.d2())) break;
        a[hole] = a[parent];
        hole = parent;
      }
      a[hole] = value;
      v_.pop_back();
    }
   private:
    std::vector<Node>& v_;
  };

  // A field of u16 codes that is cheap to copy from one generation to the next: a private WINDOW (the bounding box of everything the
  // lineage's passes have written since the background was made, a few thousand cells round a room's walls) over a shared, immutable
  // dense BACKGROUND.  A pass that writes outside the window grows it (values come over from the background); copying a field copies
  // the window and a pointer.  (A 400 x 400 field is 320 KB; 1000 new states per scan copied 320 MB of it — the host's memory
  // bandwidth, shared by the threads — to change a few hundred cells each.)
  class Codes {
   public:
    void assign_fill(int xs, uint16_t v) { xs_ = xs; bg_ = std::make_shared<std::vector<uint16_t>>((size_t)xs * xs, v); drop_window(); }
    void assign_dense(int xs, const uint16_t* src) { xs_ = xs; bg_ = std::make_shared<std::vector<uint16_t>>(src, src + (size_t)xs * xs); drop_window(); }
    void copy_from(const Codes& o) { xs_ = o.xs_; bg_ = o.bg_; i0_ = o.i0_; i1_ = o.i1_; j0_ = o.j0_; j1_ = o.j1_; w_ = o.w_; win_.assign(o.win_.begin(), o.win_.end()); }
    void swap(Codes& o) { std::swap(xs_, o.xs_); bg_.swap(o.bg_); std::swap(i0_, o.i0_); std::swap(i1_, o.i1_); std::swap(j0_, o.j0_); std::swap(j1_, o.j1_); std::swap(w_, o.w_); win_.swap(o.win_); }
    bool inside(int i, int j) const { return i >= i0_ && i <= i1_ && j >= j0_ && j <= j1_; }
    uint16_t get(int i, int j) const { return inside(i, j) ? win_[(size_t)(i - i0_) * w_ + (j - j0_)] : (*bg_)[(size_t)i * xs_ + j]; }
    uint16_t at(int cell) const { return get(cell / xs_, cell % xs_); }
    // the cell's slot in the window (grown to hold it if need be): read the old value, write the new one
    uint16_t* slot(int i, int j) {
      if (!inside(i, j)) grow(i, j);
      return &win_[(size_t)(i - i0_) * w_ + (j - j0_)];
    }
    void to_dense(std::vector<uint16_t>& out) const {
      out = *bg_;
      for (int i = i0_; i <= i1_; ++i) std::memcpy(&out[(size_t)i * xs_ + j0_], &win_[(size_t)(i - i0_) * w_], sizeof(uint16_t) * (size_t)w_);
    }
    size_t window_cells() const { return win_.size(); }
    void keep_capacity_only() { bg_.reset(); drop_window(); }
   private:
    void drop_window() { i0_ = j0_ = 0; i1_ = j1_ = -1; w_ = 0; win_.clear(); }
    void grow(int i, int j) {
      const int m = 16;   // (a pass spreads ring by ring: room for the next rings too)
      int a0 = std::max(0, i - m), a1 = std::min(xs_ - 1, i + m), b0 = std::max(0, j - m), b1 = std::min(xs_ - 1, j + m);
      if (i1_ >= i0_) { a0 = std::min(a0, i0_); a1 = std::max(a1, i1_); b0 = std::min(b0, j0_); b1 = std::max(b1, j1_); }
      const int nw = b1 - b0 + 1;
      std::vector<uint16_t> nwin((size_t)(a1 - a0 + 1) * nw);
      for (int r = a0; r <= a1; ++r) {
        uint16_t* dst = &nwin[(size_t)(r - a0) * nw];
        std::memcpy(dst, &(*bg_)[(size_t)r * xs_ + b0], sizeof(uint16_t) * (size_t)nw);
        if (r >= i0_ && r <= i1_) std::memcpy(dst + (j0_ - b0), &win_[(size_t)(r - i0_) * w_], sizeof(uint16_t) * (size_t)w_);
      }
      win_.swap(nwin);
      i0_ = a0; i1_ = a1; j0_ = b0; j1_ = b1; w_ = nw;
    }
    int xs_ = 0;
    std::shared_ptr<const std::vector<uint16_t>> bg_;
    int i0_ = 0, i1_ = -1, j0_ = 0, j1_ = -1, w_ = 0;
    std::vector<uint16_t> win_;
  };

  struct State;
  using StatePtr = std::shared_ptr<State>;

  // The occupied-set changes of the generations since a lineage's last exact ancestor (`base`, held by the root only), newest last.
  // A tree: lineages that split share their common past; nodes die with the last state that descends from them.
  struct Hist {
    std::shared_ptr<const Hist> parent;
    std::shared_ptr<const State> base;
    std::vector<int> events;
    std::shared_ptr<std::atomic<long long>> bytes;   // the owner's count of history bytes alive
    ~Hist() {
      if (bytes) bytes->fetch_sub((long long)(sizeof(Hist) + events.capacity() * sizeof(int)));
      // (a lineage's chain is as long as the run: unlink iteratively, or the destructors recurse once per scan)
      std::shared_ptr<const Hist> p = std::move(parent);
      while (p && p.use_count() == 1) { std::shared_ptr<const Hist> next = std::move(const_cast<Hist*>(p.get())->parent); p = std::move(next); }
    }
  };

  // What the device has to be told: (cell, code) pairs, code == kPending where a cell of the parent's image is not (yet) written
  struct JEntry { uint32_t cell, code; };

  // Dead states are handed to the next scan WHOLE (a scan makes up to N new states and drops as many): the field and the pass's
  // vectors keep their capacity — through malloc each is an mmap, ~100 page faults and a munmap — and the occupied set keeps its
  // thousand list nodes, which std::unordered_set's copy assignment re-uses (bits/hashtable.h, _M_assign_elements: same _M_assign as the
  // copy constructor, so the copy has the source's iteration order, bucket count and rehash-policy state either way).
  struct Pool {
    std::mutex mu;
    std::vector<State*> free_states;
    size_t keep = 64;
    ~Pool();
  };
  struct Recycle {
    std::shared_ptr<Pool> pool;
    void operator()(State* s) const;
  };
  StatePtr new_state() {
    State* raw = nullptr;
    {
      std::lock_guard<std::mutex> lk(pool_->mu);
      if (!pool_->free_states.empty()) { raw = pool_->free_states.back(); pool_->free_states.pop_back(); }
    }
    if (!raw) raw = new State();
    raw->id = next_id_++;
    return StatePtr(raw, Recycle{pool_});
  }

  // One particle's occupied set (with its iteration order: its whole history) and field (with its stale cells), and the pass that
  // produced the field — possibly stopped early.  Shared by the particles that are copies of one another; only the pass advances.
  struct State {
    uint64_t id = 0;
    std::unordered_set<int> occ;
    Codes code;                     // [G] the newest value any pass of the lineage has written (the reference's value in every cell of `mark`)
    std::vector<uint16_t> dense;    // the same as one array, made when somebody asks for the whole field (codes())
    bool dense_valid = false;       // ... and still what `code` holds
    bool fresh = false;             // code is what the brushfire leaves for exactly this set (false once either was written from outside)
    // the pass (euclideanSignedDistanceField for exactly this set)
    std::vector<uint64_t> mark;     // [G / 64] cell written by this pass (empty: no pass ran for this state)
    std::vector<int> band;          // the marked cells in the order they were written
    std::vector<Node> heap;         // the queue where the pass stands (empty: complete)
    bool complete = true;
    bool exact = true;              // the cells outside `mark` hold the reference's values too
    bool from_dense = true;         // derived from a state whose whole field was the reference's (exact and complete)
    bool hist_lost = false;         // the history budget was exceeded somewhere up the lineage: replay is impossible
    std::shared_ptr<const Hist> hist;  // generations since the last exact ancestor (null: this state is exact, or the history is lost)
    // the device image: parent's image (of id from_id, at journal length from_len) + journal; reset: from "everything pending"
    uint64_t from_id = 0; size_t from_len = 0;
    bool j_reset = false;
    std::vector<JEntry> journal;
    bool marked(int c) const { return !mark.empty() && ((mark[(size_t)c >> 6] >> (c & 63)) & 1ull); }
    bool dense_image() const { return exact && complete; }
    State() = default;
    State(const State&) = delete;
    State& operator=(const State&) = delete;
    // back to "a new state" — the containers keep what they have allocated (the set its nodes: the next owner assigns over them)
    void recycle() {
      id = 0; fresh = false; complete = true; exact = true; from_dense = true; hist_lost = false; hist.reset();
      from_id = 0; from_len = 0; j_reset = false;
      mark.clear(); band.clear(); heap.clear(); journal.clear(); dense.clear(); dense_valid = false; code.keep_capacity_only();
    }
  };

  // (a state built by hand — tests, replay — is a plain object; the particles' states come from new_state())
  struct Counters {
    long long passes = 0;       // brushfires started (one per distinct (state, event sequence) of a scan)
    long long pops = 0;         // iterations of grid_mapper.cpp:399-433 run, all passes
    long long resumes = 0;      // states advanced because a lookup landed on a pending cell
    long long completions = 0;  // passes run to the end
    long long replays = 0;      // lineages replayed from their base (stale cells wanted)
    long long replay_generations = 0;
    long long us_group = 0, us_work = 0, us_bury = 0;   // step(): wall microseconds grouping the particles | in the passes | releasing the old states
    long long us_busy = 0;      // step(): microseconds the threads spent in the passes, summed over the threads (the CPU time a scan's fields cost)
  };

  RefField(int n_particles, int xsize, int radius) : xs_(xsize), radius_(radius), hist_bytes_(std::make_shared<std::atomic<long long>>(0)),
                                                     pool_(std::make_shared<Pool>()) {
    pool_->keep = (size_t)n_particles + 64;
    auto s = new_state();
    initial_id_ = s->id;
    s->code.assign_fill(xsize, kUnreached);
    st_.assign((size_t)n_particles, s);  // "N deep copies of the prototype" (particle_filter.cpp:125-138): equal, hence shared
    dev_.assign((size_t)n_particles, Slot{});
  }

  int particles() const { return (int)st_.size(); }
  size_t occupied(int p) const { return st_[p]->occ.size(); }
  const State* state(int p) const { return st_[p].get(); }
  // distinct states among the particles / brushfires run by the last step() (what the sharing saved: tests, bench)
  int distinct_states() const { std::unordered_set<const State*> u; for (auto& s : st_) u.insert(s.get()); return (int)u.size(); }
  int last_step_brushfires() const { return last_brushfires_; }
  long long last_step_busy_us() const { return last_busy_us_; }
  long long total_brushfires() const { return total_brushfires_; }
  const Counters& counters() const { return cnt_; }
  long long history_bytes() const { return hist_bytes_->load(); }
  // reach: how far (cells) a scan's pass runs before it stops — 0: to the end, as up to round 5 (every state complete and exact)
  void set_reach(int cells) { reach_ = cells < 0 ? 0 : cells; }
  int reach() const { return reach_; }
  void set_history_budget(long long bytes) { hist_budget_ = bytes; }

  // The whole field of particle p as the reference holds it (export hooks, tests): finishes the pass and, where stale cells are not
  // known, replays the lineage.  nullptr: the history needed for that is gone (budget) — nothing is returned rather than a guess.
  const uint16_t* codes(int p) {
    State& s = *st_[p];
    if (!s.complete) { advance(s, ~0u, -1); }
    if (!s.exact && make_exact(s) != 0) return nullptr;
    if (!s.dense_valid) { s.code.to_dense(s.dense); s.dense_valid = true; }
    return s.dense.data();
  }

  // One scan for particles [first, first + count): evs[i] = the logged set changes of particle first + i in the reference's
  // call order (cell index, bit 31 set = the cell left the occupied state) -> updateCellHash (grid_mapper.cpp:480-546) for each,
  // then euclideanSignedDistanceField — up to `reach` cells.  One replay + brushfire per distinct (state, sequence), on up to
  // `threads` host threads.  (all / off: the particles' sequences back to back, particle first + i's at all[off[i]] .. all[off[i + 1]])
  void step(int first, int count, int threads, const int* all, const size_t* off) {
    struct Group { StatePtr from; const int* ev; size_t n; std::vector<int> members; StatePtr to; bool steal = false; };
    const auto t_a = std::chrono::steady_clock::now();
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
      if (g < 0) { g = (int)groups.size(); groups.push_back(Group{st_[p], ev, n, {}, nullptr, false}); cand.push_back(g); }
      groups[g].members.push_back(p);
    }
    for (Group& gr : groups) if (!(gr.n == 0 && (gr.from->occ.empty() || gr.from->fresh))) {
      gr.to = new_state();
      // every reference to the parent is one of this group's members' (or the group's own): it dies when they move on
      gr.steal = !gr.from->dense_image() && (size_t)gr.from.use_count() == gr.members.size() + 1;
    }
    std::atomic<int> next{0}, fires{0};
    std::atomic<long long> pops{0}, done{0};
    const uint32_t limit2 = reach_ > 0 ? (uint32_t)reach_ * (uint32_t)reach_ : ~0u;
    const bool keep_history = hist_bytes_->load() <= hist_budget_;
    std::atomic<long long> busy{0};
    auto work = [&] {
      long long my_pops = 0, my_done = 0;
      const auto t_in = std::chrono::steady_clock::now();
      struct Busy { std::atomic<long long>& acc; std::chrono::steady_clock::time_point t0;
                    ~Busy() { acc.fetch_add((long long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count()); } } busy_guard{busy, t_in};
      for (int g = next.fetch_add(1); g < (int)groups.size(); g = next.fetch_add(1)) {
        Group& gr = groups[g];
        // no set operation at all: nothing occupied (the reference returns at :338), or the same set in the same order whose
        // brushfire the field already is — every reached cell would get the value it has, every other keeps it
        if (!gr.to) { gr.to = gr.from; continue; }
        const State& P = *gr.from;
        State& s = *gr.to;
        // (copy-construct, as GridMapper's copy does: std::unordered_set's copy keeps the iteration order, the bucket layout and
        //  the rehash policy's state, so the copy behaves like the original from here on.  Where the parent dies with this step —
        //  nobody but this group's members holds it, and it is not an exact state a lineage's history may start from — its set is
        //  MOVED instead: the very object the reference would have gone on using, and a thousand list nodes not walked)
        if (gr.steal) s.occ = std::move(const_cast<State&>(P).occ);
        else s.occ = P.occ;
        s.code.copy_from(P.code);
        apply(s.occ, gr.ev, (int)gr.n);
        s.fresh = true;
        s.from_dense = P.dense_image();
        s.from_id = P.id; s.from_len = P.dense_image() ? kDenseLen : P.journal.size();
        s.j_reset = s.from_dense;
        s.hist_lost = P.hist_lost;
        if (!s.hist_lost) {
          if (!keep_history && !s.from_dense) s.hist_lost = true;
          else {
            auto hnode = std::make_shared<Hist>();
            hnode->events.assign(gr.ev, gr.ev + gr.n);
            if (s.from_dense) hnode->base = gr.from; else hnode->parent = P.hist;
            hnode->bytes = hist_bytes_;
            hist_bytes_->fetch_add((long long)(sizeof(Hist) + hnode->events.capacity() * sizeof(int)));
            s.hist = std::move(hnode);
          }
        }
        if (s.occ.empty()) {
          // (:335-338: nothing to do; every cell keeps what it has)
          s.complete = true; s.exact = s.from_dense;
          if (!s.from_dense) for (int c : P.band) s.journal.push_back(JEntry{(uint32_t)c, kPending});
        } else {
          s.complete = false; s.exact = false;
          s.mark.assign((G() + 63) / 64, 0ull);
          seed(s, P.dense_image() ? nullptr : &P);
          my_pops += advance(s, limit2, -1, P.dense_image() ? nullptr : &P);
          if (!s.from_dense) for (int c : P.band) if (!s.marked(c)) s.journal.push_back(JEntry{(uint32_t)c, kPending});
          if (s.complete) ++my_done;
          fires.fetch_add(1);
        }
        if (s.dense_image()) s.hist.reset();
      }
      pops.fetch_add(my_pops); done.fetch_add(my_done);
    };
    const auto t_b = std::chrono::steady_clock::now();
    run_threads(threads, (int)groups.size(), work);
    const auto t_c = std::chrono::steady_clock::now();
    // the particles move to their new states; the old ones go back to the pool
    std::vector<StatePtr> grave;
    grave.reserve((size_t)count);
    for (Group& gr : groups) { for (int p : gr.members) { grave.push_back(std::move(st_[p])); st_[p] = gr.to; } gr.from.reset(); gr.to.reset(); }
    grave.clear();   // (recycled, not destroyed: see Pool)
    const auto t_d = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    cnt_.us_group += us(t_a, t_b); cnt_.us_work += us(t_b, t_c); cnt_.us_bury += us(t_c, t_d);
    cnt_.us_busy += busy.load(); last_busy_us_ = busy.load();
    last_brushfires_ = fires.load();
    total_brushfires_ += last_brushfires_;
    cnt_.passes += last_brushfires_; cnt_.pops += pops.load(); cnt_.completions += done.load();
  }

  // Lookups of particles ps[0..n) landed on pending cells cs[0..n) (one cell reported per particle): resume exactly those states
  // until the cells are written — and a little beyond, the neighbouring beams' cells are about as far — or their passes end; a
  // cell the finished pass has not written needs the lineage's stale value: replay.  0: done (flush and look again);
  // -1: a stale cell is wanted whose history is gone (budget); -2: a replay disagreed with the pass it re-ran (internal error).
  int ensure(const int* ps, const int* cs, int n, int threads) {
    std::unordered_map<State*, std::vector<int>> want;
    for (int q = 0; q < n; ++q) want[st_[ps[q]].get()].push_back(cs[q]);
    std::vector<std::pair<State*, std::vector<int>*>> jobs;
    for (auto& kv : want) jobs.emplace_back(kv.first, &kv.second);
    std::atomic<int> next{0}, bad{0};
    std::atomic<long long> pops{0}, done{0};
    auto work = [&] {
      for (int j = next.fetch_add(1); j < (int)jobs.size(); j = next.fetch_add(1)) {
        State& s = *jobs[j].first;
        for (int c : *jobs[j].second) {
          if (s.marked(c) || s.dense_image()) continue;
          if (!s.complete) {
            pops.fetch_add(advance(s, ~0u, c));
            if (s.marked(c) && !s.complete) {   // one more ring: (sqrt(d2) + 1)^2
              const uint32_t d2 = s.code.at(c);
              const uint32_t r1 = (uint32_t)std::ceil(std::sqrt((double)d2)) + 1u;
              pops.fetch_add(advance(s, r1 * r1, -1));
            }
            if (s.complete) done.fetch_add(1);
          }
          if (!s.marked(c) && s.complete && !s.exact) {
            const int rc = make_exact(s);
            if (rc != 0) { bad.store(rc); continue; }
          }
        }
        if (s.dense_image()) s.hist.reset();
      }
    };
    run_threads(threads, (int)jobs.size(), work);
    cnt_.resumes += (long long)jobs.size(); cnt_.pops += pops.load(); cnt_.completions += done.load();
    return bad.load();
  }

  // ---- what the device's copy of the fields needs: per slot, either nothing, a range of journal entries (shared by the slots of
  // one state), or a whole image.  The device image of a state: its marked cells' codes, kPending elsewhere — the whole field once
  // the state is exact and complete.
  struct Flush {
    struct Job { uint32_t off = 0, count = 0, reset = 0; };
    std::vector<JEntry> entries;   // packed ranges
    std::vector<Job> jobs;         // [N]; count == 0 and reset == 0: nothing to do
    std::vector<int> dense_slot;   // slots that take a whole image ...
    std::vector<int> dense_src;    // ... from the image uploaded for slot dense_src (itself: upload images[dense_img])
    std::vector<int> dense_img;
    std::vector<std::vector<uint16_t>> images;
    bool any_job = false;
  };
  void plan_flush(Flush& f) {
    const int N = particles();
    f.entries.clear(); f.jobs.assign((size_t)N, Flush::Job{}); f.dense_slot.clear(); f.dense_src.clear(); f.dense_img.clear(); f.images.clear(); f.any_job = false;
    struct Key { uint64_t id; size_t start; bool operator==(const Key& o) const { return id == o.id && start == o.start; } };
    struct KeyHash { size_t operator()(const Key& k) const { return (size_t)(k.id * 0x9E3779B97F4A7C15ull) ^ k.start; } };
    std::unordered_map<Key, Flush::Job, KeyHash> packed;
    std::unordered_map<uint64_t, int> dense_first;   // state id -> the slot its image was uploaded to
    for (int p = 0; p < N; ++p) {
      State& T = *st_[p];
      Slot& d = dev_[p];
      if (T.dense_image()) {
        if (d.id == T.id && d.len == kDenseLen) continue;
        need_dense(f, dense_first, p, T);
        d = Slot{T.id, kDenseLen};
        continue;
      }
      size_t start;
      bool reset = false;
      if (d.id == T.id && d.len != kDenseLen && d.len <= T.journal.size()) start = d.len;
      else if (d.id != 0 && d.id == T.from_id && d.len == T.from_len) { start = 0; reset = T.j_reset; }
      else { need_dense(f, dense_first, p, T); d = Slot{T.id, T.journal.size()}; continue; }
      if (start < T.journal.size() || reset) {
        auto it = packed.find(Key{T.id, start});
        if (it == packed.end()) {
          Flush::Job job;
          job.off = (uint32_t)f.entries.size(); job.reset = reset ? 1u : 0u;
          for (size_t q = start; q < T.journal.size(); ++q) {
            const JEntry e = T.journal[q];
            if (e.code == kPending && T.marked((int)e.cell)) continue;   // (written since: the later entry carries its code)
            f.entries.push_back(e);
          }
          job.count = (uint32_t)f.entries.size() - job.off;
          it = packed.emplace(Key{T.id, start}, job).first;
        }
        f.jobs[p] = it->second;
        f.any_job = f.any_job || it->second.count || it->second.reset;
      }
      d = Slot{T.id, T.journal.size()};
    }
  }
  void forget_slot(int p) { dev_[p] = Slot{}; }
  void forget_all_slots() { for (auto& d : dev_) d = Slot{}; }
  // the device's slots were allocated holding kUnreached everywhere: the initial state's image
  void slots_hold_initial_image() { for (size_t p = 0; p < dev_.size(); ++p) if (st_[p]->id == initial_id_) dev_[p] = Slot{initial_id_, kDenseLen}; }

  // The occupied set of a particle whose map was written from outside (tbnav_rbpf_set_log_odds): the history is
  // unknown, the cells go in in ascending order.
  void reset(int p, const std::vector<int>& cells_ascending) {
    const uint16_t* old = codes(p);
    auto s = new_state();
    s->occ = std::unordered_set<int>();   // (a NEW set — one bucket, growing as the cells go in — not a cleared one that keeps its bucket count)
    for (int c : cells_ascending) s->occ.insert(c);
    if (old) s->code.assign_dense(xs_, old); else s->code.assign_fill(xs_, kUnreached);
    s->fresh = false;
    st_[p] = s;
  }
  // One particle of another filter, copied the way a GridMapper is copied: the set with its history, the field with its stale cells.
  // (a copy of its own: the two filters may be driven from two threads, and a state's pass advances)
  void copy_slot(int p, RefField& from, int q) {
    const uint16_t* c = from.codes(q);   // (finishes the pass there; replays if stale cells are unknown)
    const State& o = *from.st_[q];
    auto s = new_state();
    s->occ = o.occ;
    s->fresh = o.fresh;
    if (c) s->code.assign_dense(xs_, c);
    else {   // the source's history is gone: the copy inherits what is known, and the hole
      s->code.copy_from(o.code); s->mark = o.mark; s->band = o.band; s->heap = o.heap; s->complete = o.complete; s->exact = false; s->from_dense = false; s->hist_lost = true;
    }
    st_[p] = s;
    dev_[p] = Slot{};
  }
  void set_codes(int p, const uint16_t* codes_in) {
    auto s = new_state();
    s->occ = st_[p]->occ;
    s->code.assign_dense(xs_, codes_in);
    s->fresh = false;
    st_[p] = s;
  }

  // lowVarianceResampling's copies (particle_filter.cpp:495-499): every slot becomes a copy of its parent — on the device too
  // (the handle's gather moves the field slots the same way).
  void resample(const int* parent) {
    const int n = particles();
    std::vector<StatePtr> tmp((size_t)n);
    std::vector<Slot> dtmp((size_t)n);
    for (int m = 0; m < n; ++m) { tmp[m] = st_[parent[m]]; dtmp[m] = dev_[parent[m]]; }
    st_.swap(tmp); dev_.swap(dtmp);
  }

  // The device image of particle p's state as it stands (tests; the dense path of plan_flush)
  void image(int p, std::vector<uint16_t>& out) const { image_of(*st_[p], out); }

 private:
  struct Slot { uint64_t id = 0; size_t len = 0; };   // the state (and how much of its journal) the device slot holds; id 0: unknown
  size_t G() const { return (size_t)xs_ * xs_; }

  static void image_of(const State& s, std::vector<uint16_t>& out) {
    if (s.dense_image()) { s.code.to_dense(out); return; }
    s.code.to_dense(out);   // (for its size)
    std::fill(out.begin(), out.end(), kPending);
    for (int c : s.band) out[c] = s.code.at(c);
  }
  void need_dense(Flush& f, std::unordered_map<uint64_t, int>& first, int p, const State& T) {
    auto it = first.find(T.id);
    f.dense_slot.push_back(p);
    if (it == first.end()) {
      first.emplace(T.id, p);
      f.dense_src.push_back(p);
      f.dense_img.push_back((int)f.images.size());
      f.images.emplace_back();
      image_of(T, f.images.back());
    } else { f.dense_src.push_back(it->second); f.dense_img.push_back(-1); }
  }

  template <class Work>
  void run_threads(int threads, int jobs, Work& work) {
    int nt = threads < 1 ? 1 : threads;
    if (nt > jobs) nt = jobs;
    if (nt <= 1) { if (jobs > 0) work(); return; }
    workers_.run(nt, work);
  }

  // updateCellHash (grid_mapper.cpp:480-546) for the logged changes of one scan, in the reference's call order
  static void apply(std::unordered_set<int>& occ, const int* ev, int n) {
    for (int q = 0; q < n; ++q) {
      const int idx = ev[q] & 0x7FFFFFFF;
      if (ev[q] < 0) { if (occ.find(idx) != occ.end()) occ.erase(idx); }
      else { if (occ.find(idx) == occ.end()) occ.insert(idx); }
    }
  }

  // A cell's code goes to the device unless the slot is known to hold it already: the parent's image has the cell written with
  // this very code (the child's `code` starts as the parent's).
  static bool parent_has(const State* P, int idx, uint16_t d2, uint16_t old) { return P && old == d2 && P->marked(idx); }

  // euclideanSignedDistanceField, grid_mapper.cpp:348-362: the sources, in the set's iteration order
  void seed(State& st, const State* P) const {
    uint64_t* const mk = st.mark.data();
    if (st.heap.capacity() < (size_t)xs_ * 8) st.heap.reserve((size_t)xs_ * 8);
    Heap Q(st.heap);
    for (int key : st.occ) {
      const uint16_t ki = (uint16_t)(key / xs_), kj = (uint16_t)(key % xs_);
      uint16_t* const cs = st.code.slot(ki, kj);
      if (!parent_has(P, key, 0, *cs)) st.journal.push_back(JEntry{(uint32_t)key, 0u});
      *cs = 0;
      mk[(size_t)key >> 6] |= 1ull << (key & 63);
      st.band.push_back(key);
      Q.push(Node::make(0, ki, kj, ki, kj));
    }
  }

  // grid_mapper.cpp:399-433 from where the pass stands: top, push the four neighbours, THEN pop — until the queue is empty
  // (complete), its top is farther than limit2, or cell `until` (>= 0) has been written.  Returns the iterations run.
  long long advance(State& st, uint32_t limit2, int until, const State* P = nullptr) const {
    if (st.complete) return 0;
    st.dense_valid = false;
    Codes& code = st.code;
    uint64_t* const mk = st.mark.data();
    Heap Q(st.heap, true);
    const int xs = xs_, r = radius_, r2 = radius_ * radius_;
    std::vector<int>& band = st.band;
    std::vector<JEntry>& jr = st.journal;
    // enqueueCell, grid_mapper.cpp:272-329
    auto enqueue = [&](int i, int j, int si, int sj) {
      const int idx = i * xs + j;
      uint64_t& w = mk[(size_t)idx >> 6];
      const uint64_t bit = 1ull << (idx & 63);
      if (w & bit) return;
      const int di = std::abs(i - si), dj = std::abs(j - sj);
      if (di >= r || dj >= r) return;  // distances_ is cell_radius_ x cell_radius_: .at() throws, caught, return (:300-308)
      const int d2 = di * di + dj * dj;
      if (d2 > r2) return;             // dist > cell_radius_ (:311-314); sqrt(d2) > r <=> d2 > r^2 exactly
      uint16_t* const cs = code.slot(i, j);
      if (!parent_has(P, idx, (uint16_t)d2, *cs)) jr.push_back(JEntry{(uint32_t)idx, (uint32_t)d2});
      *cs = (uint16_t)d2;
      Q.push(Node::make((uint32_t)d2, i, j, si, sj));
      w |= bit;
      band.push_back(idx);
    };
    long long it = 0;
    const uint64_t* uw = until >= 0 ? &mk[(size_t)until >> 6] : nullptr;
    const uint64_t ubit = until >= 0 ? 1ull << (until & 63) : 0ull;
    while (!Q.empty()) {
      const Node c = Q.top();
      if (c.d2() > limit2) return it;
      if (uw && (*uw & ubit)) return it;
      const int ci = c.i(), cj = c.j(), si = c.si(), sj = c.sj();
      if (ci > 0) enqueue(ci - 1, cj, si, sj);
      if (cj > 0) enqueue(ci, cj - 1, si, sj);
      if (ci < xs - 1) enqueue(ci + 1, cj, si, sj);
      if (cj < xs - 1) enqueue(ci, cj + 1, si, sj);
      Q.pop();
      ++it;
    }
    st.complete = true;
    // every cell of the grid written by this pass, or a complete pass over an exact complete field: nothing stale is unknown
    if (band.size() == G() || st.from_dense) { st.exact = true; }
    return it;
  }

  // The stale cells of a complete state: run the lineage again from its base, every pass to the end (the eager algorithm).
  int make_exact(State& s) {
    if (s.exact) return 0;
    if (!s.complete) advance(s, ~0u, -1);
    if (s.exact) { s.hist.reset(); return 0; }
    if (!s.hist) return -1;
    std::vector<const Hist*> chain;
    for (const Hist* h = s.hist.get(); h; h = h->parent.get()) chain.push_back(h);
    const State* base = chain.back()->base.get();
    if (!base) return -1;
    State cur;
    cur.occ = base->occ;
    cur.code.copy_from(base->code);
    long long pops = 0;
    for (size_t g = chain.size(); g-- > 0;) {
      apply(cur.occ, chain[g]->events.data(), (int)chain[g]->events.size());
      cur.band.clear();
      if (cur.occ.empty()) continue;
      cur.mark.assign((G() + 63) / 64, 0ull);
      cur.journal.clear();
      cur.complete = false; cur.from_dense = true;
      seed(cur, nullptr);
      pops += advance(cur, ~0u, -1);
    }
    // the last generation's pass IS this state's pass: same set, same order — it must have written the same cells the same way
    if (cur.occ.size() != s.occ.size() || cur.band.size() != s.band.size()) return -2;
    for (int c : s.band) if (cur.code.at(c) != s.code.at(c)) return -2;
    s.code.swap(cur.code);
    s.dense_valid = false;
    s.exact = true;
    s.hist.reset();
    std::lock_guard<std::mutex> lk(cnt_mu_);
    cnt_.replays += 1; cnt_.replay_generations += (long long)chain.size(); cnt_.pops += pops;
    return 0;
  }

  int xs_, radius_;
  int reach_ = 1;
  long long hist_budget_ = (long long)1 << 30;
  int last_brushfires_ = 0;
  long long last_busy_us_ = 0;
  long long total_brushfires_ = 0;
  Counters cnt_;
  std::mutex cnt_mu_;
  std::atomic<uint64_t> next_id_{1};
  uint64_t initial_id_ = 0;
  std::shared_ptr<std::atomic<long long>> hist_bytes_;
  std::shared_ptr<Pool> pool_;
  Workers workers_;
  std::vector<StatePtr> st_;
  std::vector<Slot> dev_;
};

inline RefField::Pool::~Pool() { for (State* s : free_states) delete s; }
inline void RefField::Recycle::operator()(State* s) const {
  s->recycle();
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    if (pool->free_states.size() < pool->keep) { pool->free_states.push_back(s); return; }
  }
  delete s;
}

}  // namespace tbnav
