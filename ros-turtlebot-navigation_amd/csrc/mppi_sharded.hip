// mppi_sharded.hip — sharded MPPI ensembles behind the same entry points: attaching a handle to a
// communicator, the direct exchange's set-up (host side; its kernels are in mppi_softmin.hip), the sharded tick, and the groups of
// handles one process drives (what controller::MPPI built with n_gpus > 1 holds).  The reference has no counterpart: its rollouts
// are one loop (mppi.cpp:81-109); the soft-min (mppi.cpp:112-126) is what the ranks' records are combined into.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "mppi_host.hpp"

using namespace tbnav_mh;

namespace { int direct_setup(tbnav_mppi* h); bool exchange_words_alloc(tbnav_mppi* h); }

int tbnav_mh::exchange_error(const tbnav_mppi* h) {
  if (!h->comm || !h->h_dx_err || !*h->h_dx_err) return TBNAV_OK;
  tbnav::last_hip_error_slot() = (*h->h_dx_err & 2) ? "sharded tick: a rank's own rollouts failed — the time steps that saw its records were not updated (latched; re-attach the communicator)"
                                                    : "direct exchange: a peer's records did not arrive in time (latched; re-attach the communicator)";
  return TBNAV_ERR_HIP;
}
void tbnav_mh::exchange_words_free(tbnav_mppi* h) {
  (void)hipFree(h->d_dx_dead); h->d_dx_dead = nullptr;
  if (h->h_dx_err) (void)hipHostFree(h->h_dx_err);
  h->h_dx_err = nullptr; h->d_dx_err = nullptr;
}

extern "C" {
int tbnav_mppi_attach_comm(tbnav_mppi* h, tbnav_comm* comm) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipDeviceSynchronize());
  int rc_detach = TBNAV_OK;
  if (h->comm && h->direct_on && tbnav::comm_is_multiprocess(h->comm)) {
    // detaching from a direct exchange is COLLECTIVE, like attaching: a peer's publish kernel may still be storing into this
    // rank's buffer (it can be one tick ahead) — every rank synchronises its device (above), then all meet here, and only then
    // are the peers' mappings closed and the buffer freed.  A communicator that fails here is reported (the return value); the
    // teardown still runs.  tbnav_mppi_destroy of an attached handle comes through here too.
    int mine = 1;
    std::vector<int> all((size_t)tbnav::comm_size(h->comm), 0);
    rc_detach = tbnav::comm_all_gather_host(h->comm, &mine, all.data(), sizeof(int));
  }
  direct_teardown(h);
  exchange_words_free(h);
  (void)hipFree(h->d_records_all);
  h->d_records_all = nullptr;
  h->comm = nullptr;
  ++h->cfg_epoch;
  if (!comm) { h->k0 = 0; h->k_global = (uint64_t)h->K; return rc_detach; }
  if (rc_detach != TBNAV_OK) return rc_detach;   // (the old attachment is gone either way; the new one is not made)
  if (tbnav_comm_device(comm) != h->device) return TBNAV_ERR_INVALID_ARG;
  const int P = tbnav::comm_size(comm), r = tbnav::comm_rank(comm);
  TBNAV_HIP(hipMalloc((void**)&h->d_records_all, sizeof(double) * (size_t)P * h->T * h->S * TBNAV_MPPI_REC));
  h->comm = comm;
  if (!exchange_words_alloc(h)) { const int rc = tbnav::hip_fail(hipErrorOutOfMemory, "exchange error words", __FILE__, __LINE__); (void)tbnav_mppi_attach_comm(h, nullptr); return rc; }
  // this shard's place in the ensemble's noise counter space (equal shards: every rank holds K rollouts)
  h->k0 = (uint64_t)r * (uint64_t)h->K;
  h->k_global = (uint64_t)P * (uint64_t)h->K;
  // ranks in separate processes of one node: the records can go straight into the peers' buffers (collective: every rank
  // of the communicator attaches, with the same option)
  if (tbnav::comm_is_multiprocess(comm) && h->direct_want) return direct_setup(h);
  return TBNAV_OK;
}

int tbnav_mppi_exchange_kind(const tbnav_mppi* h) { return !h ? -1 : (!h->comm ? 0 : (h->direct_on ? 2 : 1)); }

}  // extern "C"

namespace {
// one rank's tick: its rollouts and records (written in place into its slot of the gather buffer), the all-gather, the combine
int sharded_partials(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, const uint64_t* seed, uint64_t tick, void* stream) {
  double* mine = h->d_records_all + (size_t)tbnav::comm_rank(h->comm) * h->T * h->S * TBNAV_MPPI_REC;
  if (h->fail_next) { h->fail_next = false; tbnav::last_hip_error_slot() = "fault injection: this rank's rollouts failed"; return TBNAV_ERR_HIP; }
  return seed ? tbnav_mppi_shard_partials_rng(h, x0, *seed, tick, stream, mine) : tbnav_mppi_shard_partials(h, x0, d_duL, d_duR, stream, mine);
}
// this rank's freshly written records -> every rank's buffer; then wait for everybody's and unpack them into d_records_all
// this rank's freshly written records (its slot of d_records_all) -> every rank's buffer, under the next sequence number
int direct_publish(tbnav_mppi* h, hipStream_t st, bool withhold) {
  const int P = tbnav::comm_size(h->comm), me = tbnav::comm_rank(h->comm), n = h->T * h->S * TBNAV_MPPI_REC;
  const unsigned int seq = ++h->dx_seq;
  const double* mine = h->d_records_all + (size_t)me * n;
  const int bx = std::min(8, (n + 255) / 256);
  hipLaunchKernelGGL(mppi_direct_publish, dim3(bx, P), dim3(256), 0, st, mine, n, h->d_dx_peers, me, P, (int)(seq & 1u), seq, withhold ? 1 : 0);
  TBNAV_HIP(hipGetLastError());
  return TBNAV_OK;
}
// wait for everybody's records of the current sequence number and unpack them into d_records_all (the self-tests; the tick's
// combine polls for the words itself — one launch fewer)
int direct_collect(tbnav_mppi* h, hipStream_t st, unsigned long long budget) {
  const int P = tbnav::comm_size(h->comm), n = h->T * h->S * TBNAV_MPPI_REC;
  const int bc = (int)std::min<size_t>(64, ((size_t)P * n + 255) / 256);
  hipLaunchKernelGGL(mppi_direct_collect, dim3(bc), dim3(256), 0, st, h->d_dx, n, P, (int)(h->dx_seq & 1u), h->dx_seq, h->d_records_all, h->d_dx_err, budget);
  TBNAV_HIP(hipGetLastError());
  return TBNAV_OK;
}
// the tick's two halves on one member: rollouts + records + their publication; the combine that polls for everybody's
int direct_partials_and_publish(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, const uint64_t* seed, uint64_t tick, void* stream);
int direct_combine(tbnav_mppi* h, hipStream_t st) {
  const int P = tbnav::comm_size(h->comm);
  const DirectSrc ds{h->d_dx + (size_t)(h->dx_seq & 1u) * P * 2 * ((size_t)h->T * h->S * TBNAV_MPPI_REC), h->dx_budget, h->d_dx_err, h->d_dx_dead, h->dx_seq};
  return launch_combine(h, h->d_records_all, P, st, -1, &ds);
}
// the buffer, the error words and the table of peers of one member (host side of both set-ups)
bool exchange_words_alloc(tbnav_mppi* h) {
  if (h->h_dx_err) { *h->h_dx_err = 0; return hipMemset(h->d_dx_dead, 0, sizeof(int)) == hipSuccess; }
  const bool ok = hipHostMalloc((void**)&h->h_dx_err, sizeof(int), hipHostMallocMapped) == hipSuccess &&
                  hipHostGetDevicePointer((void**)&h->d_dx_err, h->h_dx_err, 0) == hipSuccess &&
                  hipMalloc((void**)&h->d_dx_dead, sizeof(int)) == hipSuccess && hipMemset(h->d_dx_dead, 0, sizeof(int)) == hipSuccess;
  if (h->h_dx_err) *h->h_dx_err = 0;
  return ok;
}
bool direct_alloc(tbnav_mppi* h) {
  const int P = tbnav::comm_size(h->comm), n = h->T * h->S * TBNAV_MPPI_REC;
  const size_t words = (size_t)2 * P * 2 * n;
  return exchange_words_alloc(h) &&
         hipExtMallocWithFlags((void**)&h->d_dx, sizeof(unsigned long long) * words, hipDeviceMallocFinegrained) == hipSuccess &&
         hipMemset(h->d_dx, 0, sizeof(unsigned long long) * words) == hipSuccess &&
         hipMalloc((void**)&h->d_dx_peers, sizeof(unsigned long long*) * P) == hipSuccess;
}
double direct_pattern(int q, int it, int j) {  // the self-tests' records: every bit in play
  unsigned long long z = 0x9E3779B97F4A7C15ull * (unsigned long long)(q * 1000003 + it * 7919 + j + 1);
  z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
  double d; std::memcpy(&d, &z, sizeof d); return d;
}

}  // namespace
void tbnav_mh::direct_teardown(tbnav_mppi* h) {
  if (!h) return;
  h->direct_on = false;
  // a fresh attachment starts from a zeroed buffer and tag 1 on every rank: the self-tests end at the first local failure, so
  // ranks may leave a set-up with different counts (round-3 advisor finding)
  h->dx_seq = 0;
  h->pub_pending = false;
  for (void* p : h->dx_opened) (void)hipIpcCloseMemHandle(p);
  h->dx_opened.clear();
  (void)hipFree(h->d_dx); h->d_dx = nullptr;
  (void)hipFree(h->d_dx_peers); h->d_dx_peers = nullptr;
  if (h->h_dx_err) { *h->h_dx_err = 0; if (h->d_dx_dead) (void)hipMemset(h->d_dx_dead, 0, sizeof(int)); }   // (the words themselves live as long as the attachment)
}
namespace {

// Called by tbnav_mppi_attach_comm on every rank of a multi-process communicator.  Every step that can fail on one rank is
// followed by an agreement (an all-gather of status words through the communicator), so that all ranks end in the same
// state: direct exchange on, or off (the communicator's all-gather carries the records) — never a mixture.
int direct_setup(tbnav_mppi* h) {
  const int P = tbnav::comm_size(h->comm), me = tbnav::comm_rank(h->comm), n = h->T * h->S * TBNAV_MPPI_REC;
  struct Hello { int ok; int pad; hipIpcMemHandle_t handle; };
  auto agree = [&](int mine_ok, bool& all_ok) {   // collective
    std::vector<int> all(P, 0);
    const int rc = tbnav::comm_all_gather_host(h->comm, &mine_ok, all.data(), sizeof(int));
    all_ok = rc == TBNAV_OK;
    for (int q = 0; q < P; ++q) all_ok = all_ok && all[q] == 1;
    return rc;
  };
  // 1. the buffer (fine-grained: written by other devices while kernels of this one poll it), its IPC handle
  Hello hello{};
  hello.ok = direct_alloc(h) && hipIpcGetMemHandle(&hello.handle, h->d_dx) == hipSuccess;
  std::vector<Hello> all(P);
  { const int rc = tbnav::comm_all_gather_host(h->comm, &hello, all.data(), sizeof(Hello)); if (rc != TBNAV_OK) { direct_teardown(h); return rc; } }
  bool everybody = true;
  for (int q = 0; q < P; ++q) everybody = everybody && all[q].ok == 1;
  if (!everybody) { direct_teardown(h); return TBNAV_OK; }
  // 2. map every peer's buffer
  std::vector<unsigned long long*> peers(P, nullptr);
  int ok = 1;
  for (int q = 0; q < P && ok; ++q) {
    if (q == me) { peers[q] = h->d_dx; continue; }
    void* base = nullptr;
    if (hipIpcOpenMemHandle(&base, all[q].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { ok = 0; break; }
    h->dx_opened.push_back(base);
    peers[q] = static_cast<unsigned long long*>(base);
  }
  if (ok && hipMemcpy(h->d_dx_peers, peers.data(), sizeof(unsigned long long*) * P, hipMemcpyHostToDevice) != hipSuccess) ok = 0;
  { const int rc = agree(ok, everybody); if (rc != TBNAV_OK) { direct_teardown(h); return rc; } }
  if (!everybody) { direct_teardown(h); return TBNAV_OK; }
  // 3. self-test: rounds of pattern records through the very kernels the tick uses, every rank checking every rank's block
  //    (a stale cache line, a store that never becomes visible to the peer, a torn word would show here, not in a tick)
  hipStream_t st = nullptr;
  ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess ? 1 : 0;
  std::vector<double> pat((size_t)P * n), got((size_t)P * n);
  for (int it = 0; it < 24 && ok; ++it) {
    for (int q = 0; q < P; ++q) for (int j = 0; j < n; ++j) pat[(size_t)q * n + j] = direct_pattern(q, it, j);
    if (hipMemcpyAsync(h->d_records_all + (size_t)me * n, pat.data() + (size_t)me * n, sizeof(double) * n, hipMemcpyHostToDevice, st) != hipSuccess) { ok = 0; break; }
    // (2 s: the first touch of a fresh peer mapping may take its time; the loop ends at the first failure)
    if (direct_publish(h, st, false) != TBNAV_OK || direct_collect(h, st, 200000000ull) != TBNAV_OK) { ok = 0; break; }
    if (hipMemcpyAsync(got.data(), h->d_records_all, sizeof(double) * P * n, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { ok = 0; break; }
    if (*h->h_dx_err || std::memcmp(got.data(), pat.data(), sizeof(double) * P * n) != 0) ok = 0;
  }
  if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  { const int rc = agree(ok, everybody); if (rc != TBNAV_OK) { direct_teardown(h); return rc; } }
  if (!everybody) { direct_teardown(h); return TBNAV_OK; }
  *h->h_dx_err = 0;
  h->direct_on = true;
  return TBNAV_OK;
}

int direct_partials_and_publish(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, const uint64_t* seed, uint64_t tick, void* stream) {
  // (the sequence number is drawn here: the kernel that produces the records may publish them itself — mppi_merge_records)
  const unsigned int seq = h->dx_seq + 1u;
  h->pub_next = DirectPub{h->d_dx_peers, tbnav::comm_rank(h->comm), tbnav::comm_size(h->comm), (int)(seq & 1u), seq, h->dx_withhold ? 1 : 0};
  h->pub_pending = true;
  const int rc = sharded_partials(h, x0, d_duL, d_duR, seed, tick, stream);
  const bool published = !h->pub_pending;
  h->pub_pending = false;
  if (rc != TBNAV_OK) return rc;
  if (published) { ++h->dx_seq; return TBNAV_OK; }
  DeviceGuard guard(h->device);
  return direct_publish(h, static_cast<hipStream_t>(stream), h->dx_withhold);
}

}  // namespace
int tbnav_mh::sharded_tick(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, const uint64_t* seed, uint64_t tick, void* stream) {
  // an earlier tick's exchange failed (a bound expired, a rank's rollouts failed): say so now, not only at the next last_controls
  const int rc_latched = exchange_error(h);
  if (h->direct_on) {
    // (the direct exchange has a bound: no peer waits for ever for a rank that has stopped)
    if (rc_latched != TBNAV_OK) return rc_latched;
    const int rc = direct_partials_and_publish(h, x0, d_duL, d_duR, seed, tick, stream);
    if (rc != TBNAV_OK) {
      // A local failure publishes nothing: the peers' combines run into the bound, leave their controls as they were and latch the
      // error.  THIS rank latches too (round-5 advisor finding: it did not, and its next tick — same sequence number, the peers'
      // records of the tick it never ran already in its buffer under that number — combined two different ticks and returned OK):
      // bit 1 of its error words, as the all-gather path raises on every rank, and the sequence number moves on so that a stale
      // tag can never match.
      DeviceGuard guard(h->device);
      if (h->h_dx_err) *h->h_dx_err |= 2;
      if (h->d_dx_dead) { const int dead = 2; (void)hipMemcpy(h->d_dx_dead, &dead, sizeof dead, hipMemcpyHostToDevice); (void)hipGetLastError(); }
      ++h->dx_seq;
      return rc;
    }
    DeviceGuard guard(h->device);
    return direct_combine(h, static_cast<hipStream_t>(stream));
  }
  // Through the communicator's all-gather, which has NO bound: a rank that knows of a failure — its own rollouts' now, or the latch an
  // earlier tick's combine raised (a mapped host word: every rank sees it at its own time, round-5 advisor finding) — must not stay
  // away from a collective its peers may already be in.  It JOINS, with records that say so, every tick, for as long as it is
  // attached; the error is what it returns.
  const int rc_local = rc_latched != TBNAV_OK ? rc_latched : sharded_partials(h, x0, d_duL, d_duR, seed, tick, stream);
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t block = sizeof(double) * (size_t)h->T * h->S * TBNAV_MPPI_REC;
  void* mine = reinterpret_cast<char*>(h->d_records_all) + (size_t)tbnav::comm_rank(h->comm) * block;
  if (rc_local != TBNAV_OK) {
    // This rank's rollouts failed.  Its peers are about to enter (or sit in) an all-gather that has no timeout: JOIN it, with
    // records that say so — count kPoison (negative), every other field zero.  A combine that meets one leaves its time step's
    // controls as they were and raises the error words, on EVERY rank including this one (it runs the combine too, so that the
    // ranks' warm starts stay identical); every rank then returns the error from its next enqueue / last_controls / synchronize.
    // (Round 3: an early return left the peers in ncclAllGather for good.  Round 4: NaN records — which the clamp's fmin / fmax
    // turned into u = -max_wheel_vel on the healthy ranks, advisor finding.)
    std::vector<double> bad((size_t)h->T * h->S * TBNAV_MPPI_REC, 0.0);
    for (size_t q = 6; q < bad.size(); q += TBNAV_MPPI_REC) bad[q] = kPoison;
    (void)hipMemcpyAsync(mine, bad.data(), block, hipMemcpyHostToDevice, st);
    (void)hipStreamSynchronize(st);   // (`bad` is pageable host memory: do not let it go out of scope under the copy)
    (void)hipGetLastError();
  }
  const void* send = mine;
  void* recv = h->d_records_all;
  const int rc = tbnav::comm_all_gather(1, &h->comm, &send, &recv, block, &st);
  if (rc != TBNAV_OK) return rc_local != TBNAV_OK ? rc_local : rc;
  const int rc_combine = launch_combine(h, h->d_records_all, tbnav::comm_size(h->comm), st);
  return rc_local != TBNAV_OK ? rc_local : rc_combine;
}
namespace {
}  // namespace

extern "C" {
// `rounds` exchanges of this handle's record block and nothing else, timed with HIP events on `stream` (collective: every rank of
// the communicator calls it with the same count) — what the exchange costs by itself on the node at hand
int tbnav_mppi_exchange_probe(tbnav_mppi* h, int32_t rounds, void* stream, double* us_per_round) {
  if (!h || !h->comm || rounds <= 0 || !us_per_round || h->comm == nullptr) return TBNAV_ERR_INVALID_ARG;
  if (!tbnav::comm_is_multiprocess(h->comm) && tbnav::comm_size(h->comm) != 1) return TBNAV_ERR_UNSUPPORTED;  // (a group's members are driven together)
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev[2];
  for (auto& e : ev) TBNAV_HIP(hipEventCreate(&e));
  const size_t block = sizeof(double) * (size_t)h->T * h->S * TBNAV_MPPI_REC;
  const void* send = reinterpret_cast<const char*>(h->d_records_all) + (size_t)tbnav::comm_rank(h->comm) * block;
  void* recv = h->d_records_all;
  int rc = TBNAV_OK;
  if (hipEventRecord(ev[0], st) != hipSuccess) rc = TBNAV_ERR_HIP;
  for (int r = 0; r < rounds && rc == TBNAV_OK; ++r) {
    if (h->direct_on) { rc = direct_publish(h, st, false); if (rc == TBNAV_OK) rc = direct_collect(h, st, h->dx_budget); }
    else rc = tbnav::comm_all_gather(1, &h->comm, &send, &recv, block, &st);
  }
  float ms = 0.f;
  if (rc == TBNAV_OK && (hipEventRecord(ev[1], st) != hipSuccess || hipEventSynchronize(ev[1]) != hipSuccess || hipEventElapsedTime(&ms, ev[0], ev[1]) != hipSuccess)) rc = TBNAV_ERR_HIP;
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (rc == TBNAV_OK) rc = exchange_error(h);
  *us_per_round = rc == TBNAV_OK ? (double)ms * 1e3 / rounds : 0.0;
  return rc;
}

}  // extern "C"

namespace {
// The direct exchange for the members of one process (a ROS node driving several GPUs): the members' buffers are plain device
// pointers of this process — peer access between distinct devices, nothing to map — and one host thread enqueues, per tick,
// every member's rollouts + publication and then every member's polling combine (what a kernel polls for was enqueued before
// it, on every stream).  Same kernels, same words, same self-test as between processes.
int group_direct_setup(tbnav_mppi_group* g) {
  const int P = g->n;
  // every member's device idle BEFORE any member's buffer is freed: a member's publish kernel, still in flight on its own device,
  // stores into every other member's buffer (round-3 advisor finding: one member at a time freed a buffer under such stores)
  auto quiesce_all = [&]() { for (tbnav_mppi* h : g->m) if (h) { DeviceGuard guard(h->device); (void)hipDeviceSynchronize(); } };
  auto teardown_all = [&]() { quiesce_all(); for (tbnav_mppi* h : g->m) if (h) { DeviceGuard guard(h->device); direct_teardown(h); } };
  teardown_all();
  bool want = P > 1;
  for (tbnav_mppi* h : g->m) want = want && h->direct_want && h->comm;
  if (!want) return TBNAV_OK;
  auto give_up = [&]() { teardown_all(); return (int)TBNAV_OK; };
  for (int r = 0; r < P; ++r)
    for (int q = 0; q < P; ++q) {
      if (g->m[r]->device == g->m[q]->device) continue;
      DeviceGuard guard(g->m[r]->device);
      const hipError_t e = hipDeviceEnablePeerAccess(g->m[q]->device, 0);
      (void)hipGetLastError();
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return give_up();
    }
  for (tbnav_mppi* h : g->m) { DeviceGuard guard(h->device); if (!direct_alloc(h)) return give_up(); }
  std::vector<unsigned long long*> peers(P);
  for (int q = 0; q < P; ++q) peers[q] = g->m[q]->d_dx;
  for (tbnav_mppi* h : g->m) {
    DeviceGuard guard(h->device);
    if (hipMemcpy(h->d_dx_peers, peers.data(), sizeof(unsigned long long*) * P, hipMemcpyHostToDevice) != hipSuccess) return give_up();
  }
  const int n = g->m[0]->T * g->m[0]->S * TBNAV_MPPI_REC;
  std::vector<double> pat((size_t)P * n), got((size_t)P * n);
  for (int it = 0; it < 12; ++it) {
    for (int q = 0; q < P; ++q) for (int j = 0; j < n; ++j) pat[(size_t)q * n + j] = direct_pattern(q, it, j);
    for (int r = 0; r < P; ++r) {
      tbnav_mppi* h = g->m[r];
      DeviceGuard guard(h->device);
      if (hipMemcpyAsync(h->d_records_all + (size_t)r * n, pat.data() + (size_t)r * n, sizeof(double) * n, hipMemcpyHostToDevice, g->st[r]) != hipSuccess ||
          direct_publish(h, g->st[r], false) != TBNAV_OK) return give_up();
    }
    for (int r = 0; r < P; ++r) { DeviceGuard guard(g->m[r]->device); if (direct_collect(g->m[r], g->st[r], 200000000ull) != TBNAV_OK) return give_up(); }
    for (int r = 0; r < P; ++r) {
      tbnav_mppi* h = g->m[r];
      DeviceGuard guard(h->device);
      if (hipMemcpyAsync(got.data(), h->d_records_all, sizeof(double) * P * n, hipMemcpyDeviceToHost, g->st[r]) != hipSuccess || hipStreamSynchronize(g->st[r]) != hipSuccess ||
          *h->h_dx_err || std::memcmp(got.data(), pat.data(), sizeof(double) * P * n) != 0) return give_up();
    }
  }
  for (tbnav_mppi* h : g->m) { *h->h_dx_err = 0; h->direct_on = true; }
  return TBNAV_OK;
}
}  // namespace

extern "C" {

void tbnav_mppi_group_destroy(tbnav_mppi_group* g) {
  if (!g) return;
  // (all members idle before the first one's buffers go: their kernels store into each other's exchange buffers)
  for (tbnav_mppi* h : g->m) if (h) { DeviceGuard guard(h->device); (void)hipDeviceSynchronize(); }
  for (int r = 0; r < g->n; ++r) {
    if (r < (int)g->m.size()) tbnav_mppi_destroy(g->m[r]);
    if (r < (int)g->c.size()) tbnav_comm_destroy(g->c[r]);
    if (r < (int)g->st.size() && g->st[r]) (void)hipStreamDestroy(g->st[r]);
  }
  delete g;
}

int tbnav_mppi_group_create(const tbnav_mppi_params* params, int32_t n_gpus, const int32_t* devices, tbnav_mppi_group** out) {
  if (!params || !out || n_gpus <= 0 || params->rollouts <= 0 || params->rollouts % n_gpus != 0) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  tbnav_mppi_group* g = new (std::nothrow) tbnav_mppi_group();
  if (!g) return TBNAV_ERR_INVALID_ARG;
  g->n = n_gpus; g->K_global = params->rollouts;
  g->m.assign(n_gpus, nullptr); g->c.assign(n_gpus, nullptr); g->st.assign(n_gpus, nullptr);
  int rc = tbnav_comm_create_local(n_gpus, devices, g->c.data());
  for (int r = 0; r < n_gpus && rc == TBNAV_OK; ++r) {
    tbnav_mppi_params p = *params;
    p.rollouts = params->rollouts / n_gpus;
    p.device = tbnav_comm_device(g->c[r]);
    rc = tbnav_mppi_create(&p, &g->m[r]);
    if (rc == TBNAV_OK) rc = tbnav_mppi_attach_comm(g->m[r], g->c[r]);
    if (rc == TBNAV_OK) { DeviceGuard guard(p.device); if (hipStreamCreateWithFlags(&g->st[r], hipStreamNonBlocking) != hipSuccess) rc = TBNAV_ERR_HIP; }
  }
  if (rc == TBNAV_OK) rc = group_direct_setup(g);
  if (rc != TBNAV_OK) { tbnav_mppi_group_destroy(g); return rc; }
  *out = g;
  return TBNAV_OK;
}

int tbnav_mppi_group_size(const tbnav_mppi_group* g) { return g ? g->n : -1; }
int tbnav_mppi_group_member(tbnav_mppi_group* g, int32_t rank, tbnav_mppi** out) {
  if (!g || !out || rank < 0 || rank >= g->n) return TBNAV_ERR_INVALID_ARG;
  *out = g->m[rank];
  return TBNAV_OK;
}
int tbnav_mppi_group_set_waypoint(tbnav_mppi_group* g, double x, double y, double theta) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_mppi* h : g->m) { const int rc = tbnav_mppi_set_waypoint(h, x, y, theta); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int tbnav_mppi_group_set_initial_controls(tbnav_mppi_group* g, double uL, double uR) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_mppi* h : g->m) { const int rc = tbnav_mppi_set_initial_controls(h, uL, uR); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int tbnav_mppi_group_set_controls(tbnav_mppi_group* g, const double* u_host) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_mppi* h : g->m) { const int rc = tbnav_mppi_set_controls(h, u_host); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int tbnav_mppi_group_get_controls(tbnav_mppi_group* g, double* u_host) { return g ? tbnav_mppi_get_controls(g->m[0], u_host) : TBNAV_ERR_INVALID_ARG; }
int tbnav_mppi_group_set_dynamics(tbnav_mppi_group* g, int32_t model) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_mppi* h : g->m) { const int rc = tbnav_mppi_set_dynamics(h, model); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int tbnav_mppi_group_set_option(tbnav_mppi_group* g, int32_t option, int32_t value) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_mppi* h : g->m) { const int rc = tbnav_mppi_set_option(h, option, value); if (rc != TBNAV_OK) return rc; }
  if (option == TBNAV_MPPI_OPT_DIRECT_EXCHANGE) return group_direct_setup(g);   // (a group is attached already: the choice is made here)
  return TBNAV_OK;
}

}  // extern "C"

namespace {
// every member's partials, ONE grouped all-gather, every member's combine; member 0 publishes when asked to
int group_tick(tbnav_mppi_group* g, const double x0[3], bool own_noise, const uint64_t* seed, uint64_t tick, bool publish) {
  const int n = g->n;
  if (g->m[0]->direct_on) {
    for (int r = 0; r < n; ++r) {
      const int rc = direct_partials_and_publish(g->m[r], x0, own_noise ? g->m[r]->d_duL : nullptr, own_noise ? g->m[r]->d_duR : nullptr, seed, tick, g->st[r]);
      if (rc != TBNAV_OK) return rc;
    }
    for (int r = 0; r < n; ++r) {
      DeviceGuard guard(g->m[r]->device);
      g->m[r]->publish_next = publish && r == 0;
      const int rc = direct_combine(g->m[r], g->st[r]);
      g->m[r]->publish_next = false;
      if (rc != TBNAV_OK) return rc;
    }
    return TBNAV_OK;
  }
  for (int r = 0; r < n; ++r) {
    const int rc = sharded_partials(g->m[r], x0, own_noise ? g->m[r]->d_duL : nullptr, own_noise ? g->m[r]->d_duR : nullptr, seed, tick, g->st[r]);
    if (rc != TBNAV_OK) return rc;
  }
  std::vector<const void*> send(n);
  std::vector<void*> recv(n);
  const size_t block = sizeof(double) * (size_t)g->m[0]->T * g->m[0]->S * TBNAV_MPPI_REC;
  for (int r = 0; r < n; ++r) { recv[r] = g->m[r]->d_records_all; send[r] = reinterpret_cast<const char*>(g->m[r]->d_records_all) + (size_t)r * block; }
  { const int rc = tbnav::comm_all_gather(n, g->c.data(), send.data(), recv.data(), block, g->st.data()); if (rc != TBNAV_OK) return rc; }
  for (int r = 0; r < n; ++r) {
    DeviceGuard guard(g->m[r]->device);
    g->m[r]->publish_next = publish && r == 0;
    const int rc = launch_combine(g->m[r], g->m[r]->d_records_all, n, g->st[r]);
    g->m[r]->publish_next = false;
    if (rc != TBNAV_OK) return rc;
  }
  return TBNAV_OK;
}
}  // namespace

extern "C" {

int tbnav_mppi_group_enqueue_rng(tbnav_mppi_group* g, const double x0[3], uint64_t seed, uint64_t tick) {
  if (!g || !x0) return TBNAV_ERR_INVALID_ARG;
  return group_tick(g, x0, false, &seed, tick, false);
}
int tbnav_mppi_group_enqueue_rng_batch(tbnav_mppi_group* g, const double* x0s, int32_t x0_stride, uint64_t seed, uint64_t first_tick, int32_t n_ticks) {
  if (!g || !x0s || n_ticks < 0 || (x0_stride != 0 && x0_stride < 3)) return TBNAV_ERR_INVALID_ARG;
  for (int32_t i = 0; i < n_ticks; ++i) {
    const int rc = group_tick(g, x0s + (size_t)i * x0_stride, false, &seed, first_tick + (uint64_t)i, false);
    if (rc != TBNAV_OK) return rc;
  }
  return TBNAV_OK;
}
int tbnav_mppi_group_last_controls(tbnav_mppi_group* g, double u_out[2]) {
  if (!g || !u_out) return TBNAV_ERR_INVALID_ARG;
  return tbnav_mppi_last_controls(g->m[0], g->st[0], u_out);
}
int tbnav_mppi_group_synchronize(tbnav_mppi_group* g) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (int r = 0; r < g->n; ++r) { DeviceGuard guard(g->m[r]->device); TBNAV_HIP(hipStreamSynchronize(g->st[r])); }
  for (const tbnav_mppi* h : g->m)  // (any member's combine that ran out of time waiting for a peer's records)
    { const int rc = exchange_error(h); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int tbnav_mppi_group_new_controls_rng(tbnav_mppi_group* g, const double x0[3], uint64_t seed, uint64_t tick, double u_out[2]) {
  if (!g || !x0 || !u_out) return TBNAV_ERR_INVALID_ARG;
  const int rc = group_tick(g, x0, false, &seed, tick, true);
  return rc != TBNAV_OK ? rc : tbnav_mppi_last_controls(g->m[0], g->st[0], u_out);
}
// parity mode: host noise in the reference's draw order for the WHOLE ensemble, noise[(k * T + i) * 2 + c]; member r takes
// rollouts [r * K/n, (r + 1) * K/n)
int tbnav_mppi_group_new_controls(tbnav_mppi_group* g, const double x0[3], const double* noise_host, double u_out[2]) {
  if (!g || !x0 || !noise_host || !u_out) return TBNAV_ERR_INVALID_ARG;
  for (int r = 0; r < g->n; ++r) {
    tbnav_mppi* h = g->m[r];
    DeviceGuard guard(h->device);
    const size_t nk = (size_t)h->T * h->K;
    if (!h->d_raw) TBNAV_HIP(hipMalloc((void**)&h->d_raw, 2 * nk * sizeof(double)));
    TBNAV_HIP(hipMemcpyAsync(h->d_raw, noise_host + (size_t)r * 2 * nk, 2 * nk * sizeof(double), hipMemcpyHostToDevice, g->st[r]));
    const int blocks = (int)((nk + 255) / 256 < 4096 ? (nk + 255) / 256 : 4096);
    hipLaunchKernelGGL(mppi_unpack_noise, dim3(blocks), dim3(256), 0, g->st[r], h->T, h->K, h->d_raw, h->d_duL, h->d_duR);
    TBNAV_HIP(hipGetLastError());
  }
  const int rc = group_tick(g, x0, true, nullptr, 0, true);
  return rc != TBNAV_OK ? rc : tbnav_mppi_last_controls(g->m[0], g->st[0], u_out);
}

}  // extern "C"
