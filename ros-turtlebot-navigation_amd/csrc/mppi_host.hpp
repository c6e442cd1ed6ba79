// mppi_host.hpp — the MPPI handle and what the two host files share (mppi.hip: create / options / launchers / single-GPU C-ABI;
// mppi_sharded.hip: communicator attachment, direct exchange, sharded tick, groups).  Kernels and launch structs: mppi_device.hpp.
#ifndef TBNAV_MPPI_HOST_HPP
#define TBNAV_MPPI_HOST_HPP
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "comm.hpp"
#include "common.hpp"
#include "mppi_device.hpp"
#include "tbnav_mppi.h"

using tbnav_mk::DirectPub;

// =================================================================================================
// Handle
// =================================================================================================
struct tbnav_mppi {
  tbnav_mppi_params p;
  int T = 0, K = 0, S = 0, device = 0;
  int lk_rollout[5] = {0, 0, 0, 0, 0};  // the instantiation the last rollout launch picked: kind (1 fused, 2 scan, 3 prefix, 4 cost), template arguments
  int lk_combine[2] = {0, 0};           // ... and the last combine: KEEP, DIRECT  (tbnav_mppi_last_kernel_names)
  double xd[3] = {0, 0, 0};
  double uinit[2] = {0, 0};
  double* d_u[2] = {nullptr, nullptr};  // [2][T] each; d_u[ucur] holds the controls, d_u[1-ucur] receives the next update
  int ucur = 0;
  bool pending_shift = false;   // d_u[ucur] is an updated, not yet shifted vector (the shift is applied on read)
  double* d_J = nullptr;        // [T][K]
  double* d_duL = nullptr;      // [T][K] own noise buffers (host-noise upload / device RNG)
  double* d_duR = nullptr;
  double* d_raw = nullptr;      // [K][T][2] staging for host-order noise (lazy)
  double* d_records = nullptr;  // [T][S][8]
  double* d_out = nullptr;      // [2] device copy of the last controls
  double* d_out_host = nullptr; // device view of h_out
  double* h_out = nullptr;      // mapped pinned [4]: ul, ur, tick number of the combine that published them
  uint64_t seq = 0;             // combines enqueued so far
  uint64_t published = 0;       // tick number of the last combine that was asked to publish to h_out
  bool publish_next = false;    // set by the synchronous entry points round their enqueue
  // tbnav_mppi_enqueue_rng_batch replays captured hipGraphs of ticks (two launches each) instead of launching them one by one:
  // chunks of kGraphTicks (~0.5 us less per tick of a 8-9 us tick) and, for what is left — or a batch shorter than a chunk —,
  // ONE graph of the batch's own length (tgs): a batch of 20 ticks is one submission instead of forty, 9.3 us per tick whatever
  // the host's launch rate is doing (9.8-12.1 launched one by one)
  bool graph_on = true;         // TBNAV_MPPI_OPT_BATCH_GRAPH; cleared for good if a capture ever fails
  struct TickGraph {
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
    int len = 0; uint64_t seed = 0; double x0[3] = {0, 0, 0}; hipStream_t stream = nullptr; int ucur = -1; uint64_t epoch = ~0ull;
  };
  TickGraph tg, tgs;
  int tgs_wish_len = 0, tgs_wish_ucur = -1;   // the short graph is built by the SECOND batch in a row that could use the same one
  // The graph's kernel nodes hold BY VALUE everything launch_fused / launch_combine read from the handle when it was captured
  // (waypoint, uinit, lambda, dynamics, trig, keep_j + the J pointer, the rng shard, fused_S and the record buffer).  Every
  // setter that changes one of those bumps cfg_epoch; a graph captured under another epoch is rebuilt, never replayed.
  uint64_t cfg_epoch = 0;
  uint64_t graph_ticks = 0;  // ticks enqueued through graph replays so far (tbnav_mppi_graph_replayed_ticks: what a bench line should say ran)
  uint64_t* d_tick0 = nullptr;
  uint64_t tg_dev_tick = ~0ull;  // what *d_tick0 holds once everything enqueued so far has run (each replay's last node adds the chunk)
  int lds_from = 0;           // first time step whose loss is staged in LDS (0 = all of them)
  int prefix_rg = 0;          // > 0: mppi_rollout_prefix (the large-K default): exact suffix sums for the last 4*prefix_rg steps, exclusive prefixes before
  int prefix_rows = 0;        // rows of d_J that hold exclusive prefixes after the LAST rollout launch (0: every row is J)
  double* d_total = nullptr;  // [K] whole cost of every rollout (mppi_rollout_prefix)
  int scan_tc = 0;            // steps per thread of the time-parallel rollout kernel (0 = sequential kernel)
  int fused_r = 0;            // rollouts per workgroup of the fused rollout+partials kernel (0 = off: three kernels)
  int fused_S = 0;            // its records per time step, ceil(K / fused_r)
  // which ticks take the fused kernel: resident-noise ticks (tbnav_mppi_enqueue_dev, new_controls*) and device-noise ticks
  // (tbnav_mppi_enqueue_rng ...: the perturbations are drawn inside it) cross over to the three-kernel tick at different K
  bool fused_dev = false, fused_rng = false;
  double* d_records_f = nullptr;  // [T][fused_S][8]
  int trig = 1;               // sincos evaluations per RK4 step (1 = angle addition, 3 = the reference's three)
  int dyn = 0;                // rollout dynamics: 0 = the reference's RK4 cart, 1 = exact arcs (tbnav_mppi_set_dynamics)
  bool keep_j = false;        // the fused kernel also writes J to HBM (parity hook tbnav_mppi_get_cost_to_go); other kernels always do
  bool j_valid = false;       // d_J holds the last tick's cost-to-go
  uint64_t k0 = 0, k_global = 0;  // device noise source: this handle's rollouts are [k0, k0 + K) of k_global (sharded ensembles)
  // sharded ensemble (tbnav_mppi_attach_comm / tbnav_mppi_group_*): every tick is shard partials -> ONE all-gather of the
  // records (RCCL) -> the combine of all shards' records, all enqueued on the tick's stream
  tbnav_comm* comm = nullptr;
  double* d_records_all = nullptr;  // [nranks][T][S][8]; this rank's records are written in place at [rank]
  // direct exchange (mppi_direct_publish / _collect): set up at attach for multi-process communicators when every rank can
  // (fine-grained memory, IPC mapping, a self-test); otherwise the communicator's all-gather carries the records
  bool direct_want = true, direct_on = false;   // TBNAV_MPPI_OPT_DIRECT_EXCHANGE
  unsigned long long* d_dx = nullptr;           // [2 parities][nranks][2 * n] tagged words (n = T * S * 8), fine-grained
  unsigned long long** d_dx_peers = nullptr;    // [nranks] every rank's d_dx as mapped into this process
  std::vector<void*> dx_opened;                 // the mappings of the peers' buffers (closed at detach)
  // the exchange's error words, alive while a communicator is attached (either exchange): mapped pinned, raised by a combine that ran
  // out of time waiting for a peer's words (bit 0) or met a poisoned record — a rank whose own rollouts failed (bit 1); latched
  int* h_dx_err = nullptr; int* d_dx_err = nullptr;
  int* d_dx_dead = nullptr;                          // its device twin: later ticks see it without a trip over PCIe
  bool fail_next = false;                            // fault injection (TBNAV_MPPI_OPT_FAULT_INJECT, tests): the next sharded tick's local half fails
  bool wide_combine = true;                          // TBNAV_MPPI_OPT_WIDE_COMBINE: four waves per time step when a step has more than 256 records
  int sampler = 1;                                   // TBNAV_MPPI_OPT_SAMPLER: 1 (default) = fp64 Box-Muller on 52-bit uniforms, 0 = fp32 on 24-bit uniforms
  unsigned long long dx_budget = 200000000ull;       // 2 s of the 100 MHz clock (host-side skew between ranks is legitimate — a control loop's is milliseconds; longer: the peer has failed)
  bool dx_withhold = false;                          // fault injection (TBNAV_MPPI_OPT_DIRECT_EXCHANGE = 2, tests): this rank's records never reach its peers
  unsigned int dx_seq = 0;
  // set by the sharded tick round its call of the shard-partials entry point: if that ends in mppi_merge_records, the kernel
  // publishes the records itself (and clears this); otherwise the tick launches mppi_direct_publish
  bool pub_pending = false;
  DirectPub pub_next{nullptr, 0, 0, 0, 0u, 0};
};

// One process driving several GPUs: the whole ensemble behind one object (what controller::MPPI built with n_gpus > 1 holds).
struct tbnav_mppi_group {
  int n = 0;
  std::vector<tbnav_mppi*> m;
  std::vector<tbnav_comm*> c;
  std::vector<hipStream_t> st;
  std::vector<double*> d_raw;  // per member: staging of its slice of host-order noise
  int K_global = 0;
};

namespace tbnav_mh {
using namespace tbnav_mk;

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DeviceGuard() { if (ok && prev >= 0) (void)hipSetDevice(prev); }
};

// (mppi.hip)
Lam lam_of(double lambda);
inline Lam lam_of(const tbnav_mppi* h) { return lam_of(h->p.lambda); }
int launch_combine(tbnav_mppi* h, const double* d_records, int G, hipStream_t st, int S = -1, const DirectSrc* direct = nullptr);
// (mppi_sharded.hip)
int sharded_tick(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, const uint64_t* seed, uint64_t tick, void* stream);
void direct_teardown(tbnav_mppi* h);
void exchange_words_free(tbnav_mppi* h);
// a latched error of the attached exchange (the stream has been waited for, or the caller accepts an earlier tick's): TBNAV_OK or
// TBNAV_ERR_HIP with the text set
int exchange_error(const tbnav_mppi* h);

}  // namespace tbnav_mh
#endif
