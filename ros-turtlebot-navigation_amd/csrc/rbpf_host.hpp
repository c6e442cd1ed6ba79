// rbpf_host.hpp — the RBPF handle and what the host files share (round 6: csrc/rbpf.hip was one 2 500-line file):
//   rbpf.hip           the launch sequence of one scan (scan_enqueue / scan_finish / slam_impl), create / destroy, tbnav_rbpf_slam
//   rbpf_reffield.hip  the reference-field mode's plumbing: logs of occupied-set changes in, journal of field cells out (ref_field.hpp)
//   rbpf_batch.hip     tbnav_rbpf_slam_batch (two scans in the stream) and the shard-level entry points of the Python-driven exchange
//   rbpf_sharded.hip   the sharded filter inside the library: sharded_scan, tbnav_rbpf_attach_comm, tbnav_rbpf_group_*
//   rbpf_io.hip        particle blobs (export / import, one and many), particles / log-odds / distance fields in and out
//   rbpf_api.hip       queries and options: trace, best state / map, one-particle GridMapper calls, tbnav_rbpf_set_option, measurement hooks
// Kernels and launch structs: rbpf_device.hpp.  Everything in namespace tbnav_rh is internal to libtbnav_hip.so.
#ifndef TBNAV_RBPF_HOST_HPP
#define TBNAV_RBPF_HOST_HPP
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <thread>
#include <vector>
#include <sched.h>

#include "comm.hpp"
#include "common.hpp"
#include "ref_field.hpp"
#include "tbnav_rbpf.h"
#include "rbpf_device.hpp"

using namespace tbnav_rk;  // the launch-argument structs and the kernels (rbpf_device.hpp)

using namespace tbnav_rk;  // the launch-argument structs and the kernels (rbpf_device.hpp)

// =================================================================================================
// Handle
// =================================================================================================
struct tbnav_rbpf {
  tbnav_rbpf_params p;
  int device = 0, N = 0, k = 0, xsize = 0, ysize = 0, words = 0, radius = 0, edt_cols = 64;
  size_t G = 0;
  double l_prior = 0, l_occ = 0, l_free = 0, cut_occ = 0, max_occ_dist = 10.0;
  // particle state: [N][7] = pose(3), prev_pose(3), weight — double-buffered with the maps
  double* d_state[2] = {nullptr, nullptr};
  // log-odds: tiled, copy-on-write (see TilePool).  The tables are double-buffered with the rest of the particle state.
  TilePool pool{};
  unsigned int* d_table[2] = {nullptr, nullptr};  // [N][TT]
  unsigned int* d_shed = nullptr;                 // [N][TT]
  int TW = 0, TT = 0;
  double* d_dense = nullptr;   // [G] staging of one particle's dense log-odds (get/set_log_odds), allocated on first use
  double* d_cs = nullptr;      // [N] prefix scratch of the normalise kernel (N > kNormChunk)
  unsigned int* d_tile_scratch = nullptr;  // [TT] tile ids of a particle being exported
  // sharded filter: normalise / select over the all-gathered weights (tbnav_rbpf_resample_global_dev)
  double* d_gw = nullptr; double* d_gcs = nullptr; int* d_gparent = nullptr; double* d_gz = nullptr; size_t g_cap = 0;
  unsigned long long* d_touched = nullptr;  // [2] measurement hook: cell updates / distinct cells written (tbnav_rbpf_scan_counts)
  // rbpf_raycast_box's LDS array sized by what the particles' boxes needed in the last scans (device feedback, see the kernel)
  int* d_box_need = nullptr;      // [3] words of LDS array the largest box of a launch needed; the slots take turns
  int* h_box_need = nullptr;      // mapped pinned: the last complete launch's maximum
  int* d_box_need_host = nullptr; // device view of h_box_need
  unsigned int rc_launches = 0;   // box-counter launches so far (which slot accumulates)
  int raycast_adapt = 1;          // TBNAV_RBPF_OPT_RAYCAST_ADAPT: 0 = size the array for the worst case of the scan's longest beam
  bool count_touched = false;
  // stored distance field, u16 [N][G] x 2: allocated on first need (injection, materialisation, the stored-field
  // modes); the default query mode never touches it.  NULL until then.
  uint16_t* d_code[2] = {nullptr, nullptr};
  int* d_nocc[2] = {nullptr, nullptr};
  int cur = 0;
  int* d_trow[2] = {nullptr, nullptr};                   // [N][TW] occupied cells per tile row, kept current by the raycast kernel (the bits themselves live in the tiles)
  unsigned long long* d_bm_dense = nullptr;              // [N][xsize][words] dense rows for the exact-transform kernels, rebuilt from the tiles on demand
  int* d_rc_dense = nullptr;                             // [N][xsize]        (allocated with the stored field)
  double2* d_beams = nullptr;  // capacity max_beams
  int max_beams = 0;
  double* d_normals = nullptr;
  size_t normals_cap = 0;
  const double* last_normals = nullptr;  // the normals the last scan used (d_normals, or an entry of the batch ring); NULL: drawn inside rbpf_propose
  size_t last_z_index = 0;               // where in them its resampling offset sits (N * stride)
  const double* last_z_ptr = nullptr;    // the resampling offset's normal of the last scan, wherever it is (last_normals + last_z_index, or d_zslot)
  // device noise drawn inside rbpf_propose (round 5; TBNAV_RBPF_OPT_NOISE_IN_KERNEL, default on): nothing is stored but the
  // resampling offset's normal; workgroup 0 of the proposal launch carries the beam table over and publishes beam_seq (NoiseSrc)
  int noise_in_kernel = 0;   // TBNAV_RBPF_OPT_NOISE_IN_KERNEL (round 6: off by default — the stored-first form is the faster kernel and has no hand-over inside a launch)
  double* d_zslot = nullptr;
  unsigned int* d_beam_ready = nullptr;   // fine-grained
  double2* d_beams_fg = nullptr; int fg_beams_cap = 0;   // fine-grained copy of the beam table (NoiseSrc::fg_beams)
  unsigned int beam_seq = 0;
  struct { unsigned long long seed = 0, scan = 0; size_t base = 0, z_index = 0, n = 0; bool valid = false; } last_drawn;  // what tbnav_rbpf_get_normals regenerates from
  // tbnav_rbpf_slam_batch draws the noise of a few scans ahead in one launch: normals and beam tables of ring_scans scans
  double* d_norm_ring = nullptr; size_t norm_ring_stride = 0;
  double2* d_beam_ring = nullptr; double2* h_beam_ring = nullptr; size_t beam_ring_stride = 0;
  int ring_scans = 0;
  int* d_parent = nullptr;     // [2][N]: the parent of every slot | how many slots chose each particle
  ExportCuts cuts{};           // host-derived (glibc) log-odds break points of the int8 map export
  int* d_best = nullptr;       // arg-max particle index
  double* d_best_pose = nullptr;
  int8_t* d_export = nullptr;  // [G]
  bool sm_on = false;          // N1 option: per-particle scan matching before sampling (tbnav_rbpf_set_scan_matching)
  ScanMatchC sm{0.05, 0.05, 5, 64};
  double* d_center = nullptr;  // [N][3] matched poses of the last call
  // scratch of the batched export / import (tbnav_rbpf_export_batch_dev ...): grown on demand
  int* d_bslots = nullptr; int2* d_bcount = nullptr; BatchItem* d_bitems = nullptr; BlobHeader* d_bhdr = nullptr; size_t batch_cap = 0;
  std::vector<int2> batch_counts;  // tiles / field state of the slots counted last
  double* d_mixlut = nullptr;  // [kMixLut] mixture term per distance code (constants of the handle: tabulated once at create)
  double* d_score = nullptr;   // [N]
  bool timing = false;         // record HIP events round the kernels (tbnav_rbpf_set_timing): each costs device time, so off by default
  int tile_cap = 0;            // cells of the raycast LDS tile (0 = use the beam-ordered kernel)
  int raycast_threads = 0;     // block size of the tile raycast: 0 = 1024 (TBNAV_RBPF_OPT_RAYCAST_THREADS)
  std::vector<double2> beam_cs;  // (cos, sin) of every beam's angle in the sensor frame, kept between scans
  std::vector<double2> beams_tmp;
  int raycast_band_rows = 0;   // > 0: cap the LDS array of rbpf_raycast_box at about this many box rows (TBNAV_RBPF_OPT_RAYCAST_BAND_ROWS, tests)
  int raycast_cell16 = 1;      // 0: never the 16-bit cell form; 1: where it buys a higher residency (default); 2: wherever it can run (TBNAV_RBPF_OPT_RAYCAST_CELL16)
  int lk_raycast = -1, lk_raycast_wps = 0, lk_raycast_c16 = 0, lk_raycast_ev = 8, lk_raycast_grid = 0, lk_propose = 0, lk_propose_dn = 0, lk_box_need = 0, lk_box_cap = 0;  // the instantiations the last launches were (tbnav_rbpf_last_kernel_names): raycast threads (0 = beam-ordered), its workgroups, propose threads
  double* d_sens = nullptr;    // [N][4] sensor transform (X, Y, sin, cos) of each particle's new pose, left by the proposal kernel
  uint64_t seed = 0x5EEDull, scan_index = 0;  // device noise source (normals == NULL)
  uint64_t rng_first = 0, rng_n_global = 0;   // sharded filters: this handle's particles are [rng_first, rng_first + N) of rng_n_global (0 = unsharded)
  // sharded filter inside the library (tbnav_rbpf_attach_comm / tbnav_rbpf_group_*): the weights' all-gather and the global
  // normalise / select run on a SECOND stream beside the local map update
  tbnav_comm* comm = nullptr;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_w = nullptr, ev_g = nullptr;   // "the proposal kernel has left the weights" / (ev_g: unused since round 5 — the weights come back on the main stream)
  int shard_latched = TBNAV_OK;                // a rank-local failure after a scan's last agreement: carried into the next scan's, where every rank stops with it
  double* d_gw_raw = nullptr;                  // [n_global] all-gathered raw weights
  char* d_sendbuf = nullptr; char* d_recvbuf = nullptr; size_t send_cap = 0, recv_cap = 0;   // particle blobs of a cross-rank resample
  unsigned long long* d_sizes = nullptr;       // [n_local + n_global] blob size of every particle this rank sends | of every particle
  int* d_status = nullptr;                     // [1 + nranks] this rank's status | everybody's
  bool full_edt = false;       // distance-field mode 0 (TBNAV_RBPF_DF=full): whole-map transform after every map update
  int df_mode = 2;             // 0 full, 1 windowed refresh before the update (TBNAV_RBPF_DF=window), 2 exact query at lookup (default)
  int* d_fstate = nullptr;     // [N] distance-field state: 0 stale, 1 window fresh, 2 whole field fresh / injected
  int* d_fstate_alt = nullptr; // [N] the other buffer of the resample gather
  // reference distance-field mode (tbnav_rbpf_set_option DF_MODE = REFERENCE): host-side brushfire state + the
  // device log of occupied-set changes it is fed from
  bool ref_field = false;
  tbnav::RefField* ref = nullptr;
  // (which state — and how much of its journal — every field slot of d_code holds is the RefField's own bookkeeping: plan_flush)
  tbnav::RefField::Flush ref_flush;       // the last flush's plan (buffers kept between scans)
  int* d_pend = nullptr;                  // [N] cell + 1 of a lookup that landed on a cell the particle's pass has not written yet (kCodePending), else 0
  int* h_pend = nullptr;                  // [N] pinned copy
  double* d_state_snap = nullptr;         // [7 N] pose / prev_pose / weight before the proposal: a proposal that met pending cells is run again from here
  uint2* d_jentries = nullptr; size_t jentries_cap = 0;   // the flush's packed (cell, code) pairs
  uint3* d_jjobs = nullptr;               // [N] (offset, count, reset) per slot
  int ref_reach = 1;                      // TBNAV_RBPF_OPT_REF_REACH: how far (cells) a scan's brushfire runs before it stops (0: to the end)
  long long ref_reruns = 0;               // proposals run again because a lookup met a pending cell
  long long ref_us[6] = {0, 0, 0, 0, 0, 0}; // host microseconds spent: fetching the logs | RefField::step | resample copies | flushes | before the proposal | the settle look
  int* d_log_pack = nullptr; unsigned long long* d_log_off = nullptr; size_t log_pack_cap = 0, log_off_cap = 0;  // the scan's logs, packed
  int* d_code_src = nullptr;              // [N] slot to copy the field from (rbpf_copy_codes)
  // Reference-field mode, TBNAV_RBPF_OPT_HOST_THREADS 0 (automatic): the passes of a scan are a BURST — 60 ms of CPU time in 4 ms, then
  // nothing until the next scan's — so the number of threads follows the CPU TIME the container is granted (its cgroup quota), not
  // the number of CPUs that quota would keep busy all the time: as many threads as keep the average over a scan period under
  // kRefCpuShare of the quota, between the quota's own count and four times it (ref_field_after_scan).
  bool host_threads_auto = true;
  int host_affinity = 1; double host_quota_cpus = 1.0;   // CPUs this process may run on | CPU time per wall time it may use (= affinity without a quota)
  int ref_threads_used = 0; double ref_prev_step_start_s = -1.0, ref_last_step_wall_s = 0.0;
  int host_threads = 1;        // host threads of the reference-field mode's per-particle work (TBNAV_RBPF_OPT_HOST_THREADS; set at create)
  int* d_log_ev = nullptr;     // [N][log_cap]
  int* d_log_cnt = nullptr;    // [N]
  int log_cap = 0;
  uint64_t scans_done = 0;
  int* d_skip = nullptr;       // [N] scratch: 1 = no refresh needed this call
  int4* d_win = nullptr;       // [N] refreshed window (i0, i1, j0, j1), inclusive
  int* d_tier = nullptr;       // [N] which distance-field kernel handles the particle this scan
  int* d_err = nullptr;
  NormOut* d_norm = nullptr;
  // pinned host staging: the scan going in, the error flags and the normalisation result coming out (pageable
  // buffers make every one of those small copies a blocking, staged transfer)
  double2* h_beams = nullptr;  // [kScanSlots] x capacity max_beams
  // error flags and normalisation result live in mapped pinned host memory: the kernels write them over the
  // fabric (a handful of bytes per scan) and the host reads them after the stream sync — no copy kernels, no memset
  // kScanSlots of each: a scan in flight owns slot (scan number % kScanSlots) — tbnav_rbpf_slam_batch keeps two scans in the
  // stream; every other entry point uses slot 0
  int* h_err = nullptr;        // [kScanSlots][4] host view; d_err is the device view of the same bytes
  NormOut* h_norm = nullptr;   // [kScanSlots] host view of d_norm
  int* d_gate = nullptr;       // [kScanSlots] device memory: 1 = that scan resamples (NormArgs::gate)
  unsigned int* h_seq = nullptr;  // [kScanSlots] mapped: the scan number whose normalisation result the slot holds (NormArgs::seq)
  unsigned int* d_seq = nullptr;  // device view of h_seq
  bool fstate_dirty = true;    // some d_fstate entry may be non-zero
  int batch_pipeline = 1;      // tbnav_rbpf_slam_batch keeps two scans in the stream (TBNAV_RBPF_OPT_BATCH_PIPELINE)
  double* d_trace = nullptr;   // sampled, p_scan, p_pose, mu, sigma, eta, new_pose, weight_raw
  Trace tr{};
  hipStream_t stream = nullptr;
  hipEvent_t ev[TBNAV_RBPF_NKERNELS + 2] = {};  // 0..5 bracket kernels 0..4; 6,7 bracket the gather
  float last_ms[TBNAV_RBPF_NKERNELS] = {0};
  std::vector<int> h_parent;
};

// One process driving several GPUs: the whole filter behind one object (what bmapping::ParticleFilter built with n_gpus > 1 holds).
struct tbnav_rbpf_group {
  int n = 0, n_global = 0;
  std::vector<tbnav_rbpf*> m;
  std::vector<tbnav_comm*> c;
  std::vector<std::vector<double>> normals;  // parity mode: each member's slice of the ensemble's draw stream + the offset
};

namespace tbnav_rh {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true; }
  ~DeviceGuard() { if (ok && prev >= 0) (void)hipSetDevice(prev); }
};

// state layout helpers: the kernels take pose / prev_pose / weight pointers with [N][3] / [N] strides,
// so the 7-double record is split into three arrays inside one allocation.
struct StatePtrs { double *pose, *prev, *weight; };

struct UsTimer {   // adds the enclosing scope's wall time to a counter
  long long& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit UsTimer(long long& a) : acc(a) {}
  ~UsTimer() { acc += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// One scan = scan_enqueue (everything up to and including the map update, on the handle's stream) + scan_finish (wait,
// read the stats, run the resampling copies if the scan decided to resample).  `slot`: which of the kScanSlots result slots
// the scan owns.  gate_prev (device pointer or NULL): the resampling decision of the scan enqueued before this one, when the
// host has not seen it yet — the kernels of this scan do nothing if it is set (tbnav_rbpf_slam_batch).
struct ScanTicket { int slot = 0; int n_valid = 0; bool local_only = false; bool poll = false; unsigned int seq = 0; };
// A scan whose constants, beam table and noise are on the device already (tbnav_rbpf_slam_batch prepares a few scans at a time).
struct Prefetched { ScanC c; int rc = TBNAV_OK; const double2* d_beams = nullptr; const double* d_normals = nullptr; };

StatePtrs state_ptrs(double* base, int N);
MapT map_of(const tbnav_rbpf* h);
int ensure_codes(tbnav_rbpf* h);
double logodds_to_prob(double l);
double find_occ_cut(double l_occ_nominal, double p_occ);
int export_value_host(double l);
double bisect_first(double lo, double hi, bool (*pred)(double, int), int arg);
ExportCuts derive_export_cuts(double cut_occ);
size_t edt_lds_bytes(int xs, int words, int C);
bool mixture_consts(const tbnav_rbpf* h, ScanC& c);
int build_scan_consts(tbnav_rbpf* h, ScanC& c, const float* scan, int n_beams, const double u[3],
                      const double cur_odom[3], const double prev_odom[3], int icp_ok, const double T_icp[3],
                      std::vector<double2>& beams);
int status_from_err(const int err[4]);
int run_distance_field(tbnav_rbpf* h, const GridC& g, int p0, int count, int tiles64);
GridC grid_of(const tbnav_rbpf* h);
int ensure_full_field(tbnav_rbpf* h, int particle);
int resample_on_device(tbnav_rbpf* h);
int ref_field_prepare_log(tbnav_rbpf* h, int Bv, OccLog& log);
int ref_field_flush(tbnav_rbpf* h);
int ref_field_materialize(tbnav_rbpf* h, int particle);
int ref_field_before_propose(tbnav_rbpf* h);
int ref_field_settle(tbnav_rbpf* h, int* h_err, const std::function<int()>& relaunch);
int ref_field_after_scan(tbnav_rbpf* h, bool resampled, int p_first = 0, int p_count = -1);
int launch_raycast(tbnav_rbpf* h, const ScanC& c, int count, const double* sens, const NormArgs* nz = nullptr, int* err = nullptr,
                   const double2* beams_dev = nullptr);
int upload_beams(tbnav_rbpf* h, const std::vector<double2>& beams, int n_beams, int Bv, bool stage_only = false, int slot = 0);
int scan_enqueue(tbnav_rbpf* h, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* normals,
                 tbnav_rbpf_stats* out, bool local_only, int slot, const int* gate_prev, ScanTicket& tk,
                 const Prefetched* pre = nullptr, hipEvent_t weights_ready = nullptr);
int scan_finish(tbnav_rbpf* h, const ScanTicket& tk, tbnav_rbpf_stats* out);
int slam_impl(tbnav_rbpf* h, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
              const double prev_odom[3], int icp_ok, const double T_icp[3], const double* normals,
              tbnav_rbpf_stats* out, bool local_only);
int default_host_threads();
int host_cpu_budget(double& quota_cpus);   // the affinity mask's CPUs; quota_cpus: the cgroup's CPU-time quota in CPUs (the same number without one)
int create_impl(const tbnav_rbpf_params* P, uint64_t max_pool_bytes, tbnav_rbpf** out);
unsigned int pool_lists_for(size_t cap_tiles);   // 1 or kPoolShards (TBNAV_POOL_SHARD_MIN: a test hook, see rbpf.hip)
int pool_free_tiles(tbnav_rbpf* h, uint64_t* free_tiles);   // summed over the pool's free lists (the caller has synchronised the stream)
int batch_scratch(tbnav_rbpf* h, size_t n);
BlobLayout blob_layout(const tbnav_rbpf* h, uint32_t n_tiles, bool has_codes);
int slot_tiles(tbnav_rbpf* h, int slot, std::vector<uint32_t>& tidx, std::vector<uint32_t>& ids, int& fstate);
int count_batch(tbnav_rbpf* h, int32_t n, const int32_t* slots);
int ensure_shard_state(tbnav_rbpf* h);
int grow(char*& buf, size_t& cap, size_t need);
int sharded_scan(int n, tbnav_rbpf* const* hs, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* const* normals, tbnav_rbpf_stats* out,
                 tbnav_rbpf_stats* local_out);
int one_particle_consts(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, ScanC& c);
int sharded_scan(int n, tbnav_rbpf* const* hs, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* const* normals, tbnav_rbpf_stats* out,
                 tbnav_rbpf_stats* local_out);

}  // namespace tbnav_rh
using namespace tbnav_rh;

#endif
