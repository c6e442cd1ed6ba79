// rbpf.hip — bmapping::ParticleFilter::SLAM on MI355X (gfx950) behind the C-ABI of include/tbnav_rbpf.h: the handle (device
// state, tile pool, pinned result slots), the launch sequence of one scan, the reference-field mode's host side, the sharded
// scan (tbnav_rbpf_attach_comm / tbnav_rbpf_group_*) and every extern "C" entry point.  Reference (paths relative to the
// reference tree):
//   bmapping/src/bmapping/particle_filter.cpp:141-251 (SLAM), :295-322, :383-437, :442-500, :504-599
//   bmapping/src/bmapping/grid_mapper.cpp:69-182 (likelihood field, integrateScan), :549-898
//   bmapping/src/bmapping/sensor_model.cpp:43-112 (laserEndPoints)
// The kernels live in their families' files (all -ffp-contract=off; DESIGN.md section 4 has the reasoning and the numbers):
//   rbpf_propose.hip   rbpf_mix_lut, rbpf_sample_normals, rbpf_likelihood_one, rbpf_field_by_query, rbpf_scanmatch, rbpf_propose
//   rbpf_raycast.hip   rbpf_raycast_box<512 / 1024> (default map update), rbpf_raycast (beam-ordered), rbpf_add_repeated_test
//   rbpf_field.hip     rbpf_densify, rbpf_window, rbpf_edt<C>, rbpf_edt_compact<R> (stored-field modes, on-demand fields)
//   rbpf_resample.hip  rbpf_normalize, rbpf_resample_apply, rbpf_pool_init, dense <-> tiles, rbpf_argmax, rbpf_export_map
//   rbpf_migrate.hip   particle blobs for the sharded filter's cross-rank resample
//   rbpf_device.hpp    what they share: launch-argument structs, map accessors, reductions, kernel declarations
//   rbpf_normalize.hpp the normalise / selection body (a kernel of its own and workgroup 0 of the map update)
// (two small kernels of the reference-field mode's plumbing, rbpf_pack_logs / rbpf_copy_codes, stay here beside their only caller)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <atomic>
#include <map>
#include <thread>
#include <vector>
#include <sched.h>

#include "comm.hpp"
#include "common.hpp"
#include <chrono>
#include "ref_field.hpp"
#include "tbnav_rbpf.h"
#include "rbpf_device.hpp"

using namespace tbnav_rk;  // the launch-argument structs and the kernels (rbpf_device.hpp)


// =================================================================================================
// Handle + C-ABI
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
  int ref_reach = 3;                      // TBNAV_RBPF_OPT_REF_REACH: how far (cells) a scan's brushfire runs before it stops (0: to the end)
  long long ref_reruns = 0;               // proposals run again because a lookup met a pending cell
  long long ref_us[6] = {0, 0, 0, 0, 0, 0}; // host microseconds spent: fetching the logs | RefField::step | resample copies | flushes | before the proposal | the settle look
  int* d_log_pack = nullptr; unsigned long long* d_log_off = nullptr; size_t log_pack_cap = 0, log_off_cap = 0;  // the scan's logs, packed
  int* d_code_src = nullptr;              // [N] slot to copy the field from (rbpf_copy_codes)
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

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true; }
  ~DeviceGuard() { if (ok && prev >= 0) (void)hipSetDevice(prev); }
};

// state layout helpers: the kernels take pose / prev_pose / weight pointers with [N][3] / [N] strides,
// so the 7-double record is split into three arrays inside one allocation.
struct StatePtrs { double *pose, *prev, *weight; };
StatePtrs state_ptrs(double* base, int N) { return {base, base + (size_t)3 * N, base + (size_t)6 * N}; }
MapT map_of(const tbnav_rbpf* h) { return MapT{h->d_table[h->cur], h->d_shed, h->TW, h->TT}; }

// The stored u16 distance field exists only once something needs it: 2 * N * G * 2 bytes (double-buffered like the
// rest of the particle state).  Refused beyond 32 GB — BASELINE configs[4]-sized handles run in query mode only.
int ensure_codes(tbnav_rbpf* h) {
  if (h->d_code[0]) return TBNAV_OK;
  const size_t bytes = sizeof(uint16_t) * h->G * (size_t)h->N;
  if (2 * bytes > ((size_t)32 << 30) || h->N > 65535) return TBNAV_ERR_UNSUPPORTED;
  for (int b = 0; b < 2; ++b) {
    TBNAV_HIP(hipMalloc((void**)&h->d_code[b], bytes));
    TBNAV_HIP(hipMemset(h->d_code[b], 0xFF, bytes));  // occ_dist = max_occ_dist_ (grid_mapper.cpp:49,58)
  }
  TBNAV_HIP(hipMalloc((void**)&h->d_bm_dense, sizeof(unsigned long long) * (size_t)h->N * h->xsize * h->words));
  TBNAV_HIP(hipMalloc((void**)&h->d_rc_dense, sizeof(int) * (size_t)h->N * h->xsize));
  return TBNAV_OK;
}

double logodds_to_prob(double l) { return 1 - (1 / (1 + std::exp(l))); }  // grid_mapper.hpp:27-30 (glibc on the host)

// smallest l with prob(l) >= p_occ, found by bisection on the host (prob is monotone in l)
double find_occ_cut(double l_occ_nominal, double p_occ) {
  double lo = l_occ_nominal - 1.0, hi = l_occ_nominal + 1.0;  // prob(lo) < p_occ <= prob(hi)
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (logodds_to_prob(mid) >= p_occ) hi = mid; else lo = mid;
  }
  return hi;
}

// Break points of the exported map value as a function of the log-odds, found with the HOST libm by bisection
// (the exported value is monotone in l apart from the prob == 0.5 plateau, which maps to -1).
int export_value_host(double l) {  // GridMapper::gridMap after updateCellState, evaluated as the reference does
  const double prob = logodds_to_prob(l);
  if (prob == 0.5) return -1;
  if (prob >= 0.90) return 100;
  if (prob <= 0.35) return 0;
  return (int)(int8_t)(prob * 100);
}
double bisect_first(double lo, double hi, bool (*pred)(double, int), int arg) {  // pred(lo) false, pred(hi) true, monotone
  for (int it = 0; it < 300; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (pred(mid, arg)) hi = mid; else lo = mid;
  }
  return hi;
}
ExportCuts derive_export_cuts(double cut_occ) {
  ExportCuts c{};
  c.occ_cut = cut_occ;
  // largest l with prob <= 0.35: the predecessor of the first l with prob > 0.35
  const double first_above = bisect_first(-3.0, 0.0, [](double l, int) { return logodds_to_prob(l) > 0.35; }, 0);
  c.free_cut = std::nextafter(first_above, -1.0e9);
  c.half_lo = bisect_first(-1.0, 1.0, [](double l, int) { return logodds_to_prob(l) >= 0.5; }, 0);
  const double first_gt = bisect_first(-1.0, 1.0, [](double l, int) { return logodds_to_prob(l) > 0.5; }, 0);
  c.half_hi = std::nextafter(first_gt, -1.0e9);
  c.n_steps = 0;
  for (int k = 36; k <= 89; ++k)  // smallest l (outside the 0.5 plateau) whose exported value is >= k
    c.step[c.n_steps++] = bisect_first(c.free_cut, cut_occ, [](double l, int kk) { const int v = export_value_host(l); return v == 100 || (v >= kk); }, k);
  return c;
}

size_t edt_lds_bytes(int xs, int words, int C) { return (size_t)xs * words * 8 + (size_t)xs * C * 5; }

// the part of ScanC the beam mixture term needs (also what the handle's table of it is built from at create)
bool mixture_consts(const tbnav_rbpf* h, ScanC& c) {
  const tbnav_rbpf_params& P = h->p;
  c.g = GridC{P.xmin, P.xmax, P.ymin, P.ymax, P.resolution, h->xsize, h->ysize, h->words, h->max_occ_dist, 1.0 / P.resolution};
  c.z_hit = P.z_hit;
  c.var_hit = P.sigma_hit * P.sigma_hit;                       // grid_mapper.cpp:77
  if (almost_equal(c.var_hit, 0.0)) return false;
  c.sqrt_inv_hit = 1.0 / std::sqrt(2.0 * kPI * c.var_hit);    // pdfNormal, grid_mapper.cpp:25
  c.rand_term = P.z_rand / P.z_max;                            // grid_mapper.cpp:121
  return true;
}

int build_scan_consts(tbnav_rbpf* h, ScanC& c, const float* scan, int n_beams, const double u[3],
                      const double cur_odom[3], const double prev_odom[3], int icp_ok, const double T_icp[3],
                      std::vector<double2>& beams) {
  const tbnav_rbpf_params& P = h->p;
  c.N = h->N; c.k = h->k; c.icp_ok = icp_ok ? 1 : 0;
  for (int q = 0; q < 3; ++q) { c.Trs[q] = P.Trs[q]; c.Ld[q] = std::sqrt(P.sample_range[q]); c.Lm[q] = std::sqrt(P.motion_noise[q]);
                                c.Ticp[q] = T_icp[q]; c.u[q] = u[q]; }
  if (!mixture_consts(h, c)) return TBNAV_ERR_PDF_VARIANCE;
  c.scan_min = P.scan_likelihood_min; c.scan_max = P.scan_likelihood_max;
  c.pose_min = P.pose_likelihood_min; c.pose_max = P.pose_likelihood_max;
  c.a1 = P.srr; c.a2 = P.srt; c.a3 = P.str_; c.a4 = P.stt;
  // odometry deltas (particle_filter.cpp:393-403), identical for every particle and sample
  c.rot1 = std::atan2(cur_odom[2] - prev_odom[2], cur_odom[1] - prev_odom[1]) - prev_odom[0];
  const double dxo = cur_odom[1] - prev_odom[1], dyo = cur_odom[2] - prev_odom[2];
  c.trans = std::sqrt(dxo * dxo + dyo * dyo);
  c.rot2 = normalize_angle_PI(normalize_angle_PI(cur_odom[0]) - normalize_angle_PI(prev_odom[0]) - c.rot1);
  c.d_free = h->l_free - h->l_prior;
  c.d_occ = h->l_occ - h->l_prior;
  c.cut_occ = h->cut_occ;
  c.stride_normals = icp_ok ? 3 * h->k + 3 : 3;
  c.p0 = 0;
  // valid beams in the sensor frame, sensor_model.cpp:73-108 (float limits, double angle accumulation)
  // (the angle of beam i does not depend on the scan: its cosine and sine — glibc's, in the reference's accumulation
  //  order — are kept from one call to the next; 2 x 360 libm calls were a tenth of the host's time per scan)
  if ((int)h->beam_cs.size() != n_beams) {
    h->beam_cs.resize(n_beams);
    double beam_angle = P.beam_min;
    for (int i = 0; i < n_beams; ++i) {
      h->beam_cs[i] = double2{std::cos(beam_angle), std::sin(beam_angle)};
      beam_angle += P.beam_delta;
      if (P.beam_max < 0.0 && beam_angle <= P.beam_max) beam_angle = P.beam_min;
      else if (P.beam_max >= 0.0 && beam_angle >= P.beam_max) beam_angle = P.beam_min;
    }
  }
  beams.clear();
  c.rmax = 0.0;
  for (int i = 0; i < n_beams; ++i) {
    const double range = scan[i];
    if (range >= P.range_min && range < P.range_max) {
      beams.push_back(double2{range * h->beam_cs[i].x, range * h->beam_cs[i].y});
      c.rmax = std::max(c.rmax, range);
    }
  }
  c.Bv = (int)beams.size();
  return TBNAV_OK;
}

int status_from_err(const int err[4]) {
  // (first: a scan whose beam table never arrived has computed nothing — whatever else is raised would be an artefact of that)
  if (err[3] & 16) { tbnav::last_hip_error_slot() = "rbpf_propose: the scan's beam table never reached the device (the launch's leading workgroup never published it)"; return TBNAV_ERR_HIP; }
  if (err[0]) return TBNAV_ERR_OUT_OF_WORLD;
  if (err[2]) return TBNAV_ERR_PDF_VARIANCE;
  if (err[1]) return TBNAV_ERR_ETA_ZERO;
  if (err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;  // no free log-odds tile left (the scan of that particle was not applied)
  if (err[3] & 4) return TBNAV_ERR_UNSUPPORTED;  // a likelihood lookup left the particle's refreshed window (cannot happen: see rbpf_window)
  if (err[3]) return TBNAV_ERR_BRESENHAM;
  return TBNAV_OK;
}

// The three tiers of the exact distance transform for particles [p0, p0 + count): windowed (tiles_x = the
// tiles a window can span) or whole-map (tiles_x = every tile; the caller has set win/skip accordingly).
int run_distance_field(tbnav_rbpf* h, const GridC& g, int p0, int count, int tiles64) {
  hipStream_t st = h->stream;
  // dense bitmap rows + row counts of these particles, from their tiles
  hipLaunchKernelGGL(rbpf_densify, dim3((h->xsize + 3) / 4, count), dim3(256), 0, st, g, p0, h->pool, map_of(h), h->d_trow[h->cur],
                     h->d_bm_dense, h->d_rc_dense);
  TBNAV_HIP(hipGetLastError());
  // tier 0: <= kEdtRowsA non-empty rows, tier 1: <= kEdtRowsB, tier 2: the general kernel (decided on the device)
  TBNAV_HIP(hipMemsetAsync(h->d_tier + p0, 0, sizeof(int) * count, st));
  const EdtJob job{h->d_win, h->d_skip, p0};
  const dim3 gridc(tiles64, count);
  hipLaunchKernelGGL(rbpf_edt_compact<kEdtRowsA>, gridc, dim3(kWave), edt_compact_lds(kEdtRowsA), st, g, h->radius,
                     h->d_bm_dense, h->d_rc_dense, h->d_code[h->cur], h->d_tier, 0, job);
  TBNAV_HIP(hipGetLastError());
  hipLaunchKernelGGL(rbpf_edt_compact<kEdtRowsB>, gridc, dim3(kWave), edt_compact_lds(kEdtRowsB), st, g, h->radius,
                     h->d_bm_dense, h->d_rc_dense, h->d_code[h->cur], h->d_tier, 1, job);
  TBNAV_HIP(hipGetLastError());
  const int C = h->edt_cols;
  const size_t lds = edt_lds_bytes(h->xsize, h->words, C);
  const dim3 grid(tiles64 * (kWave / C), count);
  if (C == 64) hipLaunchKernelGGL(rbpf_edt<64>, grid, dim3(64), lds, st, g, h->radius, h->d_bm_dense, h->d_code[h->cur], h->d_tier, 2, job);
  else hipLaunchKernelGGL(rbpf_edt<32>, grid, dim3(32), lds, st, g, h->radius, h->d_bm_dense, h->d_code[h->cur], h->d_tier, 2, job);
  TBNAV_HIP(hipGetLastError());
  return TBNAV_OK;
}

GridC grid_of(const tbnav_rbpf* h) {
  return GridC{h->p.xmin, h->p.xmax, h->p.ymin, h->p.ymax, h->p.resolution, h->xsize, h->ysize, h->words, h->max_occ_dist, 1.0 / h->p.resolution};
}

// Whole-field refresh of ONE particle, on demand (state 2 afterwards).
int ensure_full_field(tbnav_rbpf* h, int particle) {
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  hipStream_t st = h->stream;
  int stt = 0;
  TBNAV_HIP(hipStreamSynchronize(st));
  TBNAV_HIP(hipMemcpy(&stt, h->d_fstate + particle, sizeof(int), hipMemcpyDeviceToHost));
  if (stt == 2) return TBNAV_OK;
  const int zero = 0, two = 2;
  if (h->edt_cols == 0) {
    const GridC g = grid_of(h);
    hipLaunchKernelGGL(rbpf_field_by_query, dim3((unsigned)((h->G + 255) / 256)), dim3(256), 0, st, g, h->radius, particle,
                       h->pool, map_of(h), h->d_trow[h->cur], h->d_code[h->cur]);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipStreamSynchronize(st));
    TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
    h->fstate_dirty = true;
    return TBNAV_OK;
  }
  const int4 full = make_int4(0, h->xsize - 1, 0, h->ysize - 1);
  TBNAV_HIP(hipMemcpy(h->d_win + particle, &full, sizeof full, hipMemcpyHostToDevice));
  TBNAV_HIP(hipMemcpy(h->d_skip + particle, &zero, sizeof zero, hipMemcpyHostToDevice));
  int rc = run_distance_field(h, grid_of(h), particle, 1, (h->ysize + kWave - 1) / kWave);
  if (rc != TBNAV_OK) return rc;
  TBNAV_HIP(hipStreamSynchronize(st));
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
  h->fstate_dirty = true;
  return TBNAV_OK;
}

// lowVarianceResampling's copies on the device: d_parent holds the parent of every slot and, behind them, how many slots
// chose each particle.  One launch (rbpf_resample_apply): tables and reference counts, and state / counts / field state
// into the alternate buffers (the occupancy bits travel with the tiles: nothing of map size is copied).
int resample_on_device(tbnav_rbpf* h) {
  const int N = h->N, nxt = 1 - h->cur;
  hipStream_t st = h->stream;
  const size_t n = (size_t)N * h->TT;
  const int blocks = (int)std::min<size_t>((n + kResampleThreads - 1) / kResampleThreads, 8192);
  const size_t work = h->d_code[0] ? h->G / 4 : (size_t)0;
  const int chunks = (int)std::min<size_t>(std::max<size_t>(work / 2048, 1), 64);
  const GatherArgs ga{h->G, h->TW, h->d_state[h->cur], h->d_state[nxt], h->d_trow[h->cur], h->d_trow[nxt], h->d_nocc[h->cur], h->d_nocc[nxt],
                      h->d_fstate, h->d_fstate_alt, h->d_code[h->cur], h->d_code[nxt], (h->df_mode != 2 || h->ref_field) ? 1 : 0};
  hipLaunchKernelGGL(rbpf_resample_apply, dim3(blocks + N * chunks), dim3(kResampleThreads), 0, st, N, h->TT, h->d_parent, h->d_parent + N,
                     h->d_table[h->cur], h->d_table[nxt], h->d_shed, h->pool, blocks, chunks, ga);
  TBNAV_HIP(hipGetLastError());
  std::swap(h->d_fstate, h->d_fstate_alt);
  h->cur = nxt;
  return TBNAV_OK;
}

// ---- reference distance-field mode (ref_field.hpp) ------------------------------------------------------------
// Before the raycast: a log big enough for every cell update of the scan (a cell can enter and leave the occupied set
// more than once in one scan).
int ref_field_prepare_log(tbnav_rbpf* h, int Bv, OccLog& log) {
  const double reach = (double)h->p.range_max + std::hypot(h->p.Trs[1], h->p.Trs[2]);
  const long per_ray = (long)std::ceil(reach / h->p.resolution) + 4;
  const long cap = (long)std::max(Bv, 1) * per_ray;
  if ((size_t)cap * h->N * sizeof(int) > ((size_t)1 << 30)) return TBNAV_ERR_UNSUPPORTED;
  if (cap > h->log_cap) {
    (void)hipFree(h->d_log_ev); h->d_log_ev = nullptr; h->log_cap = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_log_ev, sizeof(int) * (size_t)cap * h->N));
    h->log_cap = (int)cap;
  }
  if (!h->d_log_cnt) TBNAV_HIP(hipMalloc((void**)&h->d_log_cnt, sizeof(int) * h->N));
  TBNAV_HIP(hipMemsetAsync(h->d_log_cnt, 0, sizeof(int) * h->N, h->stream));
  log = OccLog{h->d_log_ev, h->d_log_cnt, h->log_cap};
  return TBNAV_OK;
}
// After the scan (and its resample, if one fired): replay the logged set changes, run the reference's brushfire for
// every particle as it was BEFORE the resample (the reference integrates the scan in the particle loop and resamples
// afterwards, particle_filter.cpp:158-249), copy like the resample did, and make the result the authoritative field.
// the particles' logged sequences packed back to back (one copy to the host instead of one per particle)
__global__ __launch_bounds__(256) void rbpf_pack_logs(const int* __restrict__ ev, int log_cap, int p_first, const unsigned long long* __restrict__ off,
                                                      int* __restrict__ out) {
  const int i = blockIdx.x;
  const unsigned long long o = off[i], n = off[i + 1] - o;
  const int* src = ev + (size_t)(p_first + i) * log_cap;
  for (unsigned long long q = threadIdx.x; q < n; q += blockDim.x) out[o + q] = src[q];
}
// slot p takes the field slot src[p] holds (src[p] == p: keep) — the particles that share a state with one whose whole image was uploaded
__global__ __launch_bounds__(256) void rbpf_copy_codes(uint16_t* __restrict__ code, size_t G, const int* __restrict__ src) {
  const int p = blockIdx.y, q = src[p];
  if (q == p) return;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  if ((G & 7) == 0) {  // (every slot starts on a 16-byte boundary)
    const uint4* s = reinterpret_cast<const uint4*>(code + (size_t)q * G);
    uint4* d = reinterpret_cast<uint4*>(code + (size_t)p * G);
    for (size_t i = i0; i < G / 8; i += step) d[i] = s[i];
  } else {
    for (size_t i = i0; i < G; i += step) code[(size_t)p * G + i] = code[(size_t)q * G + i];
  }
}
// The journal of field slot p (ref_field.hpp, plan_flush): optionally "everything pending" first, then `count` (cell, code) pairs —
// every cell at most once per launch.  One workgroup per slot.
__global__ __launch_bounds__(256) void rbpf_field_journal(uint16_t* __restrict__ code, size_t G, const uint3* __restrict__ jobs, const uint2* __restrict__ entries) {
  const int p = blockIdx.x;
  const uint3 j = jobs[p];
  if (!j.y && !j.z) return;
  uint16_t* const slot = code + (size_t)p * G;
  if (j.z) {
    const unsigned int fill = (unsigned int)kCodePending * 0x10001u;
    if ((G & 7) == 0) {
      uint4* d = reinterpret_cast<uint4*>(slot);
      for (size_t i = threadIdx.x; i < G / 8; i += blockDim.x) d[i] = make_uint4(fill, fill, fill, fill);
    } else {
      for (size_t i = threadIdx.x; i < G; i += blockDim.x) slot[i] = kCodePending;
    }
    __threadfence();
    __syncthreads();
  }
  for (unsigned int e = threadIdx.x; e < j.y; e += blockDim.x) {
    const uint2 v = entries[j.x + e];
    slot[v.x] = (uint16_t)v.y;
  }
}
// Bring the device's field slots to where the host's states are: whole images for slots whose content is unknown (or whose state
// has become exact and complete), journal ranges for the others.  Synchronises the stream (the plan's host buffers are read).
struct UsTimer {   // adds the enclosing scope's wall time to a counter
  long long& acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit UsTimer(long long& a) : acc(a) {}
  ~UsTimer() { acc += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
int ref_field_flush(tbnav_rbpf* h) {
  UsTimer ut(h->ref_us[3]);
  const int N = h->N;
  hipStream_t st = h->stream;
  tbnav::RefField::Flush& f = h->ref_flush;
  h->ref->plan_flush(f);
  if (f.dense_slot.empty() && !f.any_job) return TBNAV_OK;
  bool any_copy = false;
  std::vector<int> src;
  for (size_t q = 0; q < f.dense_slot.size(); ++q) {
    if (f.dense_img[q] >= 0)
      TBNAV_HIP(hipMemcpyAsync(h->d_code[h->cur] + (size_t)f.dense_slot[q] * h->G, f.images[f.dense_img[q]].data(), sizeof(uint16_t) * h->G, hipMemcpyHostToDevice, st));
    else {
      if (src.empty()) { src.resize(N); for (int p = 0; p < N; ++p) src[p] = p; }
      src[f.dense_slot[q]] = f.dense_src[q];
      any_copy = true;
    }
  }
  if (any_copy) {
    if (!h->d_code_src) TBNAV_HIP(hipMalloc((void**)&h->d_code_src, sizeof(int) * N));
    TBNAV_HIP(hipMemcpyAsync(h->d_code_src, src.data(), sizeof(int) * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rbpf_copy_codes, dim3(64, N), dim3(256), 0, st, h->d_code[h->cur], h->G, h->d_code_src);
    TBNAV_HIP(hipGetLastError());
  }
  if (f.any_job) {
    static_assert(sizeof(tbnav::RefField::JEntry) == sizeof(uint2) && sizeof(tbnav::RefField::Flush::Job) == sizeof(uint3), "the plan's records are what the kernel reads");
    if (f.entries.size() > h->jentries_cap) {
      TBNAV_HIP(hipStreamSynchronize(st));
      (void)hipFree(h->d_jentries); h->d_jentries = nullptr; h->jentries_cap = 0;
      const size_t cap = f.entries.size() + f.entries.size() / 2 + 4096;
      TBNAV_HIP(hipMalloc((void**)&h->d_jentries, sizeof(uint2) * cap));
      h->jentries_cap = cap;
    }
    if (!h->d_jjobs) TBNAV_HIP(hipMalloc((void**)&h->d_jjobs, sizeof(uint3) * N));
    if (!f.entries.empty()) TBNAV_HIP(hipMemcpyAsync(h->d_jentries, f.entries.data(), sizeof(uint2) * f.entries.size(), hipMemcpyHostToDevice, st));
    TBNAV_HIP(hipMemcpyAsync(h->d_jjobs, f.jobs.data(), sizeof(uint3) * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rbpf_field_journal, dim3(N), dim3(256), 0, st, h->d_code[h->cur], h->G, h->d_jjobs, h->d_jentries);
    TBNAV_HIP(hipGetLastError());
  }
  TBNAV_HIP(hipStreamSynchronize(st));
  return TBNAV_OK;
}
// The whole field of one particle as the reference holds it, on the device (exports, the one-particle entry points): the pass is run
// to the end, stale cells are recovered by replaying the lineage where they are not known (ref_field.hpp).
int ref_field_materialize(tbnav_rbpf* h, int particle) {
  if (!h->ref->codes(particle)) {
    tbnav::last_hip_error_slot() = "reference-field mode: a whole field was asked for whose stale cells need history beyond the history budget";
    return TBNAV_ERR_UNSUPPORTED;
  }
  const int rc = ref_field_flush(h);
  if (rc != TBNAV_OK) return rc;
  const int two = 2;
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
  h->fstate_dirty = true;
  return TBNAV_OK;
}
// Before the proposal of a scan: the slots in step with the states (imports and exports since the last scan), the particle state
// kept for a second run, the pending flags cleared.
int ref_field_before_propose(tbnav_rbpf* h) {
  UsTimer ut(h->ref_us[4]);
  const int N = h->N;
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  if (!h->d_pend) {
    TBNAV_HIP(hipMalloc((void**)&h->d_pend, sizeof(int) * N));
    TBNAV_HIP(hipHostMalloc((void**)&h->h_pend, sizeof(int) * N, hipHostMallocDefault));
    TBNAV_HIP(hipMalloc((void**)&h->d_state_snap, sizeof(double) * 7 * N));
  }
  if (h->sm_on) {   // the per-particle scan matcher reads whole fields: every pass to the end (the option is not the reference's filter)
    for (int p = 0; p < N; ++p) if (!h->ref->codes(p)) return TBNAV_ERR_UNSUPPORTED;
  }
  { const int rc = ref_field_flush(h); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_fstate), 2, N, h->stream));
  h->fstate_dirty = true;
  TBNAV_HIP(hipMemcpyAsync(h->d_state_snap, h->d_state[h->cur], sizeof(double) * 7 * N, hipMemcpyDeviceToDevice, h->stream));
  TBNAV_HIP(hipMemsetAsync(h->d_pend, 0, sizeof(int) * N, h->stream));
  return TBNAV_OK;
}
// After the proposal: did a lookup land on a cell its particle's pass has not written?  Then exactly those states are resumed on the
// host (RefField::ensure), the new cells go to the device, and the proposal runs again from the kept particle state — until none
// does.  (A closed room never gets here: its beams end within a cell or two of the obstacles the last scans integrated.)
template <class Relaunch>
int ref_field_settle(tbnav_rbpf* h, int* h_err, Relaunch relaunch) {
  const int N = h->N;
  hipStream_t st = h->stream;
  std::vector<int> ps, cs;
  for (int round = 0; round < 4096; ++round) {
    {
      UsTimer ut(h->ref_us[5]);
      TBNAV_HIP(hipMemcpyAsync(h->h_pend, h->d_pend, sizeof(int) * N, hipMemcpyDeviceToHost, st));
      TBNAV_HIP(hipStreamSynchronize(st));
    }
    ps.clear(); cs.clear();
    for (int p = 0; p < N; ++p) if (h->h_pend[p]) { ps.push_back(p); cs.push_back(h->h_pend[p] - 1); }
    if (ps.empty()) return TBNAV_OK;
    const int rc = h->ref->ensure(ps.data(), cs.data(), (int)ps.size(), h->host_threads);
    if (rc == -1) {
      tbnav::last_hip_error_slot() = "reference-field mode: a lookup needs a stale cell whose history is beyond the history budget";
      return TBNAV_ERR_UNSUPPORTED;
    }
    if (rc != 0) { tbnav::last_hip_error_slot() = "reference-field mode: a replayed pass differs from the pass it re-ran (internal error)"; return TBNAV_ERR_HIP; }
    { const int rf = ref_field_flush(h); if (rf != TBNAV_OK) return rf; }
    TBNAV_HIP(hipMemcpyAsync(h->d_state[h->cur], h->d_state_snap, sizeof(double) * 7 * N, hipMemcpyDeviceToDevice, st));
    TBNAV_HIP(hipMemsetAsync(h->d_pend, 0, sizeof(int) * N, st));
    for (int q = 0; q < 4; ++q) h_err[q] = 0;   // (mapped; the stream is idle)
    ++h->ref_reruns;
    const int rl = relaunch();
    if (rl != TBNAV_OK) return rl;
  }
  return TBNAV_ERR_UNSUPPORTED;
}
int ref_field_after_scan(tbnav_rbpf* h, bool resampled, int p_first = 0, int p_count = -1) {
  const int N = h->N;
  if (p_count < 0) p_count = N;
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  hipStream_t st = h->stream;
  TBNAV_HIP(hipStreamSynchronize(st));
  auto t_log0 = std::chrono::steady_clock::now();
  std::vector<int> cnt(N);
  TBNAV_HIP(hipMemcpy(cnt.data(), h->d_log_cnt, sizeof(int) * N, hipMemcpyDeviceToHost));
  // the logs: packed on the device, ONE copy (a few thousand events per particle; one small copy each was 10-20 ms per 1000)
  std::vector<size_t> off((size_t)p_count + 1, 0);
  for (int i = 0; i < p_count; ++i) {
    if (cnt[p_first + i] > h->log_cap) return TBNAV_ERR_UNSUPPORTED;  // cannot happen: the log holds every cell update
    off[i + 1] = off[i] + (size_t)cnt[p_first + i];
  }
  const size_t total = off[p_count];
  std::vector<int> all(total ? total : 1);
  if (total) {
    if (total > h->log_pack_cap || (size_t)p_count + 1 > h->log_off_cap) {
      (void)hipFree(h->d_log_pack); (void)hipFree(h->d_log_off); h->d_log_pack = nullptr; h->d_log_off = nullptr; h->log_pack_cap = h->log_off_cap = 0;
      const size_t cap = total + total / 2, ocap = (size_t)N + 1;
      TBNAV_HIP(hipMalloc((void**)&h->d_log_pack, sizeof(int) * cap));
      TBNAV_HIP(hipMalloc((void**)&h->d_log_off, sizeof(unsigned long long) * ocap));
      h->log_pack_cap = cap; h->log_off_cap = ocap;
    }
    std::vector<unsigned long long> off64(off.begin(), off.end());
    TBNAV_HIP(hipMemcpy(h->d_log_off, off64.data(), sizeof(unsigned long long) * off64.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rbpf_pack_logs, dim3(p_count), dim3(256), 0, st, h->d_log_ev, h->log_cap, p_first, h->d_log_off, h->d_log_pack);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipMemcpyAsync(all.data(), h->d_log_pack, sizeof(int) * total, hipMemcpyDeviceToHost, st));
    TBNAV_HIP(hipStreamSynchronize(st));
  }
  // one replay + brushfire per distinct (state, sequence) — ref_field.hpp — side by side on the host's cores: inside one state the
  // order of every set and heap operation is the reference's
  h->ref_us[0] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_log0).count();
  h->ref->set_reach(h->ref_reach);
  { UsTimer ut(h->ref_us[1]); h->ref->step(p_first, p_count, h->host_threads, all.data(), off.data()); }
  if (resampled) {
    UsTimer ut(h->ref_us[2]);
    h->h_parent.resize(N);
    TBNAV_HIP(hipMemcpy(h->h_parent.data(), h->d_parent, sizeof(int) * N, hipMemcpyDeviceToHost));
    h->ref->resample(h->h_parent.data());   // (the device's gather has moved the field slots the same way: resample_on_device)
  }
  // to the device: what each pass wrote, as a journal on top of the parent's image the slot holds
  { const int rc = ref_field_flush(h); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_fstate), 2, N, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  h->fstate_dirty = true;
  return TBNAV_OK;
}

// GridMapper::integrateScan's map update (grid_mapper.cpp:140-178) for particles [c.p0, c.p0 + count) at their poses.
// sens: the sensor transforms the proposal kernel left for exactly these poses (NULL: the raycast derives them).
// nz (optional): the weights' normalise / select step to run with this update — inside the box-counter kernel's launch as
// workgroup 0 (no second stream, no event), behind the other map-update kernels as a launch of its own.
int launch_raycast(tbnav_rbpf* h, const ScanC& c, int count, const double* sens, const NormArgs* nz = nullptr, int* err = nullptr,
                   const double2* beams_dev = nullptr) {
  if (!err) err = h->d_err;
  if (!beams_dev) beams_dev = h->d_beams;
  const int* gp = nz ? nz->gate_prev : nullptr;
  hipStream_t st = h->stream;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  const int bvn = c.Bv > 0 ? c.Bv : 1;
  const MapT M = map_of(h);
  int nt = h->raycast_threads;
  const bool nt_auto = nt == 0;
  if (nt == 0) nt = 1024;
  // rbpf_raycast_box: one u32 per cell of the box, the box padded to whole groups of 8 cells along y
  long cap_win = 0, cap4 = 0;
  if (h->tile_cap > 0) {
    // every end point lies within `reach` of the robot's position: at most floor(2 reach / res) + 2 rows or columns (+1 spare);
    // along y the box is padded to whole pairs of cells
    const double reach = c.rmax + std::hypot(h->p.Trs[1], h->p.Trs[2]);
    const long side = (long)std::floor(2.0 * reach / h->p.resolution) + 3;
    cap_win = (side * ((side + 2) & ~1L) + 7) & ~7L;
    // ... but no more than lets TWO workgroups share a CU's 160 KB (the kernel works a larger box through in bands of rows;
    // at least one padded row must fit)
    const long cap_fit = ((78L * 1024 - (long)box_lds_bytes(0, (size_t)bvn) - (long)kBoxStaticLds) / 4) & ~7L;
    if (cap_win > cap_fit) cap_win = std::max(cap_fit, (side + 9) & ~7L);
    // (test hook: at most about this many rows of the box per band, to drive the band loop on small maps)
    if (h->raycast_band_rows > 0) cap_win = std::min(cap_win, (h->raycast_band_rows * ((side + 2) & ~1L) + 7) & ~7L);
    // ... and no more than the particles' boxes needed lately (+ 1/8 + 512 words for what a scan's motion changes): the bound
    // above is the scan's longest beam in every direction from every pose, a room's box is a fraction of that — the array is
    // what keeps a CU at two workgroups.  A box that outgrows the guess costs its particle a second band, not correctness.
    const int need = (h->raycast_adapt && h->h_box_need) ? *reinterpret_cast<volatile int*>(h->h_box_need) : 0;
    if (need > 0) {
      const long want = ((long)need + need / 8 + 512 + 7) & ~7L;
      cap_win = std::min(cap_win, std::max(want, (side + 9) & ~7L));
      // FOUR 512-thread workgroups per CU — every one of 1000 particles resident at once instead of 768 and a second, partial
      // round (round 4; measured 24.6 / 30.8 / 37.1 us with 1 / 2 / 3 workgroups per CU, 48.3 for 1000 particles on 768 slots) —
      // when the boxes' need plus a margin of three rows or so fits a quarter of the CU's LDS.  The margin is tighter than the
      // three-per-CU form's 1/8 + 512: a box that outgrows it costs its particle a second band, never correctness.
      cap4 = std::max(((long)need + 256 + 7) & ~7L, (side + 9) & ~7L);
    }
  }
  // Residency the launch gets: R workgroups per CU need R x (dynamic + static LDS) <= 160 KB; 4 and 3 per CU are 512-thread
  // workgroups (64 / 80 registers a lane), 2 per CU 1024 threads (64).  Two cell formats: 32-bit words (slot in the word) and,
  // when that buys a higher residency, 16-bit words + a slot table (C16: a few more instructions per walk step).  Measured at
  // cfg3, N = 1000 / 4000, 32-bit words: 1024 x 2: 55.8 / 191 us; 512 x 2: 54.4 / 199; 512 x 3: 47.9 / 163; 512 x 4 (four
  // events a slot instead of eight: what lets the bench room's 8272-cell boxes in): 41.7 / 143.
  const size_t nz_lds = nz ? sizeof(double) * 2 * kNormChunk : (size_t)0;
  auto fits = [&](int R, size_t bytes) { return (size_t)R * (std::max(bytes, nz_lds) + kBoxStaticLds) <= (size_t)kMaxLds; };
  const bool may4 = h->raycast_adapt != 2 && cap4 > 0 && cap4 <= cap_win;
  int wps = 8;
  bool c16 = false;
  const bool pick = nt_auto || nt == 512;
  const bool c16_ok = h->raycast_cell16 != 0 && cap_win > 0 && cap_win < 65528 && c.Bv + 64 < 32768;
  // (measured, N = 1000: the 16-bit form costs ~15 % at EQUAL residency — 57.2 against 49.8 us at three per CU: twice the
  //  same-dword collisions of the LDS adds where rays converge, the sub-word arithmetic of every step — so going from three to
  //  four per CU with it loses, 51.5 against 49.8 us, and it is used only where the 32-bit form is stuck at TWO per CU: the
  //  SURVEY room's 13 860-cell boxes, 63.7 -> 60.7 us)
  // (four per CU: with eight events a slot where they fit, with four where only they do — measured in a 3 x 1.9 m room whose cells take
  //  5-10 events: 39.4 us with four-event slots against 37.7 with three workgroups per CU and eight)
  size_t ev_slot = kBoxEv;  // (what the instantiation launched below holds per slot)
  const bool force4 = h->raycast_adapt == 3;  // (tests: four-event slots wherever four workgroups fit)
  if (pick && may4 && !force4 && fits(4, box_lds_bytes((size_t)cap4, (size_t)bvn, kBoxEv))) { nt = 512; cap_win = cap4; }
  else if (pick && may4 && fits(4, box_lds_bytes((size_t)cap4, (size_t)bvn, kBoxEvFour))) { nt = 512; cap_win = cap4; ev_slot = kBoxEvFour; }
  else if (pick && cap_win > 0 && fits(3, box_lds_bytes((size_t)cap_win, (size_t)bvn))) { nt = 512; wps = 6; }
  else if (pick && may4 && c16_ok && fits(4, box16_lds_bytes((size_t)cap4, (size_t)bvn, kBoxEv))) { nt = 512; cap_win = cap4; c16 = true; }
  // (no 16-bit form with four-event slots: its niche — boxes that fit four per CU only with BOTH economies — is a few hundred cells wide, and
  //  the instantiation was the one map-update kernel left with a spill; such boxes run three per CU in <512, 6, true, 8>)
  else if (pick && c16_ok && fits(3, box16_lds_bytes((size_t)cap_win, (size_t)bvn))) { nt = 512; wps = 6; c16 = true; }
  else if (nt == 512) wps = 6;
  if (h->raycast_cell16 == 2 && c16_ok && nt == 512) c16 = true;   // (tests / A-B: the 16-bit form wherever it can run)
  if (c16) ev_slot = kBoxEv;   // (the 16-bit form has eight-event slots only)
  const size_t lds_win = c16 ? box16_lds_bytes((size_t)cap_win, (size_t)bvn, ev_slot) : box_lds_bytes((size_t)cap_win, (size_t)bvn, ev_slot);
  if (cap_win > 0 && !h->ref_field && c.Bv < 32768 - kWave && nt >= 512 && lds_win <= (size_t)kMaxLds - 4096) {
    // default: box counters (rbpf_raycast_box)
    unsigned long long* touched = h->count_touched ? h->d_touched : nullptr;
    const NormArgs na = nz ? *nz : NormArgs{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr};
    const int blocks = count + (nz ? 1 : 0);
    const size_t lds_launch = nz ? std::max(lds_win, sizeof(double) * 2 * kNormChunk) : lds_win;  // (workgroup 0's two arrays)
    const int need_slot = (int)(h->rc_launches++ % 3u);
    h->lk_raycast = nt == 512 ? 512 : 1024; h->lk_raycast_wps = wps; h->lk_raycast_c16 = c16 ? 1 : 0; h->lk_raycast_ev = (int)ev_slot; h->lk_raycast_grid = blocks;
    h->lk_box_cap = (int)cap_win; h->lk_box_need = (h->raycast_adapt && h->h_box_need) ? *reinterpret_cast<volatile int*>(h->h_box_need) : 0;
    const int hash_words = c16 ? (int)box16_hash_words((size_t)bvn) : 0;
#define TBNAV_BOX(NT_, WPS_, C16_, EV_) hipLaunchKernelGGL((rbpf_raycast_box<NT_, WPS_, C16_, EV_>), dim3(blocks), dim3(NT_), lds_launch, st, c, h->pool, M, beams_dev, sp.pose, sens, \
                                                h->d_trow[h->cur], h->d_nocc[h->cur], err, (int)cap_win, touched, na, h->d_box_need, h->d_box_need_host, need_slot, hash_words)
    const bool ev4 = ev_slot == (size_t)kBoxEvFour;
    if (nt == 512 && wps == 8 && c16) TBNAV_BOX(512, 8, true, 8);
    else if (nt == 512 && wps == 8 && ev4) TBNAV_BOX(512, 8, false, 4);
    else if (nt == 512 && wps == 8) TBNAV_BOX(512, 8, false, 8);
    else if (nt == 512 && c16) TBNAV_BOX(512, 6, true, 8);
    else if (nt == 512) TBNAV_BOX(512, 6, false, 8);
    else TBNAV_BOX(1024, 8, false, 8);
#undef TBNAV_BOX
    TBNAV_HIP(hipGetLastError());
    return TBNAV_OK;
  }
  {
    // beam-ordered kernel: scans the LDS tile cannot hold, and the reference distance-field mode (it logs the
    // occupied-set changes in the reference's order)
    OccLog log{nullptr, nullptr, 0};
    if (h->ref_field) {
      const int rc2 = ref_field_prepare_log(h, c.Bv, log);
      if (rc2 != TBNAV_OK) return rc2;
    }
    h->lk_raycast = 0; h->lk_raycast_grid = count;
    hipLaunchKernelGGL(rbpf_raycast, dim3(count), dim3(kWave), sizeof(int) * (2 * bvn + (h->TT + 31) / 32), st, c, h->pool, M, beams_dev,
                       sp.pose, h->d_trow[h->cur], h->d_nocc[h->cur], err, log, gp);
  }
  TBNAV_HIP(hipGetLastError());
  if (nz) {
    hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, st, nz->N, nz->zp, nz->weight, nz->weight_out, nz->cs, nz->parent, nz->out,
                       nz->gate, nz->gate_prev, nz->seq, nz->seq_val, nz->children);
    TBNAV_HIP(hipGetLastError());
  }
  return TBNAV_OK;
}

// the scan's valid beams into d_beams (shared by slam_impl and the one-particle entry points)
int upload_beams(tbnav_rbpf* h, const std::vector<double2>& beams, int n_beams, int Bv, bool stage_only = false, int slot = 0) {
  if (n_beams > h->max_beams) {
    TBNAV_HIP(hipStreamSynchronize(h->stream));  // (a scan still in flight reads the buffers about to go)
    (void)hipFree(h->d_beams);
    (void)hipHostFree(h->h_beams);
    h->d_beams = nullptr; h->h_beams = nullptr; h->max_beams = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_beams, sizeof(double2) * n_beams));
    TBNAV_HIP(hipHostMalloc((void**)&h->h_beams, sizeof(double2) * n_beams * kScanSlots, hipHostMallocDefault));
    h->max_beams = n_beams;
  }
  if (Bv) {
    double2* hb = h->h_beams + (size_t)slot * h->max_beams;
    std::memcpy(hb, beams.data(), sizeof(double2) * Bv);
    if (!stage_only) TBNAV_HIP(hipMemcpyAsync(h->d_beams, hb, sizeof(double2) * Bv, hipMemcpyHostToDevice, h->stream));
  }
  return TBNAV_OK;
}

// One scan = scan_enqueue (everything up to and including the map update, on the handle's stream) + scan_finish (wait,
// read the stats, run the resampling copies if the scan decided to resample).  `slot`: which of the kScanSlots result slots
// the scan owns.  gate_prev (device pointer or NULL): the resampling decision of the scan enqueued before this one, when the
// host has not seen it yet — the kernels of this scan do nothing if it is set (tbnav_rbpf_slam_batch).
struct ScanTicket { int slot = 0; int n_valid = 0; bool local_only = false; bool poll = false; unsigned int seq = 0; };
// A scan whose constants, beam table and noise are on the device already (tbnav_rbpf_slam_batch prepares a few scans at a time).
struct Prefetched { ScanC c; int rc = TBNAV_OK; const double2* d_beams = nullptr; const double* d_normals = nullptr; };
int scan_enqueue(tbnav_rbpf* h, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* normals,
                 tbnav_rbpf_stats* out, bool local_only, int slot, const int* gate_prev, ScanTicket& tk,
                 const Prefetched* pre = nullptr, hipEvent_t weights_ready = nullptr) {
  hipStream_t st = h->stream;
  int* const d_err = h->d_err + 4 * slot;
  int* const h_err = h->h_err + 4 * slot;
  ++h->scans_done;
  ScanC c;
  std::vector<double2>& beams = h->beams_tmp;  // (kept between calls: no allocation per scan)
  int rc;
  if (pre) { c = pre->c; rc = pre->rc; }
  else rc = build_scan_consts(h, c, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, beams);
  std::memset(out, 0, sizeof *out);
  if (rc != TBNAV_OK) { out->status = rc; return rc; }
  out->n_valid_beams = c.Bv;
  tk.slot = slot; tk.n_valid = c.Bv; tk.local_only = local_only;
  if (!pre) {
    rc = upload_beams(h, beams, n_beams, c.Bv, /*stage_only=*/normals == nullptr, slot);  // device noise: the noise kernel carries the beams over
    if (rc != TBNAV_OK) return rc;
  }
  const size_t n_norm = (size_t)h->N * c.stride_normals + 1;
  // device noise: drawn inside the proposal kernel, whose leading workgroup also carries the beam table over (NoiseSrc) — unless a
  // kernel in front of it needs the table on the device (the per-particle scan matcher), the scan comes prepared with its chunk
  // (tbnav_rbpf_slam_batch), or the option is off: then rbpf_sample_normals stores the same values first, as up to round 4
  // (... or the reference-field mode may have to run the proposal twice: the stored stream is there for the second run)
  bool dn = !pre && !normals && !(h->sm_on && c.icp_ok) && h->noise_in_kernel == 1 && !h->ref_field;
  NoiseSrc ns{};
  if (dn) {
    // (the hand-over needs fine-grained device memory: where the platform does not give any, the option switches itself off for good —
    //  the stored-first form computes the same values)
    hipError_t ea = hipSuccess;
    if (!h->d_zslot) {
      ea = hipMalloc((void**)&h->d_zslot, sizeof(double));
      if (ea == hipSuccess) ea = hipExtMallocWithFlags((void**)&h->d_beam_ready, sizeof(unsigned int) * kReadyCopies * kReadyStride, hipDeviceMallocFinegrained);
      if (ea == hipSuccess) ea = hipMemsetAsync(h->d_beam_ready, 0, sizeof(unsigned int) * kReadyCopies * kReadyStride, st);
      h->beam_seq = 0;
    }
    if (ea == hipSuccess && h->fg_beams_cap < h->max_beams) {
      ea = hipStreamSynchronize(st);
      (void)hipFree(h->d_beams_fg); h->d_beams_fg = nullptr; h->fg_beams_cap = 0;
      if (ea == hipSuccess) ea = hipExtMallocWithFlags((void**)&h->d_beams_fg, sizeof(double2) * h->max_beams, hipDeviceMallocFinegrained);
      if (ea == hipSuccess) h->fg_beams_cap = h->max_beams;
    }
    if (ea != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(h->d_zslot); (void)hipFree(h->d_beam_ready); (void)hipFree(h->d_beams_fg);
      h->d_zslot = nullptr; h->d_beam_ready = nullptr; h->d_beams_fg = nullptr; h->fg_beams_cap = 0;
      h->noise_in_kernel = 0;
      dn = false;
    }
  }
  if (dn) {
    if (++h->beam_seq == 0u) ++h->beam_seq;   // (0 is the cleared word)
    ns.seed = h->seed; ns.scan = h->scan_index;
    ns.base = h->rng_n_global ? (size_t)h->rng_first * c.stride_normals : 0;
    ns.z_index = h->rng_n_global ? (size_t)h->rng_n_global * c.stride_normals : (size_t)h->N * c.stride_normals;
    ns.z_out = h->d_zslot;
    ns.host_beams = (const double2*)(h->h_beams + (size_t)slot * h->max_beams); ns.dev_beams = h->d_beams; ns.fg_beams = h->d_beams_fg;
    ns.ready = h->d_beam_ready; ns.seq = h->beam_seq;
  }
  h->last_drawn.valid = false;
  if (dn) {
    h->last_drawn.seed = ns.seed; h->last_drawn.scan = ns.scan; h->last_drawn.base = ns.base; h->last_drawn.z_index = ns.z_index;
    h->last_drawn.n = n_norm; h->last_drawn.valid = true;
  } else {
  if (!pre && n_norm > h->normals_cap) {
    TBNAV_HIP(hipStreamSynchronize(st));  // (a scan still in flight reads the old buffer)
    (void)hipFree(h->d_normals);
    h->d_normals = nullptr;
    TBNAV_HIP(hipMalloc((void**)&h->d_normals, sizeof(double) * n_norm));
    h->normals_cap = n_norm;
  }
  if (pre) {
    // (drawn with the rest of its chunk)
  } else if (normals) {
    TBNAV_HIP(hipMemcpyAsync(h->d_normals, normals, sizeof(double) * n_norm, hipMemcpyHostToDevice, st));
  } else {
    const int blocks = (int)std::min<size_t>((n_norm / 2 + 255) / 256, 4096);
    if (h->rng_n_global)  // this shard's slice of the ensemble's stream + the ensemble's resampling offset (same on every rank)
      hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, st, n_norm - 1, (unsigned long long)h->seed,
                         (unsigned long long)h->scan_index, h->d_normals, (const double2*)(h->h_beams + (size_t)slot * h->max_beams), h->d_beams, c.Bv,
                         (size_t)0, (size_t)0, (size_t)h->rng_first * c.stride_normals, (size_t)h->rng_n_global * c.stride_normals, n_norm - 1);
    else
    hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, st, n_norm, (unsigned long long)h->seed,
                       (unsigned long long)h->scan_index, h->d_normals, (const double2*)(h->h_beams + (size_t)slot * h->max_beams), h->d_beams, c.Bv);
    TBNAV_HIP(hipGetLastError());
  }
  }
  ++h->scan_index;
  const double2* const beams_dev = pre ? pre->d_beams : h->d_beams;
  const double* const normals_dev = dn ? nullptr : (pre ? pre->d_normals : h->d_normals);
  h->last_normals = normals_dev;
  h->last_z_index = (size_t)h->N * c.stride_normals;
  h->last_z_ptr = dn ? h->d_zslot : normals_dev + h->last_z_index;
  for (int q = 0; q < 4; ++q) h_err[q] = 0;  // mapped: the scan that last owned the slot has been waited for
  h->h_norm[slot] = NormOut{};
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);

  // ---- distance-field refresh for this call's lookups (windowed), then the particle update
  if (h->ref_field) { rc = ref_field_before_propose(h); if (rc != TBNAV_OK) { out->status = rc; return rc; } }
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[0], st));
  {
    // every lookup of this call lies within `half` metres of the particle's CURRENT position: the sampled poses
    // sit at T(pose)*T_icp (or the motion-model pose) +- the sampling noise, the laser at |Trs| from them, and
    // a valid beam ends less than range_max from the laser
    double sig = 0.0;
    for (int q = 1; q < 3; ++q) sig = std::max(sig, std::max(h->p.sample_range[q], h->p.motion_noise[q]));
    const double move = std::max(std::hypot(T_icp[1], T_icp[2]), std::fabs(u[1]));
    const double half = (double)h->p.range_max + std::hypot(h->p.Trs[1], h->p.Trs[2]) + move + 8.0 * std::sqrt(sig);
    int half_cells = (int)std::ceil(half / h->p.resolution) + 3;
    if (h->full_edt || half_cells > h->xsize) half_cells = h->xsize;  // whole map
    if (h->df_mode != 2) {  // query mode needs neither windows nor skip flags: the proposal kernel reads the field state itself
      hipLaunchKernelGGL(rbpf_window, dim3((h->N + 255) / 256), dim3(256), 0, st, c.g, h->N, half_cells, h->df_mode == 1 ? 1 : 0,
                         sp.pose, h->d_fstate, h->d_skip, h->d_win);
      TBNAV_HIP(hipGetLastError());
    }
    if (h->df_mode == 1) {
      const int tiles = std::min((2 * half_cells + 1 + kWave - 1) / kWave + 1, (h->ysize + kWave - 1) / kWave);
      rc = run_distance_field(h, c.g, 0, h->N, tiles);
      if (rc != TBNAV_OK) return rc;
    }
  }
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[1], st));
  size_t propose_lds = sizeof(double) * ((12 + kUnCap) * h->k + 3 * (c.Bv > 0 ? c.Bv : 1)) + sizeof(unsigned int) * 4 * (c.Bv > 0 ? c.Bv : 1);
  const size_t propose_lds_base = propose_lds;  // (what follows adds the LDS slice of the occupancy bitmap)
  int occ_half = 0;
  if (h->df_mode == 2) {
    // LDS copy of the occupancy bitmap round each particle's sensor: reach of a lookup (range_max + sampling
    // spread) plus a margin for the walk to the nearest obstacle; shrunk, then dropped, if it would not fit
    double sig = 0.0;
    for (int q = 1; q < 3; ++q) sig = std::max(sig, std::max(h->p.sample_range[q], h->p.motion_noise[q]));
    const int reach = (int)std::ceil(((double)h->p.range_max + 8.0 * std::sqrt(sig)) / h->p.resolution) + 2;
    for (int margin : {48, 16, 0}) {
      const int half = reach + margin;
      const int rows = std::min(h->xsize, 2 * half + 1), nw = std::min(h->words, (2 * half + 1 + 63) / 64 + 1);
      const size_t bytes = (size_t)rows * nw * 8 + (size_t)rows * 4;
      if (bytes <= 48 * 1024) { occ_half = half; propose_lds += bytes; break; }
    }
  }
  if (propose_lds > (size_t)kMaxLds - 3072) return TBNAV_ERR_UNSUPPORTED;  // scan x samples too large for one workgroup's LDS
  const int* skip_arr = h->df_mode == 2 ? h->d_fstate : h->d_skip;
  const int skip_eq = h->df_mode == 2 ? 2 : 1;
  const double* center = nullptr;
  if (h->sm_on && c.icp_ok) {
    // N1 option: every particle refines T(pose) * T_icp against its own map first; the samples are drawn round that
    const size_t sm_lds = sizeof(double2) * (c.Bv > 0 ? c.Bv : 1) + sizeof(double) * kMixLut + sizeof(unsigned long long) * 4 * (c.Bv > 0 ? c.Bv : 1) +
                          (propose_lds - propose_lds_base);
    hipLaunchKernelGGL(rbpf_scanmatch, dim3(h->N), dim3(kMatchThreads), sm_lds, st, c, h->sm, beams_dev, h->d_code[h->cur],
                       h->pool, map_of(h), h->d_trow[h->cur], skip_arr, skip_eq, h->df_mode, h->radius, occ_half,
                       h->d_nocc[h->cur], h->d_win, sp.pose, h->d_center, h->d_score, d_err, gate_prev, h->d_mixlut);
    TBNAV_HIP(hipGetLastError());
    center = h->d_center;
  }
  // workgroup size: four waves when four workgroups fit a CU's LDS (the 360-beam scans: 38 KB each), eight when the scan's
  // tables leave room for two or three only (1080 beams: 58 KB) — measured: 360 beams 32 us per 1000 particles with 256
  // threads against 43 with 512; the configs[4] shard 0.80 ms with 256 against 0.61 with 512
  h->lk_propose = (propose_lds + 3072 > (size_t)kMaxLds / 4) ? 2 * kProposeThreads : kProposeThreads;
  h->lk_propose_dn = dn ? 1 : 0;
  int* const pend = h->ref_field ? h->d_pend : nullptr;
#define TBNAV_PROPOSE(NT_, DN_) hipLaunchKernelGGL((rbpf_propose<NT_, DN_>), dim3(h->N + (DN_ ? 1 : 0)), dim3(NT_), propose_lds, st, c, beams_dev,                    \
                     h->d_code[h->cur], h->pool, map_of(h), h->d_trow[h->cur], skip_arr, skip_eq, h->df_mode, h->radius, occ_half,                       \
                     h->d_nocc[h->cur], h->d_win, normals_dev, center, sp.pose, sp.prev, sp.weight, h->tr, h->d_sens, d_err, gate_prev, h->d_mixlut, ns, pend)
  auto launch_propose = [&]() -> int {
    if (propose_lds + 3072 > (size_t)kMaxLds / 4) { if (dn) TBNAV_PROPOSE(2 * kProposeThreads, true); else TBNAV_PROPOSE(2 * kProposeThreads, false); }
    else { if (dn) TBNAV_PROPOSE(kProposeThreads, true); else TBNAV_PROPOSE(kProposeThreads, false); }
    TBNAV_HIP(hipGetLastError());
    return TBNAV_OK;
  };
#undef TBNAV_PROPOSE
  rc = launch_propose();
  if (rc != TBNAV_OK) return rc;
  if (h->ref_field) {  // lookups that met cells the lazy brushfire has not written: resume those states, run the proposal again
    rc = ref_field_settle(h, h_err, launch_propose);
    if (rc != TBNAV_OK) { out->status = rc; return rc; }
  }
  if (weights_ready) TBNAV_HIP(hipEventRecord(weights_ready, st));  // (sharded filter: the exchange starts here, beside the map update)
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[2], st));
  // normalise / select needs only the weights the proposal kernel left: it rides in the map update's launch as one extra
  // workgroup (the chain of adds it is made of would otherwise sit on the critical path, and a second stream costs an
  // event and a dependent boundary).  With event timing on it is a launch of its own, so that the intervals mean what
  // they say.
  const double* z_norm = h->last_z_ptr;
  tk.seq = (unsigned int)h->scans_done;
  const NormArgs nz{h->N, z_norm, sp.weight, sp.weight, h->d_cs, h->d_parent, h->d_norm + slot, h->d_gate + slot, gate_prev,
                    tk.poll ? h->d_seq + slot : nullptr, tk.seq, h->d_parent + h->N};
  auto launch_normalize = [&](hipStream_t s2) -> int {
    hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, s2, h->N, z_norm, sp.weight, sp.weight, h->d_cs, h->d_parent, h->d_norm + slot,
                       nullptr, nullptr, nullptr, 0u, h->d_parent + h->N);
    TBNAV_HIP(hipGetLastError());
    return TBNAV_OK;
  };
  const bool overlap = !local_only && !h->timing;
  if ((gate_prev || tk.poll) && !overlap) return TBNAV_ERR_INVALID_ARG;  // (a gated or polled scan is a batch scan: never local-only or timed)
  rc = launch_raycast(h, c, h->N, h->d_sens, overlap ? &nz : nullptr, d_err, beams_dev);
  if (rc != TBNAV_OK) return rc;
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[3], st));
  if (h->full_edt) {
    // legacy placement (TBNAV_RBPF_FULL_EDT=1): whole field of every particle right after the map update,
    // where the reference runs its brushfire (grid_mapper.cpp:181)
    TBNAV_HIP(hipMemsetAsync(h->d_skip, 0, sizeof(int) * h->N, st));
    rc = run_distance_field(h, c.g, 0, h->N, (h->ysize + kWave - 1) / kWave);
    if (rc != TBNAV_OK) return rc;
  }
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[4], st));
  if (!local_only && !overlap) {
    rc = launch_normalize(st);
    if (rc != TBNAV_OK) return rc;
  }
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[5], st));
  // the map changed: every field is stale until the next refresh (whole-map mode: fresh everywhere).  In query mode
  // the states are all zero already unless a field was injected or materialised since the last call.
  if (h->full_edt) {
    TBNAV_HIP(hipMemsetD32Async((hipDeviceptr_t)h->d_fstate, 2, h->N, st));
    h->fstate_dirty = true;
  } else if (h->fstate_dirty || h->df_mode != 2) {
    TBNAV_HIP(hipMemsetD32Async((hipDeviceptr_t)h->d_fstate, 0, h->N, st));
    h->fstate_dirty = h->df_mode != 2;
  }
  return TBNAV_OK;
}

int scan_finish(tbnav_rbpf* h, const ScanTicket& tk, tbnav_rbpf_stats* out) {
  hipStream_t st = h->stream;
  const bool local_only = tk.local_only;
  int rc = TBNAV_OK;
  if (tk.poll) {
    // the normalise / select workgroup raises the slot's flag as soon as its result is in host memory — the host need not
    // wait for the rest of the map update (error flags raised later in that launch: see tbnav_rbpf_slam_batch)
    volatile unsigned int* flag = h->h_seq + tk.slot;
    for (unsigned long spins = 1; *flag != tk.seq; ++spins) {
      __builtin_ia32_pause();
      if ((spins & 0xFFFF) == 0) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess && *flag != tk.seq) return TBNAV_ERR_HIP;  // the stream drained and the flag never came
        if (q != hipSuccess && q != hipErrorNotReady) TBNAV_HIP(q);
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    TBNAV_HIP(hipStreamSynchronize(st));
  }
  const int* err = h->h_err + 4 * tk.slot;
  const NormOut no = h->h_norm[tk.slot];
  out->status = status_from_err(err);
  out->sum_w = no.sum_w; out->sq_sum = no.sq_sum; out->neff = no.neff; out->resampled = no.resampled;
  bool gathered = false;
  if (!local_only && no.resampled && out->status == TBNAV_OK) {
    // lowVarianceResampling's deep copies (particle_filter.cpp:495): tables and state move on the device
    if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[6], st));
    rc = resample_on_device(h);
    if (rc != TBNAV_OK) return rc;
    if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[7], st));
    gathered = true;
  }
  if (gathered && !tk.poll) TBNAV_HIP(hipStreamSynchronize(st));  // (a batch goes straight on: the next scan is behind the copies in the stream)
  for (float& v : h->last_ms) v = 0.f;
  if (h->timing) {
    float e01, e12, e23, e34, e45;
    TBNAV_HIP(hipEventElapsedTime(&e01, h->ev[0], h->ev[1]));
    TBNAV_HIP(hipEventElapsedTime(&e12, h->ev[1], h->ev[2]));
    TBNAV_HIP(hipEventElapsedTime(&e23, h->ev[2], h->ev[3]));
    TBNAV_HIP(hipEventElapsedTime(&e34, h->ev[3], h->ev[4]));
    TBNAV_HIP(hipEventElapsedTime(&e45, h->ev[4], h->ev[5]));
    h->last_ms[0] = e12;        // propose
    h->last_ms[1] = e23;        // raycast
    h->last_ms[2] = 0.f;        // (occupancy pass: folded into the raycast)
    h->last_ms[3] = e01 + e34;  // distance field (windowed refresh before the update, or whole-map after it)
    h->last_ms[4] = e45;        // normalise / select
    if (gathered) TBNAV_HIP(hipEventElapsedTime(&h->last_ms[5], h->ev[6], h->ev[7]));
  }
  if (h->ref_field && out->status == TBNAV_OK) {
    rc = ref_field_after_scan(h, gathered);
    if (rc != TBNAV_OK) return rc;
  }
  return out->status;
}

int sharded_scan(int n, tbnav_rbpf* const* hs, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* const* normals, tbnav_rbpf_stats* out,
                 tbnav_rbpf_stats* local_out);

int slam_impl(tbnav_rbpf* h, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
              const double prev_odom[3], int icp_ok, const double T_icp[3], const double* normals,
              tbnav_rbpf_stats* out, bool local_only) {
  if (!h || !scan || n_beams <= 0 || !u || !cur_odom || !prev_odom || !T_icp || !out) return TBNAV_ERR_INVALID_ARG;
  if (h->ref_field && (local_only || h->comm)) return TBNAV_ERR_UNSUPPORTED;  // the reference-field mode is a single-handle mode
  if (h->comm && !local_only) {  // this process's rank of a sharded filter: the exchange is issued from here (sharded_scan)
    const double* nr[1] = {normals};
    return sharded_scan(1, &h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, nr, out, nullptr);
  }
  DeviceGuard guard(h->device);
  ScanTicket tk;
  const int rc = scan_enqueue(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals, out, local_only, 0, nullptr, tk);
  if (rc != TBNAV_OK) return rc;
  return scan_finish(h, tk, out);
}

}  // namespace

extern "C" {

namespace {
// host threads for the reference-field mode: the cores this process may run on — its affinity mask, and under a cgroup CPU quota
// (cpu.max: a container that sees 128 CPUs but may use 32 of them) no more than that — at most 128; TBNAV_RBPF_OPT_HOST_THREADS
// overrides.  (Round 4 capped this at 32: the bench box has more.)
int default_host_threads() {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
    long long quota = 0, period = 0;
    if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
      const int q = (int)((quota + period - 1) / period);
      if (q >= 1 && q < n) n = q;
    }
    std::fclose(f);
  }
  return n < 1 ? 1 : (n > 128 ? 128 : n);
}
int create_impl(const tbnav_rbpf_params* P, uint64_t max_pool_bytes, tbnav_rbpf** out) {
  if (!P || !out) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  if (P->num_particles <= 0 || P->num_samples_mode <= 0 || !(P->resolution > 0.0) || !(P->xmax > P->xmin) || !(P->ymax > P->ymin))
    return TBNAV_ERR_INVALID_ARG;
  const int xsize = (int)static_cast<unsigned int>(std::ceil((P->xmax - P->xmin) / P->resolution));  // mapSize, grid_mapper.cpp:31-34
  const int ysize = (int)static_cast<unsigned int>(std::ceil((P->ymax - P->ymin) / P->resolution));
  if (xsize != ysize) return TBNAV_ERR_UNSUPPORTED;  // the reference indexes both axes with xsize_ (grid_mapper.cpp:195-197,896)
  if (xsize < 4 || xsize > 32000 || (xsize & 1)) return TBNAV_ERR_UNSUPPORTED;  // vectorised code copies need G % 4 == 0
  if (P->num_particles > (1 << 20)) return TBNAV_ERR_UNSUPPORTED;
  const int radius = (int)static_cast<unsigned int>(std::ceil((10.0 - 0.0) / P->resolution));         // cell_radius_, grid_mapper.cpp:50
  if (radius > 254) return TBNAV_ERR_UNSUPPORTED;  // row-pass distances are stored as u8 (and radius^2 must fit the u16 code)
  const int words = (ysize + 63) / 64;
  int C = 64;
  if (edt_lds_bytes(xsize, words, C) > (size_t)kMaxLds) C = 32;
  // larger maps (xsize > ~640, e.g. BASELINE configs[4]'s 2000 x 2000): no LDS distance transform.  The SLAM path then
  // always answers lookups by query, and an on-demand field is produced cell by cell with the same query.
  if (edt_lds_bytes(xsize, words, C) > (size_t)kMaxLds) C = 0;
  int ndev = 0;
  {
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      return tbnav::hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount", __FILE__, __LINE__);
  }
  int dev = P->device;
  if (dev < 0) TBNAV_HIP(hipGetDevice(&dev));
  if (dev >= ndev) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(dev);
  if (!guard.ok) return TBNAV_ERR_NO_DEVICE;

  tbnav_rbpf* h = new (std::nothrow) tbnav_rbpf();
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->p = *P; h->device = dev; h->N = P->num_particles; h->k = P->num_samples_mode;
  h->host_threads = default_host_threads();
  h->xsize = xsize; h->ysize = ysize; h->words = words; h->radius = radius; h->edt_cols = C;
  h->G = (size_t)xsize * ysize;
  h->TW = (xsize + kTS - 1) / kTS; h->TT = h->TW * h->TW;
  {
    // every beam shorter than range_max ends within this many cells of the robot cell (+2 for the laser offset / rounding)
    const double reach = (double)P->range_max + std::sqrt(P->Trs[1] * P->Trs[1] + P->Trs[2] * P->Trs[2]);
    const long side = 2 * ((long)std::ceil(reach / P->resolution) + 2) + 1;
    h->tile_cap = (side * side <= 30000) ? (int)(side * side) : 0;
    h->df_mode = 2;  // exact query at lookup; the other modes are selected with tbnav_rbpf_set_option
    h->full_edt = false;
  }
  // log-odds constants with the host libm, exactly as the reference's ctor (grid_mapper.cpp:42-47)
  h->l_prior = std::log(0.5 / (1 - 0.5));
  h->l_occ = std::log(0.90 / (1 - 0.90));
  h->l_free = std::log(0.35 / (1 - 0.35));
  h->cut_occ = find_occ_cut(h->l_occ, 0.90);
  h->cuts = derive_export_cuts(h->cut_occ);
  const int N = h->N;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
  const size_t table_entries = (size_t)N * h->TT;
  for (int b = 0; b < 2; ++b) {
    A((void**)&h->d_state[b], sizeof(double) * 7 * N);
    A((void**)&h->d_table[b], sizeof(unsigned int) * table_entries);
    A((void**)&h->d_nocc[b], sizeof(int) * N);
    A((void**)&h->d_trow[b], sizeof(int) * (size_t)N * h->TW);
  }
  A((void**)&h->d_shed, sizeof(unsigned int) * table_entries);
  A((void**)&h->d_cs, sizeof(double) * N);
  A((void**)&h->d_sens, sizeof(double) * 4 * N);
  A((void**)&h->d_tile_scratch, sizeof(unsigned int) * h->TT);
  A((void**)&h->d_touched, sizeof(unsigned long long) * 2);
  A((void**)&h->d_box_need, sizeof(int) * 3);
  A((void**)&h->d_parent, sizeof(int) * 2 * N);
  A((void**)&h->d_best, sizeof(int));
  A((void**)&h->d_best_pose, sizeof(double) * 3);
  A((void**)&h->d_export, h->G);
  A((void**)&h->d_fstate, sizeof(int) * N);
  A((void**)&h->d_fstate_alt, sizeof(int) * N);
  A((void**)&h->d_center, sizeof(double) * 3 * N);
  A((void**)&h->d_score, sizeof(double) * N);
  A((void**)&h->d_skip, sizeof(int) * N);
  A((void**)&h->d_win, sizeof(int4) * N);
  A((void**)&h->d_tier, sizeof(int) * N);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_err, sizeof(int) * 4 * kScanSlots, hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_norm, sizeof(NormOut) * kScanSlots, hipHostMallocMapped);
  if (e == hipSuccess) { std::memset(h->h_err, 0, sizeof(int) * 4 * kScanSlots); std::memset((void*)h->h_norm, 0, sizeof(NormOut) * kScanSlots); }
  A((void**)&h->d_gate, sizeof(int) * kScanSlots);
  if (e == hipSuccess) e = hipMemset(h->d_gate, 0, sizeof(int) * kScanSlots);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_seq, sizeof(unsigned int) * kScanSlots, hipHostMallocMapped);
  if (e == hipSuccess) { std::memset(h->h_seq, 0, sizeof(unsigned int) * kScanSlots); e = hipHostGetDevicePointer((void**)&h->d_seq, h->h_seq, 0); }
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->d_err, h->h_err, 0);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->d_norm, h->h_norm, 0);
  const size_t kk = (size_t)h->k;
  const size_t trace_doubles = (size_t)N * (kk * 3 + kk + kk + 3 + 9 + 1 + 3 + 1);
  A((void**)&h->d_trace, sizeof(double) * trace_doubles);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  for (auto& ev : h->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
  // ---- the tile pool: everything the particles' maps can ever need if that fits the budget, else the budget.
  //      Default budget: half of the memory that is free now (several handles can live side by side).
  if (e == hipSuccess) {
    size_t free_b = 0, total_b = 0;
    e = hipMemGetInfo(&free_b, &total_b);
    const size_t per_tile = sizeof(double) * kTileCells + sizeof(unsigned int) * kTS + sizeof(int) + sizeof(unsigned int);
    const size_t budget = max_pool_bytes ? (size_t)max_pool_bytes : free_b / 2;
    // worst case: every table entry its own tile, plus the tiles the entries left since the last resample (still named by
    // the shed notes until the next resample settles them), plus the zero tile
    size_t cap = 2 * table_entries + 1;
    if (cap * per_tile > budget) cap = budget / per_tile;
    if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
    if (cap < (size_t)N + 2 && e == hipSuccess) e = hipErrorOutOfMemory;  // not even one tile per particle
    h->pool.cap = (unsigned int)cap;
    A((void**)&h->pool.lo, sizeof(double) * kTileCells * cap);
    A((void**)&h->pool.bm, sizeof(unsigned int) * kTS * cap);
    A((void**)&h->pool.ref, sizeof(int) * cap);
    A((void**)&h->pool.ring, sizeof(unsigned int) * cap);
    A((void**)&h->pool.ctr, sizeof(unsigned long long) * 2);
  }
  if (e == hipSuccess) {
    double* t = h->d_trace;
    h->tr.sampled = t; t += (size_t)N * kk * 3;
    h->tr.p_scan = t; t += (size_t)N * kk;
    h->tr.p_pose = t; t += (size_t)N * kk;
    h->tr.mu = t; t += (size_t)N * 3;
    h->tr.sigma = t; t += (size_t)N * 9;
    h->tr.eta = t; t += (size_t)N;
    h->tr.new_pose = t; t += (size_t)N * 3;
    h->tr.weight_raw = t;
    e = hipMemset(h->d_trace, 0, sizeof(double) * trace_doubles);
  }
  if (e == hipSuccess) {  // initParticleSet, particle_filter.cpp:125-138
    std::vector<double> s((size_t)7 * N);
    for (int i = 0; i < N; ++i) {
      for (int q = 0; q < 3; ++q) { s[(size_t)i * 3 + q] = P->pose0[q]; s[(size_t)3 * N + i * 3 + q] = P->pose0[q]; }
      s[(size_t)6 * N + i] = 1.0 / N;
    }
    e = hipMemcpy(h->d_state[0], s.data(), sizeof(double) * 7 * N, hipMemcpyHostToDevice);
    // empty maps: every table entry names the shared zero tile (log_odds_prior_ = log(1) = 0)
    if (e == hipSuccess) e = hipMemset(h->d_table[0], 0, sizeof(unsigned int) * table_entries);
    if (e == hipSuccess) e = hipMemset(h->d_shed, 0, sizeof(unsigned int) * table_entries);
    if (e == hipSuccess) e = hipMemset(h->pool.lo, 0, sizeof(double) * kTileCells);  // tile 0
    if (e == hipSuccess) e = hipMemset(h->pool.ref, 0, sizeof(int) * h->pool.cap);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rbpf_pool_init, dim3(1024), dim3(256), 0, h->stream, h->pool);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemset(h->d_touched, 0, sizeof(unsigned long long) * 2);
    if (e == hipSuccess) {  // the queue's scratch memory, before the first map update that needs it (see the kernel)
      hipLaunchKernelGGL(rbpf_warm_scratch, dim3(1024), dim3(512), 0, h->stream, reinterpret_cast<int*>(h->d_touched), 0);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemset(h->d_box_need, 0, sizeof(int) * 3);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_box_need, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) { *h->h_box_need = 0; e = hipHostGetDevicePointer((void**)&h->d_box_need_host, h->h_box_need, 0); }
    if (e == hipSuccess) e = hipMemset(h->d_nocc[0], 0, sizeof(int) * N);
    if (e == hipSuccess) e = hipMemset(h->d_skip, 0, sizeof(int) * N);
    // empty maps: the field "everything unreached" is what any lookup computes, no stored field needed (state 0)
    if (e == hipSuccess) e = hipMemset(h->d_fstate, 0, sizeof(int) * N);
    if (e == hipSuccess) e = hipMemset(h->d_fstate_alt, 0, sizeof(int) * N);
    h->fstate_dirty = false;
    if (e == hipSuccess) {
      std::vector<int4> w(N, make_int4(0, xsize - 1, 0, ysize - 1));
      e = hipMemcpy(h->d_win, w.data(), sizeof(int4) * N, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(h->d_trow[0], 0, sizeof(int) * (size_t)N * h->TW);
    if (e == hipSuccess) e = hipMemset(h->pool.bm, 0, sizeof(unsigned int) * kTS);  // tile 0
  }
  if (e == hipSuccess && C > 0) {
    const int lds = (int)edt_lds_bytes(xsize, words, C);
    e = (C == 64) ? hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                  : hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt<32>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  // (2.3 KB of static LDS: the embedded normalise's scan scratch)
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 6, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 6, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 8, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 8, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 8, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<1024, 8, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 1024);
  // the proposal / scan-match kernels carry the scan, the per-sample data and the bitmap slice: more than the 64 KB
  // default for long scans or many samples
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<kProposeThreads, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);  // (2.3 KB static)
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<2 * kProposeThreads, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<kProposeThreads, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<2 * kProposeThreads, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_scanmatch), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt_compact<kEdtRowsA>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)edt_compact_lds(kEdtRowsA));
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt_compact<kEdtRowsB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)edt_compact_lds(kEdtRowsB));
  A((void**)&h->d_mixlut, sizeof(double) * kMixLut);
  if (e == hipSuccess) {
    ScanC cm{};
    if (mixture_consts(h, cm)) {  // (a zero variance is reported by the first scan, as the reference throws there)
      hipLaunchKernelGGL(rbpf_mix_lut, dim3(kMixLut / 256), dim3(256), 0, h->stream, cm, h->d_mixlut);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    const int rc = tbnav::hip_fail(e, "tbnav_rbpf_create allocation", __FILE__, __LINE__);
    tbnav_rbpf_destroy(h);
    return rc;
  }
  *out = h;
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_create(const tbnav_rbpf_params* P, tbnav_rbpf** out) { return create_impl(P, 0, out); }
int tbnav_rbpf_create_pool(const tbnav_rbpf_params* P, uint64_t max_pool_bytes, tbnav_rbpf** out) { return create_impl(P, max_pool_bytes, out); }

int tbnav_rbpf_pool_stats(tbnav_rbpf* h, uint64_t* capacity_tiles, uint64_t* free_tiles, uint64_t* tile_bytes) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  unsigned long long ctr[2] = {0, 0};
  TBNAV_HIP(hipMemcpy(ctr, h->pool.ctr, sizeof ctr, hipMemcpyDeviceToHost));
  if (capacity_tiles) *capacity_tiles = h->pool.cap - 1;  // tile 0 is the shared zero tile
  if (free_tiles) *free_tiles = ctr[1] - ctr[0];
  if (tile_bytes) *tile_bytes = sizeof(double) * kTileCells;
  return TBNAV_OK;
}

void tbnav_rbpf_destroy(tbnav_rbpf* h) {
#ifdef TBNAV_PHASE_PROF
  rbpf_prof_print_propose();
  rbpf_prof_print_raycast();
#endif
  if (!h) return;
  DeviceGuard guard(h->device);
  for (int b = 0; b < 2; ++b) { (void)hipFree(h->d_state[b]); (void)hipFree(h->d_table[b]); (void)hipFree(h->d_code[b]); (void)hipFree(h->d_nocc[b]); (void)hipFree(h->d_trow[b]); }
  (void)hipFree(h->d_bm_dense); (void)hipFree(h->d_rc_dense);
  (void)hipFree(h->pool.lo); (void)hipFree(h->pool.bm); (void)hipFree(h->pool.ref); (void)hipFree(h->pool.ring); (void)hipFree(h->pool.ctr);
  (void)hipFree(h->d_sens); (void)hipFree(h->d_shed); (void)hipFree(h->d_dense); (void)hipFree(h->d_cs); (void)hipFree(h->d_touched); (void)hipFree(h->d_box_need); if (h->h_box_need) (void)hipHostFree(h->h_box_need); (void)hipFree(h->d_fstate_alt);
  (void)hipFree(h->d_log_ev); (void)hipFree(h->d_log_cnt); (void)hipFree(h->d_tile_scratch); (void)hipFree(h->d_log_pack); (void)hipFree(h->d_log_off); (void)hipFree(h->d_code_src);
  (void)hipFree(h->d_pend); (void)hipHostFree(h->h_pend); (void)hipFree(h->d_state_snap); (void)hipFree(h->d_jentries); (void)hipFree(h->d_jjobs);
  (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); (void)hipFree(h->d_gz);
  (void)hipFree(h->d_gw_raw); (void)hipFree(h->d_sendbuf); (void)hipFree(h->d_recvbuf); (void)hipFree(h->d_sizes); (void)hipFree(h->d_status);
  if (h->ev_w) (void)hipEventDestroy(h->ev_w);
  if (h->ev_g) (void)hipEventDestroy(h->ev_g);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  (void)hipFree(h->d_zslot); (void)hipFree(h->d_beam_ready); (void)hipFree(h->d_beams_fg);
  (void)hipFree(h->d_beams); (void)hipFree(h->d_normals); (void)hipFree(h->d_parent); (void)hipFree(h->d_best); (void)hipFree(h->d_best_pose); (void)hipFree(h->d_export); (void)hipFree(h->d_tier); (void)hipFree(h->d_fstate); (void)hipFree(h->d_skip); (void)hipFree(h->d_win); (void)hipFree(h->d_center); (void)hipFree(h->d_mixlut); (void)hipFree(h->d_bslots); (void)hipFree(h->d_bcount); (void)hipFree(h->d_bitems); (void)hipFree(h->d_bhdr); (void)hipFree(h->d_score);
  (void)hipFree(h->d_trace);
  (void)hipHostFree(h->h_beams); (void)hipHostFree(h->h_err); (void)hipHostFree(h->h_norm); (void)hipFree(h->d_gate);
  for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
  (void)hipHostFree(h->h_seq); (void)hipHostFree(h->h_beam_ring); (void)hipFree(h->d_beam_ring); (void)hipFree(h->d_norm_ring);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h->ref;
  delete h;
}

int tbnav_rbpf_grid_size(const tbnav_rbpf* h, int32_t* xsize, int32_t* ysize) {
  if (!h || !xsize || !ysize) return TBNAV_ERR_INVALID_ARG;
  *xsize = h->xsize; *ysize = h->ysize;
  return TBNAV_OK;
}

int tbnav_rbpf_set_seed(tbnav_rbpf* h, uint64_t seed) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->seed = seed;
  h->scan_index = 0;
  return TBNAV_OK;
}

int tbnav_rbpf_set_rng_shard(tbnav_rbpf* h, uint64_t first_particle, uint64_t particles_global) {
  if (!h || (particles_global && particles_global < first_particle + (uint64_t)h->N)) return TBNAV_ERR_INVALID_ARG;
  h->rng_first = first_particle;
  h->rng_n_global = particles_global;
  return TBNAV_OK;
}

int tbnav_rbpf_get_normals(tbnav_rbpf* h, double* out, int64_t n) {
  if (!h || !out || n <= 0) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (h->last_drawn.valid) {
    // the last scan drew its noise inside the proposal kernel and stored none of it: the same counters through rbpf_sample_normals
    // give the same values (one definition: normal_pair) — what the kernel used, regenerated for whoever asks
    if ((size_t)n > h->last_drawn.n) return TBNAV_ERR_INVALID_ARG;
    const size_t nn = h->last_drawn.n;
    if (nn > h->normals_cap) {
      (void)hipFree(h->d_normals); h->d_normals = nullptr; h->normals_cap = 0;
      TBNAV_HIP(hipMalloc((void**)&h->d_normals, sizeof(double) * nn));
      h->normals_cap = nn;
    }
    const int blocks = (int)std::min<size_t>((nn / 2 + 255) / 256, 4096);
    const bool sharded = h->last_drawn.z_index != nn - 1 || h->last_drawn.base != 0;
    if (sharded)
      hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, h->stream, nn - 1, h->last_drawn.seed, h->last_drawn.scan, h->d_normals,
                         (const double2*)nullptr, (double2*)nullptr, 0, (size_t)0, (size_t)0, h->last_drawn.base, h->last_drawn.z_index, nn - 1);
    else
      hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, h->stream, nn, h->last_drawn.seed, h->last_drawn.scan, h->d_normals,
                         (const double2*)nullptr, (double2*)nullptr, 0, (size_t)0, (size_t)0, (size_t)0, ~(size_t)0, (size_t)0);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipStreamSynchronize(h->stream));
    TBNAV_HIP(hipMemcpy(out, h->d_normals, sizeof(double) * n, hipMemcpyDeviceToHost));
    return TBNAV_OK;
  }
  if ((size_t)n > std::max(h->normals_cap, h->norm_ring_stride)) return TBNAV_ERR_INVALID_ARG;
  TBNAV_HIP(hipMemcpy(out, h->last_normals ? h->last_normals : h->d_normals, sizeof(double) * n, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int64_t tbnav_rbpf_num_normals(const tbnav_rbpf* h, int32_t icp_ok) {
  if (!h) return -1;
  return (int64_t)h->N * (icp_ok ? 3 * h->k + 3 : 3) + 1;
}

int tbnav_rbpf_slam(tbnav_rbpf* h, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                    const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals,
                    tbnav_rbpf_stats* out) {
  return slam_impl(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals, out, false);
}

int tbnav_rbpf_slam_batch(tbnav_rbpf* h, const float* scans, int32_t n_beams, int32_t n_scans, const double* u, const double* odom,
                          const int32_t* icp_ok, const double* T_icp, tbnav_rbpf_stats* out) {
  if (!h || !scans || n_scans <= 0 || !u || !odom || !T_icp || !out) return TBNAV_ERR_INVALID_ARG;
  // Two scans in the stream at a time: scan s + 1 is enqueued BEFORE the host waits for scan s, on the assumption that scan
  // s does not resample — its kernels check scan s's decision on the device (NormArgs::gate) and do nothing if it does; the
  // host then runs the copies and enqueues scan s + 1 again.  Between scans the device waits for nothing, and the results
  // are those of n_scans synchronous calls, bit for bit.  Only in the default configuration (distance look-ups by query: no
  // per-scan field refresh on the stream; no event timing; not the reference-field mode).
  const bool pipelined = n_scans > 1 && h->batch_pipeline && h->df_mode == 2 && !h->full_edt && !h->ref_field && !h->timing && !h->rng_n_global;
  if (!pipelined) {
    for (int s = 0; s < n_scans; ++s) {
      const int rc = slam_impl(h, scans + (size_t)s * n_beams, n_beams, u + 3 * s, odom + 3 * (s + 1), odom + 3 * s, icp_ok ? icp_ok[s] : 1,
                               T_icp + 3 * s, nullptr, out + s, false);
      if (rc != TBNAV_OK) return rc;
    }
    return TBNAV_OK;
  }
  if (n_beams <= 0) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  // The noise and the beam tables of the next few scans are put on the device by ONE launch per chunk (same Philox counters
  // as one launch per scan: same values), so that a scan is two launches — proposal, map update — back to back.
  const size_t norm_stride = (((size_t)h->N * (3 * (size_t)h->k + 3) + 1) + 1) & ~(size_t)1;
  int chunk = 8;
  while (chunk > 2 && (size_t)chunk * norm_stride * sizeof(double) > ((size_t)512 << 20)) --chunk;
  const bool ahead = chunk >= 3;  // (the host rewrites the pinned staging of chunk c + 1 once scan 0 of chunk c is through)
  if (ahead && (h->ring_scans != chunk || h->norm_ring_stride != norm_stride || h->beam_ring_stride != (size_t)n_beams)) {
    TBNAV_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_norm_ring); (void)hipFree(h->d_beam_ring); (void)hipHostFree(h->h_beam_ring);
    h->d_norm_ring = nullptr; h->d_beam_ring = nullptr; h->h_beam_ring = nullptr; h->ring_scans = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_norm_ring, sizeof(double) * norm_stride * chunk));
    TBNAV_HIP(hipMalloc((void**)&h->d_beam_ring, sizeof(double2) * (size_t)n_beams * chunk));
    TBNAV_HIP(hipHostMalloc((void**)&h->h_beam_ring, sizeof(double2) * (size_t)n_beams * chunk, hipHostMallocDefault));
    std::memset(h->h_beam_ring, 0, sizeof(double2) * (size_t)n_beams * chunk);
    h->ring_scans = chunk; h->norm_ring_stride = norm_stride; h->beam_ring_stride = (size_t)n_beams;
  }
  const unsigned long long scan0 = h->scan_index;  // noise counter of the batch's first scan
  std::vector<Prefetched> pre(ahead ? chunk : 0);
  int prepared_to = 0;  // scans [0, prepared_to) have had their chunk prepared
  int chunk_first = 0;  // the first scan of the chunk prepared last: scan s of it uses slot s - chunk_first of the rings
  auto prepare = [&](int first) -> int {
    // The call's FIRST chunk is two scans: the host's share of a chunk (the scans' beam tables, ~5 us each) sits in front of the
    // call's first launch, where nothing hides it — a call of 6 scans cost 40 us on top of its scans, one of 3 cost 21.  Later chunks
    // are prepared while two scans are in the stream.  (Two, not one: the pinned staging of a chunk is rewritten when the next is
    // prepared, during the iteration of its last scan — by then the scan before that has been waited for, and with it the launch
    // that read the staging, only if the chunk had two scans at least.)
    const int m = std::min(first == 0 ? 2 : chunk, n_scans - first);
    chunk_first = first;
    for (int j = 0; j < m; ++j) {
      const int s = first + j;
      Prefetched& q = pre[j];
      q.rc = build_scan_consts(h, q.c, scans + (size_t)s * n_beams, n_beams, u + 3 * s, odom + 3 * (s + 1), odom + 3 * s,
                               icp_ok ? icp_ok[s] : 1, T_icp + 3 * s, h->beams_tmp);
      q.d_beams = h->d_beam_ring + (size_t)j * n_beams;
      q.d_normals = h->d_norm_ring + (size_t)j * norm_stride;
      if (q.rc == TBNAV_OK && q.c.Bv) std::memcpy(h->h_beam_ring + (size_t)j * n_beams, h->beams_tmp.data(), sizeof(double2) * q.c.Bv);
    }
    const int blocks = (int)std::min<size_t>((norm_stride / 2 + 255) / 256, 4096);
    hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks, m), dim3(256), 0, h->stream, norm_stride, (unsigned long long)h->seed,
                       scan0 + (unsigned long long)first, h->d_norm_ring, (const double2*)h->h_beam_ring, h->d_beam_ring, n_beams,
                       norm_stride, (size_t)n_beams);
    TBNAV_HIP(hipGetLastError());
    prepared_to = first + m;
    return TBNAV_OK;
  };
  ScanTicket tk[2];
  auto enqueue = [&](int s, const int* gate_prev) -> int {
    if (ahead && s >= prepared_to) { const int rc = prepare(s); if (rc != TBNAV_OK) return rc; }
    ScanTicket& t = tk[s & 1];
    t = ScanTicket{};
    t.poll = true;
    h->scan_index = scan0 + (unsigned long long)s;  // (scan_enqueue counts it)
    return scan_enqueue(h, scans + (size_t)s * n_beams, n_beams, u + 3 * s, odom + 3 * (s + 1), odom + 3 * s, icp_ok ? icp_ok[s] : 1,
                        T_icp + 3 * s, nullptr, out + s, false, s % kScanSlots, gate_prev, t, ahead ? &pre[s - chunk_first] : nullptr);
  };
  int rc = enqueue(0, nullptr);
  if (rc != TBNAV_OK) return rc;
  for (int s = 0; s < n_scans; ++s) {
    const int rc_next = s + 1 < n_scans ? enqueue(s + 1, h->d_gate + s % kScanSlots) : TBNAV_OK;
    rc = scan_finish(h, tk[s & 1], out + s);
    if (rc == TBNAV_OK && s > 0) {
      // scan s - 1 was finished when its weights were normalised, while its map update was still running; that launch is
      // complete now (scan s ran behind it): anything it flagged after that?
      const int late = status_from_err(h->h_err + 4 * ((s - 1) % kScanSlots));
      if (late != TBNAV_OK) {
        (void)hipStreamSynchronize(h->stream);
        out[s - 1].status = late;
        std::memset(out + s, 0, sizeof(tbnav_rbpf_stats) * (size_t)(n_scans - s));
        return late;
      }
    }
    if (rc != TBNAV_OK || rc_next != TBNAV_OK) {
      (void)hipStreamSynchronize(h->stream);  // (whatever of scan s + 1 is in the stream: the filter's state after an error is unspecified)
      return rc != TBNAV_OK ? rc : rc_next;
    }
    if (out[s].resampled && s + 1 < n_scans) {
      // scan s + 1's launches did nothing: same scan number, same noise, again — on the resampled particles
      --h->scans_done;
      rc = enqueue(s + 1, nullptr);
      if (rc != TBNAV_OK) { (void)hipStreamSynchronize(h->stream); return rc; }
    }
  }
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // the last scan's map update
  out[n_scans - 1].status = status_from_err(h->h_err + 4 * ((n_scans - 1) % kScanSlots));
  return out[n_scans - 1].status;
}

int tbnav_rbpf_slam_local(tbnav_rbpf* h, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                          const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals,
                          tbnav_rbpf_stats* out) {
  return slam_impl(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals, out, true);
}

// Host-side, sequential, bit-faithful: O(n_global) double adds — the exchange step of the sharded
// filter; every rank runs it on the same all-gathered weights.
int tbnav_rbpf_resample_global(const double* w, int64_t n, double z, int32_t* parents, double* wn, tbnav_rbpf_stats* out) {
  if (!w || n <= 0 || !parents || !wn || !out) return TBNAV_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof *out);
  double sum = 0.0;
  for (int64_t i = 0; i < n; ++i) sum += w[i];
  double sq = 0.0;
  for (int64_t i = 0; i < n; ++i) { wn[i] = w[i] / sum; sq += wn[i] * wn[i]; }
  out->sum_w = sum; out->sq_sum = sq;
  out->neff = static_cast<int>(1.0 / sq);
  const int N = (int)n;
  out->resampled = (out->neff < (N / 2)) ? 1 : 0;
  if (!out->resampled) { for (int m = 0; m < N; ++m) parents[m] = m; return TBNAV_OK; }
  const double r = z / static_cast<double>(N);
  double c = wn[0];
  int i = 0;
  for (int m = 0; m < N; ++m) {
    const double U = r + static_cast<double>(m * (1.0 / (N - 1)));
    while (U > c) {
      i++;
      if (i > N - 1) { i = N - 1; break; }
      c += wn[i];
    }
    parents[m] = i;
  }
  return TBNAV_OK;
}

int tbnav_rbpf_add_repeated(const double* x, const double* d, const int32_t* n, double* out, int64_t count) {
  if (!x || !d || !n || !out || count <= 0 || count > (1 << 26)) return TBNAV_ERR_INVALID_ARG;
  double *dx = nullptr, *dd = nullptr, *dout = nullptr;
  int* dn = nullptr;
  int rc = TBNAV_OK;
  auto body = [&]() -> int {
    TBNAV_HIP(hipMalloc((void**)&dx, sizeof(double) * count)); TBNAV_HIP(hipMalloc((void**)&dd, sizeof(double) * count));
    TBNAV_HIP(hipMalloc((void**)&dout, sizeof(double) * count)); TBNAV_HIP(hipMalloc((void**)&dn, sizeof(int) * count));
    TBNAV_HIP(hipMemcpy(dx, x, sizeof(double) * count, hipMemcpyHostToDevice)); TBNAV_HIP(hipMemcpy(dd, d, sizeof(double) * count, hipMemcpyHostToDevice));
    TBNAV_HIP(hipMemcpy(dn, n, sizeof(int) * count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rbpf_add_repeated_test, dim3((unsigned int)((count + 255) / 256)), dim3(256), 0, 0, dx, dd, dn, dout, (int)count);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipMemcpy(out, dout, sizeof(double) * count, hipMemcpyDeviceToHost));
    return TBNAV_OK;
  };
  rc = body();
  (void)hipFree(dx); (void)hipFree(dd); (void)hipFree(dout); (void)hipFree(dn);
  return rc;
}

int tbnav_rbpf_gather_local(tbnav_rbpf* h, const int32_t* local_parent) {
  if (!h || !local_parent) return TBNAV_ERR_INVALID_ARG;
  if (h->ref_field) return TBNAV_ERR_UNSUPPORTED;  // the reference-field mode is a single-handle mode
  DeviceGuard guard(h->device);
  const int N = h->N;
  // slots with parent -1 keep their own content: copy self
  std::vector<int> par(local_parent, local_parent + N);
  par.resize(2 * (size_t)N, 0);  // [N, 2N): how many slots chose each particle
  for (int m = 0; m < N; ++m) { if (par[m] < 0) par[m] = m; if (par[m] >= N) return TBNAV_ERR_INVALID_ARG; ++par[N + par[m]]; }
  TBNAV_HIP(hipMemcpyAsync(h->d_parent, par.data(), sizeof(int) * 2 * N, hipMemcpyHostToDevice, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // par is a local
  const int rc = resample_on_device(h);
  if (rc != TBNAV_OK) return rc;
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}

// ---- device-side exchange for the sharded filter -------------------------------------------------------------
int tbnav_rbpf_copy_weights_dev(tbnav_rbpf* h, double* d_dst) {
  if (!h || !d_dst) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipMemcpyAsync(d_dst, sp.weight, sizeof(double) * h->N, hipMemcpyDeviceToDevice, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}

int tbnav_rbpf_resample_global_dev(tbnav_rbpf* h, const double* d_weights_all, int64_t n_global, int64_t offset, double z,
                                   int32_t* parents_out, tbnav_rbpf_stats* out) {
  if (!h || !d_weights_all || n_global <= 0 || offset < 0 || offset + h->N > n_global || !out || n_global > (1 << 24)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  if ((size_t)n_global > h->g_cap) {
    (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); h->d_gw = h->d_gcs = nullptr; h->d_gparent = nullptr; h->g_cap = 0;
    (void)hipFree(h->d_gw_raw); h->d_gw_raw = nullptr;  // (the in-library sharded scan sizes its buffers with the same capacity: it re-creates them)
    TBNAV_HIP(hipMalloc((void**)&h->d_gw, sizeof(double) * n_global));
    TBNAV_HIP(hipMalloc((void**)&h->d_gcs, sizeof(double) * n_global));
    TBNAV_HIP(hipMalloc((void**)&h->d_gparent, sizeof(int) * n_global));
    h->g_cap = (size_t)n_global;
  }
  if (!h->d_gz) TBNAV_HIP(hipMalloc((void**)&h->d_gz, sizeof(double)));
  if (z != z) {  // NaN: the offset the last scan's device noise carries (with tbnav_rbpf_set_rng_shard: the ENSEMBLE's, same on every rank)
    if (!h->last_z_ptr) return TBNAV_ERR_INVALID_ARG;
    TBNAV_HIP(hipMemcpyAsync(h->d_gz, h->last_z_ptr, sizeof z, hipMemcpyDeviceToDevice, st));
  } else
  TBNAV_HIP(hipMemcpyAsync(h->d_gz, &z, sizeof z, hipMemcpyHostToDevice, st));
  *h->h_norm = NormOut{};
  // the reference's sequential normalise / Neff / selection (particle_filter.cpp:442-500) over the GLOBAL vector:
  // every rank runs the same kernel on the same values, so all ranks agree bit for bit
  hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, st, (int)n_global, h->d_gz, d_weights_all, h->d_gw, h->d_gcs, h->d_gparent, h->d_norm);
  TBNAV_HIP(hipGetLastError());
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipMemcpyAsync(sp.weight, h->d_gw + offset, sizeof(double) * h->N, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  const NormOut no = *h->h_norm;
  std::memset(out, 0, sizeof *out);
  out->sum_w = no.sum_w; out->sq_sum = no.sq_sum; out->neff = no.neff; out->resampled = no.resampled;
  if (no.resampled && parents_out) TBNAV_HIP(hipMemcpy(parents_out, h->d_gparent, sizeof(int) * n_global, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

namespace {
int batch_scratch(tbnav_rbpf* h, size_t n) {
  if (n <= h->batch_cap) return TBNAV_OK;
  (void)hipFree(h->d_bslots); (void)hipFree(h->d_bcount); (void)hipFree(h->d_bitems); (void)hipFree(h->d_bhdr);
  h->d_bslots = nullptr; h->d_bcount = nullptr; h->d_bitems = nullptr; h->d_bhdr = nullptr; h->batch_cap = 0;
  const size_t cap = n + n / 2 + 64;
  TBNAV_HIP(hipMalloc((void**)&h->d_bslots, sizeof(int) * cap));
  TBNAV_HIP(hipMalloc((void**)&h->d_bcount, sizeof(int2) * cap));
  TBNAV_HIP(hipMalloc((void**)&h->d_bitems, sizeof(BatchItem) * cap));
  TBNAV_HIP(hipMalloc((void**)&h->d_bhdr, sizeof(BlobHeader) * cap));
  h->batch_cap = cap;
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_set_weights_from_global_dev(tbnav_rbpf* h, const int32_t* global_parent_of_slot /*[N]*/) {
  // after a resample every slot carries its parent's normalised weight (weights are NOT reset, particle_filter.cpp:495)
  if (!h || !global_parent_of_slot || !h->d_gw) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  for (int m = 0; m < h->N; ++m)
    if (global_parent_of_slot[m] < 0 || (size_t)global_parent_of_slot[m] >= h->g_cap) return TBNAV_ERR_INVALID_ARG;
  { const int rc = batch_scratch(h, (size_t)h->N); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipMemcpyAsync(h->d_bslots, global_parent_of_slot, sizeof(int) * h->N, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(rbpf_gather_weights, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->N, h->d_gw, h->d_bslots, sp.weight);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // (the parent list is the caller's)
  return TBNAV_OK;
}

namespace {
BlobLayout blob_layout(const tbnav_rbpf* h, uint32_t n_tiles, bool has_codes) { return blob_layout_hd(h->TW, h->G, n_tiles, has_codes); }
int slot_tiles(tbnav_rbpf* h, int slot, std::vector<uint32_t>& tidx, std::vector<uint32_t>& ids, int& fstate) {
  std::vector<uint32_t> row(h->TT);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(row.data(), h->d_table[h->cur] + (size_t)slot * h->TT, sizeof(uint32_t) * h->TT, hipMemcpyDeviceToHost));
  TBNAV_HIP(hipMemcpy(&fstate, h->d_fstate + slot, sizeof(int), hipMemcpyDeviceToHost));
  tidx.clear(); ids.clear();
  for (int t = 0; t < h->TT; ++t) if (row[t]) { tidx.push_back((uint32_t)t); ids.push_back(row[t]); }
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_export_size(tbnav_rbpf* h, int32_t slot, uint64_t* bytes) {
  if (!h || !bytes || slot < 0 || slot >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  std::vector<uint32_t> tidx, ids; int fs = 0;
  { const int rc = slot_tiles(h, slot, tidx, ids, fs); if (rc != TBNAV_OK) return rc; }
  *bytes = blob_layout(h, (uint32_t)tidx.size(), fs == 2 && h->d_code[0]).total;
  return TBNAV_OK;
}

int tbnav_rbpf_export_particle_dev(tbnav_rbpf* h, int32_t slot, void* d_buf, uint64_t capacity, uint64_t* bytes) {
  if (!h || !d_buf || slot < 0 || slot >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  std::vector<uint32_t> tidx, ids; int fs = 0;
  { const int rc = slot_tiles(h, slot, tidx, ids, fs); if (rc != TBNAV_OK) return rc; }
  const bool has_codes = fs == 2 && h->d_code[0];
  const uint32_t n = (uint32_t)tidx.size();
  const BlobLayout L = blob_layout(h, n, has_codes);
  if (bytes) *bytes = L.total;
  if (L.total > capacity) return TBNAV_ERR_INVALID_ARG;
  char* b = static_cast<char*>(d_buf);
  BlobHeader hd{kBlobMagic, n, has_codes ? 1u : 0u, 0, fs, (uint32_t)h->xsize, (uint32_t)h->TT};
  TBNAV_HIP(hipMemcpy(&hd.nocc, h->d_nocc[h->cur] + slot, sizeof(int), hipMemcpyDeviceToHost));
  TBNAV_HIP(hipMemcpy(b, &hd, sizeof hd, hipMemcpyHostToDevice));
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  double* bs = reinterpret_cast<double*>(b + L.state);
  TBNAV_HIP(hipMemcpyAsync(bs, sp.pose + (size_t)slot * 3, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(bs + 3, sp.prev + (size_t)slot * 3, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(bs + 6, sp.weight + slot, sizeof(double), hipMemcpyDeviceToDevice, st));
  if (n) {
    TBNAV_HIP(hipMemcpy(b + L.tidx, tidx.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    TBNAV_HIP(hipMemcpy(h->d_tile_scratch, ids.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rbpf_pack_tiles, dim3(n), dim3(256), 0, st, h->pool, h->d_tile_scratch, reinterpret_cast<double*>(b + L.tiles),
                       reinterpret_cast<unsigned int*>(b + L.tile_bm));
    TBNAV_HIP(hipGetLastError());
  }
  TBNAV_HIP(hipMemcpyAsync(b + L.trow, h->d_trow[h->cur] + (size_t)slot * h->TW, sizeof(int) * h->TW, hipMemcpyDeviceToDevice, st));
  if (has_codes) TBNAV_HIP(hipMemcpyAsync(b + L.codes, h->d_code[h->cur] + (size_t)slot * h->G, sizeof(uint16_t) * h->G, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  return TBNAV_OK;
}

int tbnav_rbpf_import_particle_dev(tbnav_rbpf* h, int32_t slot, const void* d_buf, uint64_t bytes) {
  if (!h || !d_buf || slot < 0 || slot >= h->N || bytes < sizeof(BlobHeader)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  const char* b = static_cast<const char*>(d_buf);
  BlobHeader hd{};
  TBNAV_HIP(hipStreamSynchronize(st));
  TBNAV_HIP(hipMemcpy(&hd, b, sizeof hd, hipMemcpyDeviceToHost));
  if (hd.magic != kBlobMagic || hd.xsize != (uint32_t)h->xsize || hd.TT != (uint32_t)h->TT || hd.n_tiles > (uint32_t)h->TT) return TBNAV_ERR_INVALID_ARG;
  const BlobLayout L = blob_layout(h, hd.n_tiles, hd.has_codes != 0);
  if (L.total > bytes) return TBNAV_ERR_INVALID_ARG;
  if (hd.has_codes) { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  const MapT M = map_of(h);
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  hipLaunchKernelGGL(rbpf_release_slot, dim3((h->TT + 255) / 256), dim3(256), 0, st, h->pool, M, slot);
  TBNAV_HIP(hipGetLastError());
  if (hd.n_tiles) {
    hipLaunchKernelGGL(rbpf_unpack_tiles, dim3(hd.n_tiles), dim3(256), 0, st, h->pool, M, slot, reinterpret_cast<const unsigned int*>(b + L.tidx),
                       reinterpret_cast<const double*>(b + L.tiles), reinterpret_cast<const unsigned int*>(b + L.tile_bm), h->d_err);
    TBNAV_HIP(hipGetLastError());
  }
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  const double* bs = reinterpret_cast<const double*>(b + L.state);
  TBNAV_HIP(hipMemcpyAsync(sp.pose + (size_t)slot * 3, bs, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(sp.prev + (size_t)slot * 3, bs + 3, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(sp.weight + slot, bs + 6, sizeof(double), hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(h->d_trow[h->cur] + (size_t)slot * h->TW, b + L.trow, sizeof(int) * h->TW, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(h->d_nocc[h->cur] + slot, &hd.nocc, sizeof(int), hipMemcpyHostToDevice, st));
  const int fs = hd.has_codes ? 2 : 0;
  if (hd.has_codes) {
    TBNAV_HIP(hipMemcpyAsync(h->d_code[h->cur] + (size_t)slot * h->G, b + L.codes, sizeof(uint16_t) * h->G, hipMemcpyDeviceToDevice, st));
    h->fstate_dirty = true;
  }
  TBNAV_HIP(hipMemcpyAsync(h->d_fstate + slot, &fs, sizeof(int), hipMemcpyHostToDevice, st));
  TBNAV_HIP(hipStreamSynchronize(st));  // hd / fs are locals
  if (h->h_err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;
  return TBNAV_OK;
}

// ---- the same for many particles at once (what a cross-rank resample needs: hundreds of particles per rank) ------------
namespace {
int count_batch(tbnav_rbpf* h, int32_t n, const int32_t* slots) {
  for (int i = 0; i < n; ++i) if (slots[i] < 0 || slots[i] >= h->N) return TBNAV_ERR_INVALID_ARG;
  { const int rc = batch_scratch(h, (size_t)n); if (rc != TBNAV_OK) return rc; }
  h->batch_counts.resize(n);
  TBNAV_HIP(hipMemcpyAsync(h->d_bslots, slots, sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(rbpf_count_tiles, dim3(n), dim3(256), 0, h->stream, map_of(h), h->d_bslots, h->d_fstate, h->d_bcount);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(h->batch_counts.data(), h->d_bcount, sizeof(int2) * n, hipMemcpyDeviceToHost, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_export_batch_sizes(tbnav_rbpf* h, int32_t n, const int32_t* slots, uint64_t* sizes_out) {
  if (!h || n < 0 || (n && (!slots || !sizes_out))) return TBNAV_ERR_INVALID_ARG;
  if (n == 0) return TBNAV_OK;
  DeviceGuard guard(h->device);
  { const int rc = count_batch(h, n, slots); if (rc != TBNAV_OK) return rc; }
  for (int i = 0; i < n; ++i)
    sizes_out[i] = blob_layout(h, (uint32_t)h->batch_counts[i].x, h->batch_counts[i].y == 2 && h->d_code[0]).total;
  return TBNAV_OK;
}

int tbnav_rbpf_export_batch_dev(tbnav_rbpf* h, int32_t n, const int32_t* slots, void* d_buf, uint64_t capacity, uint64_t* offsets_out) {
  if (!h || n < 0 || (n && (!slots || !d_buf || !offsets_out))) return TBNAV_ERR_INVALID_ARG;
  if (n == 0) { if (offsets_out) offsets_out[0] = 0; return TBNAV_OK; }
  DeviceGuard guard(h->device);
  // (counted again rather than trusting what tbnav_rbpf_export_batch_sizes saw: a scan in between would change the tables;
  //  a tiny launch and one 8-byte-per-particle copy)
  { const int rc = count_batch(h, n, slots); if (rc != TBNAV_OK) return rc; }
  std::vector<BatchItem> items(n);
  uint64_t off = 0;
  for (int i = 0; i < n; ++i) {
    const bool has_codes = h->batch_counts[i].y == 2 && h->d_code[0];
    items[i] = BatchItem{slots[i], (unsigned int)h->batch_counts[i].x, has_codes ? 1 : 0, 0, off};
    offsets_out[i] = off;
    off += blob_layout(h, items[i].n_tiles, has_codes).total;
  }
  offsets_out[n] = off;
  if (off > capacity) return TBNAV_ERR_INVALID_ARG;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipMemcpyAsync(h->d_bitems, items.data(), sizeof(BatchItem) * n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(rbpf_pack_batch, dim3(n), dim3(256), 0, h->stream, h->pool, map_of(h), sp.pose, sp.prev, sp.weight, h->d_trow[h->cur],
                     h->d_nocc[h->cur], h->d_fstate, h->d_code[h->cur], h->G, h->xsize, h->d_bitems, static_cast<char*>(d_buf));
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // (items is a local; the caller sends the buffer next)
  return TBNAV_OK;
}

int tbnav_rbpf_import_batch_dev(tbnav_rbpf* h, int32_t n, const int32_t* slots, const void* d_buf, uint64_t bytes, const uint64_t* offsets) {
  if (!h || n < 0 || (n && (!slots || !d_buf || !offsets))) return TBNAV_ERR_INVALID_ARG;
  if (n == 0) return TBNAV_OK;
  DeviceGuard guard(h->device);
  std::vector<char> seen(h->N, 0);
  std::vector<BatchItem> items(n);
  for (int i = 0; i < n; ++i) {
    if (slots[i] < 0 || slots[i] >= h->N || seen[slots[i]] || offsets[i] + sizeof(BlobHeader) > bytes || (offsets[i] & 7)) return TBNAV_ERR_INVALID_ARG;
    seen[slots[i]] = 1;  // (a slot receives one particle; one particle may fill several slots)
    items[i] = BatchItem{slots[i], 0u, 0, 0, offsets[i]};
  }
  { const int rc = batch_scratch(h, (size_t)n); if (rc != TBNAV_OK) return rc; }
  hipStream_t st = h->stream;
  const char* b = static_cast<const char*>(d_buf);
  TBNAV_HIP(hipMemcpyAsync(h->d_bitems, items.data(), sizeof(BatchItem) * n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(rbpf_blob_headers, dim3((n + 255) / 256), dim3(256), 0, st, h->d_bitems, b, h->d_bhdr, n);
  TBNAV_HIP(hipGetLastError());
  std::vector<BlobHeader> hd(n);
  TBNAV_HIP(hipMemcpyAsync(hd.data(), h->d_bhdr, sizeof(BlobHeader) * n, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  bool any_codes = false;
  for (int i = 0; i < n; ++i) {
    if (hd[i].magic != kBlobMagic || hd[i].xsize != (uint32_t)h->xsize || hd[i].TT != (uint32_t)h->TT || hd[i].n_tiles > (uint32_t)h->TT) return TBNAV_ERR_INVALID_ARG;
    if (offsets[i] + blob_layout(h, hd[i].n_tiles, hd[i].has_codes != 0).total > bytes) return TBNAV_ERR_INVALID_ARG;
    any_codes |= hd[i].has_codes != 0;
  }
  if (any_codes) { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; h->fstate_dirty = true; }
  {
    // Does the pool hold what is coming?  Checked BEFORE the destination slots give their tiles up: the incoming tiles against
    // the free ones plus every tile the slots name now (an upper bound of what releasing them returns).  Beyond that the import
    // cannot succeed and nothing is touched; inside the bound it goes ahead (tiles the slots share with particles that stay do
    // not come back: the unpack kernel then reports the exhaustion, with the slots' maps already released — see the header).
    uint64_t incoming = 0;
    for (int i = 0; i < n; ++i) incoming += hd[i].n_tiles;
    unsigned long long ctr[2] = {0, 0};
    TBNAV_HIP(hipMemcpy(ctr, h->pool.ctr, sizeof ctr, hipMemcpyDeviceToHost));
    const uint64_t free_now = ctr[1] - ctr[0];
    if (incoming > free_now) {
      std::vector<int> sl(n);
      for (int i = 0; i < n; ++i) sl[i] = slots[i];
      { const int rc = count_batch(h, n, sl.data()); if (rc != TBNAV_OK) return rc; }
      uint64_t named = 0;
      for (int i = 0; i < n; ++i) named += (uint64_t)h->batch_counts[i].x;
      if (incoming > free_now + named) return TBNAV_ERR_POOL_EXHAUSTED;
      TBNAV_HIP(hipMemcpyAsync(h->d_bitems, items.data(), sizeof(BatchItem) * n, hipMemcpyHostToDevice, st));  // (count_batch reused the scratch's slot list only)
    }
  }
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  const MapT M = map_of(h);
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  hipLaunchKernelGGL(rbpf_release_slots, dim3(n), dim3(256), 0, st, h->pool, M, h->d_bitems);  // pushes: all before the first pop
  TBNAV_HIP(hipGetLastError());
  hipLaunchKernelGGL(rbpf_unpack_batch, dim3(n), dim3(256), 0, st, h->pool, M, sp.pose, sp.prev, sp.weight, h->d_trow[h->cur], h->d_nocc[h->cur],
                     h->d_fstate, h->d_code[h->cur], h->G, h->d_bitems, b, h->d_err);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(st));
  if (h->h_err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;
  return TBNAV_OK;
}

}  // extern "C"

// =================================================================================================
// The sharded filter inside the library (include/tbnav_comm.h)
// =================================================================================================
namespace {

int ensure_shard_state(tbnav_rbpf* h) {
  const int P = tbnav::comm_size(h->comm);
  const size_t ng = (size_t)P * h->N;
  if (!h->stream2) TBNAV_HIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  if (!h->ev_w) TBNAV_HIP(hipEventCreateWithFlags(&h->ev_w, hipEventDisableTiming));
  if (!h->ev_g) TBNAV_HIP(hipEventCreateWithFlags(&h->ev_g, hipEventDisableTiming));
  if (ng > h->g_cap || !h->d_gw_raw) {
    TBNAV_HIP(hipStreamSynchronize(h->stream));
    TBNAV_HIP(hipStreamSynchronize(h->stream2));
    (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); (void)hipFree(h->d_gw_raw); (void)hipFree(h->d_sizes); (void)hipFree(h->d_status);
    h->d_gw = h->d_gcs = h->d_gw_raw = nullptr; h->d_gparent = nullptr; h->d_sizes = nullptr; h->d_status = nullptr; h->g_cap = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_gw, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gcs, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gw_raw, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gparent, sizeof(int) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_sizes, sizeof(unsigned long long) * ((size_t)h->N + ng)));
    TBNAV_HIP(hipMalloc((void**)&h->d_status, sizeof(int) * (1 + (size_t)P)));
    h->g_cap = ng;
  }
  return TBNAV_OK;
}

int grow(char*& buf, size_t& cap, size_t need) {
  if (need <= cap) return TBNAV_OK;
  (void)hipFree(buf); buf = nullptr; cap = 0;
  const size_t want = need + need / 4 + 4096;
  TBNAV_HIP(hipMalloc((void**)&buf, want));
  cap = want;
  return TBNAV_OK;
}

// ParticleFilter::SLAM over the members' shards.  n == 1: this process's rank of a multi-process filter; n > 1: every member of a
// one-process group, in rank order.  Per scan and member, on the device:
//   main stream   noise -> propose -> [event: weights final] -> map update ........................ -> (resample: migration)
//   second stream                      wait -> ONE all-gather of the raw weights -> the reference's sequential normalise /
//                                      Neff / selection over the GLOBAL vector (identical on every rank) -> own slice back
// so the chain of adds of the global normalise (which grows with the ensemble, not with the shard) runs BESIDE the local map
// update, and the host waits once, for both streams.  Only when resampling fires do particles move: one all-gather of blob
// sizes, one batched export per rank, one message per (source, destination) pair, one batched import (tbnav_rbpf_export_batch_*
// / _import_batch_dev), and an all-gather of the ranks' statuses so that a rank whose pool is exhausted stops everybody.
int sharded_scan(int n, tbnav_rbpf* const* hs, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* const* normals, tbnav_rbpf_stats* out,
                 tbnav_rbpf_stats* local_out) {
  if (n <= 0 || !hs || !scan || n_beams <= 0 || !u || !cur_odom || !prev_odom || !T_icp || !out) return TBNAV_ERR_INVALID_ARG;
  for (int r = 0; r < n; ++r) if (!hs[r] || !hs[r]->comm || hs[r]->N != hs[0]->N || hs[r]->ref_field) return TBNAV_ERR_INVALID_ARG;
  // (made at attach and never resized while attached — tbnav_rbpf_attach_comm; a handle without them was never attached)
  for (int r = 0; r < n; ++r) if (!hs[r]->d_gw_raw || !hs[r]->d_status || !hs[r]->stream2 || (size_t)tbnav::comm_size(hs[r]->comm) * hs[r]->N > hs[r]->g_cap) return TBNAV_ERR_INVALID_ARG;
  const int P = tbnav::comm_size(hs[0]->comm), nl = hs[0]->N;
  const size_t ng = (size_t)P * nl;
  if (ng > ((size_t)1 << 24)) return TBNAV_ERR_UNSUPPORTED;
  // (everything above is a function of arguments every rank shares: all ranks return together.  From here on a failure that
  //  only THIS rank sees — a launch that fails, an allocation, a pool that runs dry — must not make it leave while its peers
  //  wait in a collective that has no timeout: the rank notes the code in lerr[], skips its own work, KEEPS JOINING the
  //  collectives, and the ranks agree on a status before anyone acts on data that may be missing.  Round-3 advisor finding.)
  std::vector<tbnav_comm*> comms(n);
  std::vector<hipStream_t> s1(n), s2(n);
  std::vector<ScanTicket> tk(n);
  std::vector<tbnav_rbpf_stats> lst(n);
  std::vector<int> lerr(n, TBNAV_OK);
  auto note = [&](int r, int rc) { if (rc != TBNAV_OK && lerr[r] == TBNAV_OK) lerr[r] = rc; };
  auto hipok = [&](int r, hipError_t e, const char* what, int line) { if (e != hipSuccess) note(r, tbnav::hip_fail(e, what, __FILE__, line)); return e == hipSuccess; };
#define TBNAV_L(r, call) hipok(r, (call), #call, __LINE__)
  // the ranks' codes -> one status, the same on every rank (the lowest rank's failure); collective when ranks live elsewhere
  auto agree = [&](const std::vector<int>& codes, int& status) -> int {
    status = TBNAV_OK;
    if (n == P) { for (int r = 0; r < n; ++r) if (codes[r] != TBNAV_OK && status == TBNAV_OK) status = codes[r]; return TBNAV_OK; }
    // (the same rule inside the agreement itself: a copy that fails on this rank is a code this rank contributes — if its word
    //  cannot even be uploaded, the word it holds is whatever the last agreement left, and the rank still reports its own
    //  failure below — never a return before the all-gather its peers are entering)
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    int local_fail = TBNAV_OK;
    for (int r = 0; r < n; ++r) {
      DeviceGuard guard(hs[r]->device);
      const hipError_t e = hipMemcpyAsync(hs[r]->d_status, &codes[r], sizeof(int), hipMemcpyHostToDevice, hs[r]->stream);
      if (e != hipSuccess && local_fail == TBNAV_OK) local_fail = tbnav::hip_fail(e, "agree: status upload", __FILE__, __LINE__);
      send[r] = hs[r]->d_status; recv[r] = hs[r]->d_status + 1;
    }
    { const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(int), s1.data()); if (rc != TBNAV_OK) return rc; }
    std::vector<int> all(P, TBNAV_OK);
    { DeviceGuard guard(hs[0]->device);
      hipError_t e = hipMemcpyAsync(all.data(), hs[0]->d_status + 1, sizeof(int) * P, hipMemcpyDeviceToHost, hs[0]->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(hs[0]->stream);
      if (e != hipSuccess && local_fail == TBNAV_OK) local_fail = tbnav::hip_fail(e, "agree: status download", __FILE__, __LINE__); }
    for (int q = 0; q < P; ++q) if (all[q] != TBNAV_OK) { status = all[q]; break; }
    // A failure of the agreement's own copies on THIS rank is not this scan's status (round-5 advisor finding: the rank returned
    // while its peers, who agreed on OK, went on into the scan's next collectives and waited for it).  It is latched: the rank goes
    // on with the agreed status, keeps joining this scan's collectives, and contributes the failure to the NEXT scan's first
    // agreement, where every rank stops with it.
    if (local_fail != TBNAV_OK) for (int r = 0; r < n; ++r) if (hs[r]->shard_latched == TBNAV_OK) hs[r]->shard_latched = local_fail;
    return TBNAV_OK;
  };
  std::memset(out, 0, sizeof *out);
  // ---- A: every member's local scan (no normalise tail), its "weights are final" event recorded behind the proposal kernel
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    comms[r] = h->comm; s1[r] = h->stream; s2[r] = h->stream2;
    note(r, h->shard_latched);
    if (lerr[r] == TBNAV_OK)
      note(r, scan_enqueue(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals ? normals[r] : nullptr, &lst[r], true, 0, nullptr, tk[r], nullptr, h->ev_w));
    if (lerr[r] == TBNAV_OK) TBNAV_L(r, hipStreamWaitEvent(h->stream2, h->ev_w, 0));
  }
  // ---- B: the ONE collective of the update + the global normalise / selection, on the second streams
  //         (a member that failed above still takes part — with whatever its weight buffer holds: nobody will use the result)
  {
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    for (int r = 0; r < n; ++r) { send[r] = state_ptrs(hs[r]->d_state[hs[r]->cur], nl).weight; recv[r] = hs[r]->d_gw_raw; }
    const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(double) * nl, s2.data());
    if (rc != TBNAV_OK) return rc;   // (the communicator itself failed: it reports on every rank)
  }
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    if (lerr[r] != TBNAV_OK) continue;
    DeviceGuard guard(h->device);
    h->h_norm[1] = NormOut{};
    // the resampling offset: the scan's last normal — with tbnav_rbpf_set_rng_shard (device noise) the ENSEMBLE's, identical on every rank
    const double* zp = h->last_z_ptr;
    hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, h->stream2, (int)ng, zp, h->d_gw_raw, h->d_gw, h->d_gcs, h->d_gparent, h->d_norm + 1,
                       nullptr, nullptr, nullptr, 0u);
    (void)TBNAV_L(r, hipGetLastError());
    // (the normalised weights go back into the shard only once the ranks have AGREED that this scan succeeded everywhere — below:
    //  a failed rank's slice of the gathered vector is whatever its buffer held)
  }
  // ---- C: the host waits once per member (the reference's SLAM() is synchronous)
  std::vector<int> lstat(n, TBNAV_OK);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    TBNAV_L(r, hipStreamSynchronize(h->stream2));
    // what the reference reports by throwing (a particle left the world, eta is 0 ...) happens to the rank that holds the particle
    lstat[r] = lerr[r] != TBNAV_OK ? lerr[r] : scan_finish(h, tk[r], &lst[r]);
    if (local_out) local_out[r] = lst[r];
  }
  // Every rank must stop at the SAME scan with the same status — a rank that went on alone would sit in the next scan's
  // all-gather for ever: one all-gather of the ranks' statuses per scan (4 bytes each; ~1 % of a scan) when ranks live elsewhere.
  int status = TBNAV_OK;
  { const int rc = agree(lstat, status); if (rc != TBNAV_OK) return rc; }
  const NormOut no = hs[0]->h_norm[1];
  out->sum_w = no.sum_w; out->sq_sum = no.sq_sum; out->neff = no.neff; out->resampled = no.resampled;
  out->n_valid_beams = lst[0].n_valid_beams;
  out->status = status;
  if (status != TBNAV_OK) return status;
  // the scan stands on every rank: each shard takes its slice of the globally normalised weights (on its main stream — the host
  // has waited for the second one above; whatever the main stream does next sees them)
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const size_t off = (size_t)tbnav::comm_rank(h->comm) * nl;
    TBNAV_L(r, hipMemcpyAsync(state_ptrs(h->d_state[h->cur], nl).weight, h->d_gw + off, sizeof(double) * nl, hipMemcpyDeviceToDevice, h->stream));
  }
  // (a copy that could not even be enqueued: the ranks have already agreed on this scan — the code is latched and stops every rank
  //  at the next scan's agreement, or at this one's if a resampling follows)
  for (int r = 0; r < n; ++r) if (lerr[r] != TBNAV_OK) hs[r]->shard_latched = lerr[r];
  if (!no.resampled) return TBNAV_OK;
  // ---- D: lowVarianceResampling's copies across shards.  Slot m (global) takes particle parents[m].
  std::vector<int> parents(ng);
  { DeviceGuard guard(hs[0]->device); if (!TBNAV_L(0, hipMemcpy(parents.data(), hs[0]->d_gparent, sizeof(int) * ng, hipMemcpyDeviceToHost))) std::fill(parents.begin(), parents.end(), 0); }
  struct Plan { std::vector<std::pair<int, int>> sends, recvs; std::vector<int32_t> send_slots; std::vector<uint64_t> send_sizes, send_offs; std::vector<unsigned long long> sizes_local; };
  std::vector<Plan> plan(n);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const int me = tbnav::comm_rank(h->comm), lo = me * nl;
    Plan& pl = plan[r];
    for (size_t m = 0; m < ng; ++m) {  // (dst, q): every particle of mine some other rank's slot chose — once per destination
      const int q = parents[m], dst = (int)(m / nl);
      if (q / nl == me && dst != me) pl.sends.emplace_back(dst, q);
    }
    std::sort(pl.sends.begin(), pl.sends.end());
    pl.sends.erase(std::unique(pl.sends.begin(), pl.sends.end()), pl.sends.end());
    for (int m = lo; m < lo + nl; ++m) { const int q = parents[m]; if (q / nl != me) pl.recvs.emplace_back(q / nl, q); }
    std::sort(pl.recvs.begin(), pl.recvs.end());
    pl.recvs.erase(std::unique(pl.recvs.begin(), pl.recvs.end()), pl.recvs.end());
    pl.send_slots.resize(pl.sends.size());
    for (size_t i = 0; i < pl.sends.size(); ++i) pl.send_slots[i] = pl.sends[i].second - lo;
    pl.send_sizes.assign(pl.sends.size(), 0);
    if (lerr[r] == TBNAV_OK) note(r, tbnav_rbpf_export_batch_sizes(h, (int32_t)pl.sends.size(), pl.send_slots.data(), pl.send_sizes.data()));
    if (lerr[r] != TBNAV_OK) std::fill(pl.send_sizes.begin(), pl.send_sizes.end(), 0);
    // what a particle of mine weighs, for whoever receives it (a particle sent to several ranks weighs the same for each)
    pl.sizes_local.assign(nl, 0ull);
    for (size_t i = 0; i < pl.sends.size(); ++i) pl.sizes_local[pl.sends[i].second - lo] = pl.send_sizes[i];
    TBNAV_L(r, hipMemcpyAsync(h->d_sizes, pl.sizes_local.data(), sizeof(unsigned long long) * nl, hipMemcpyHostToDevice, h->stream));
  }
  {
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    for (int r = 0; r < n; ++r) { send[r] = hs[r]->d_sizes; recv[r] = hs[r]->d_sizes + nl; }
    const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(unsigned long long) * nl, s1.data());
    if (rc != TBNAV_OK) return rc;
  }
  std::vector<std::vector<tbnav::P2P>> p_send(n), p_recv(n);
  std::vector<std::vector<uint64_t>> recv_offs(n);
  std::vector<unsigned long long> sizes_all(ng);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    Plan& pl = plan[r];
    if (!(TBNAV_L(r, hipMemcpyAsync(sizes_all.data(), h->d_sizes + nl, sizeof(unsigned long long) * ng, hipMemcpyDeviceToHost, h->stream)) &&
          TBNAV_L(r, hipStreamSynchronize(h->stream)))) std::fill(sizes_all.begin(), sizes_all.end(), 0ull);
    // everything this rank sends: ONE export, the blobs back to back in (destination, particle) order
    uint64_t total = 0;
    for (uint64_t b : pl.send_sizes) total += b;
    pl.send_offs.assign(pl.sends.size() + 1, 0);
    if (lerr[r] == TBNAV_OK) note(r, grow(h->d_sendbuf, h->send_cap, (size_t)total));
    if (lerr[r] == TBNAV_OK) note(r, tbnav_rbpf_export_batch_dev(h, (int32_t)pl.sends.size(), pl.send_slots.data(), h->d_sendbuf, total, pl.send_offs.data()));
    for (size_t i = 0; i < pl.sends.size();) {  // one message per destination
      size_t j = i;
      while (j < pl.sends.size() && pl.sends[j].first == pl.sends[i].first) ++j;
      p_send[r].push_back(tbnav::P2P{pl.sends[i].first, h->d_sendbuf + pl.send_offs[i], (size_t)(pl.send_offs[j] - pl.send_offs[i])});
      i = j;
    }
    // everything it receives: one buffer, the blobs in (source, particle) order
    recv_offs[r].assign(pl.recvs.size() + 1, 0);
    for (size_t i = 0; i < pl.recvs.size(); ++i) recv_offs[r][i + 1] = recv_offs[r][i] + sizes_all[pl.recvs[i].second];
    if (lerr[r] == TBNAV_OK) note(r, grow(h->d_recvbuf, h->recv_cap, (size_t)recv_offs[r].back()));
    for (size_t i = 0; i < pl.recvs.size();) {
      size_t j = i;
      while (j < pl.recvs.size() && pl.recvs[j].first == pl.recvs[i].first) ++j;
      p_recv[r].push_back(tbnav::P2P{pl.recvs[i].first, h->d_recvbuf + recv_offs[r][i], (size_t)(recv_offs[r][j] - recv_offs[r][i])});
      i = j;
    }
  }
  // is every rank ready to send what the sizes promised and to receive it?  A rank whose export or allocation failed cannot
  // honour its messages (its peers would wait for bytes that never come): agree BEFORE the exchange; nobody has touched a slot yet
  { const int rc = agree(lerr, status); if (rc != TBNAV_OK) return rc; }
  if (status != TBNAV_OK) { out->status = status; return status; }
  { const int rc = tbnav::comm_exchange(n, comms.data(), p_send.data(), p_recv.data(), s1.data()); if (rc != TBNAV_OK) return rc; }
  // local parents inside the handle (tile tables + reference counts), then the imported ones; weights are NOT reset by the
  // reference: every slot carries its parent's normalised weight
  std::vector<int> mstat(n, TBNAV_OK);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const int me = tbnav::comm_rank(h->comm), lo = me * nl;
    Plan& pl = plan[r];
    std::vector<int32_t> local_parent(nl), imp_slots;
    std::vector<uint64_t> imp_offs;
    for (int m = 0; m < nl; ++m) {
      const int q = parents[lo + m];
      if (q / nl == me) local_parent[m] = q - lo;
      else {
        local_parent[m] = -1;
        const auto it = std::lower_bound(pl.recvs.begin(), pl.recvs.end(), std::make_pair(q / nl, q));
        imp_slots.push_back(m);
        imp_offs.push_back(recv_offs[r][(size_t)(it - pl.recvs.begin())]);
      }
    }
    int rc = tbnav_rbpf_gather_local(h, local_parent.data());
    if (rc == TBNAV_OK && !imp_slots.empty())
      rc = tbnav_rbpf_import_batch_dev(h, (int32_t)imp_slots.size(), imp_slots.data(), h->d_recvbuf, recv_offs[r].back(), imp_offs.data());
    if (rc == TBNAV_OK) rc = tbnav_rbpf_set_weights_from_global_dev(h, parents.data() + lo);
    mstat[r] = rc;
  }
  // a rank that failed (tile pool exhausted) must not leave the others waiting in the next scan's collective: agree on it
  { const int rc = agree(mstat, status); if (rc != TBNAV_OK) return rc; }
#undef TBNAV_L
  out->status = status;
  return status;
}

}  // namespace

// One process driving several GPUs: the whole filter behind one object (what bmapping::ParticleFilter built with n_gpus > 1 holds).
struct tbnav_rbpf_group {
  int n = 0, n_global = 0;
  std::vector<tbnav_rbpf*> m;
  std::vector<tbnav_comm*> c;
  std::vector<std::vector<double>> normals;  // parity mode: each member's slice of the ensemble's draw stream + the offset
};

extern "C" {

int tbnav_rbpf_attach_comm(tbnav_rbpf* h, tbnav_comm* comm) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (comm && (h->ref_field || tbnav_comm_device(comm) != h->device)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  h->comm = comm;
  h->shard_latched = TBNAV_OK;
  if (!comm) { h->rng_first = 0; h->rng_n_global = 0; return TBNAV_OK; }
  // equal shards: this rank's particles are [rank * N, (rank + 1) * N) of nranks * N — also for the device noise source
  h->rng_first = (uint64_t)tbnav::comm_rank(comm) * (uint64_t)h->N;
  h->rng_n_global = (uint64_t)tbnav::comm_size(comm) * (uint64_t)h->N;
  // the buffers every collective of a scan works on exist from here on (nranks * N is fixed for the attachment): a handle
  // whose shard state cannot be made is NOT attached — sharded_scan's precondition, so that no rank finds itself without
  // something to join a collective with in the middle of a scan (round-4 advisor finding)
  const int rc = ensure_shard_state(h);
  if (rc != TBNAV_OK) { h->comm = nullptr; h->rng_first = 0; h->rng_n_global = 0; }
  return rc;
}

void tbnav_rbpf_group_destroy(tbnav_rbpf_group* g) {
  if (!g) return;
  for (int r = 0; r < g->n; ++r) {
    if (r < (int)g->m.size()) tbnav_rbpf_destroy(g->m[r]);
    if (r < (int)g->c.size()) tbnav_comm_destroy(g->c[r]);
  }
  delete g;
}

int tbnav_rbpf_group_create(const tbnav_rbpf_params* params, int32_t n_gpus, const int32_t* devices, uint64_t max_pool_bytes_per_member,
                            tbnav_rbpf_group** out) {
  if (!params || !out || n_gpus <= 0 || params->num_particles <= 0 || params->num_particles % n_gpus != 0) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  tbnav_rbpf_group* g = new (std::nothrow) tbnav_rbpf_group();
  if (!g) return TBNAV_ERR_INVALID_ARG;
  g->n = n_gpus; g->n_global = params->num_particles;
  g->m.assign(n_gpus, nullptr); g->c.assign(n_gpus, nullptr); g->normals.resize(n_gpus);
  int rc = tbnav_comm_create_local(n_gpus, devices, g->c.data());
  for (int r = 0; r < n_gpus && rc == TBNAV_OK; ++r) {
    tbnav_rbpf_params p = *params;
    p.num_particles = params->num_particles / n_gpus;
    p.device = tbnav_comm_device(g->c[r]);
    rc = create_impl(&p, max_pool_bytes_per_member, &g->m[r]);
    if (rc == TBNAV_OK) {
      // initParticleSet gives every particle weight 1 / N of the WHOLE filter (particle_filter.cpp:134)
      std::vector<double> w((size_t)p.num_particles, 1.0 / params->num_particles);
      rc = tbnav_rbpf_set_particles(g->m[r], nullptr, nullptr, w.data());
    }
    if (rc == TBNAV_OK) rc = tbnav_rbpf_attach_comm(g->m[r], g->c[r]);
  }
  if (rc != TBNAV_OK) { tbnav_rbpf_group_destroy(g); return rc; }
  *out = g;
  return TBNAV_OK;
}

int tbnav_rbpf_group_size(const tbnav_rbpf_group* g) { return g ? g->n : -1; }
int tbnav_rbpf_group_member(tbnav_rbpf_group* g, int32_t rank, tbnav_rbpf** out) {
  if (!g || !out || rank < 0 || rank >= g->n) return TBNAV_ERR_INVALID_ARG;
  *out = g->m[rank];
  return TBNAV_OK;
}
int tbnav_rbpf_group_set_seed(tbnav_rbpf_group* g, uint64_t seed) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_rbpf* h : g->m) { const int rc = tbnav_rbpf_set_seed(h, seed); if (rc != TBNAV_OK) return rc; }  // one seed: the members draw disjoint slices of its stream
  return TBNAV_OK;
}
int tbnav_rbpf_group_set_option(tbnav_rbpf_group* g, int32_t option, int32_t value) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_rbpf* h : g->m) { const int rc = tbnav_rbpf_set_option(h, option, value); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int64_t tbnav_rbpf_group_num_normals(const tbnav_rbpf_group* g, int32_t icp_ok) {
  if (!g) return -1;
  return (int64_t)g->n_global * (icp_ok ? 3 * g->m[0]->k + 3 : 3) + 1;
}

// normals: the ENSEMBLE's draw stream in the reference's order (tbnav_rbpf_group_num_normals values: particle-major, the
// resampling offset last) or NULL (device noise: every member draws its slice of one stream).
int tbnav_rbpf_group_slam(tbnav_rbpf_group* g, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                          const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals, tbnav_rbpf_stats* out) {
  if (!g || !out) return TBNAV_ERR_INVALID_ARG;
  std::vector<const double*> nr(g->n, nullptr);
  if (normals) {
    const size_t stride = icp_ok ? 3 * (size_t)g->m[0]->k + 3 : 3, nl = (size_t)g->m[0]->N;
    for (int r = 0; r < g->n; ++r) {
      std::vector<double>& v = g->normals[r];
      v.resize(nl * stride + 1);
      std::memcpy(v.data(), normals + (size_t)r * nl * stride, sizeof(double) * nl * stride);
      v[nl * stride] = normals[(size_t)g->n_global * stride];
      nr[r] = v.data();
    }
  }
  return sharded_scan(g->n, g->m.data(), scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals ? nr.data() : nullptr, out, nullptr);
}

// ParticleFilter::getRobotState over the ensemble: strict >, first wins (particle_filter.cpp:255-274) — members in rank order
int tbnav_rbpf_group_best_state(tbnav_rbpf_group* g, double pose[3], int32_t* best_index) {
  if (!g || !pose) return TBNAV_ERR_INVALID_ARG;
  double best_w = 0.0; int best_r = 0, best_i = 0; double best_pose[3] = {0, 0, 0};
  bool have = false;
  for (int r = 0; r < g->n; ++r) {
    double p[3]; int32_t idx = 0;
    int rc = tbnav_rbpf_best_state(g->m[r], p, &idx);
    if (rc != TBNAV_OK) return rc;
    double w = 0.0;
    { DeviceGuard guard(g->m[r]->device); TBNAV_HIP(hipMemcpy(&w, state_ptrs(g->m[r]->d_state[g->m[r]->cur], g->m[r]->N).weight + idx, sizeof w, hipMemcpyDeviceToHost)); }
    // (a member whose weights are all <= 0.0 reports its slot 0, as the reference's loop would keep index 0)
    if (!have || w > best_w) { best_w = w; best_r = r; best_i = idx; std::memcpy(best_pose, p, sizeof p); have = true; }
  }
  std::memcpy(pose, best_pose, sizeof best_pose);
  if (best_index) *best_index = best_r * g->m[0]->N + best_i;
  return TBNAV_OK;
}
int tbnav_rbpf_group_best_map(tbnav_rbpf_group* g, int8_t* map) {
  if (!g || !map) return TBNAV_ERR_INVALID_ARG;
  double pose[3]; int32_t idx = 0;
  const int rc = tbnav_rbpf_group_best_state(g, pose, &idx);
  if (rc != TBNAV_OK) return rc;
  const int nl = g->m[0]->N;
  return tbnav_rbpf_particle_map(g->m[idx / nl], idx % nl, map);
}

int tbnav_rbpf_copy_particle(tbnav_rbpf* dst, int32_t dst_slot, tbnav_rbpf* src, int32_t src_slot) {
  if (!dst || !src || dst_slot < 0 || dst_slot >= dst->N || src_slot < 0 || src_slot >= src->N) return TBNAV_ERR_INVALID_ARG;
  if (dst->xsize != src->xsize || dst->ref_field != src->ref_field || dst->device != src->device) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(src->device);
  uint64_t bytes = 0;
  int rc = tbnav_rbpf_export_size(src, src_slot, &bytes);
  if (rc != TBNAV_OK) return rc;
  void* buf = nullptr;
  TBNAV_HIP(hipMalloc(&buf, bytes));
  rc = tbnav_rbpf_export_particle_dev(src, src_slot, buf, bytes, nullptr);
  if (rc == TBNAV_OK) rc = tbnav_rbpf_import_particle_dev(dst, dst_slot, buf, bytes);
  (void)hipFree(buf);
  if (rc == TBNAV_OK && src->ref_field) {  // the set with its history, the field with its stale cells
    dst->ref->copy_slot(dst_slot, *src->ref, src_slot);   // (the slot's device content counts as unknown: the next flush uploads the state's image)
  }
  return rc;
}

int tbnav_rbpf_get_particles(tbnav_rbpf* h, double* pose, double* prev_pose, double* weight) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const int N = h->N;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], N);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (pose) TBNAV_HIP(hipMemcpy(pose, sp.pose, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
  if (prev_pose) TBNAV_HIP(hipMemcpy(prev_pose, sp.prev, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
  if (weight) TBNAV_HIP(hipMemcpy(weight, sp.weight, sizeof(double) * N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_set_particles(tbnav_rbpf* h, const double* pose, const double* prev_pose, const double* weight) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const int N = h->N;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], N);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (pose) TBNAV_HIP(hipMemcpy(sp.pose, pose, sizeof(double) * 3 * N, hipMemcpyHostToDevice));
  if (prev_pose) TBNAV_HIP(hipMemcpy(sp.prev, prev_pose, sizeof(double) * 3 * N, hipMemcpyHostToDevice));
  if (weight) TBNAV_HIP(hipMemcpy(sp.weight, weight, sizeof(double) * N, hipMemcpyHostToDevice));
  return TBNAV_OK;
}

int tbnav_rbpf_get_log_odds(tbnav_rbpf* h, int32_t particle, double* out) {
  if (!h || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  if (!h->d_dense) TBNAV_HIP(hipMalloc((void**)&h->d_dense, sizeof(double) * h->G));
  const int blocks = (int)std::min<size_t>((h->G + 255) / 256, 4096);
  hipLaunchKernelGGL(rbpf_tiles_to_dense, dim3(blocks), dim3(256), 0, h->stream, h->xsize, h->G, h->pool, map_of(h), particle, h->d_dense);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(out, h->d_dense, sizeof(double) * h->G, hipMemcpyDeviceToHost, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}

int tbnav_rbpf_set_log_odds(tbnav_rbpf* h, int32_t particle, const double* in) {
  if (!h || !in || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  if (!h->d_dense) TBNAV_HIP(hipMalloc((void**)&h->d_dense, sizeof(double) * h->G));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(h->d_dense, in, sizeof(double) * h->G, hipMemcpyHostToDevice));
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  // the tiles take the new log-odds and the occupancy bits they imply; the particle's occupied counts are rebuilt
  TBNAV_HIP(hipMemsetAsync(h->d_nocc[h->cur] + particle, 0, sizeof(int), h->stream));
  TBNAV_HIP(hipMemsetAsync(h->d_trow[h->cur] + (size_t)particle * h->TW, 0, sizeof(int) * h->TW, h->stream));
  hipLaunchKernelGGL(rbpf_dense_to_tiles, dim3(h->TT), dim3(kWave), 0, h->stream, h->xsize, h->cut_occ, h->pool, map_of(h), particle, h->d_dense,
                     h->d_trow[h->cur], h->d_nocc[h->cur], h->d_err);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (h->h_err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;
  const int zero = 0;  // the distance field no longer matches the map
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &zero, sizeof zero, hipMemcpyHostToDevice));
  if (h->ref_field) {  // the occupied set's history is unknown from here on: ascending order (documented in tbnav_rbpf.h)
    std::vector<int> cells;
    for (size_t c = 0; c < h->G; ++c) if (in[c] >= h->cut_occ) cells.push_back((int)c);
    h->ref->reset(particle, cells);
    h->ref->forget_slot(particle);
  }
  return TBNAV_OK;
}

int tbnav_rbpf_get_dist_code(tbnav_rbpf* h, int32_t particle, uint16_t* out) {
  if (!h || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  if (h->ref_field) { const int rc = ref_field_materialize(h, particle); if (rc != TBNAV_OK) return rc; }  // the pass to its end, stale cells by replay
  else { const int rc = ensure_full_field(h, particle); if (rc != TBNAV_OK) return rc; }  // whole field on demand
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(out, h->d_code[h->cur] + (size_t)particle * h->G, sizeof(uint16_t) * h->G, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_get_occ_dist(tbnav_rbpf* h, int32_t particle, double* out) {
  if (!h || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  std::vector<uint16_t> code(h->G);
  int rc = tbnav_rbpf_get_dist_code(h, particle, code.data());
  if (rc != TBNAV_OK) return rc;
  for (size_t c = 0; c < h->G; ++c)
    out[c] = code[c] == kCodeUnreached ? h->max_occ_dist : std::sqrt((double)code[c]) * h->p.resolution;
  return TBNAV_OK;
}

int tbnav_rbpf_set_occ_dist(tbnav_rbpf* h, int32_t particle, const double* in) {
  if (!h || !in || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  std::vector<uint16_t> code(h->G);
  const double res = h->p.resolution;
  for (size_t c = 0; c < h->G; ++c) {
    const double v = in[c];
    const double cells = v / res;
    const long d2 = std::lround(cells * cells);
    if (d2 >= 0 && d2 < 65535 && std::sqrt((double)d2) * res == v) { code[c] = (uint16_t)d2; continue; }
    if (v == h->max_occ_dist) { code[c] = kCodeUnreached; continue; }
    return TBNAV_ERR_INVALID_ARG;
  }
  DeviceGuard guard(h->device);
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (h->ref_field) { h->ref->set_codes(particle, code.data()); h->ref->forget_slot(particle); }
  TBNAV_HIP(hipMemcpy(h->d_code[h->cur] + (size_t)particle * h->G, code.data(), sizeof(uint16_t) * h->G, hipMemcpyHostToDevice));
  const int two = 2;  // an injected field is authoritative: the next call does not refresh it
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
  h->fstate_dirty = true;
  return TBNAV_OK;
}

int tbnav_rbpf_get_occupied_count(tbnav_rbpf* h, int32_t* counts) {
  if (!h || !counts) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(counts, h->d_nocc[h->cur], sizeof(int) * h->N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_get_trace(tbnav_rbpf* h, double* sampled, double* p_scan, double* p_pose, double* mu, double* sigma,
                         double* eta, double* new_pose, double* weight_raw, int32_t* resample_parent) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const size_t N = h->N, k = h->k;
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  auto get = [&](double* dst, const double* src, size_t n) -> hipError_t {
    return dst ? hipMemcpy(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost) : hipSuccess;
  };
  TBNAV_HIP(get(sampled, h->tr.sampled, N * k * 3));
  TBNAV_HIP(get(p_scan, h->tr.p_scan, N * k));
  TBNAV_HIP(get(p_pose, h->tr.p_pose, N * k));
  TBNAV_HIP(get(mu, h->tr.mu, N * 3));
  TBNAV_HIP(get(sigma, h->tr.sigma, N * 9));
  TBNAV_HIP(get(eta, h->tr.eta, N));
  TBNAV_HIP(get(new_pose, h->tr.new_pose, N * 3));
  TBNAV_HIP(get(weight_raw, h->tr.weight_raw, N));
  if (resample_parent) TBNAV_HIP(hipMemcpy(resample_parent, h->d_parent, sizeof(int) * N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_best_state(tbnav_rbpf* h, double pose[3], int32_t* best_index) {
  if (!h || !pose) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  hipLaunchKernelGGL(rbpf_argmax, dim3(1), dim3(256), 0, st, h->N, sp.weight, sp.pose, h->d_best, h->d_best_pose);
  TBNAV_HIP(hipGetLastError());
  int idx = 0;
  TBNAV_HIP(hipMemcpyAsync(pose, h->d_best_pose, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipMemcpyAsync(&idx, h->d_best, sizeof(int), hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  if (best_index) *best_index = idx;
  return TBNAV_OK;
}

int tbnav_rbpf_best_map(tbnav_rbpf* h, int8_t* map) {
  if (!h || !map) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  hipLaunchKernelGGL(rbpf_argmax, dim3(1), dim3(256), 0, st, h->N, sp.weight, sp.pose, h->d_best, h->d_best_pose);
  TBNAV_HIP(hipGetLastError());
  const int blocks = (int)std::min<size_t>((h->G + 255) / 256, 2048);
  hipLaunchKernelGGL(rbpf_export_map, dim3(blocks), dim3(256), 0, st, h->xsize, h->G, h->cuts, h->d_best, h->pool, map_of(h),
                     h->d_export);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(map, h->d_export, h->G, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  return TBNAV_OK;
}

int tbnav_rbpf_set_scan_matching(tbnav_rbpf* h, int32_t enable, double lstep, double astep, int32_t iterations) {
  if (!h || (enable && (!(lstep > 0.0) || !(astep > 0.0) || iterations < 1 || iterations > 32))) return TBNAV_ERR_INVALID_ARG;
  h->sm_on = enable != 0;
  if (enable) { h->sm.lstep = lstep; h->sm.astep = astep; h->sm.iters = iterations; h->sm.max_moves = 64; }
  return TBNAV_OK;
}

int tbnav_rbpf_get_scan_match(tbnav_rbpf* h, double* centers, double* scores) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (centers) TBNAV_HIP(hipMemcpy(centers, h->d_center, sizeof(double) * 3 * h->N, hipMemcpyDeviceToHost));
  if (scores) TBNAV_HIP(hipMemcpy(scores, h->d_score, sizeof(double) * h->N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

// ---- one particle's GridMapper, for the host class bmapping::GridMapper (grid_mapper.hpp:128-140) ---------------
namespace {
int one_particle_consts(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, ScanC& c) {
  const double zero[3] = {0.0, 0.0, 0.0};
  std::vector<double2> beams;
  int rc = build_scan_consts(h, c, scan, n_beams, zero, zero, zero, 1, zero, beams);
  if (rc != TBNAV_OK) return rc;
  c.p0 = particle;
  return upload_beams(h, beams, n_beams, c.Bv);
}
}  // namespace

int tbnav_rbpf_integrate_scan(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, const double pose[3]) {
  if (!h || !scan || n_beams <= 0 || !pose || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  ++h->scans_done;
  ScanC c;
  int rc = one_particle_consts(h, particle, scan, n_beams, c);
  if (rc != TBNAV_OK) return rc;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(sp.pose + (size_t)particle * 3, pose, sizeof(double) * 3, hipMemcpyHostToDevice));
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  rc = launch_raycast(h, c, 1, nullptr);
  if (rc != TBNAV_OK) return rc;
  const int zero = 0;  // the map changed: a stored field of this particle is stale
  TBNAV_HIP(hipMemcpyAsync(h->d_fstate + particle, &zero, sizeof zero, hipMemcpyHostToDevice, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  rc = status_from_err(h->h_err);
  if (rc != TBNAV_OK) return rc;
  if (h->ref_field) return ref_field_after_scan(h, false, particle, 1);
  if (h->df_mode != 2) return ensure_full_field(h, particle);  // stored-field modes: the whole field after the update
  return TBNAV_OK;
}

int tbnav_rbpf_likelihood(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, const double pose[3], double* out) {
  if (!h || !scan || n_beams <= 0 || !pose || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  ScanC c;
  int rc = one_particle_consts(h, particle, scan, n_beams, c);
  if (rc != TBNAV_OK) return rc;
  if (h->ref_field) { rc = ref_field_materialize(h, particle); if (rc != TBNAV_OK) return rc; }  // (a lookup anywhere: the whole field)
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  hipLaunchKernelGGL(rbpf_likelihood_one, dim3(1), dim3(kWave), 0, h->stream, c, h->d_beams, h->d_code[h->cur], h->pool, map_of(h),
                     h->d_trow[h->cur], h->d_fstate, h->radius, h->d_nocc[h->cur], pose[0], pose[1], pose[2], h->d_score, h->d_err, h->d_mixlut);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(out, h->d_score, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return status_from_err(h->h_err);
}

int tbnav_rbpf_particle_map(tbnav_rbpf* h, int32_t particle, int8_t* map) {
  if (!h || !map || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  TBNAV_HIP(hipMemcpyAsync(h->d_best, &particle, sizeof(int), hipMemcpyHostToDevice, st));
  const int blocks = (int)std::min<size_t>((h->G + 255) / 256, 2048);
  hipLaunchKernelGGL(rbpf_export_map, dim3(blocks), dim3(256), 0, st, h->xsize, h->G, h->cuts, h->d_best, h->pool, map_of(h), h->d_export);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(map, h->d_export, h->G, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  return TBNAV_OK;
}

int tbnav_rbpf_set_option(tbnav_rbpf* h, int32_t option, int32_t value) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  switch (option) {
    case TBNAV_RBPF_OPT_DF_MODE: {
      if (value < TBNAV_RBPF_DF_FULL || value > TBNAV_RBPF_DF_REFERENCE) return TBNAV_ERR_INVALID_ARG;
      if (h->scans_done) return TBNAV_ERR_INVALID_ARG;  // the mode belongs to the filter's whole life
      if (value == TBNAV_RBPF_DF_REFERENCE) {
        if (h->N > 4096) return TBNAV_ERR_UNSUPPORTED;  // serial host brushfire per particle: small ensembles only
        if (h->xsize > tbnav::RefField::kMaxSide || (long)h->radius * h->radius >= 65534) return TBNAV_ERR_UNSUPPORTED;  // (the host queue's nodes: 12-bit coordinates, 16-bit squared distances)
        { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
        delete h->ref;
        h->ref = new (std::nothrow) tbnav::RefField(h->N, h->xsize, h->radius);
        if (!h->ref) return TBNAV_ERR_INVALID_ARG;
        h->ref->set_reach(h->ref_reach);
        // (no scan yet: the maps are empty, and the slots were allocated holding "unreached" everywhere — the initial state's image)
        TBNAV_HIP(hipStreamSynchronize(h->stream));
        TBNAV_HIP(hipMemset(h->d_code[h->cur], 0xFF, sizeof(uint16_t) * h->G * (size_t)h->N));
        h->ref->slots_hold_initial_image();
        h->ref_field = true; h->df_mode = 2; h->full_edt = false;
        return TBNAV_OK;
      }
      if (value != TBNAV_RBPF_DF_QUERY) {
        if (h->edt_cols == 0) return TBNAV_ERR_UNSUPPORTED;  // no LDS transform for this map size
        { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
        // empty maps: the stored field "everything unreached" IS the whole, fresh field
        TBNAV_HIP(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(h->d_fstate), 2, h->N));
        h->fstate_dirty = true;
      }
      h->ref_field = false; h->df_mode = value; h->full_edt = value == TBNAV_RBPF_DF_FULL;
      return TBNAV_OK;
    }
    case TBNAV_RBPF_OPT_REF_REACH:   // reference-field mode: cells a scan's brushfire runs out to before it stops (0: to the end, as up to round 5)
      if (value < 0 || value > 65535) return TBNAV_ERR_INVALID_ARG;
      h->ref_reach = value;
      if (h->ref) h->ref->set_reach(value);
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_ORDERED:
      if (value) h->tile_cap = 0;
      else {
        const double reach = (double)h->p.range_max + std::hypot(h->p.Trs[1], h->p.Trs[2]);
        const long side = 2 * ((long)std::ceil(reach / h->p.resolution) + 2) + 1;
        h->tile_cap = (side * side <= 30000) ? (int)(side * side) : 0;
      }
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_THREADS:
      if (value != 0 && value != 256 && value != 512 && value != 1024) return TBNAV_ERR_INVALID_ARG;
      h->raycast_threads = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_BAND_ROWS:
      if (value < 0) return TBNAV_ERR_INVALID_ARG;
      h->raycast_band_rows = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_CELL16:
      if (value < 0 || value > 2) return TBNAV_ERR_INVALID_ARG;
      h->raycast_cell16 = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_ADAPT:
      if (value < 0 || value > 3) return TBNAV_ERR_INVALID_ARG;
      h->raycast_adapt = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_BATCH_PIPELINE:
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->batch_pipeline = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_HOST_THREADS:
      if (value < 0 || value > 256) return TBNAV_ERR_INVALID_ARG;
      h->host_threads = value ? value : default_host_threads();
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_FORM:
      // retired: 1 named round 2's tile kernel (removed in round 4) and for a while silently meant the much slower beam-ordered
      // kernel instead — that one has its own switch, _RAYCAST_ORDERED (round-4 advisor finding)
      return value == 0 ? TBNAV_OK : TBNAV_ERR_INVALID_ARG;
    case TBNAV_RBPF_OPT_COUNT_CELLS:
      h->count_touched = value != 0;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_NOISE_IN_KERNEL:
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->noise_in_kernel = value;
      return TBNAV_OK;
    default: return TBNAV_ERR_INVALID_ARG;
  }
}

int tbnav_rbpf_scan_counts(tbnav_rbpf* h, uint64_t* cell_updates, uint64_t* distinct_cells, int32_t reset) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  unsigned long long v[2] = {0, 0};
  TBNAV_HIP(hipMemcpy(v, h->d_touched, sizeof v, hipMemcpyDeviceToHost));
  if (cell_updates) *cell_updates = v[0];
  if (distinct_cells) *distinct_cells = v[1];
  if (reset) TBNAV_HIP(hipMemset(h->d_touched, 0, sizeof v));
  return TBNAV_OK;
}

int tbnav_rbpf_reference_field_counts(tbnav_rbpf* h, int32_t* distinct_states, int32_t* last_brushfires, int64_t* total_brushfires) {
  if (!h || !h->ref_field || !h->ref) return TBNAV_ERR_INVALID_ARG;
  if (distinct_states) *distinct_states = h->ref->distinct_states();
  if (last_brushfires) *last_brushfires = h->ref->last_step_brushfires();
  if (total_brushfires) *total_brushfires = h->ref->total_brushfires();
  return TBNAV_OK;
}

int tbnav_rbpf_reference_field_stats(tbnav_rbpf* h, int64_t out[16]) {
  if (!h || !out || !h->ref_field || !h->ref) return TBNAV_ERR_INVALID_ARG;
  const tbnav::RefField::Counters& k = h->ref->counters();
  out[0] = k.passes; out[1] = k.pops; out[2] = k.resumes; out[3] = k.completions; out[4] = k.replays; out[5] = k.replay_generations;
  out[6] = h->ref->history_bytes(); out[7] = h->ref_reruns;
  for (int q = 0; q < 6; ++q) out[8 + q] = h->ref_us[q];
  out[14] = k.us_group; out[15] = k.us_bury;
  return TBNAV_OK;
}

int tbnav_rbpf_set_timing(tbnav_rbpf* h, int32_t enable) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->timing = enable != 0;
  return TBNAV_OK;
}

int tbnav_rbpf_last_kernel_names(const tbnav_rbpf* h, char* propose, int32_t propose_cap, char* raycast, int32_t raycast_cap, int32_t* raycast_workgroups) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (propose && propose_cap > 0) { if (h->lk_propose) snprintf(propose, (size_t)propose_cap, "rbpf_propose<%d, %s>", h->lk_propose, h->lk_propose_dn ? "true" : "false"); else propose[0] = 0; }
  if (raycast && raycast_cap > 0) {
    if (h->lk_raycast > 0) snprintf(raycast, (size_t)raycast_cap, "rbpf_raycast_box<%d, %d, %s, %d>", h->lk_raycast, h->lk_raycast_wps, h->lk_raycast_c16 ? "true" : "false", h->lk_raycast_ev);
    else if (h->lk_raycast == 0) snprintf(raycast, (size_t)raycast_cap, "rbpf_raycast");
    else raycast[0] = 0;
  }
  if (raycast_workgroups) *raycast_workgroups = h->lk_raycast_grid;
  return TBNAV_OK;
}

int tbnav_rbpf_raycast_box_cells(const tbnav_rbpf* h, int32_t* need_cells, int32_t* array_cells) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (need_cells) *need_cells = h->lk_box_need;
  if (array_cells) *array_cells = h->lk_box_cap;
  return TBNAV_OK;
}

int tbnav_rbpf_last_kernel_ms(tbnav_rbpf* h, float ms[TBNAV_RBPF_NKERNELS]) {
  if (!h || !ms) return TBNAV_ERR_INVALID_ARG;
  for (int i = 0; i < TBNAV_RBPF_NKERNELS; ++i) ms[i] = h->last_ms[i];
  return TBNAV_OK;
}

}  // extern "C"
