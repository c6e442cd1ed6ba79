// mppi.hip — MI355X (gfx950) implementation of controller::MPPI::newControls behind the C-ABI of include/tbnav_mppi.h: the handle,
// which kernel a (K, T) takes, the launchers and the single-GPU entry points.  Reference: controller/src/controller/mppi.cpp:72-140,
// rk4.cpp:49-115, controller/include/controller/mppi.hpp:41-105 (paths relative to the reference tree).
// Kernels: mppi_rollout.hip (four rollout families), mppi_softmin.hip (records, combine, exchange words, noise) over mppi_device.hpp;
// communicator attachment, the sharded tick and groups: mppi_sharded.hip; the handle: mppi_host.hpp.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <chrono>
#include <new>
#include <type_traits>
#include <vector>

#include "mppi_host.hpp"

using namespace tbnav_mh;

namespace tbnav_mh {
Lam lam_of(double lambda) {
  const double inv = 1.0 / lambda;
  unsigned long long bits; std::memcpy(&bits, &lambda, sizeof bits);
  const bool all_ones = (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;   // the one significand Markstein's theorem excludes
  return Lam{lambda, (std::isnormal(inv) && std::isnormal(lambda) && !all_ones) ? inv : 0.0};
}
}  // namespace tbnav_mh

namespace {

// the publication the sharded tick has asked for, taken over by the launch that can carry it out (see pub_pending)
DirectPub take_pub(tbnav_mppi* h) {
  if (!h->pub_pending) return DirectPub{nullptr, 0, 0, 0, 0u, 0};
  h->pub_pending = false;
  return h->pub_next;
}

int launch_rollout(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR,
                   hipStream_t st) {
  RolloutArgs a;
  a.half_r = h->p.wheel_radius / 2.0;
  a.r_over_b = h->p.wheel_radius / h->p.wheel_base;
  a.r_d = h->p.wheel_radius * (1 / h->p.wheel_base);
  a.h = h->p.dt;
  a.h6 = h->p.dt / 6.0;
  for (int c = 0; c < 3; ++c) { a.x0[c] = x0[c]; a.xd[c] = h->xd[c]; a.Q[c] = h->p.Q[c]; a.P1[c] = h->p.P1[c]; }
  a.R[0] = h->p.R[0]; a.R[1] = h->p.R[1];
  a.T = h->T; a.K = h->K;
  const dim3 grid((h->K + kWave - 1) / kWave), block(kWave);
  const USrc usrc{h->d_u[h->ucur], h->pending_shift ? 1 : 0, h->uinit[0], h->uinit[1]};
  if (h->scan_tc > 0 && h->dyn == 0) {  // (the arc dynamics live in the fused and the sequential kernels)
    h->prefix_rows = 0;
    const int TCv = h->scan_tc, C = (h->T + TCv - 1) / TCv;
    const size_t lds = ((size_t)2 * h->T + (size_t)4 * C * kWave) * sizeof(double);
    const dim3 blk(kWave, C);
#define TBNAV_SCAN(TR, TCC, MW) do { h->lk_rollout[0] = 2; h->lk_rollout[1] = TR; h->lk_rollout[2] = TCC; h->lk_rollout[3] = MW; \
                                 hipLaunchKernelGGL((mppi_rollout_scan<TR, TCC, MW>), grid, blk, lds, st, a, d_duL, d_duR, usrc, h->d_J); } while (0)
#define TBNAV_SCAN_TC(TR)                                                                                          \
  switch (TCv) {                                                                                                   \
    case 4: if (C > 12) TBNAV_SCAN(TR, 4, 16); else TBNAV_SCAN(TR, 4, 12); break;                                  \
    case 5: TBNAV_SCAN(TR, 5, 12); break;   case 6: TBNAV_SCAN(TR, 6, 12); break;                                  \
    case 7: TBNAV_SCAN(TR, 7, 16); break;                                                                          \
    case 8: if (C > 12) TBNAV_SCAN(TR, 8, 16); else TBNAV_SCAN(TR, 8, 12); break;                                  \
    case 10: TBNAV_SCAN(TR, 10, 12); break; case 12: TBNAV_SCAN(TR, 12, 12); break;                                \
    case 16: TBNAV_SCAN(TR, 16, 12); break; default: TBNAV_SCAN(TR, 20, 12); break;                                \
  }
    // (TRIG 2 — a fresh sincos every step — is an A-B setting of the sequential / fused kernels; here it takes the three-evaluation form:
    //  its eleven instantiations spilled up to 1.2 KB per lane and nothing selected them)
    if (h->trig == 1) { TBNAV_SCAN_TC(1) } else { TBNAV_SCAN_TC(3) }
#undef TBNAV_SCAN_TC
#undef TBNAV_SCAN
  } else {
    if (h->prefix_rg > 0 && h->dyn == 0 && h->trig == 1) {
      const size_t ldsp = (size_t)2 * h->T * sizeof(double);
      a.lds_from = 0;
      h->lk_rollout[0] = 3; h->lk_rollout[1] = h->prefix_rg < 3 ? h->prefix_rg : 3;
      switch (h->prefix_rg) {
        case 1: hipLaunchKernelGGL((mppi_rollout_prefix<1>), grid, block, ldsp, st, a, d_duL, d_duR, usrc, h->d_J, h->d_total); break;
        case 2: hipLaunchKernelGGL((mppi_rollout_prefix<2>), grid, block, ldsp, st, a, d_duL, d_duR, usrc, h->d_J, h->d_total); break;
        default: hipLaunchKernelGGL((mppi_rollout_prefix<3>), grid, block, ldsp, st, a, d_duL, d_duR, usrc, h->d_J, h->d_total); break;
      }
      TBNAV_HIP(hipGetLastError());
      h->j_valid = true;
      h->prefix_rows = h->T - kGroup * h->prefix_rg;
      return TBNAV_OK;
    }
    h->prefix_rows = 0;
    const size_t lds = (size_t)2 * h->T * sizeof(double) + (size_t)(h->T - h->lds_from) * kWave * sizeof(double);
    a.lds_from = h->lds_from;
    h->lk_rollout[0] = 4; h->lk_rollout[1] = h->dyn == 1 ? 4 : (h->trig >= 1 && h->trig <= 2 ? h->trig : 3);
    if (h->dyn == 1) hipLaunchKernelGGL((mppi_rollout_cost<4>), grid, block, lds, st, a, d_duL, d_duR, usrc, h->d_J);
    else if (h->trig == 1) hipLaunchKernelGGL((mppi_rollout_cost<1>), grid, block, lds, st, a, d_duL, d_duR, usrc, h->d_J);
    else if (h->trig == 2) hipLaunchKernelGGL((mppi_rollout_cost<2>), grid, block, lds, st, a, d_duL, d_duR, usrc, h->d_J);
    else hipLaunchKernelGGL((mppi_rollout_cost<3>), grid, block, lds, st, a, d_duL, d_duR, usrc, h->d_J);
  }
  TBNAV_HIP(hipGetLastError());
  h->j_valid = true;
  return TBNAV_OK;
}


RolloutArgs rollout_args(const tbnav_mppi* h, const double x0[3]) {
  RolloutArgs a;
  a.half_r = h->p.wheel_radius / 2.0;
  a.r_over_b = h->p.wheel_radius / h->p.wheel_base;
  a.r_d = h->p.wheel_radius * (1 / h->p.wheel_base);
  a.h = h->p.dt;
  a.h6 = h->p.dt / 6.0;
  for (int c = 0; c < 3; ++c) { a.x0[c] = x0[c]; a.xd[c] = h->xd[c]; a.Q[c] = h->p.Q[c]; a.P1[c] = h->p.P1[c]; }
  a.R[0] = h->p.R[0]; a.R[1] = h->p.R[1];
  a.T = h->T; a.K = h->K;
  a.lds_from = 0;
  return a;
}

size_t fused_lds_bytes(int T, int R) { return (size_t)3 * T * (R + 1) * sizeof(double); }

// rollout + partial records in one launch (small K); the caller follows with launch_combine(..., fused_S)
int launch_fused(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, hipStream_t st,
                 const RngArgs* rng = nullptr) {
  const RolloutArgs a = rollout_args(h, x0);
  const USrc usrc{h->d_u[h->ucur], h->pending_shift ? 1 : 0, h->uinit[0], h->uinit[1]};
  const int R = h->fused_r, TL = (h->T + kWave - 1) / kWave;
  const dim3 grid(h->fused_S), block(kWave * R);
  const size_t lds = fused_lds_bytes(h->T, R);
  const RngArgs g = rng ? *rng : RngArgs{0, 0, 0.0, 0.0};
#define TBNAV_FUSED(TR, RR, TLL, RG) do { h->lk_rollout[0] = 1; h->lk_rollout[1] = TR; h->lk_rollout[2] = RR; h->lk_rollout[3] = TLL; h->lk_rollout[4] = RG;     \
                                     hipLaunchKernelGGL((mppi_rollout_fused<TR, RR, TLL, RG>), grid, block, lds, st, a, d_duL, d_duR, usrc, \
                                                        lam_of(h), h->keep_j ? h->d_J : nullptr, h->d_records_f, h->fused_S, g); } while (0)
#define TBNAV_FUSED_R(TR)                                                                                \
  if (R == 8) { if (TL == 1) TBNAV_FUSED(TR, 8, 1, 0); else TBNAV_FUSED(TR, 8, 2, 0); }          \
  else if (R == 4) { if (TL == 1) TBNAV_FUSED(TR, 4, 1, 0); else TBNAV_FUSED(TR, 4, 2, 0); }     \
  else { if (TL == 1) TBNAV_FUSED(TR, 16, 1, 0); else TBNAV_FUSED(TR, 16, 2, 0); }
#define TBNAV_FUSED_RNG(TR, RG)                                                                              \
  if (R == 16) { if (TL == 1) TBNAV_FUSED(TR, 16, 1, RG); else TBNAV_FUSED(TR, 16, 2, RG); }                   \
  else { if (TL == 1) TBNAV_FUSED(TR, 8, 1, RG); else TBNAV_FUSED(TR, 8, 2, RG); }
  if (rng) {  // in-kernel noise: instantiated for 8 and 16 rollouts per workgroup (the handle's own choices) — the caller checks
    // (RNG 2 = the fp64 sampler, the handle's default; RNG 1 = the fp32 one, TBNAV_MPPI_OPT_SAMPLER = 0)
    if (h->sampler == 1) {
      if (h->dyn == 1) { TBNAV_FUSED_RNG(4, 2); } else if (h->trig == 3) { TBNAV_FUSED_RNG(3, 2); } else { TBNAV_FUSED_RNG(2, 2); }
    } else if (h->dyn == 1) { TBNAV_FUSED_RNG(4, 1); } else if (h->trig == 3) { TBNAV_FUSED_RNG(3, 1); } else { TBNAV_FUSED_RNG(2, 1); }
  } else if (h->dyn == 1) { TBNAV_FUSED_R(4) } else if (h->trig == 3) { TBNAV_FUSED_R(3) } else { TBNAV_FUSED_R(2) }
#undef TBNAV_FUSED_RNG
#undef TBNAV_FUSED_R
#undef TBNAV_FUSED
  TBNAV_HIP(hipGetLastError());
  h->j_valid = h->keep_j;
  h->prefix_rows = 0;
  return TBNAV_OK;
}

int launch_partials(tbnav_mppi* h, const double* d_duL, const double* d_duR, double* d_records, hipStream_t st) {
  const dim3 grid(h->S, h->T), block(kSliceThreads);
  hipLaunchKernelGGL(mppi_partials, grid, block, 0, st, h->T, h->K, h->S, lam_of(h), h->d_J, d_duL, d_duR, d_records, h->prefix_rows, h->d_total);
  TBNAV_HIP(hipGetLastError());
  return TBNAV_OK;
}

}  // namespace
int tbnav_mh::launch_combine(tbnav_mppi* h, const double* d_records, int G, hipStream_t st, int S, const DirectSrc* direct) {
  if (S < 0) S = h->S;
  int tpr = 1;
  while (tpr < G * S && tpr < kWave) tpr <<= 1;
  const int wpb = TBNAV_COMBINE_WAVES;  // waves per workgroup
  const int steps_per_block = wpb * (kWave / tpr);
  const int blocks = (h->T + steps_per_block - 1) / steps_per_block;
  const USrc usrc{h->d_u[h->ucur], h->pending_shift ? 1 : 0, h->uinit[0], h->uinit[1]};
  // (records that came through an all-gather: only the error words — a poisoned record raises them, see mppi_combine)
  const DirectSrc ds = direct ? *direct : DirectSrc{nullptr, 0ull, (G > 1 && h->comm) ? h->d_dx_err : nullptr, (G > 1 && h->comm) ? h->d_dx_dead : nullptr, 0u};
#define TBNAV_COMBINE(KEEP, MODE) do { h->lk_combine[0] = KEEP; h->lk_combine[1] = MODE; hipLaunchKernelGGL((mppi_combine<KEEP, MODE>), dim3(blocks), dim3(wpb * kWave), 0, st, h->T, G, S, lam_of(h), h->p.max_wheel_vel, usrc, \
                                                    d_records, h->d_u[1 - h->ucur], h->d_out, h->publish_next ? h->d_out_host : nullptr, (double)(h->seq + 1), ds); } while (0)
#define TBNAV_COMBINE_KEEP(MODE) do { if (G * S > 4 * kWave && G * S <= 8 * kWave) TBNAV_COMBINE(8, MODE); else if (G * S > 2 * kWave && G * S <= 4 * kWave) TBNAV_COMBINE(4, MODE); \
                                      else TBNAV_COMBINE(2, MODE); } while (0)
  // (which form: the direct exchange polls; records that came through an all-gather may carry a failed rank's poison; one group has neither —
  //  and with more than 256 records per step, the fused kernel's at K = 4097 ... 8192, it gets four waves per time step)
  if (direct) TBNAV_COMBINE_KEEP(2);
  else if (G > 1 && ds.err) TBNAV_COMBINE_KEEP(1);
  else if (G == 1 && S > 4 * kWave && S <= 16 * kWave && h->wide_combine) {
    h->lk_combine[0] = -4; h->lk_combine[1] = 0;
    hipLaunchKernelGGL(mppi_combine_wide, dim3(h->T), dim3(4 * kWave), 0, st, h->T, S, lam_of(h), h->p.max_wheel_vel, usrc, d_records, h->d_u[1 - h->ucur], h->d_out,
                       h->publish_next ? h->d_out_host : nullptr, (double)(h->seq + 1));
  }
  else TBNAV_COMBINE_KEEP(0);
#undef TBNAV_COMBINE_KEEP
#undef TBNAV_COMBINE
  TBNAV_HIP(hipGetLastError());
  ++h->seq;
  if (h->publish_next) h->published = h->seq;
  h->ucur = 1 - h->ucur;       // the freshly written vector is current ...
  h->pending_shift = true;     // ... and its shift is still owed
  return TBNAV_OK;
}
namespace {

// Apply an owed shift for real (host side; only the state accessors need the materialised vector).
int materialize_controls(tbnav_mppi* h, double* u_host /*[2][T], may be null*/) {
  const int T = h->T;
  std::vector<double> tmp((size_t)2 * T);
  TBNAV_HIP(hipDeviceSynchronize());
  TBNAV_HIP(hipMemcpy(tmp.data(), h->d_u[h->ucur], sizeof(double) * 2 * T, hipMemcpyDeviceToHost));
  if (h->pending_shift) {
    for (int i = 0; i + 1 < T; ++i) { tmp[i] = tmp[i + 1]; tmp[T + i] = tmp[T + i + 1]; }
    tmp[T - 1] = h->uinit[0];
    tmp[2 * T - 1] = h->uinit[1];
    TBNAV_HIP(hipMemcpy(h->d_u[h->ucur], tmp.data(), sizeof(double) * 2 * T, hipMemcpyHostToDevice));
    h->pending_shift = false;
  }
  if (u_host) std::memcpy(u_host, tmp.data(), sizeof(double) * 2 * T);
  return TBNAV_OK;
}

// counter of rollout 0, step 0 of this handle at tick `tick`: counter(k, i) = tick*T*K_global + (k0 + k)*T + i, so the
// shards of one ensemble draw disjoint perturbations from one seed (and the same ones as the unsharded ensemble)
uint64_t rng_base(const tbnav_mppi* h, uint64_t tick) { return tick * (uint64_t)h->T * h->k_global + h->k0 * (uint64_t)h->T; }

bool pick_noise(tbnav_mppi* h, const double*& d_duL, const double*& d_duR) {
  if (!d_duL && !d_duR) { d_duL = h->d_duL; d_duR = h->d_duR; return true; }
  return d_duL && d_duR;
}

// the perturbations can be drawn inside the fused kernel: its in-kernel form exists for 8 and 16 rollouts per workgroup (either sampler)
bool rng_in_kernel(const tbnav_mppi* h) {
  return h->fused_rng && (h->fused_r == 8 || h->fused_r == 16);
}

}  // namespace

extern "C" {

int tbnav_mppi_create(const tbnav_mppi_params* params, tbnav_mppi** out) {
  if (!params || !out) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  if (params->rollouts <= 0 || !(params->dt > 0.0) || !(params->horizon > 0.0) ||
      !(params->lambda > 0.0) || !(params->wheel_base != 0.0))
    return TBNAV_ERR_INVALID_ARG;
  const int T = static_cast<int>(params->horizon / params->dt);  // mppi.cpp:47
  if (T <= 0) return TBNAV_ERR_INVALID_ARG;
  int ndev = 0;
  {
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      return tbnav::hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount", __FILE__, __LINE__);
  }
  int dev = params->device;
  if (dev < 0) TBNAV_HIP(hipGetDevice(&dev));
  if (dev >= ndev) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(dev);
  if (!guard.ok) return TBNAV_ERR_NO_DEVICE;

  tbnav_mppi* h = new (std::nothrow) tbnav_mppi();
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->p = *params;
  h->T = T;
  h->K = params->rollouts;
  h->S = (h->K + kSlice - 1) / kSlice;
  h->device = dev;
  {
    // How many steps' losses fit in LDS while the whole grid stays resident in ONE round:
    // blocks per CU needed = ceil(blocks / CUs) (at most 8 considered), LDS budget per block = 160 KB / that.
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const long blocks = (params->rollouts + kWave - 1) / kWave;
    long per_cu = (blocks + cus - 1) / cus;
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    const long budget = (long)kMaxLdsBytes / per_cu - 512 - (long)(2 * T * sizeof(double));
    long steps_in_lds = budget > 0 ? budget / (long)(kWave * sizeof(double)) : 0;
    if (steps_in_lds > T) steps_in_lds = T;
    h->lds_from = T - (int)steps_in_lds;
    h->lds_from = ((h->lds_from + 3) / 4) * 4;  // whole groups of 4 steps switch staging together
    if (h->lds_from > T) h->lds_from = T;
    // round 3: the prefix-form kernel takes the streaming shape whenever a late region of 1, 2 or 3 groups leaves whole rounds
    // of three groups (one of the three always does) and at least one round — whatever T: it stages nothing in LDS
    if (T % 4 == 0 && blocks >= 2 * cus)
      for (int rg : {1, 2, 3}) if ((T / 4 - rg) % kRoundGroups == 0 && T / 4 - rg >= kRoundGroups) { h->prefix_rg = rg; break; }
  }
  // time-parallel kernel: up to 16 chunks (waves) per workgroup
  h->scan_tc = 0;
  for (int tc : {4, 5, 6, 8, 10, 12, 16, 20})
    if ((T + tc - 1) / tc <= 12) { h->scan_tc = tc; break; }
  {
    // The time-parallel kernel exists to create waves when the rollout count alone cannot fill the chip;
    // once K/64 one-wave workgroups cover >= 2 waves per CU-SIMD pair the sequential kernel (fewer
    // registers, no chunk barriers) is faster (measured on MI355X: K=65536,T=100 87 us vs 121 us;
    // K=1024,T=50 32 us vs 11 us).
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    // Measured on MI355X (tools/mppi_size_sweep.py, T = 25 / 50 / 100, tick in us, resident noise):
    //   K        fused<8>      time-parallel     sequential
    //   2048     11.3          15.2              29.4            (T = 50)
    //   4096     20.1          20.5              49.2            (T = 100)
    //   8192     32.9          23.6              52.1
    //   16384    54.6          29.3              56.1
    //   32768    100           47.1              55.1
    //   49152    -             66.2              62.1
    //   65536    165 (T = 50)  86.6              76.1
    // one wave per rollout (fused) costs 5x the instructions of one lane per rollout and only pays while the chip is
    // otherwise empty; the time-parallel kernel carries the middle; the sequential one takes over once K/64 one-wave
    // workgroups are ~2.5 per CU.  (Round 1 switched fused -> sequential at 2 per CU and never used the middle kernel
    // below T = 129: K = 16384 ran at half speed.)
    const long waves = (params->rollouts + kWave - 1) / kWave;
    if (2 * waves > 5 * cus) h->scan_tc = 0;
    // fused rollout + partials (lanes = time): T must fit two steps per lane.  With the perturbations drawn inside it the
    // fused kernel saves the sample kernel's launch and 16 B per rollout-step, so device-noise ticks stay with it longer
    // (K = 8192, T = 100: 33.1 us against 36.1 for sample + time-parallel; K = 12288, T = 50: 38.9 against 29.7).
    const bool fused_ok = T <= 2 * kWave && h->scan_tc > 0;
    // (re-measured after the combine learnt to keep a lane's records in registers and to spread over one-wave workgroups —
    //  the fused kernel's many records were what made it lose earlier: K = 8192, T = 100: fused with 16 rollouts per workgroup
    //  20.4 us, time-parallel 22.6; with device noise 21.3 against 35.5)
    h->fused_dev = fused_ok && 2 * waves <= cus;        // K <= 8192 at 256 CUs
    h->fused_rng = fused_ok && 2 * waves <= cus;
    // rollouts per workgroup: 8 spreads K = 1024 over 128 CUs (9.0 us against 10.0 with 16: latency-bound); from ~2048 up the
    // chip is covered anyway and 16 halve the records the combine reads (K = 2048: 11.6 -> 10.7 us, 3072: 14.7 -> 12.1,
    // 4096, T = 100: 20.1 -> 15.6)
    h->fused_r = (h->fused_dev || h->fused_rng) ? (8 * waves <= cus ? 8 : 16) : 0;   // 8 up to K = 2048 (K = 2048, T = 50: 9.2 us against 10.0; 3072: 11.6 against 10.5)
  }
  h->fused_S = h->fused_r ? (h->K + h->fused_r - 1) / h->fused_r : 0;
  h->k_global = (uint64_t)h->K;
  const size_t tk = (size_t)T * h->K;
  hipError_t e = hipSuccess;
  auto alloc = [&](double** p, size_t n) { if (e == hipSuccess) e = hipMalloc((void**)p, n * sizeof(double)); };
  alloc(&h->d_u[0], 2 * (size_t)T);
  alloc(&h->d_u[1], 2 * (size_t)T);
  alloc(&h->d_J, tk);
  alloc(&h->d_duL, tk);
  alloc(&h->d_duR, tk);
  alloc(&h->d_records, (size_t)T * h->S * TBNAV_MPPI_REC);
  if (h->prefix_rg) alloc(&h->d_total, (size_t)h->K);
  if (h->fused_r) alloc(&h->d_records_f, (size_t)T * h->fused_S * TBNAV_MPPI_REC);
  alloc(&h->d_out, 2);
  if (e == hipSuccess) e = hipMemset(h->d_out, 0, 2 * sizeof(double));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_out, 4 * sizeof(double), hipHostMallocMapped);
  if (e == hipSuccess) { for (int q = 0; q < 4; ++q) h->h_out[q] = 0.0; e = hipHostGetDevicePointer((void**)&h->d_out_host, h->h_out, 0); }
  if (e == hipSuccess) e = hipMemset(h->d_u[0], 0, 2 * (size_t)T * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->d_u[1], 0, 2 * (size_t)T * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->d_J, 0, tk * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->d_duL, 0, tk * sizeof(double));
  if (e == hipSuccess) e = hipMemset(h->d_duR, 0, tk * sizeof(double));
  if (e == hipSuccess) {
    const int lds_max = (int)((size_t)2 * T * sizeof(double) + (size_t)(T - h->lds_from) * kWave * sizeof(double));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mppi_rollout_cost<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mppi_rollout_cost<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mppi_rollout_cost<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mppi_rollout_cost<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    const int rc = tbnav::hip_fail(e, "tbnav_mppi_create allocation", __FILE__, __LINE__);
    tbnav_mppi_destroy(h);
    return rc;
  }
  *out = h;
  return TBNAV_OK;
}

void tbnav_mppi_destroy(tbnav_mppi* h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  // A handle that is still attached detaches first — for a multi-process direct exchange that is COLLECTIVE (every rank's destroy or
  // detach meets in tbnav_mppi_attach_comm: a peer one tick ahead may still be storing into this rank's buffer; round-4 advisor
  // finding: the buffer used to be freed here without the rendezvous) — so the communicator must still be alive: destroy handles
  // before their communicators, or detach explicitly first.
  if (h->comm) (void)tbnav_mppi_attach_comm(h, nullptr);
  (void)hipDeviceSynchronize();
  (void)hipFree(h->d_u[0]); (void)hipFree(h->d_u[1]); (void)hipFree(h->d_J); (void)hipFree(h->d_duL); (void)hipFree(h->d_duR);
  (void)hipFree(h->d_total); (void)hipFree(h->d_records_all);
  direct_teardown(h);
  exchange_words_free(h);
  (void)hipFree(h->d_raw); (void)hipFree(h->d_records); (void)hipFree(h->d_records_f); (void)hipFree(h->d_out);
  if (h->h_out) (void)hipHostFree(h->h_out);
  for (auto* g : {&h->tg, &h->tgs}) { if (g->exec) (void)hipGraphExecDestroy(g->exec); if (g->graph) (void)hipGraphDestroy(g->graph); }
  (void)hipFree(h->d_tick0);
  delete h;
}

int tbnav_mppi_steps(const tbnav_mppi* h) { return h ? h->T : -1; }
int tbnav_mppi_set_dynamics(tbnav_mppi* h, int32_t model) {
  if (!h || (model != TBNAV_MPPI_DYN_RK4 && model != TBNAV_MPPI_DYN_ARC)) return TBNAV_ERR_INVALID_ARG;
  h->dyn = model;
  ++h->cfg_epoch;
  return TBNAV_OK;
}

int tbnav_mppi_set_option(tbnav_mppi* h, int32_t option, int32_t value) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const int T = h->T;
  ++h->cfg_epoch;  // (any option may change what a captured graph of ticks has baked in)
  switch (option) {
    case TBNAV_MPPI_OPT_KEEP_J: h->keep_j = value != 0; return TBNAV_OK;
    case TBNAV_MPPI_OPT_TRIG:
      if (value < 1 || value > 3) return TBNAV_ERR_INVALID_ARG;
      h->trig = value;
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_NO_LDS_STAGING:
      if (value) { h->lds_from = T; h->prefix_rg = 0; }
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_REG_TAIL:   // (the round-2 kernel this switched is gone — mppi_rollout_prefix superseded it; accepted and ignored)
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_PREFIX_FORM:
      if (!value) h->prefix_rg = 0;
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_BATCH_GRAPH:
      h->graph_on = value != 0;
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_DIRECT_EXCHANGE:  // (takes effect at the next tbnav_mppi_attach_comm)
      h->direct_want = value != 0;
      h->dx_withhold = value == 2;                       // (tests of the bound: see dx_withhold)
      h->dx_budget = value == 2 ? 30000000ull : 200000000ull;  // 0.3 s there
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_FAULT_INJECT:   // (tests) the next sharded tick's local half reports a failure; 0 takes it back
      h->fail_next = value != 0;
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_WIDE_COMBINE:   // 0: always the one-wave-per-step combine (A-B); default 1
      h->wide_combine = value != 0;
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_SAMPLER:        // 1 (default): fp64 Box-Muller on 52-bit uniforms (utilities.cpp:20-24's width); 0: fp32 on 24-bit uniforms
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->sampler = value;
      return TBNAV_OK;
    case TBNAV_MPPI_OPT_KERNEL: {
      // 0: mppi_rollout_cost (sequential); n > 0: mppi_rollout_scan with n steps per thread; -4 / -8 / -16: fused, that many rollouts per workgroup
      int fused = 0, tc = 0;
      if (value < 0) {
        fused = -value;
        if (!(T <= 2 * kWave && (fused == 4 || fused == 8 || fused == 16))) return TBNAV_ERR_INVALID_ARG;
      } else if (value > 0) {
        const int cmax = (value == 4 || value == 7 || value == 8) ? 16 : 12;
        bool known = false;
        for (int t : {4, 5, 6, 7, 8, 10, 12, 16, 20}) known |= t == value;
        if (!known || (T + value - 1) / value > cmax) return TBNAV_ERR_INVALID_ARG;
        tc = value;
      }
      h->scan_tc = tc;
      h->fused_r = fused;
      h->fused_dev = fused > 0;          // a forced choice holds for both kinds of tick (in-kernel noise exists for 8 and 16:
      h->fused_rng = fused == 8 || fused == 16;  //  a 4-rollout workgroup samples first)
      h->fused_S = fused ? (h->K + fused - 1) / fused : 0;
      (void)hipFree(h->d_records_f);
      h->d_records_f = nullptr;
      if (fused) TBNAV_HIP(hipMalloc((void**)&h->d_records_f, sizeof(double) * (size_t)T * h->fused_S * TBNAV_MPPI_REC));
      return TBNAV_OK;
    }
    default: return TBNAV_ERR_INVALID_ARG;
  }
}

int tbnav_mppi_set_rng_shard(tbnav_mppi* h, uint64_t first_rollout, uint64_t rollouts_global) {
  if (!h || rollouts_global < first_rollout + (uint64_t)h->K) return TBNAV_ERR_INVALID_ARG;
  h->k0 = first_rollout;
  h->k_global = rollouts_global;
  ++h->cfg_epoch;
  return TBNAV_OK;
}

int tbnav_mppi_rollout_variant(const tbnav_mppi* h) { return h ? (h->fused_dev ? -h->fused_r : h->scan_tc) : 0; }
int tbnav_mppi_streaming_form(const tbnav_mppi* h) {
  if (!h) return -1;
  return (h->dyn == 0 && h->trig == 1 && h->prefix_rg > 0) ? 2 : 0;
}
int64_t tbnav_mppi_graph_replayed_ticks(const tbnav_mppi* h) { return h ? (int64_t)h->graph_ticks : -1; }
int tbnav_mppi_rollouts(const tbnav_mppi* h) { return h ? h->K : -1; }
int tbnav_mppi_records_per_step(const tbnav_mppi* h) { return h ? h->S : -1; }

int tbnav_mppi_set_initial_controls(tbnav_mppi* h, double uL, double uR) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  h->uinit[0] = uL; h->uinit[1] = uR;
  ++h->cfg_epoch;
  double* tmp = new (std::nothrow) double[2 * (size_t)h->T];
  if (!tmp) return TBNAV_ERR_INVALID_ARG;
  for (int i = 0; i < h->T; ++i) { tmp[i] = uL; tmp[h->T + i] = uR; }
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(h->d_u[h->ucur], tmp, 2 * (size_t)h->T * sizeof(double), hipMemcpyHostToDevice);
  h->pending_shift = false;
  delete[] tmp;
  TBNAV_HIP(e);
  return TBNAV_OK;
}

int tbnav_mppi_set_waypoint(tbnav_mppi* h, double x, double y, double theta) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->xd[0] = x; h->xd[1] = y; h->xd[2] = theta;
  ++h->cfg_epoch;
  return TBNAV_OK;
}

int tbnav_mppi_get_controls(tbnav_mppi* h, double* u_host) {
  if (!h || !u_host) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  return materialize_controls(h, u_host);
}

int tbnav_mppi_set_controls(tbnav_mppi* h, const double* u_host) {
  if (!h || !u_host) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipDeviceSynchronize());
  TBNAV_HIP(hipMemcpy(h->d_u[h->ucur], u_host, 2 * (size_t)h->T * sizeof(double), hipMemcpyHostToDevice));
  h->pending_shift = false;
  return TBNAV_OK;
}

int tbnav_mppi_shard_partials(tbnav_mppi* h, const double x0[3], const double* d_duL,
                              const double* d_duR, void* stream, double* d_records_out) {
  if (!h || !x0 || !d_records_out || !pick_noise(h, d_duL, d_duR)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->fused_dev && kSlice % h->fused_r == 0) {
    // small K: the fused rollout+partials kernel, then its fine records folded into the K-slice records that the
    // ranks exchange (two short launches instead of three)
    const int rcf = launch_fused(h, x0, d_duL, d_duR, st);
    if (rcf != TBNAV_OK) return rcf;
    hipLaunchKernelGGL(mppi_merge_records, dim3(h->S, h->T), dim3(kWave), 0, st, h->T, h->fused_S, kSlice / h->fused_r, h->S,
                       lam_of(h), h->d_records_f, d_records_out, take_pub(h));
    TBNAV_HIP(hipGetLastError());
    return TBNAV_OK;
  }
  int rc = launch_rollout(h, x0, d_duL, d_duR, st);
  if (rc != TBNAV_OK) return rc;
  return launch_partials(h, d_duL, d_duR, d_records_out, st);
}

// Same, with this shard's perturbations drawn on the device (tbnav_mppi_set_rng_shard gives the shard its place in
// the ensemble's counter space): inside the fused kernel when that is the handle's kernel, else sampled first.
int tbnav_mppi_shard_partials_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream, double* d_records_out) {
  if (!h || !x0 || !d_records_out) return TBNAV_ERR_INVALID_ARG;
  if (!(rng_in_kernel(h) && kSlice % h->fused_r == 0)) {
    const int rc = tbnav_mppi_sample_noise(h, seed, tick, stream);
    return rc != TBNAV_OK ? rc : tbnav_mppi_shard_partials(h, x0, nullptr, nullptr, stream, d_records_out);
  }
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const RngArgs g{seed, rng_base(h, tick), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var)};
  const int rcf = launch_fused(h, x0, h->d_duL, h->d_duR, st, &g);
  if (rcf != TBNAV_OK) return rcf;
  hipLaunchKernelGGL(mppi_merge_records, dim3(h->S, h->T), dim3(kWave), 0, st, h->T, h->fused_S, kSlice / h->fused_r, h->S,
                     lam_of(h), h->d_records_f, d_records_out, take_pub(h));
  TBNAV_HIP(hipGetLastError());
  return TBNAV_OK;
}

int tbnav_mppi_shard_combine(tbnav_mppi* h, const double* d_records_all, int32_t n_shards, void* stream) {
  if (!h || !d_records_all || n_shards <= 0) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  return launch_combine(h, d_records_all, n_shards, static_cast<hipStream_t>(stream));
}

int tbnav_mppi_enqueue_dev(tbnav_mppi* h, const double x0[3], const double* d_duL,
                           const double* d_duR, void* stream) {
  if (!h || !x0 || !pick_noise(h, d_duL, d_duR)) return TBNAV_ERR_INVALID_ARG;
  if (h->comm) return sharded_tick(h, x0, d_duL, d_duR, nullptr, 0, stream);
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->fused_dev) {
    const int rcf = launch_fused(h, x0, d_duL, d_duR, st);
    if (rcf != TBNAV_OK) return rcf;
    return launch_combine(h, h->d_records_f, 1, st, h->fused_S);
  }
  int rc = launch_rollout(h, x0, d_duL, d_duR, st);
  if (rc != TBNAV_OK) return rc;
  // (fusing the combine into the partials kernel's last workgroup was measured and rejected: DESIGN.md section 4)
  rc = launch_partials(h, d_duL, d_duR, h->d_records, st);
  if (rc != TBNAV_OK) return rc;
  return launch_combine(h, h->d_records, 1, st);
}

int tbnav_mppi_profile_tick(tbnav_mppi* h, const double x0[3], const double* d_duL,
                            const double* d_duR, void* stream, float ms[TBNAV_MPPI_NKERNELS]) {
  if (!h || !x0 || !ms || !pick_noise(h, d_duL, d_duR)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev[TBNAV_MPPI_NKERNELS + 1];
  for (auto& e : ev) TBNAV_HIP(hipEventCreate(&e));
  int rc = TBNAV_OK;
  TBNAV_HIP(hipEventRecord(ev[0], st));
  if (h->fused_dev) {  // rollout and partials are one kernel: its time is reported under [0], [1] is the empty interval
    rc = launch_fused(h, x0, d_duL, d_duR, st);
    if (rc == TBNAV_OK) { TBNAV_HIP(hipEventRecord(ev[1], st)); TBNAV_HIP(hipEventRecord(ev[2], st)); rc = launch_combine(h, h->d_records_f, 1, st, h->fused_S); }
  } else {
    rc = launch_rollout(h, x0, d_duL, d_duR, st);
    if (rc == TBNAV_OK) { TBNAV_HIP(hipEventRecord(ev[1], st)); rc = launch_partials(h, d_duL, d_duR, h->d_records, st); }
    if (rc == TBNAV_OK) { TBNAV_HIP(hipEventRecord(ev[2], st)); rc = launch_combine(h, h->d_records, 1, st); }
  }
  if (rc == TBNAV_OK) {
    TBNAV_HIP(hipEventRecord(ev[3], st));
    TBNAV_HIP(hipEventSynchronize(ev[3]));
    for (int i = 0; i < TBNAV_MPPI_NKERNELS; ++i) TBNAV_HIP(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}


// Per-kernel durations without the event overhead: every kernel of the tick is launched `reps` times back to back
// between ONE pair of events (a single launch of a 5 us kernel between two events measures the events as much as
// the kernel).  The controller state advances as if `reps` ticks had run on the same inputs.
int tbnav_mppi_profile_kernels(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, void* stream,
                               int32_t reps, float ms[TBNAV_MPPI_NKERNELS]) {
  if (!h || !x0 || !ms || reps < 2 || (reps & 1) || !pick_noise(h, d_duL, d_duR)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev[2];
  for (auto& e : ev) TBNAV_HIP(hipEventCreate(&e));
  int rc = TBNAV_OK;
  auto timed = [&](int which, auto&& launch) {
    if (rc != TBNAV_OK) return;
    if (hipEventRecord(ev[0], st) != hipSuccess) { rc = TBNAV_ERR_HIP; return; }
    for (int r = 0; r < reps && rc == TBNAV_OK; ++r) rc = launch();
    if (rc != TBNAV_OK) return;
    float t = 0.f;
    if (hipEventRecord(ev[1], st) != hipSuccess || hipEventSynchronize(ev[1]) != hipSuccess ||
        hipEventElapsedTime(&t, ev[0], ev[1]) != hipSuccess) { rc = TBNAV_ERR_HIP; return; }
    ms[which] = t / (float)reps;
  };
  for (int i = 0; i < TBNAV_MPPI_NKERNELS; ++i) ms[i] = 0.f;
  if (h->fused_dev) {
    timed(0, [&] { return launch_fused(h, x0, d_duL, d_duR, st); });
    timed(2, [&] { return launch_combine(h, h->d_records_f, 1, st, h->fused_S); });
  } else {
    timed(0, [&] { return launch_rollout(h, x0, d_duL, d_duR, st); });
    timed(1, [&] { return launch_partials(h, d_duL, d_duR, h->d_records, st); });
    timed(2, [&] { return launch_combine(h, h->d_records, 1, st); });
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int tbnav_mppi_profile_kernels_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream, int32_t reps,
                                   float ms[TBNAV_MPPI_NKERNELS]) {
  if (!h || !x0 || !ms || reps < 2 || (reps & 1)) return TBNAV_ERR_INVALID_ARG;
  if (!(h->fused_dev && rng_in_kernel(h))) {
    // no in-kernel noise for this configuration: the production tick samples into the handle's buffers and runs the plain kernels
    const int rc = tbnav_mppi_sample_noise(h, seed, tick, stream);
    return rc != TBNAV_OK ? rc : tbnav_mppi_profile_kernels(h, x0, nullptr, nullptr, stream, reps, ms);
  }
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev[2];
  for (auto& e : ev) TBNAV_HIP(hipEventCreate(&e));
  int rc = TBNAV_OK;
  auto timed = [&](int which, auto&& launch) {
    if (rc != TBNAV_OK) return;
    if (hipEventRecord(ev[0], st) != hipSuccess) { rc = TBNAV_ERR_HIP; return; }
    for (int r = 0; r < reps && rc == TBNAV_OK; ++r) rc = launch(r);
    if (rc != TBNAV_OK) return;
    float t = 0.f;
    if (hipEventRecord(ev[1], st) != hipSuccess || hipEventSynchronize(ev[1]) != hipSuccess ||
        hipEventElapsedTime(&t, ev[0], ev[1]) != hipSuccess) { rc = TBNAV_ERR_HIP; return; }
    ms[which] = t / (float)reps;
  };
  for (int i = 0; i < TBNAV_MPPI_NKERNELS; ++i) ms[i] = 0.f;
  // the launches of tbnav_mppi_enqueue_rng, kernel by kernel: the RNG = true instantiation of the fused kernel
  timed(0, [&](int r) {
    const RngArgs g{seed, rng_base(h, tick + (uint64_t)r), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var)};
    return launch_fused(h, x0, h->d_duL, h->d_duR, st, &g);
  });
  timed(2, [&](int) { return launch_combine(h, h->d_records_f, 1, st, h->fused_S); });
  for (auto& e : ev) (void)hipEventDestroy(e);
  return rc;
}

int tbnav_mppi_last_kernel_names(const tbnav_mppi* h, char* rollout, int32_t rollout_cap, char* combine, int32_t combine_cap) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  const int* k = h->lk_rollout;
  if (rollout && rollout_cap > 0) {
    switch (k[0]) {
      case 1: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_fused<%d, %d, %d, %d>", k[1], k[2], k[3], k[4]); break;   // (as the profiler spells the instantiation)
      case 2: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_scan<%d, %d, %d>", k[1], k[2], k[3]); break;
      case 3: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_prefix<%d>", k[1]); break;
      case 4: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_cost<%d>", k[1]); break;
      default: rollout[0] = 0;
    }
  }
  if (combine && combine_cap > 0) {
    if (h->lk_combine[0] == -4) snprintf(combine, (size_t)combine_cap, "mppi_combine_wide");
    else if (h->lk_combine[0]) snprintf(combine, (size_t)combine_cap, "mppi_combine<%d, %d>", h->lk_combine[0], h->lk_combine[1]);
    else combine[0] = 0;
  }
  return TBNAV_OK;
}

int tbnav_mppi_last_controls(tbnav_mppi* h, void* stream, double u_out[2]) {
  if (!h || !u_out) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto exchange_ok = [&]() -> int { return exchange_error(h); };  // (the stream has been waited for) an exchange that failed left its mark in host memory
  if (h->published != h->seq) {  // the last tick was enqueue-only: fetch the device copy
    TBNAV_HIP(hipMemcpyAsync(h->h_out, h->d_out, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    TBNAV_HIP(hipStreamSynchronize(st));
    u_out[0] = h->h_out[0];
    u_out[1] = h->h_out[1];
    return exchange_ok();
  }
  // The last combine published (ul, ur, tick number) in mapped host memory.  Poll for its number for a short while (a
  // control loop calls this right behind the enqueue: the answer is microseconds away and a stream synchronisation
  // costs more than the tick), then fall back to waiting on the stream.
  const double want = (double)h->seq;
  volatile double* vo = h->h_out;
  const auto t0 = std::chrono::steady_clock::now();
  bool seen = vo[2] == want;
  while (!seen && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(200)) seen = vo[2] == want;
  if (!seen) TBNAV_HIP(hipStreamSynchronize(st));
  std::atomic_thread_fence(std::memory_order_acquire);
  u_out[0] = vo[0];
  u_out[1] = vo[1];
  return exchange_ok();
}

int tbnav_mppi_new_controls_dev(tbnav_mppi* h, const double x0[3], const double* d_duL,
                                const double* d_duR, void* stream, double u_out[2]) {
  if (!u_out) return TBNAV_ERR_INVALID_ARG;
  if (h) h->publish_next = true;
  int rc = tbnav_mppi_enqueue_dev(h, x0, d_duL, d_duR, stream);
  if (h) h->publish_next = false;
  if (rc != TBNAV_OK) return rc;
  return tbnav_mppi_last_controls(h, stream, u_out);
}

int tbnav_mppi_new_controls(tbnav_mppi* h, const double x0[3], const double* noise_host,
                            double u_out[2]) {
  if (!h || !x0 || !noise_host || !u_out) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const size_t n = (size_t)h->T * h->K;
  if (!h->d_raw) TBNAV_HIP(hipMalloc((void**)&h->d_raw, 2 * n * sizeof(double)));
  TBNAV_HIP(hipMemcpyAsync(h->d_raw, noise_host, 2 * n * sizeof(double), hipMemcpyHostToDevice, nullptr));
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(mppi_unpack_noise, dim3(blocks), dim3(256), 0, nullptr, h->T, h->K, h->d_raw,
                     h->d_duL, h->d_duR);
  TBNAV_HIP(hipGetLastError());
  return tbnav_mppi_new_controls_dev(h, x0, nullptr, nullptr, nullptr, u_out);
}

int tbnav_mppi_sample_noise(tbnav_mppi* h, uint64_t seed, uint64_t tick, void* stream) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const size_t n = (size_t)h->T * h->K;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(mppi_sample_noise, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     h->T, h->K, seed, rng_base(h, tick), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var), h->sampler, h->d_duL,
                     h->d_duR);
  TBNAV_HIP(hipGetLastError());
  return TBNAV_OK;
}

// Production tick: the perturbations of tick `tick` under `seed` are the ones tbnav_mppi_sample_noise would write, but
// for the fused small-K kernel they are generated inside it and never stored (the soft-min reads them from the
// workgroup's LDS tile).  Other configurations sample into the handle's buffers first — same values, same result.
int tbnav_mppi_enqueue_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream) {
  if (!h || !x0) return TBNAV_ERR_INVALID_ARG;
  if (h->comm) return sharded_tick(h, x0, nullptr, nullptr, &seed, tick, stream);
  if (!rng_in_kernel(h)) {
    const int rc = tbnav_mppi_sample_noise(h, seed, tick, stream);
    return rc != TBNAV_OK ? rc : tbnav_mppi_enqueue_dev(h, x0, nullptr, nullptr, stream);
  }
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const RngArgs g{seed, rng_base(h, tick), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var)};
  const int rc = launch_fused(h, x0, h->d_duL, h->d_duR, st, &g);
  return rc != TBNAV_OK ? rc : launch_combine(h, h->d_records_f, 1, st, h->fused_S);
}

namespace {
using TickGraph = tbnav_mppi::TickGraph;
void drop_graph(TickGraph& g) {
  if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  if (g.graph) { (void)hipGraphDestroy(g.graph); g.graph = nullptr; }
}
bool graph_usable(const tbnav_mppi* h, const TickGraph& g, const double* x0, uint64_t seed, hipStream_t st) {
  return g.exec && g.epoch == h->cfg_epoch && g.seed == seed && g.stream == st && std::memcmp(g.x0, x0, sizeof g.x0) == 0;
}
// Capture `len` ticks (len even: the controls' double buffer is back where it was after a replay) and a last node that advances
// the device's tick word by len.  A capture that fails turns the replays off for good (plain launches from there on).
void build_graph(tbnav_mppi* h, TickGraph& g, int len, const double* x0, uint64_t seed, hipStream_t st) {
  // (another stream may not have run the previous replay's tick-advance node yet: what the device word holds is unknown)
  h->tg_dev_tick = ~0ull;
  drop_graph(g);
  if (!h->d_tick0 && hipMalloc((void**)&h->d_tick0, sizeof(uint64_t)) != hipSuccess) { h->d_tick0 = nullptr; h->graph_on = false; }
  if (!h->graph_on) return;
  if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) { h->graph_on = false; (void)hipGetLastError(); return; }
  const int ucur0 = h->ucur; const uint64_t seq0 = h->seq;
  int rc = TBNAV_OK;
  for (int t = 0; t < len && rc == TBNAV_OK; ++t) {
    RngArgs ra{seed, rng_base(h, (uint64_t)t), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var)};
    ra.tick0 = h->d_tick0; ra.per_tick = (uint64_t)h->T * h->k_global;
    rc = launch_fused(h, x0, h->d_duL, h->d_duR, st, &ra);
    if (rc == TBNAV_OK) rc = launch_combine(h, h->d_records_f, 1, st, h->fused_S);
  }
  if (rc == TBNAV_OK) { hipLaunchKernelGGL(mppi_tick_advance, dim3(1), dim3(1), 0, st, h->d_tick0, (uint64_t)len); if (hipGetLastError() != hipSuccess) rc = TBNAV_ERR_HIP; }
  hipGraph_t gr = nullptr;
  const hipError_t e_end = hipStreamEndCapture(st, &gr);
  h->ucur = ucur0; h->seq = seq0;  // nothing ran: the host-side state goes back
  if (rc == TBNAV_OK && e_end == hipSuccess && gr && hipGraphInstantiate(&g.exec, gr, nullptr, nullptr, 0) == hipSuccess) {
    g.graph = gr; g.len = len; g.seed = seed; g.stream = st; g.ucur = h->ucur; g.epoch = h->cfg_epoch; std::memcpy(g.x0, x0, sizeof g.x0);
  } else {
    if (gr) (void)hipGraphDestroy(gr);
    g.exec = nullptr; h->graph_on = false; (void)hipGetLastError();
  }
}
int replay_graph(tbnav_mppi* h, const TickGraph& g, uint64_t t0, hipStream_t st) {
  // (consecutive replays need no copy: the last node of the one before has advanced the device word)
  if (t0 != h->tg_dev_tick) {
    h->tg_dev_tick = ~0ull;
    hipLaunchKernelGGL(mppi_tick_set, dim3(1), dim3(1), 0, st, h->d_tick0, t0);
    TBNAV_HIP(hipGetLastError());
  }
  { const hipError_t eg = hipGraphLaunch(g.exec, st); if (eg != hipSuccess) { h->tg_dev_tick = ~0ull; TBNAV_HIP(eg); } }
  h->tg_dev_tick = t0 + (uint64_t)g.len;
  h->seq += (uint64_t)g.len;  // ucur: unchanged after an even number of ticks; the shift stays owed
  h->graph_ticks += (uint64_t)g.len;
  return TBNAV_OK;
}
}  // namespace

int tbnav_mppi_enqueue_rng_batch(tbnav_mppi* h, const double* x0s, int32_t x0_stride, uint64_t seed, uint64_t first_tick, int32_t n_ticks,
                                 void* stream) {
  if (!h || !x0s || n_ticks < 0 || (x0_stride != 0 && x0_stride < 3)) return TBNAV_ERR_INVALID_ARG;
  int32_t i = 0;
  constexpr int kGraphTicks = 100;  // (even: the controls' double buffer is back where it was after a chunk)
  constexpr int kShortMin = 8;      // the shortest batch worth a graph of its own
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (only where the tick is short enough for the launches themselves to matter: K = 1024: 8.25 -> 8.15 us per tick on a fast host, 8.9 -> 8.3
  //  on a slower one; from K = 2048 up the device is the bound and the replay is 1-3 % slower than plain launches)
  if (h->graph_on && !h->comm && x0_stride == 0 && st != nullptr && rng_in_kernel(h) && h->fused_r == 8 && h->K <= 1536 && n_ticks >= 2) {
    DeviceGuard guard(h->device);
    // (the chunk graph is built by the first batch call that could use one, however short — a warm-up call, typically — so that a
    //  later long call does not pay the ~1 ms of capture + instantiation)
    // the first tick after set_controls / set_initial_controls reads the vector unshifted: keep it out of the graphs
    if (!h->pending_shift) { const int rc = tbnav_mppi_enqueue_rng(h, x0s, seed, first_tick, stream); if (rc != TBNAV_OK) return rc; ++i; }
    const bool usable = graph_usable(h, h->tg, x0s, seed, st);
    // (a graph has the controls' double buffer baked in as it stood at capture: on the other parity ONE plain tick brings it
    //  back — a rebuild would cost ~0.5 ms inside the caller's batch)
    if (usable && h->tg.ucur != h->ucur && n_ticks - i > kGraphTicks) {
      const int rc = tbnav_mppi_enqueue_rng(h, x0s, seed, first_tick + (uint64_t)i, stream);
      if (rc != TBNAV_OK) return rc;
      ++i;
    }
    const bool same = usable && h->tg.ucur == h->ucur;
    if (!same && (!usable || n_ticks - i >= kGraphTicks)) build_graph(h, h->tg, kGraphTicks, x0s, seed, st);
    while (h->tg.exec && n_ticks - i >= kGraphTicks) {
      const int rc = replay_graph(h, h->tg, first_tick + (uint64_t)i, st);
      if (rc != TBNAV_OK) return rc;
      i += kGraphTicks;
    }
    // What is left (or a batch shorter than a chunk): one graph of exactly that many ticks, less one if odd.  Built by the second
    // batch in a row that asks for the same length on the same parity of the double buffer — a caller that times blocks of 20
    // ticks gets it in its second block; one with batches of ever-changing length never pays for a graph it would not reuse.
    // (Measured at K = 1024, T = 50, blocks of 20 between synchronisations: 9.3 us per tick replayed, 9.8-12.1 launched one by
    //  one — the spread is the host's launch rate, which the replay does not depend on.  An earlier note here gave a 10-tick
    //  replay 9.9-10.2 us against 8.9: that was a chunk graph plus plain ticks plus the tick-set launch in one batch.)
    // (an odd batch leaves the double buffer on the other parity: ONE plain tick in front brings the next batch back to the parity
    //  the graph was captured on, if what is left then still is the graph's length)
    if (h->graph_on && graph_usable(h, h->tgs, x0s, seed, st) && h->tgs.ucur != h->ucur && ((n_ticks - i - 1) & ~1) == h->tgs.len) {
      const int rc = tbnav_mppi_enqueue_rng(h, x0s, seed, first_tick + (uint64_t)i, stream);
      if (rc != TBNAV_OK) return rc;
      ++i;
    }
    const int L = (n_ticks - i) & ~1;
    if (h->graph_on && h->tg.exec && L >= kShortMin) {
      const bool fits = graph_usable(h, h->tgs, x0s, seed, st) && h->tgs.len == L && h->tgs.ucur == h->ucur;
      if (!fits && h->tgs_wish_len == L && h->tgs_wish_ucur == h->ucur) build_graph(h, h->tgs, L, x0s, seed, st);
      h->tgs_wish_len = L; h->tgs_wish_ucur = h->ucur;
      if (h->tgs.exec && graph_usable(h, h->tgs, x0s, seed, st) && h->tgs.len == L && h->tgs.ucur == h->ucur) {
        const int rc = replay_graph(h, h->tgs, first_tick + (uint64_t)i, st);
        if (rc != TBNAV_OK) return rc;
        i += L;
      }
    }
  }
  for (; i < n_ticks; ++i) {
    const int rc = tbnav_mppi_enqueue_rng(h, x0s + (size_t)i * x0_stride, seed, first_tick + (uint64_t)i, stream);
    if (rc != TBNAV_OK) return rc;
  }
  return TBNAV_OK;
}

int tbnav_mppi_new_controls_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream, double u_out[2]) {
  if (!h || !u_out) return TBNAV_ERR_INVALID_ARG;
  h->publish_next = true;
  const int rc = tbnav_mppi_enqueue_rng(h, x0, seed, tick, stream);
  h->publish_next = false;
  return rc != TBNAV_OK ? rc : tbnav_mppi_last_controls(h, stream, u_out);
}

int tbnav_mppi_get_noise(tbnav_mppi* h, double* duL_host, double* duR_host) {
  if (!h || !duL_host || !duR_host) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const size_t n = (size_t)h->T * h->K;
  TBNAV_HIP(hipDeviceSynchronize());
  TBNAV_HIP(hipMemcpy(duL_host, h->d_duL, n * sizeof(double), hipMemcpyDeviceToHost));
  TBNAV_HIP(hipMemcpy(duR_host, h->d_duR, n * sizeof(double), hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_mppi_debug_sincos(const double* x_host, int32_t n, double* sin_host, double* cos_host) {
  if (!x_host || !sin_host || !cos_host || n <= 0) return TBNAV_ERR_INVALID_ARG;
  double *dx = nullptr, *ds = nullptr, *dc = nullptr;
  hipError_t e = hipMalloc((void**)&dx, sizeof(double) * n);
  if (e == hipSuccess) e = hipMalloc((void**)&ds, sizeof(double) * n);
  if (e == hipSuccess) e = hipMalloc((void**)&dc, sizeof(double) * n);
  if (e == hipSuccess) e = hipMemcpy(dx, x_host, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(mppi_debug_sincos, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, dx, ds, dc); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpy(sin_host, ds, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(cos_host, dc, sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(ds); (void)hipFree(dc);
  TBNAV_HIP(e);
  return TBNAV_OK;
}

int tbnav_mppi_debug_div_lambda(const double* x_host, int32_t n, double lambda, double* out_host, int32_t* used_reciprocal) {
  if (!x_host || !out_host || n <= 0) return TBNAV_ERR_INVALID_ARG;
  const Lam lam = lam_of(lambda);
  if (used_reciprocal) *used_reciprocal = lam.inv != 0.0;
  double *dx = nullptr, *dy = nullptr;
  hipError_t e = hipMalloc((void**)&dx, sizeof(double) * n);
  if (e == hipSuccess) e = hipMalloc((void**)&dy, sizeof(double) * n);
  if (e == hipSuccess) e = hipMemcpy(dx, x_host, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(mppi_debug_div_lambda, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, dx, lam, dy); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpy(out_host, dy, sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy);
  TBNAV_HIP(e);
  return TBNAV_OK;
}

int tbnav_mppi_get_cost_to_go(tbnav_mppi* h, double* J_host) {
  if (!h || !J_host) return TBNAV_ERR_INVALID_ARG;
  if (!h->j_valid) return TBNAV_ERR_INVALID_ARG;  // the last tick ran the fused kernel without TBNAV_MPPI_OPT_KEEP_J
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipDeviceSynchronize());
  TBNAV_HIP(hipMemcpy(J_host, h->d_J, (size_t)h->T * h->K * sizeof(double), hipMemcpyDeviceToHost));
  if (h->prefix_rows > 0) {  // rows below prefix_rows hold exclusive prefixes: J(i) = S - E(i), as mppi_partials forms it
    std::vector<double> tot((size_t)h->K);
    TBNAV_HIP(hipMemcpy(tot.data(), h->d_total, tot.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->prefix_rows; ++i)
      for (int k = 0; k < h->K; ++k)   // (a total that overflowed: +inf on every row, as mppi_partials reads it)
        J_host[(size_t)i * h->K + k] = std::isinf(tot[k]) ? tot[k] : tot[k] - J_host[(size_t)i * h->K + k];
  }
  return TBNAV_OK;
}

}  // extern "C"
