// mppi_rollout.hip — the rollout half of controller::MPPI::newControls (mppi.cpp:81-109; rk4.cpp:49-115; mppi.hpp:41-105), four
// kernel families (all fp64; which one a handle takes: tbnav_mppi_create in mppi.hip):
//   mppi_rollout_fused   small K (K/64 < 2 x CUs, T <= 128): one wave per rollout with its lanes over TIME, DPP wave scans for
//                        heading / position / cost-to-go, and the soft-min partial record of every time step over the workgroup's
//                        rollouts in the same launch; the perturbations drawn in-kernel (RNG = 1: fp32 Box-Muller, 2: fp64)
//   mppi_rollout_scan    the middle (8192 < K <= 40960): time-parallel, 64 rollouts x ceil(T/TC) chunk-waves, totals through LDS
//   mppi_rollout_prefix  large K (the streaming default): one lane per rollout, one branch per round of 12 steps, exclusive
//                        prefixes E(i) written as it goes, exact suffix sums for the last 4 RG steps, total S; J = S - E
//   mppi_rollout_cost    the general one-lane-per-rollout kernel — what the large-K shapes take that the prefix form does not cover:
//                        T not a multiple of 4 or below 16, T > 240 at any K beyond the fused kernel's, the exact-arc dynamics
//                        (TRIG == 4) and the TRIG = 2 / 3 settings beyond K = 8192 (tests/test_mppi_gpu.py: T = 400, ragged K,
//                        the arc dynamics and the trig settings at K = 16384 run it)
// Rollout dynamics: the reference's CartModel + RK4 (TRIG 1..3 = how many sincos per step are evaluated afresh), or the
// exact-arc option TRIG == 4 (DiffDrive::feedforward per step).  Shared arithmetic: mppi_device.hpp.
#include "mppi_device.hpp"

namespace tbnav_mk {

// Per-step losses are staged for the backward suffix sum.  LDS ([steps][64] doubles, one column per
// lane, conflict-free 8-B accesses) is the cheap place — J is then written exactly once and the forward
// pass issues no global stores — but T*512 B per one-wave block caps residency.  So the LAST
// (T - lds_from) steps go to LDS, sized at create time so that the whole grid is resident in one round
// (tbnav_mppi_create), and the first lds_from steps use J itself as scratch (re-read from L2).
// The noise of group g+1 (G steps x 2 arrays x 512 B per wave) is requested before group g is
// integrated, so the loads fly under a group's worth of trig instead of stalling each step.
template <int TRIG, int G, bool TO_LDS>
__device__ __forceinline__ void rollout_group(const RolloutArgs& a, int i0, int lane, int k, double& x, double& y,
                                              double& th, const double (&dl)[G], const double (&dr)[G],
                                              const double* __restrict__ u, double* __restrict__ lds_loss,
                                              double* __restrict__ J, double* reg_loss = nullptr) {
  const int T = a.T, K = a.K;
  double ul[G], ur[G], thq[G], xq[G], yq[G];
#pragma unroll
  for (int q = 0; q < G; ++q) {
    ul[q] = u[i0 + q] + dl[q];        // mppi.cpp:93 — rollout controls are not clamped
    ur[q] = u[T + i0 + q] + dr[q];
  }
  if constexpr (TRIG == 4) arc_steps<G>(a, x, y, th, ul, ur, thq, xq, yq);
  else rk4_steps<TRIG, G>(a, x, y, th, ul, ur, thq, xq, yq);
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const int i = i0 + q;
    const double l = (i == T - 1) ? terminal_loss(a, xq[q], yq[q], thq[q])  // mppi.cpp:105 overwrites, not adds
                                  : lqr_loss(a, xq[q], yq[q], thq[q], ul[q], ur[q]);
    if (reg_loss) reg_loss[q] = l;  // (callers pass a statically indexed slice of a register array)
    else if (TO_LDS) lds_loss[(i - a.lds_from) * kWave + lane] = l;
    else J[(size_t)i * K + k] = l;
  }
}

// LDS carve (dynamic): u_lds [2*T] (warm-start controls, broadcast reads) then, if LDS_STAGE, the losses.
constexpr int kAhead = 3;  // groups of noise requested ahead of the one being integrated (12 steps ~ 1.5 us of trig)
template <int TRIG>
__global__ __launch_bounds__(kWave) void mppi_rollout_cost(RolloutArgs a,
                                                           const double* __restrict__ duL,
                                                           const double* __restrict__ duR,
                                                           USrc u,
                                                           double* __restrict__ J) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  const int lane = threadIdx.x;
  const int T = a.T, K = a.K;
  double* u_lds = lds_all;                 // [2*T]
  double* lds_loss = lds_all + 2 * T;      // [T - lds_from][64]
  for (int t = lane; t < 2 * T; t += kWave) u_lds[t] = u.get(t >= T, t >= T ? t - T : t, T);
  __syncthreads();
  const int k = blockIdx.x * kWave + lane;
  if (k >= K) return;
  double x = a.x0[0], y = a.x0[1], th = a.x0[2];
  const int n_full = T / kGroup;
  const double* pl = duL + k;
  const double* pr = duR + k;
  double nl[kAhead][kGroup], nr[kAhead][kGroup];
#pragma unroll
  for (int r = 0; r < kAhead; ++r) {
    if (r < n_full) {
#pragma unroll
      for (int q = 0; q < kGroup; ++q) {
        const size_t off = (size_t)(r * kGroup + q) * K;
        nl[r][q] = pl[off];
        nr[r][q] = pr[off];
      }
    }
  }
  for (int g0 = 0; g0 < n_full; g0 += kAhead) {
#pragma unroll
    for (int r = 0; r < kAhead; ++r) {   // ring slot r holds group g0 + r
      const int g = g0 + r;
      if (g < n_full) {
        double dl[kGroup], dr[kGroup];
#pragma unroll
        for (int q = 0; q < kGroup; ++q) { dl[q] = nl[r][q]; dr[q] = nr[r][q]; }
        if (g + kAhead < n_full) {
#pragma unroll
          for (int q = 0; q < kGroup; ++q) {
            const size_t off = (size_t)((g + kAhead) * kGroup + q) * K;
            nl[r][q] = pl[off];
            nr[r][q] = pr[off];
          }
        }
        if (g * kGroup >= a.lds_from) rollout_group<TRIG, kGroup, true>(a, g * kGroup, lane, k, x, y, th, dl, dr, u_lds, lds_loss, J);
        else rollout_group<TRIG, kGroup, false>(a, g * kGroup, lane, k, x, y, th, dl, dr, u_lds, lds_loss, J);
      }
    }
  }
  for (int i = n_full * kGroup; i < T; ++i) {  // ragged tail, one step at a time
    const double dl[1] = {pl[(size_t)i * K]}, dr[1] = {pr[(size_t)i * K]};
    if (i >= a.lds_from) rollout_group<TRIG, 1, true>(a, i, lane, k, x, y, th, dl, dr, u_lds, lds_loss, J);
    else rollout_group<TRIG, 1, false>(a, i, lane, k, x, y, th, dl, dr, u_lds, lds_loss, J);
  }
  // cumSumCost (mppi.cpp:15-25): J(i) = loss(i) + J(i+1), from the end.  The staged losses are fetched
  // eight at a time, the next eight already in flight while these are added in order.
  double* Jk = J + k;
  const int lds_from = a.lds_from;
  auto staged = [&](int t) -> double { return t >= lds_from ? lds_loss[(t - lds_from) * kWave + lane] : Jk[(size_t)t * K]; };
  constexpr int kB = 8;
  double acc = 0.0;
  int i = T - 1;
  double cur[kB], nxt[kB];
  if (i >= kB - 1) {
#pragma unroll
    for (int q = 0; q < kB; ++q) cur[q] = staged(i - q);
  }
  for (; i >= kB - 1; i -= kB) {
    const bool more = (i - kB) >= kB - 1;
    if (more) {
#pragma unroll
      for (int q = 0; q < kB; ++q) nxt[q] = staged(i - kB - q);
    }
#pragma unroll
    for (int q = 0; q < kB; ++q) {
      acc = (i - q == T - 1) ? cur[q] : cur[q] + acc;
      Jk[(size_t)(i - q) * K] = acc;
    }
    if (more) {
#pragma unroll
      for (int q = 0; q < kB; ++q) cur[q] = nxt[q];
    }
  }
  for (; i >= 0; --i) {
    const double l = staged(i);
    acc = (i == T - 1) ? l : l + acc;
    Jk[(size_t)i * K] = acc;
  }
}

// ---- streaming rollout, prefix form (round 3; the large-K default) --------------------------------------------------
// What held mppi_rollout_cost_reg at 46 us (K = 65536, T = 100: ONE wave per SIMD — 1024 one-wave workgroups on 1024 SIMDs, so
// all latency hiding has to come from inside the wave): (1) small_sincos's wave-uniform `if (__any(big))` sat in EVERY step and
// cut the unrolled round into ~50 basic blocks of one step each — no scheduling region held more than one step's dependent
// fp64 chain; (2) every loss made a round trip through LDS and the kernel ended with a backward pass that is pure memory.
// Here:  * ONE branch per round of 12 steps: the round's controls are formed first, `any |d| > 2^-5` is decided once, and the
//          straight-line Taylor round (no branch inside: one scheduling region, 12 independent small-angle chains + 3 fresh
//          sincos chains) or the general round (the kernels above) runs;
//        * the forward pass keeps the running sum and stores the EXCLUSIVE PREFIX E(i) = loss(0) + ... + loss(i-1) to J[i] as it
//          goes (one coalesced 512-B store per step, under the trig); the last 4*RG steps keep their losses in registers and get
//          their exact suffix sums J(i) as before, and the rollout's total S = E(T - 4 RG) + J(T - 4 RG) goes to total[k].  The
//          consumers (mppi_partials, the parity getter) form J(i) = S - E(i) for the prefix rows: its rounding error is
//          eps * S — what J(0) = S carries anyway — and the rows where S / J(i) would amplify it (the horizon's end) are the
//          exact ones.  No LDS stage, no backward pass over LDS, J written once, nothing re-read.
//        * the step itself in fewer instructions: heading += h * w (the reference's (h/6) * (((w + 2w) + 2w) + w) is the same
//          number up to one rounding), stage headings by two successive rotations, x += (h/6 v) * ((c1 + 4 c2) + c4).  Differences
//          from the reference's association are <= 2 ulp per step (J asserted within 1e-12 of the oracle as for every kernel).
template <int G>
__device__ __forceinline__ void lean_group(const RolloutArgs& a, double& x, double& y, double& th, const double (&ul)[G], const double (&ur)[G],
                                           double (&thq)[G], double (&xq)[G], double (&yq)[G]) {
  double g6[G], d[G], hth[G];
  double t = th;
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const double w = a.r_over_b * (ur[q] - ul[q]);
    g6[q] = a.h6 * (a.half_r * (ul[q] + ur[q]));
    const double hw = a.h * w;
    d[q] = 0.5 * hw;
    hth[q] = t;
    t = t + hw;
    thq[q] = t;
  }
  th = t;
  double s1, c1;
  fast_sincos(hth[0], s1, c1);   // fresh at the group's first step; the later steps carry the stage-4 pair (<= 3 steps = 6 rotations)
  double sd[G], cd[G];
#pragma unroll
  for (int q = 0; q < G; ++q) {   // straight-line Taylor pair, |d| <= 2^-5 (the caller checked the whole round)
    const double d2 = d[q] * d[q];
    double ps = fma(d2, -1.0 / 42.0, 1.0);
    ps = fma(d2 * (-1.0 / 20.0), ps, 1.0);
    ps = fma(d2 * (-1.0 / 6.0), ps, 1.0);
    sd[q] = d[q] * ps;
    double pc = fma(d2, -1.0 / 56.0, 1.0);
    pc = fma(d2 * (-1.0 / 30.0), pc, 1.0);
    pc = fma(d2 * (-1.0 / 12.0), pc, 1.0);
    cd[q] = fma(d2 * -0.5, pc, 1.0);
  }
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const double c2 = c1 * cd[q] - s1 * sd[q], s2 = s1 * cd[q] + c1 * sd[q];
    const double c4 = c2 * cd[q] - s2 * sd[q], s4 = s2 * cd[q] + c2 * sd[q];
    x = fma(g6[q], fma(4.0, c2, c1) + c4, x);
    y = fma(g6[q], fma(4.0, s2, s1) + s4, y);
    xq[q] = x; yq[q] = y;
    s1 = s4; c1 = c4;
  }
}

template <int RG>
__global__ __launch_bounds__(kWave) void mppi_rollout_prefix(RolloutArgs a_in, const double* __restrict__ duL, const double* __restrict__ duR, USrc u,
                                                             double* __restrict__ J, double* __restrict__ total) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  const int lane = threadIdx.x;
  const int T = a_in.T, K = a_in.K;
  double* u_lds = lds_all;                 // [2*T]
  for (int t = lane; t < 2 * T; t += kWave) u_lds[t] = u.get(t >= T, t >= T ? t - T : t, T);
  __shared__ double consts[20];            // (the rollout's constants through LDS into VECTOR registers: see mppi_rollout_cost_reg)
  if (lane == 0) {
    consts[0] = a_in.half_r; consts[1] = a_in.r_over_b; consts[2] = a_in.r_d; consts[3] = a_in.h; consts[4] = a_in.h6;
    for (int q = 0; q < 3; ++q) { consts[5 + q] = a_in.x0[q]; consts[8 + q] = a_in.xd[q]; consts[11 + q] = a_in.Q[q]; consts[16 + q] = a_in.P1[q]; }
    consts[14] = a_in.R[0]; consts[15] = a_in.R[1];
  }
  __syncthreads();
  RolloutArgs a;
  a.half_r = consts[0]; a.r_over_b = consts[1]; a.r_d = consts[2]; a.h = consts[3]; a.h6 = consts[4];
#pragma unroll
  for (int q = 0; q < 3; ++q) { a.x0[q] = consts[5 + q]; a.xd[q] = consts[8 + q]; a.Q[q] = consts[11 + q]; a.P1[q] = consts[16 + q]; }
  a.R[0] = consts[14]; a.R[1] = consts[15];
  a.T = T; a.K = K; a.lds_from = 0;
  const int k = blockIdx.x * kWave + lane;
  if (k >= K) return;
  double x = a.x0[0], y = a.x0[1], th = a.x0[2];
  const int n_main = T / kGroup - RG;                  // groups whose exclusive prefix goes to J: a multiple of kRoundGroups
  const size_t gstride = (size_t)kGroup * K;
  const double* pl = duL + k;
  const double* pr = duR + k;
  double* Jk = J + k;
  // |d| = |h/2 * r/b * (ur - ul)| <= 2^-5  <=>  |ur - ul| <= dmax
  const double dmax = 0.0625 / fabs(a.h * a.r_over_b);
  double nl[kRoundGroups][kGroup], nr[kRoundGroups][kGroup];
#pragma unroll
  for (int r = 0; r < kRoundGroups; ++r) {
#pragma unroll
    for (int q = 0; q < kGroup; ++q) {
      const size_t off = (size_t)(r * kGroup + q) * K;
      nl[r][q] = pl[off];
      nr[r][q] = pr[off];
    }
  }
  double acc = 0.0;  // E(i): losses of the steps before i, in step order
  const double* nxl = pl + (size_t)kRoundGroups * gstride;
  const double* nxr = pr + (size_t)kRoundGroups * gstride;
  for (int g0 = 0; g0 < n_main; g0 += kRoundGroups) {
    double ul[kRoundGroups][kGroup], ur[kRoundGroups][kGroup];
    bool big = false;
#pragma unroll
    for (int r = 0; r < kRoundGroups; ++r) {
#pragma unroll
      for (int q = 0; q < kGroup; ++q) {
        const int i = (g0 + r) * kGroup + q;
        ul[r][q] = u_lds[i] + nl[r][q];        // mppi.cpp:93 — rollout controls are not clamped
        ur[r][q] = u_lds[T + i] + nr[r][q];
        big |= !(fabs(ur[r][q] - ul[r][q]) <= dmax);
      }
    }
    // the noise of the next round (its last RG groups' worth past n_main is the late region's: same addresses, same ring)
    if (g0 + kRoundGroups < T / kGroup) {
#pragma unroll
      for (int r = 0; r < kRoundGroups; ++r) {
        if (g0 + kRoundGroups + r < T / kGroup) {
#pragma unroll
          for (int q = 0; q < kGroup; ++q) {
            nl[r][q] = nxl[(size_t)r * gstride + (size_t)q * K];
            nr[r][q] = nxr[(size_t)r * gstride + (size_t)q * K];
          }
        }
      }
      nxl += (size_t)kRoundGroups * gstride;
      nxr += (size_t)kRoundGroups * gstride;
    }
    double thq[kRoundGroups][kGroup], xq[kRoundGroups][kGroup], yq[kRoundGroups][kGroup];
    if (__any(big)) {
#pragma unroll
      for (int r = 0; r < kRoundGroups; ++r) rk4_steps<2, kGroup>(a, x, y, th, ul[r], ur[r], thq[r], xq[r], yq[r]);
    } else {
#pragma unroll
      for (int r = 0; r < kRoundGroups; ++r) lean_group<kGroup>(a, x, y, th, ul[r], ur[r], thq[r], xq[r], yq[r]);
    }
#pragma unroll
    for (int r = 0; r < kRoundGroups; ++r) {
#pragma unroll
      for (int q = 0; q < kGroup; ++q) {
        const int i = (g0 + r) * kGroup + q;   // (never the terminal step: that one is in the late region)
        Jk[(size_t)i * K] = acc;
        acc = acc + lqr_loss(a, xq[r][q], yq[r][q], thq[r][q], ul[r][q], ur[r][q]);
      }
    }
  }
  // late region: the last RG groups, losses in registers, exact suffix sums (mppi.cpp:15-25 from the end)
  static_assert(RG <= kRoundGroups, "the late region's noise is what the last round's prefetch left in the ring");
  double lreg[RG][kGroup];
#pragma unroll
  for (int j = 0; j < RG; ++j) {
    double dl[kGroup], dr[kGroup];
#pragma unroll
    for (int q = 0; q < kGroup; ++q) {
      dl[q] = nl[j][q]; dr[q] = nr[j][q];
    }
    rollout_group<2, kGroup, true>(a, (n_main + j) * kGroup, lane, k, x, y, th, dl, dr, u_lds, nullptr, J, lreg[j]);
  }
  double suf = 0.0;
#pragma unroll
  for (int j = RG - 1; j >= 0; --j) {
#pragma unroll
    for (int q = kGroup - 1; q >= 0; --q) {
      const int i = (n_main + j) * kGroup + q;
      suf = (j == RG - 1 && q == kGroup - 1) ? lreg[j][q] : lreg[j][q] + suf;
      Jk[(size_t)i * K] = suf;
    }
  }
  total[k] = acc + suf;   // S = E(T - 4 RG) + J(T - 4 RG)
}

// ---- time-parallel rollout ---------------------------------------------------------------------------
// The cart's increments do not depend on position: th_{i+1} = th_i + dth(u_i) and
// x_{i+1} = x_i + incx(th_i, u_i), so a rollout is three scans (heading, position, cost-to-go) around
// purely element-wise work — and the element-wise work is where the time goes (sincos, the loss).
// One workgroup = 64 rollouts x C time chunks (one wave per chunk of TC steps held in registers):
//   1. every thread loads its TC steps of noise (2*TC independent 512-B wave loads in flight at once),
//      forms dth, chunk-local exclusive prefix; chunk totals meet in LDS; heading at chunk start =
//      th0 + totals of the earlier chunks (added in chunk order);
//   2. TC independent trig evaluations (ILP), incx/incy, chunk-local prefix, totals through LDS;
//   3. losses, chunk-local suffix sums, totals of the LATER chunks added from the horizon backwards.
// Compared with the one-lane-per-rollout kernel this multiplies the number of waves by C (K = 1024,
// T = 50: 16 -> 208 waves; K = 65536, T = 100: 1024 -> 13312), which is what hides the fp64 dependent
// latency.  The only numerical difference is the association of the three sums (chunked instead of
// strictly sequential): <= a few 1e-16 relative on x, y, theta and J (tests assert J within 1e-11).
// MAXW = most waves (time chunks) per workgroup: 12 -> 3 waves per SIMD, up to 168 VGPRs; 16 -> 4 per SIMD, 128.
template <int TRIG, int TC, int MAXW>
__global__ __launch_bounds__(kWave * MAXW) void mppi_rollout_scan(RolloutArgs a, const double* __restrict__ duL,
                                                                           const double* __restrict__ duR,
                                                                           USrc u,
                                                                           double* __restrict__ J) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  const int lane = threadIdx.x, c = threadIdx.y, C = blockDim.y;
  const int T = a.T, K = a.K;
  double* u_lds = lds_all;                       // [2*T]
  double* tot = lds_all + 2 * T;                 // [4][C][64]: dtheta, dx, dy, loss totals per chunk
  for (int t = c * kWave + lane; t < 2 * T; t += C * kWave) u_lds[t] = u.get(t >= T, t >= T ? t - T : t, T);
  __syncthreads();
  const int k = blockIdx.x * kWave + lane;
  const bool live = k < K;
  const int kk = live ? k : K - 1;               // dead lanes shadow a valid rollout (no divergence at the barriers)
  const int i0 = c * TC;
  // Live across the phases: per step v, w (or the control cost), the chunk-local heading, then x, y.
  // The trig is done in sub-batches of kSub steps (scheduling barrier between them): kSub independent
  // chains are enough to cover the fp64 latency, and the temporaries of more would spill.
  constexpr int kSub = (TC % 5 == 0) ? 5 : 4;
  double vv[TC], ww[TC];
#pragma unroll
  for (int q = 0; q < TC; ++q) {                 // 2*TC independent 512-B wave loads in flight
    const int i = i0 + q;
    const size_t off = (size_t)(i < T ? i : T - 1) * K + kk;
    vv[q] = duL[off];
    ww[q] = duR[off];
  }
  double tha[TC], ctrl[TC];                      // chunk-local heading AFTER step q; control cost of step q
  double run = 0.0;
#pragma unroll
  for (int q = 0; q < TC; ++q) {
    const int i = i0 + q;
    const bool in = i < T;
    const double ul = in ? u_lds[i] + vv[q] : 0.0;      // mppi.cpp:93 — rollout controls are not clamped
    const double ur = in ? u_lds[T + i] + ww[q] : 0.0;
    ctrl[q] = (ul * a.R[0]) * ul + (ur * a.R[1]) * ur;
    vv[q] = a.half_r * (ul + ur);
    ww[q] = a.r_over_b * (ur - ul);
    run += in ? a.h6 * (((ww[q] + 2.0 * ww[q]) + 2.0 * ww[q]) + ww[q]) : 0.0;
    tha[q] = run;
  }
  tot[(0 * C + c) * kWave + lane] = run;
  __syncthreads();
  double th0 = a.x0[2];
  for (int cc = 0; cc < c; ++cc) th0 += tot[(0 * C + cc) * kWave + lane];
  double runx = 0.0, runy = 0.0;
  double s4 = 0.0, c4 = 1.0;
#pragma unroll
  for (int q = 0; q < TC; ++q) {
    if (q % kSub == 0 && q) __builtin_amdgcn_sched_barrier(0);
    const double hth = th0 + (q == 0 ? 0.0 : tha[q - 1]);   // heading at the START of step i0+q
    const double v = vv[q], w = ww[q];
    double s1, c1, s2, c2;
    if (TRIG == 1 && (q & 3) != 0) { s1 = s4; c1 = c4; }   // carried from the previous step's stage 4 (refresh every 4th step)
    else fast_sincos(hth, s1, c1);
    if (TRIG == 3) {
      fast_sincos(hth + a.h * (0.5 * w), s2, c2);
      fast_sincos(hth + a.h * w, s4, c4);
    } else {
      double sd, cd;
      small_sincos(a.h * (0.5 * w), sd, cd);
      c2 = c1 * cd - s1 * sd;
      s2 = s1 * cd + c1 * sd;
      const double s2d = 2.0 * sd * cd, c2d = 1.0 - 2.0 * sd * sd;
      c4 = c1 * c2d - s1 * s2d;
      s4 = s1 * c2d + c1 * s2d;
    }
    const double k1x = v * c1, k1y = v * s1, k2x = v * c2, k2y = v * s2, k4x = v * c4, k4y = v * s4;
    const bool in = i0 + q < T;
    runx += in ? a.h6 * (((k1x + 2.0 * k2x) + 2.0 * k2x) + k4x) : 0.0;
    runy += in ? a.h6 * (((k1y + 2.0 * k2y) + 2.0 * k2y) + k4y) : 0.0;
    vv[q] = runx;                                // (reuse) chunk-local x AFTER step q
    ww[q] = runy;                                //         chunk-local y
  }
  __builtin_amdgcn_sched_barrier(0);
  tot[(1 * C + c) * kWave + lane] = runx;
  tot[(2 * C + c) * kWave + lane] = runy;
  __syncthreads();
  double xs = a.x0[0], ys = a.x0[1];
  for (int cc = 0; cc < c; ++cc) { xs += tot[(1 * C + cc) * kWave + lane]; ys += tot[(2 * C + cc) * kWave + lane]; }
  double* xa = vv;
  run = 0.0;
#pragma unroll
  for (int q = TC - 1; q >= 0; --q) {
    const int i = i0 + q;
    const double e0 = (xs + vv[q]) - a.xd[0], e1 = (ys + ww[q]) - a.xd[1], e2 = (th0 + tha[q]) - a.xd[2];
    double l = (i == T - 1) ? ((e0 * a.P1[0]) * e0 + (e1 * a.P1[1]) * e1) + (e2 * a.P1[2]) * e2      // mppi.cpp:105 overwrites
                            : (((e0 * a.Q[0]) * e0 + (e1 * a.Q[1]) * e1) + (e2 * a.Q[2]) * e2) + ctrl[q];
    if (i >= T) l = 0.0;
    run = l + run;                               // chunk-local suffix sum, from the chunk's end
    xa[q] = run;                                 // (reuse: suffix value)
  }
  tot[(3 * C + c) * kWave + lane] = run;
  __syncthreads();
  double tail = 0.0;
  for (int cc = C - 1; cc > c; --cc) tail = tot[(3 * C + cc) * kWave + lane] + tail;
  if (live) {
#pragma unroll
    for (int q = 0; q < TC; ++q)
      if (i0 + q < T) J[(size_t)(i0 + q) * K + k] = xa[q] + tail;
  }
}

// ---- fused rollout + soft-min partials for small K (lanes = TIME) ------------------------------------------
// When K/64 one-wave workgroups cannot fill the chip (K = 1024: 16 of 256 CUs), the tick is three short kernels
// whose execution time is all latency.  This kernel turns the rollout round: one WAVE per rollout with its lanes
// over the time steps (TL consecutive steps per lane), so the three scans of the time-parallel formulation
// (heading, position, cost-to-go) are wave scans — no chunk totals through LDS, no barriers between them — and a
// workgroup is R rollouts (R waves), i.e. K/R workgroups spread over the chip (K = 1024, R = 8: 128 CUs).
//   1. lane t of wave r loads element (t, k0 + r) of the noise: the R waves share each row's cache line(s);
//   2. per lane: controls, dtheta -> wave scan -> heading at the start of its steps -> ONE sincos per step
//      (+ angle addition) -> RK4 increments -> wave scans -> x, y -> loss -> wave suffix scan -> J (into LDS);
//   3. J goes out coalesced, and — the tile being in LDS anyway — the soft-min partial record of every time step
//      over the workgroup's R rollouts is formed here (groups of R lanes, xor-shuffles inside the group):
//      records[T][K/R][8], the same record the partials kernel writes for a 2048-rollout slice.
// The combine then merges K/R records per step instead of K/2048.  Numerics: the sums are wave-scan trees
// instead of sequential chains (a few 1e-16 relative on x, y, theta, J — inside the 1e-11 J assertion).
template <int TRIG, int R, int TL, int RNG>
__global__ __launch_bounds__(kWave * R) void mppi_rollout_fused(RolloutArgs a, const double* __restrict__ duL,
                                                                const double* __restrict__ duR, USrc u, Lam lam,
                                                                double* __restrict__ J /* NULL: not kept */, double* __restrict__ records, int S,
                                                                RngArgs rng) {
  extern __shared__ __attribute__((aligned(16))) double lds_all[];
  constexpr int RP = R + 1;  // padded tile rows: the transposed reads of a wave hit distinct banks
  const int T = a.T, K = a.K;
  double* nL = lds_all;           // [T][RP]
  double* nR = nL + T * RP;       // [T][RP]
  double* Jl = nR + T * RP;       // [T][RP]
  const int tid = threadIdx.x, lane = tid & (kWave - 1), r = tid / kWave, nthr = kWave * R;
  const int k0 = blockIdx.x * R;
  const int kk = (k0 + r < K) ? k0 + r : K - 1;  // a ragged tail shadows a valid rollout
  {
    // lane = time: the R waves of the workgroup read the same 64*R/8-byte rows of the noise, one element each (the
    // row is one or two cache lines, fetched once and served to the other waves from L1); the values also go to
    // the LDS tile for the partials below
    double dl[TL], dr[TL], uL[TL], uR[TL];
    if constexpr (RNG != 0) { if (rng.tick0) rng.base += *rng.tick0 * rng.per_tick; }  // (a scalar load, under the warm-start loads)
#pragma unroll
    for (int q = 0; q < TL; ++q) {
      const int i = lane * TL + q, ii = i < T ? i : T - 1;
      if constexpr (RNG != 0) device_noise<RNG == 2>(rng, T, ii, kk, dl[q], dr[q]);  // production mode: the perturbations never touch HBM
      else { dl[q] = duL[(size_t)ii * K + kk]; dr[q] = duR[(size_t)ii * K + kk]; }
      uL[q] = u.get(0, ii, T);
      uR[q] = u.get(1, ii, T);
    }
    double v[TL], w[TL], ctrl[TL], pth[TL];
    double run = 0.0;
#pragma unroll
    for (int q = 0; q < TL; ++q) {
      const int i = lane * TL + q;
      const bool in = i < T;
      const double ul = in ? uL[q] + dl[q] : 0.0;  // mppi.cpp:93 — rollout controls are not clamped
      const double ur = in ? uR[q] + dr[q] : 0.0;
      if (in) { nL[i * RP + r] = dl[q]; nR[i * RP + r] = dr[q]; }
      ctrl[q] = (ul * a.R[0]) * ul + (ur * a.R[1]) * ur;
      if constexpr (TRIG == 4) {  // exact arc: (v, w) hold the step's body-frame displacement (xn, yn)
        double thn;
        arc_body_step(a, ul, ur, v[q], w[q], thn);
        run += in ? thn : 0.0;
      } else {
        v[q] = a.half_r * (ul + ur);
        w[q] = a.r_over_b * (ur - ul);
        run += in ? a.h6 * (((w[q] + 2.0 * w[q]) + 2.0 * w[q]) + w[q]) : 0.0;
      }
      pth[q] = run;  // lane-local heading change AFTER step q
    }
    const double th_lane = a.x0[2] + (tbnav::wave_scan_incl(run, lane) - run);  // heading at the start of this lane's steps
    double runx = 0.0, runy = 0.0, px[TL], py[TL];
#pragma unroll
    for (int q = 0; q < TL; ++q) {
      double hth = th_lane + (q == 0 ? 0.0 : pth[q - 1]);
      double s1, c1, s2, c2, s4, c4;
      if constexpr (TRIG == 4) {
        // feedforward builds Twb from the CURRENT heading: the raw x0 for the first step, normalised afterwards
        if (lane * TL + q > 0) hth = normalize_angle_pi(hth);
        fast_sincos(hth, s1, c1);
        const bool in4 = lane * TL + q < T;
        runx += in4 ? (c1 * v[q] - s1 * w[q]) : 0.0;
        runy += in4 ? (s1 * v[q] + c1 * w[q]) : 0.0;
        px[q] = runx;
        py[q] = runy;
        continue;
      }
      fast_sincos(hth, s1, c1);
      if (TRIG == 3) {
        fast_sincos(hth + a.h * (0.5 * w[q]), s2, c2);
        fast_sincos(hth + a.h * w[q], s4, c4);
      } else {
        double sd, cd;
        small_sincos(a.h * (0.5 * w[q]), sd, cd);
        c2 = c1 * cd - s1 * sd;
        s2 = s1 * cd + c1 * sd;
        const double s2d = 2.0 * sd * cd, c2d = 1.0 - 2.0 * sd * sd;
        c4 = c1 * c2d - s1 * s2d;
        s4 = s1 * c2d + c1 * s2d;
      }
      const double k1x = v[q] * c1, k1y = v[q] * s1, k2x = v[q] * c2, k2y = v[q] * s2, k4x = v[q] * c4, k4y = v[q] * s4;
      const bool in = lane * TL + q < T;
      runx += in ? a.h6 * (((k1x + 2.0 * k2x) + 2.0 * k2x) + k4x) : 0.0;
      runy += in ? a.h6 * (((k1y + 2.0 * k2y) + 2.0 * k2y) + k4y) : 0.0;
      px[q] = runx;
      py[q] = runy;
    }
    const double x_lane = a.x0[0] + (tbnav::wave_scan_incl(runx, lane) - runx);
    const double y_lane = a.x0[1] + (tbnav::wave_scan_incl(runy, lane) - runy);
    double suf[TL];
    run = 0.0;
#pragma unroll
    for (int q = TL - 1; q >= 0; --q) {
      const int i = lane * TL + q;
      const double th_after = (TRIG == 4) ? normalize_angle_pi(th_lane + pth[q]) : th_lane + pth[q];
      const double e0 = (x_lane + px[q]) - a.xd[0], e1 = (y_lane + py[q]) - a.xd[1], e2 = th_after - a.xd[2];
      double l = (i == T - 1) ? ((e0 * a.P1[0]) * e0 + (e1 * a.P1[1]) * e1) + (e2 * a.P1[2]) * e2      // mppi.cpp:105 overwrites
                              : (((e0 * a.Q[0]) * e0 + (e1 * a.Q[1]) * e1) + (e2 * a.Q[2]) * e2) + ctrl[q];
      if (i >= T) l = 0.0;
      run = l + run;
      suf[q] = run;  // lane-local suffix sum from the lane's last step
    }
    const double tail = tbnav::wave_scan_incl_rev(run, lane) - run;  // cost of every later lane's steps
#pragma unroll
    for (int q = 0; q < TL; ++q) {
      const int i = lane * TL + q;
      if (i < T) Jl[i * RP + r] = suf[q] + tail;
    }
  }
  __syncthreads();
  if (J) {  // parity hook only (tbnav_mppi_get_cost_to_go): the update itself needs the records, not J
    for (int idx = tid; idx < T * R; idx += nthr) {
      const int t = idx / R, rr = idx - t * R;
      if (k0 + rr < K) J[(size_t)t * K + k0 + rr] = Jl[t * RP + rr];
    }
  }
  // soft-min partial record of each time step over this workgroup's rollouts (mppi.cpp:115-121)
  const double inf = __builtin_huge_val();
  const int rr = tid % R;
  const bool ok = k0 + rr < K;
  for (int t = tid / R; t < T; t += nthr / R) {
    const double j = ok ? Jl[t * RP + rr] : inf;
    const double l = ok ? nL[t * RP + rr] : 0.0, rg = ok ? nR[t * RP + rr] : 0.0;
    // (group reductions on the DPP network: R is 4, 8 or 16 consecutive lanes — no LDS round trip per butterfly step)
    auto gmin = [](double x, double y) { return fmin(x, y); };
    auto gsum = [](double x, double y) { return x + y; };
    const double mn = tbnav::group_reduce_dpp<R>(j, gmin);
    // exp(-(J - min)/lambda) with the reference's association: (J - min) * -1.0 / lambda (mppi.cpp:117)
    const double e = ok ? exp(div_lambda((j - mn) * -1.0, lam)) : 0.0;
    const double A = tbnav::group_reduce_dpp<R>(e, gsum), B = tbnav::group_reduce_dpp<R>(e * l, gsum), C = tbnav::group_reduce_dpp<R>(e * rg, gsum);
    const double D = tbnav::group_reduce_dpp<R>(l, gsum), E = tbnav::group_reduce_dpp<R>(rg, gsum), n = tbnav::group_reduce_dpp<R>(ok ? 1.0 : 0.0, gsum);
    if (rr == 0) {
      double* rec = records + ((size_t)t * S + blockIdx.x) * TBNAV_MPPI_REC;
      rec[0] = mn; rec[1] = A; rec[2] = B; rec[3] = C; rec[4] = D; rec[5] = E; rec[6] = n; rec[7] = 0.0;
    }
  }
}

// ---- explicit instantiations: exactly what mppi.hip's launchers name -------------------------------------------------------------
#define TBNAV_ARGS_ROLLOUT RolloutArgs, const double* __restrict__, const double* __restrict__, USrc, double* __restrict__
template __global__ void mppi_rollout_cost<1>(TBNAV_ARGS_ROLLOUT);
template __global__ void mppi_rollout_cost<2>(TBNAV_ARGS_ROLLOUT);
template __global__ void mppi_rollout_cost<3>(TBNAV_ARGS_ROLLOUT);
template __global__ void mppi_rollout_cost<4>(TBNAV_ARGS_ROLLOUT);
template __global__ void mppi_rollout_prefix<1>(TBNAV_ARGS_ROLLOUT, double* __restrict__);
template __global__ void mppi_rollout_prefix<2>(TBNAV_ARGS_ROLLOUT, double* __restrict__);
template __global__ void mppi_rollout_prefix<3>(TBNAV_ARGS_ROLLOUT, double* __restrict__);
#define TBNAV_INST_SCAN(TR) \
  template __global__ void mppi_rollout_scan<TR, 4, 16>(TBNAV_ARGS_ROLLOUT); template __global__ void mppi_rollout_scan<TR, 4, 12>(TBNAV_ARGS_ROLLOUT);   \
  template __global__ void mppi_rollout_scan<TR, 5, 12>(TBNAV_ARGS_ROLLOUT); template __global__ void mppi_rollout_scan<TR, 6, 12>(TBNAV_ARGS_ROLLOUT);   \
  template __global__ void mppi_rollout_scan<TR, 7, 16>(TBNAV_ARGS_ROLLOUT); template __global__ void mppi_rollout_scan<TR, 8, 16>(TBNAV_ARGS_ROLLOUT);   \
  template __global__ void mppi_rollout_scan<TR, 8, 12>(TBNAV_ARGS_ROLLOUT); template __global__ void mppi_rollout_scan<TR, 10, 12>(TBNAV_ARGS_ROLLOUT);  \
  template __global__ void mppi_rollout_scan<TR, 12, 12>(TBNAV_ARGS_ROLLOUT); template __global__ void mppi_rollout_scan<TR, 16, 12>(TBNAV_ARGS_ROLLOUT); \
  template __global__ void mppi_rollout_scan<TR, 20, 12>(TBNAV_ARGS_ROLLOUT);
TBNAV_INST_SCAN(1)
TBNAV_INST_SCAN(3)
#undef TBNAV_INST_SCAN
#define TBNAV_ARGS_FUSED RolloutArgs, const double* __restrict__, const double* __restrict__, USrc, Lam, double* __restrict__, double* __restrict__, int, RngArgs
#define TBNAV_INST_FUSED(TR, RR, RG) \
  template __global__ void mppi_rollout_fused<TR, RR, 1, RG>(TBNAV_ARGS_FUSED); template __global__ void mppi_rollout_fused<TR, RR, 2, RG>(TBNAV_ARGS_FUSED);
#define TBNAV_INST_FUSED_TR(TR) \
  TBNAV_INST_FUSED(TR, 4, 0) TBNAV_INST_FUSED(TR, 8, 0) TBNAV_INST_FUSED(TR, 16, 0) TBNAV_INST_FUSED(TR, 8, 1) TBNAV_INST_FUSED(TR, 16, 1)
TBNAV_INST_FUSED_TR(2)
TBNAV_INST_FUSED_TR(3)
TBNAV_INST_FUSED_TR(4)
// the fp64 sampler (round 6: the handle's default, TBNAV_MPPI_OPT_SAMPLER = 1) in every dynamics' kernel
TBNAV_INST_FUSED(2, 8, 2) TBNAV_INST_FUSED(2, 16, 2) TBNAV_INST_FUSED(3, 8, 2) TBNAV_INST_FUSED(3, 16, 2) TBNAV_INST_FUSED(4, 8, 2) TBNAV_INST_FUSED(4, 16, 2)
#undef TBNAV_INST_FUSED_TR
#undef TBNAV_INST_FUSED
#undef TBNAV_ARGS_FUSED
#undef TBNAV_ARGS_ROLLOUT

}  // namespace tbnav_mk
