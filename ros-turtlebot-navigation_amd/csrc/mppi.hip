// mppi.hip — MI355X (gfx950) implementation of controller::MPPI::newControls behind the C-ABI of
// include/tbnav_mppi.h.  Reference: controller/src/controller/mppi.cpp:72-140, rk4.cpp:49-115,
// controller/include/controller/mppi.hpp:41-105 (paths relative to the reference tree).
//
// Kernels (all fp64; compiled with -ffp-contract=fast-honor-pragmas — the contract is a tolerance, csrc/Makefile and
// DESIGN.md section 4 say why; sin/cos are this file's own fast_sincos):
//   mppi_rollout_fused  small K (the default for K/64 < 2 x CUs, T <= 128): one wave per rollout with its lanes over
//                       TIME, DPP wave scans for heading / position / cost-to-go, J -> [T][K], and the soft-min partial
//                       record of every time step over the workgroup's rollouts in the same launch (mppi.cpp:81-121)
//   mppi_rollout_scan   128 < T <= 240: 64 rollouts x ceil(T/TC) waves, TC steps per thread in registers, chunk totals
//                       through LDS
//   mppi_rollout_cost   large K: one lane per rollout, T steps in groups of four independent trig chains, per-step loss
//                       staged in LDS / J, in-lane suffix sum -> J[T][K]        (mppi.cpp:81-109)
//   mppi_partials       grid (K-slices, T): per-time-step min / soft-min partial sums over one K-slice ->
//                       records[T][S][8]                                        (mppi.cpp:115-121)
//   mppi_merge_records  sharded small-K ticks: fold the fused kernel's fine records into the K-slice records
//   mppi_combine        any number of workgroups: merge the records of all slices / shards, update + clamp u, emit
//                       u(:,0); the shift is applied on read by the next tick   (mppi.cpp:118-137)
//   mppi_unpack_noise   reference draw order [K][T][2] -> duL/duR [T][K]
//   mppi_sample_noise   Philox4x32-10 + Box-Muller, production replacement of mppi.cpp:173-184
// Rollout dynamics: the reference's CartModel + RK4 (TRIG 1..3 = how many sincos per step are evaluated afresh), or the
// exact-arc option TRIG == 4 (DiffDrive::feedforward per step).
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

#include "comm.hpp"
#include "common.hpp"
#include "tbnav_mppi.h"

namespace {

constexpr int kWave = 64;
constexpr int kSliceThreads = 256;
constexpr int kSliceItems = 8;
constexpr int kSlice = kSliceThreads * kSliceItems;  // rollouts per K-slice record
constexpr int kMaxLdsBytes = 160 * 1024;

// The warm-start controls as the kernels see them.  A tick leaves its updated, UNSHIFTED controls in `p`;
// the shift of mppi.cpp:134-137 (u(:,i) <- u(:,i+1), last <- uinit) is applied on read by the next tick
// (`shift` = 1), so no single workgroup has to own the whole vector and the combine can use many.
struct USrc {
  const double* p;   // [2][T]
  int shift;
  double init_l, init_r;
  __device__ __forceinline__ double get(int row, int i, int T) const {
    if (!shift) return p[row * T + i];
    return (i + 1 < T) ? p[row * T + i + 1] : (row ? init_r : init_l);
  }
};

// lambda and fl(1 / lambda) (formed on the host by the correctly rounded division; 0: use the division).  x / lambda is needed
// once per soft-min weight and sits on the K = 1024 tick's latency chain; the quotient below is the correctly rounded one — the
// SAME bits as x / lambda (Markstein: with y = RN(1/b), q = RN(a y), e = a - b q exactly by FMA, RN(q + e y) = RN(a/b) unless b's
// significand is all ones, which the host excludes) — in three dependent instructions instead of the division's ~25.
struct Lam { double lambda, inv; };
__device__ __forceinline__ double div_lambda(double x, const Lam& l) {
  if (l.inv == 0.0) return x / l.lambda;   // (launch-uniform)
  const double q = x * l.inv;
  const double e = fma(-q, l.lambda, x);
  return (fabs(q) < __builtin_huge_val()) ? fma(e, l.inv, q) : q;   // (an infinite cost: the quotient is the infinity itself, not inf - inf)
}
struct RolloutArgs {
  double half_r;    // wheel_radius / 2.0            (mppi.hpp:45)
  double r_over_b;  // wheel_radius / wheel_base     (mppi.hpp:47)
  double r_d;       // wheel_radius * (1 / wheel_base)   (diff_drive.cpp:85-88, arc dynamics only)
  double h;         // step                          (rk4.cpp:105)
  double h6;        // step / 6.0                    (rk4.cpp:114)
  double x0[3];
  double xd[3];
  double Q[3], R[2], P1[3];
  int T, K;
  int lds_from;     // steps i >= lds_from stage their loss in LDS (row i - lds_from); earlier ones in J itself
};

// One RK4 step of the kinematic cart with zero-order-hold control (rk4.cpp:95-115).  theta-dot does
// not depend on the state, so k1.theta == k2.theta == k3.theta == k4.theta == w and the four stages
// see the headings th, th+d, th+d, th+2d with d = h*(0.5*w).  Every product/sum keeps the
// reference's association.
//
// TRIG = 3: three sincos calls per step (th, th+d, th+2d), exactly the reference's evaluations.
// TRIG = 2: ONE sincos per step (th, refreshed every step) and the other two by angle addition.
// TRIG = 1 (default): as 2, and the step's own heading is carried over from the previous step's stage-4
//           rotation, with a fresh sincos every 4th step (<= 3 chained rotations, ~5e-16 absolute).
// TRIG 1/2: ONE sincos per step (th, refreshed every step so nothing accumulates) and the other two
//           headings by angle addition with sin/cos of the small angle d (|d| <= 2^-5: degree-11/10
//           Taylor polynomials, truncation < 1e-24; larger |d|: a full sincos of d).  The rotated
//           values are within ~2 ulp of libm's, i.e. the same size as the libm-vs-ocml difference the
//           parity tolerance already absorbs; J stays within 1e-12 of the oracle (tests).
// ---- device trig for the rollout -------------------------------------------------------------------
// ocml's sincos carries a Payne-Hanek path behind a branch and ~130 fp64 instructions; a rollout
// heading is a few radians.  fast_sincos: Cody-Waite reduction by pi/2 held as three doubles, each
// step one FMA (exact product, single rounding), then the fdlibm kernel polynomials on [-pi/4, pi/4].
// Measured against libm: <= 1.1e-16 absolute for |x| <= 1e5 (tests); the reduction itself stays good to
// ~1e-16 * (|x| * 2^-40 + 1), i.e. it degrades gracefully beyond 1e12 rad instead of branching to a
// library call (a call inside the unrolled rollout spills the whole register set).  Headings that
// large are unphysical (1e12 rad = 1.6e11 revolutions within one horizon); NaN/Inf propagate.
__device__ __forceinline__ void fast_sincos(double x, double& s, double& c) {
  const double kf = rint(x * 0.6366197723675814);
  double r = fma(-kf, 0x1.921fb54442d18p+0, x);
  r = fma(-kf, 0x1.1a62633145c07p-54, r);
  r = fma(-kf, -0x1.f1976b7ed8fbcp-110, r);
  const double z = r * r;
  // sin kernel
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  const double sr = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
  // cos kernel
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double hz = 0.5 * z, wq = 1.0 - hz;
  const double cr = wq + (((1.0 - wq) - hz) + z * (z * pc));
  const int n = (int)(kf - 4.0 * rint(kf * 0.25)) & 3;   // quadrant; kf may exceed the int range
  const double sa = (n & 1) ? cr : sr, ca = (n & 1) ? sr : cr;
  s = (n & 2) ? -sa : sa;
  c = ((n + 1) & 2) ? -ca : ca;
}

__device__ __forceinline__ void small_sincos(double d, double& sd, double& cd) {
  // straight-line Taylor pair (truncation < 3e-19 relative for |d| <= 2^-5); the full evaluation is entered only if SOME
  // lane of the wave needs it (wave-uniform branch: no divergence, and never taken for physical wheel speeds)
  const double d2 = d * d;
  // sin d = d (1 - d2/6 (1 - d2/20 (1 - d2/42)))          next term d^9/9!  <= 3e-19 relative at |d| = 2^-5
  double ps = 1.0 - d2 * (1.0 / 42.0);
  ps = 1.0 - d2 * (1.0 / 20.0) * ps;
  ps = 1.0 - d2 * (1.0 / 6.0) * ps;
  sd = d * ps;
  // cos d = 1 - d2/2 (1 - d2/12 (1 - d2/30 (1 - d2/56)))  next term d^10/10! <= 3e-22
  double pc = 1.0 - d2 * (1.0 / 56.0);
  pc = 1.0 - d2 * (1.0 / 30.0) * pc;
  pc = 1.0 - d2 * (1.0 / 12.0) * pc;
  cd = 1.0 - d2 * 0.5 * pc;
  const bool big = !(fabs(d) <= 0.03125);
  if (__any(big)) {
    double sf, cf;
    fast_sincos(d, sf, cf);
    sd = big ? sf : sd;
    cd = big ? cf : cd;
  }
}


// ---- exact-arc dynamics (TRIG == 4; SURVEY.md 8-f N4 — an option, NOT the reference MPPI's RK4) ------------
// One rollout step = the plant's own update: twist = DiffDrive::wheelsToTwist(u) * dt (diff_drive.cpp:79-94),
// Transform2D::integrateTwist from the identity (rigid2d.cpp:239-303: the screw's rotation branch, the pure
// translation branch for |w dt| < 1e-12, standstill), composed onto the pose as DiffDrive::feedforward does
// (diff_drive.cpp:175-194: x += c*x' - s*y', heading normalised to (-pi, pi]).  The body-frame displacement
// (xn, yn, thn) of a step depends only on the controls, so the rollout keeps the time-parallel shape: heading =
// scan of thn, position = scan of the rotated displacements.
__device__ __forceinline__ double normalize_angle_pi(double rad) {  // rigid2d.hpp:52-64
  const double kPi = 3.14159265358979323846;
  const double q = floor((rad + kPi) / (2.0 * kPi));
  rad = (rad + kPi) - q * 2.0 * kPi;
  if (rad < 0) rad += 2.0 * kPi;
  return rad - kPi;
}
__device__ __forceinline__ void arc_body_step(const RolloutArgs& a, double ul, double ur, double& xn, double& yn, double& thn) {
  const double tw = (a.r_d * (ur - ul)) * a.h;      // twist.w * dt
  const double tv = (a.half_r * (ul + ur)) * a.h;   // twist.vx * dt   (vy == 0)
  xn = 0.0; yn = 0.0; thn = 0.0;
  if (!(fabs(tw) < 1.0e-12)) {
    const double beta = fabs(tw), Sw = tw / beta, Svx = tv / beta;
    double sb, cb;
    fast_sincos(beta, sb, cb);
    const double mw2 = -1.0 * (Sw * Sw);
    xn = Svx * (beta + (beta - sb) * mw2);
    yn = Svx * ((1.0 - cb) * Sw);
    thn = atan2(sb * Sw, 1.0 + (1.0 - cb) * mw2);
  } else if (!(fabs(tv) < 1.0e-12)) {
    xn = tv;  // S.vx * beta with beta = |tv|, S.vx = +-1; no rotation
  }
}
template <int G>
__device__ __forceinline__ void arc_steps(const RolloutArgs& a, double& x, double& y, double& th,
                                          const double (&ul)[G], const double (&ur)[G], double (&thq)[G],
                                          double (&xq)[G], double (&yq)[G]) {
  double xn[G], yn[G], hth[G];
  double t = th;
#pragma unroll
  for (int q = 0; q < G; ++q) {
    double thn;
    arc_body_step(a, ul[q], ur[q], xn[q], yn[q], thn);
    hth[q] = t;                              // heading at the START of step q (what Twb is built from)
    t = normalize_angle_pi(t + thn);
    thq[q] = t;
  }
  th = t;
#pragma unroll
  for (int q = 0; q < G; ++q) {
    double s1, c1;
    fast_sincos(hth[q], s1, c1);
    x = (c1 * xn[q] - s1 * yn[q]) + x;
    y = (s1 * xn[q] + c1 * yn[q]) + y;
    xq[q] = x;
    yq[q] = y;
  }
}

// G consecutive RK4 steps.  The heading recurrence th_{i+1} = th_i + (h/6)*(6 w_i) is a cheap serial
// chain, so the G headings are formed first and the G expensive trig evaluations that depend on them
// are INDEPENDENT: with one wave per SIMD (K = 65536 gives exactly that) the in-order issue would
// otherwise sit on each sincos's dependent chain; this way G chains are in flight at once.  x and y
// are then accumulated in step order with the reference's association, and the per-step losses formed.
template <int TRIG, int G>
__device__ __forceinline__ void rk4_steps(const RolloutArgs& a, double& x, double& y, double& th,
                                          const double (&ul)[G], const double (&ur)[G], double (&thq)[G],
                                          double (&xq)[G], double (&yq)[G]) {
  double v[G], w[G], hth[G];
#pragma unroll
  for (int q = 0; q < G; ++q) {
    v[q] = a.half_r * (ul[q] + ur[q]);
    w[q] = a.r_over_b * (ur[q] - ul[q]);
  }
  double t = th;
#pragma unroll
  for (int q = 0; q < G; ++q) {
    hth[q] = t;                                              // heading at the START of step q
    t = t + a.h6 * (((w[q] + 2.0 * w[q]) + 2.0 * w[q]) + w[q]);
    thq[q] = t;                                              // heading AFTER step q (what the loss sees)
  }
  th = t;
  double s1[G], c1[G], s2[G], c2[G], s4[G], c4[G];
#pragma unroll
  for (int q = 0; q < G; ++q) {
    // TRIG == 1: only the group's first step evaluates sincos; the heading at the start of step q is the
    // stage-4 heading of step q-1 (th + h*w vs th + (h/6)*6w: the same angle up to one rounding), so its
    // sin/cos are carried over — three chained rotations at most before the next fresh evaluation.
    if (TRIG == 1 && q > 0) { s1[q] = s4[q - 1]; c1[q] = c4[q - 1]; }
    else fast_sincos(hth[q], s1[q], c1[q]);
    if (TRIG == 3) {
      fast_sincos(hth[q] + a.h * (0.5 * w[q]), s2[q], c2[q]);
      fast_sincos(hth[q] + a.h * w[q], s4[q], c4[q]);
    } else {
      double sd, cd;
      small_sincos(a.h * (0.5 * w[q]), sd, cd);
      c2[q] = c1[q] * cd - s1[q] * sd;
      s2[q] = s1[q] * cd + c1[q] * sd;
      const double s2d = 2.0 * sd * cd, c2d = 1.0 - 2.0 * sd * sd;  // double angle: one rotation from (c1, s1)
      c4[q] = c1[q] * c2d - s1[q] * s2d;
      s4[q] = s1[q] * c2d + c1[q] * s2d;
    }
  }
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const double k1x = v[q] * c1[q], k1y = v[q] * s1[q];
    const double k2x = v[q] * c2[q], k2y = v[q] * s2[q];
    const double k4x = v[q] * c4[q], k4y = v[q] * s4[q];
    x = x + a.h6 * (((k1x + 2.0 * k2x) + 2.0 * k2x) + k4x);
    y = y + a.h6 * (((k1y + 2.0 * k2y) + 2.0 * k2y) + k4y);
    xq[q] = x;
    yq[q] = y;
  }
}

__device__ __forceinline__ double lqr_loss(const RolloutArgs& a, double x, double y, double th,
                                           double ul, double ur) {
  const double e0 = x - a.xd[0], e1 = y - a.xd[1], e2 = th - a.xd[2];
  const double state = ((e0 * a.Q[0]) * e0 + (e1 * a.Q[1]) * e1) + (e2 * a.Q[2]) * e2;
  const double ctrl = (ul * a.R[0]) * ul + (ur * a.R[1]) * ur;
  return state + ctrl;
}
__device__ __forceinline__ double terminal_loss(const RolloutArgs& a, double x, double y, double th) {
  const double e0 = x - a.xd[0], e1 = y - a.xd[1], e2 = th - a.xd[2];
  return ((e0 * a.P1[0]) * e0 + (e1 * a.P1[1]) * e1) + (e2 * a.P1[2]) * e2;
}

// Per-step losses are staged for the backward suffix sum.  LDS ([steps][64] doubles, one column per
// lane, conflict-free 8-B accesses) is the cheap place — J is then written exactly once and the forward
// pass issues no global stores — but T*512 B per one-wave block caps residency.  So the LAST
// (T - lds_from) steps go to LDS, sized at create time so that the whole grid is resident in one round
// (tbnav_mppi_create), and the first lds_from steps use J itself as scratch (re-read from L2).
// The noise of group g+1 (G steps x 2 arrays x 512 B per wave) is requested before group g is
// integrated, so the loads fly under a group's worth of trig instead of stalling each step.
constexpr int kGroup = 4;
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

constexpr int kRoundGroups = 3;
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


__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t key, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// One (duL, duR) pair of the device noise source: Philox4x32-10 keyed by the seed, counter = tick*T*K + k*T + i, then
// Box-Muller.  mppi_sample_noise fills the [T][K] arrays with it; the fused kernel can call it in place of the loads.
struct RngArgs {  // base = tick * T * K_global + k0 * T
  uint64_t seed, base; double sig_l, sig_r;
  // replayed graphs of ticks (tbnav_mppi_enqueue_rng_batch): `base` is baked for the tick's position in the chunk and the chunk's
  // first tick is read from device memory, times the counters one tick uses
  const uint64_t* tick0 = nullptr; uint64_t per_tick = 0;
};
__device__ __forceinline__ void device_noise(const RngArgs& g, int T, int i, int k, double& dl, double& dr) {
  uint32_t r[4];
  philox4x32_10(g.base + (uint64_t)k * T + i, g.seed, r);
  // Box-Muller on the fp32 transcendental units (v_log_f32, v_sin_f32 / v_cos_f32 take their argument in turns):
  // a handful of instructions instead of ~130 fp64 ones for log + sincospi + sqrt.  The perturbations are random
  // numbers, not parity quantities: 24-bit uniforms give normals on a 2^-24 grid out to 5.9 sigma, which is all a
  // sampling controller can use (the reference's own sampler is not reproducible run to run either, utilities.cpp:14).
  const float u1 = ((float)(r[0] >> 8) + 0.5f) * 0x1.0p-24f;  // (0, 1)
  const float u2 = ((float)(r[1] >> 8) + 0.5f) * 0x1.0p-24f;  // [0, 1) turns
  const float rad = __builtin_sqrtf(-2.0f * 0.69314718056f * __builtin_amdgcn_logf(u1));  // v_log_f32 is log2
  dl = g.sig_l * (double)(rad * __builtin_amdgcn_cosf(u2));
  dr = g.sig_r * (double)(rad * __builtin_amdgcn_sinf(u2));
}

#ifdef TBNAV_PHASE_PROF
// development build: per-wave timeline (10 ns ticks) of workgroup 5 of the fused kernel and of the combine, printed by
// tbnav_mppi_destroy
__device__ unsigned long long g_mtrace[2][8][8];
#define MTRACE(k, i) do { if (blockIdx.x == 5 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 8) g_mtrace[k][threadIdx.x >> 6][i] = wall_clock64(); } while (0)
#else
#define MTRACE(k, i)
#endif
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
template <int TRIG, int R, int TL, bool RNG>
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
  MTRACE(0, 0);
  {
    // lane = time: the R waves of the workgroup read the same 64*R/8-byte rows of the noise, one element each (the
    // row is one or two cache lines, fetched once and served to the other waves from L1); the values also go to
    // the LDS tile for the partials below
    double dl[TL], dr[TL], uL[TL], uR[TL];
    if constexpr (RNG) { if (rng.tick0) rng.base += *rng.tick0 * rng.per_tick; }  // (a scalar load, under the warm-start loads)
#pragma unroll
    for (int q = 0; q < TL; ++q) {
      const int i = lane * TL + q, ii = i < T ? i : T - 1;
      if constexpr (RNG) device_noise(rng, T, ii, kk, dl[q], dr[q]);  // production mode: the perturbations never touch HBM
      else { dl[q] = duL[(size_t)ii * K + kk]; dr[q] = duR[(size_t)ii * K + kk]; }
      uL[q] = u.get(0, ii, T);
      uR[q] = u.get(1, ii, T);
    }
    MTRACE(0, 1);
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
    MTRACE(0, 2);
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
    MTRACE(0, 3);
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
  MTRACE(0, 4);
  __syncthreads();
  if (J) {  // parity hook only (tbnav_mppi_get_cost_to_go): the update itself needs the records, not J
    for (int idx = tid; idx < T * R; idx += nthr) {
      const int t = idx / R, rr = idx - t * R;
      if (k0 + rr < K) J[(size_t)t * K + k0 + rr] = Jl[t * RP + rr];
    }
  }
  MTRACE(0, 5);
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
  MTRACE(0, 6);
}

__device__ __forceinline__ double block_min(double v, double* scratch) {
  v = tbnav::wave_min_dpp(v);
  const int wid = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) scratch[wid] = v;
  __syncthreads();
  double r = scratch[0];
  for (int w = 1; w < kSliceThreads / kWave; ++w) r = fmin(r, scratch[w]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = tbnav::wave_sum_dpp(v);
  const int wid = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) scratch[wid] = v;
  __syncthreads();
  double r = scratch[0];
  for (int w = 1; w < kSliceThreads / kWave; ++w) r += scratch[w];
  __syncthreads();
  return r;
}

// grid = (S, T).  Block (s, i) reduces time step i over rollouts [s*kSlice, (s+1)*kSlice).
// prefix_rows > 0 (mppi_rollout_prefix ran): rows i < prefix_rows of J hold the exclusive prefix E(i) and total[k] the
// rollout's whole cost S: J(i, k) = S - E(i) is formed here (the 8 * K bytes of `total` are re-read by every time step's
// workgroups: L2 hits).
__global__ __launch_bounds__(kSliceThreads) void mppi_partials(int T, int K, int S, Lam lam,
                                                               const double* __restrict__ J,
                                                               const double* __restrict__ duL,
                                                               const double* __restrict__ duR,
                                                               double* __restrict__ records, int prefix_rows,
                                                               const double* __restrict__ total) {
  __shared__ double scratch[kSliceThreads / kWave];
  // which rows the rollout kernel touched LAST are the ones still in L2 / the Infinity Cache: the backward suffix pass of the
  // round-2 kernels ends at row 0, the prefix-form kernel's forward pass at row T - 1 — start there (35.7 -> us at K = 65536)
  const int s = blockIdx.x, i = prefix_rows > 0 ? T - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int base = s * kSlice;
  const double inf = __builtin_huge_val();
  double j[kSliceItems], l[kSliceItems], r[kSliceItems], tot[kSliceItems];
  const bool pre = i < prefix_rows;
  double mn = inf;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < kSliceItems; ++it) {
    const int k = base + it * kSliceThreads + threadIdx.x;  // coalesced across lanes
    const bool ok = k < K;
    j[it] = ok ? J[(size_t)i * K + k] : inf;
    tot[it] = (ok && pre) ? total[k] : inf;   // (requested with the other loads; subtracted below, once everything is on its way)
    l[it] = ok ? duL[(size_t)i * K + k] : 0.0;
    r[it] = ok ? duR[(size_t)i * K + k] : 0.0;
    cnt += ok ? 1 : 0;
  }
#pragma unroll
  for (int it = 0; it < kSliceItems; ++it) {
    // J(i) = S - E(i).  A missing rollout (tot = +inf by construction) and a rollout whose total OVERFLOWED to +inf both end as
    // J = +inf, weight 0 — what the suffix-sum kernels give the latter (its prefix row alone would be a FINITE E(i) and look like
    // the cheapest rollout of the step; round-3 advisor finding)
    if (pre) j[it] = (tot[it] == inf) ? inf : tot[it] - j[it];
    mn = fmin(mn, j[it]);
  }
  mn = block_min(mn, scratch);
  double A = 0, B = 0, C = 0, D = 0, E = 0;
#pragma unroll
  for (int it = 0; it < kSliceItems; ++it) {
    // exp(-(J - min)/lambda) with the reference's association: (J - min) * -1.0 / lambda (mppi.cpp:117)
    const double e = (j[it] == inf) ? 0.0 : exp(div_lambda((j[it] - mn) * -1.0, lam));
    A += e;
    B += e * l[it];
    C += e * r[it];
    D += l[it];
    E += r[it];
  }
  A = block_sum(A, scratch);
  B = block_sum(B, scratch);
  C = block_sum(C, scratch);
  D = block_sum(D, scratch);
  E = block_sum(E, scratch);
  const double n = block_sum((double)cnt, scratch);
  if (threadIdx.x == 0) {
    double* rec = records + ((size_t)i * S + s) * TBNAV_MPPI_REC;
    rec[0] = mn; rec[1] = A; rec[2] = B; rec[3] = C; rec[4] = D; rec[5] = E; rec[6] = n; rec[7] = 0.0;
  }
}


// Fold the fused kernel's fine records ([T][Sf][8], one per R rollouts) into the K-slice records the sharding
// interface exchanges ([T][S][8], one per kSlice = 2048 rollouts): grid (S, T), one wave each.  Same algebra as the
// combine's first half — re-base every partial sum to the common minimum — so the result equals the partials
// kernel's record for that slice up to the association of the sums.
// (the direct exchange's sending side, when the records are produced here: see mppi_direct_publish further down)
struct DirectPub { unsigned long long* const* peers; int me, P, parity; unsigned int seq; int only_self; };
__global__ __launch_bounds__(kWave) void mppi_merge_records(int T, int Sf, int per_slice, int S, Lam lam,
                                                            const double* __restrict__ fine, double* __restrict__ records, DirectPub pub) {
  const int s = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
  const int r0 = s * per_slice, r1 = min(Sf, r0 + per_slice);
  const double inf = __builtin_huge_val();
  double M = inf;
  for (int r = r0 + lane; r < r1; r += kWave) {
    const double* rec = fine + ((size_t)i * Sf + r) * TBNAV_MPPI_REC;
    if (rec[6] > 0.0) M = fmin(M, rec[0]);
  }
  M = tbnav::wave_min_dpp(M);
  double A = 0, B = 0, C = 0, D = 0, E = 0, n = 0;
  for (int r = r0 + lane; r < r1; r += kWave) {
    const double* rec = fine + ((size_t)i * Sf + r) * TBNAV_MPPI_REC;
    if (rec[6] > 0.0) {
      const double sc = exp(div_lambda((rec[0] - M) * -1.0, lam));
      A += sc * rec[1]; B += sc * rec[2]; C += sc * rec[3];
      D += rec[4]; E += rec[5]; n += rec[6];
    }
  }
  A = tbnav::wave_sum_dpp(A); B = tbnav::wave_sum_dpp(B); C = tbnav::wave_sum_dpp(C);
  D = tbnav::wave_sum_dpp(D); E = tbnav::wave_sum_dpp(E); n = tbnav::wave_sum_dpp(n);
  if (lane == 0) {
    double* out = records + ((size_t)i * S + s) * TBNAV_MPPI_REC;
    out[0] = M; out[1] = A; out[2] = B; out[3] = C; out[4] = D; out[5] = E; out[6] = n; out[7] = 0.0;
  }
  if (pub.peers) {
    // direct exchange: this record goes straight into every rank's buffer as tagged words (mppi_direct_publish's layout) — the
    // fold and the publish are one launch
    __shared__ double rec8[TBNAV_MPPI_REC];
    if (lane == 0) { rec8[0] = M; rec8[1] = A; rec8[2] = B; rec8[3] = C; rec8[4] = D; rec8[5] = E; rec8[6] = n; rec8[7] = 0.0; }
    __syncthreads();
    const size_t nrec = (size_t)T * S * TBNAV_MPPI_REC, base = ((size_t)i * S + s) * TBNAV_MPPI_REC;
    const unsigned long long tag = (unsigned long long)pub.seq << 32;
    for (int j = lane; j < pub.P * 2 * TBNAV_MPPI_REC; j += kWave) {
      const int q = j / (2 * TBNAV_MPPI_REC), w = j - q * 2 * TBNAV_MPPI_REC, f = w >> 1;
      if (pub.only_self && q != pub.me) continue;
      const unsigned long long b = (unsigned long long)__double_as_longlong(rec8[f]);
      unsigned long long* dst = pub.peers[q] + ((size_t)(pub.parity * pub.P + pub.me) * nrec + base + f) * 2 + (w & 1);
      __hip_atomic_store(dst, tag | ((w & 1) ? (b >> 32) : (b & 0xFFFFFFFFull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Merge the G*S partial records of every time step (records: [G][T][S][8]) and update u(:,i)
// (mppi.cpp:118-125).  Each time step gets a group of `tpr` lanes (the power of two >= the record count, at
// most a wave): 64/tpr steps per wave, xor-shuffle reductions inside the group.  Any number of workgroups:
// the updated controls are written UNSHIFTED to u_out and the shift is applied on read by the next tick
// (USrc), u(:,0) goes to `out` (mppi.cpp:129-131).
#ifndef TBNAV_COMBINE_WAVES
#define TBNAV_COMBINE_WAVES 1  // one wave per workgroup: the groups spread over as many CUs as there are time steps (K = 1024 tick 8.9 -> 8.4 us against four waves)
#endif
// The direct exchange's receiving side (see mppi_direct_publish): a record field is two tagged 8-byte words in this rank's own
// fine-grained buffer; poll them until both carry the tick's sequence number (bounded: an error word and zeros after `budget`
// ticks of the 100 MHz clock).
struct DirectSrc { const unsigned long long* w0; unsigned long long budget; int* err; int* err_dev; unsigned int seq; };  // err: mapped host word; err_dev: its device twin (what later ticks look at)
__device__ __forceinline__ double direct_load(const DirectSrc& d, size_t idx, bool& failed) {
  unsigned long long* w = const_cast<unsigned long long*>(d.w0) + 2 * idx;
  unsigned long long lo = 0ull, hi = 0ull;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    lo = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    hi = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned int)(lo >> 32) == d.seq && (unsigned int)(hi >> 32) == d.seq) break;
    if (wall_clock64() - t0 > d.budget) {
      __hip_atomic_fetch_or(d.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_or(d.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lo = hi = 0ull;
      failed = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
}

// NR whole records (7 fields = 14 consecutive words each) at once: every word is requested before any is looked at — the
// buffer is fine-grained memory, every load a trip to the fabric, and one field after the other would be fourteen of them
// in a row per record; only records whose words do not all carry the tick's number yet are asked for again.
template <int NR>
__device__ __forceinline__ bool direct_load_records(const DirectSrc& d, const size_t (&idx)[NR], const bool (&have)[NR], double (&out)[NR][7]) {
  unsigned long long w[NR][14];
  bool done[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    done[q] = !have[q];
#pragma unroll
    for (int f = 0; f < 7; ++f) out[q][f] = 0.0;
  }
  const unsigned long long t0 = wall_clock64();
  for (;;) {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (!done[q]) {
        unsigned long long* p = const_cast<unsigned long long*>(d.w0) + 2 * idx[q];
#pragma unroll
        for (int k = 0; k < 14; ++k) w[q][k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    bool all = true;
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (!done[q]) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 14; ++k) ok = ok && (unsigned int)(w[q][k] >> 32) == d.seq;
        if (ok) {
          done[q] = true;
#pragma unroll
          for (int f = 0; f < 7; ++f) out[q][f] = __longlong_as_double((long long)((w[q][2 * f + 1] << 32) | (w[q][2 * f] & 0xFFFFFFFFull)));
        } else all = false;
      }
    if (all) break;
    if (wall_clock64() - t0 > d.budget) {
      __hip_atomic_fetch_or(d.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_or(d.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;  // (the records that never came stay zero; the caller leaves its time step's controls as they were)
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

template <int kKeep, bool DIRECT = false>
__global__ __launch_bounds__(256) void mppi_combine(int T, int G, int S, Lam lam, double umax, USrc u,
                                                    const double* __restrict__ records, double* __restrict__ u_out,
                                                    double* __restrict__ out, double* __restrict__ out_host, double seq, DirectSrc ds) {
  // (DIRECT: `records` is not read — field f of record (g, i, sl) is polled for in the exchange buffer, same index)
  // (an exchange that has timed out once stays dead: the ticks queued behind it must not each wait the whole bound again)
  const bool dead = DIRECT && __hip_atomic_load(ds.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  // (a time step whose peers' records did not all arrive — now or in an earlier tick — is NOT updated: its controls go out as they
  //  came in, never a soft-min over this rank's shard alone; the error word reaches the host with the next enqueue / last_controls)
  bool failed = dead;
  auto field = [&](const double* rec, int f) { return DIRECT ? (dead ? 0.0 : direct_load(ds, (size_t)(rec - records) + f, failed)) : rec[f]; };
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = blockDim.x / kWave;
  const int R = G * S;
  int tpr = 1;
  while (tpr < R && tpr < kWave) tpr <<= 1;
  const int spw = kWave / tpr, sub = lane / tpr, l = lane - sub * tpr;
  const int i = (blockIdx.x * nw + wid) * spw + sub;
  const bool valid = i < T;
  MTRACE(1, 0);
  // the warm-start controls do not depend on the records: fetch them first, under the record loads
  const double u_l = valid ? u.get(0, i, T) : 0.0, u_r = valid ? u.get(1, i, T) : 0.0;
  // Up to kKeep records per lane stay in registers (2: at most 128 records per step — the K = 1024 tick, whose critical
  // path should not carry idle slots; 4 / 8: up to 256 / 512 — the fused kernel with 16 rollouts per workgroup up to K = 4096 / 8192);
  // beyond that the second pass re-reads them (L1/L2 hits).
  const bool keep = R <= kKeep * tpr;
  double rk[kKeep][7];
  if constexpr (DIRECT) {
    size_t idx[kKeep];
    bool hv[kKeep];
#pragma unroll
    for (int q = 0; q < kKeep; ++q) {
      const int r = l + q * tpr;
      hv[q] = valid && keep && r < R && !dead;
      const int g = (hv[q] && G > 1) ? r / S : 0, sl = hv[q] ? r - g * S : 0;
      idx[q] = (((size_t)g * T + (valid ? i : 0)) * S + sl) * TBNAV_MPPI_REC;
    }
    if constexpr (kKeep <= 2) failed = !direct_load_records<kKeep>(ds, idx, hv, rk) || failed;   // (the K = 1024 tick: both records' words in flight together)
    else {
#pragma unroll
      for (int q = 0; q < kKeep; ++q) {
        const size_t i1[1] = {idx[q]};
        const bool h1[1] = {hv[q]};
        double o1[1][7];
        failed = !direct_load_records<1>(ds, i1, h1, o1) || failed;
#pragma unroll
        for (int f = 0; f < 7; ++f) rk[q][f] = o1[0][f];
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < kKeep; ++q) {
      const int r = l + q * tpr;
      const bool have = valid && keep && r < R;
      const int g = (have && G > 1) ? r / S : 0, sl = have ? r - g * S : 0;  // (one group — every single-GPU tick: no division)
      const double* rec = records + (((size_t)g * T + (valid ? i : 0)) * S + sl) * TBNAV_MPPI_REC;
#pragma unroll
      for (int f = 0; f < 7; ++f) rk[q][f] = have ? rec[f] : 0.0;  // n == 0 marks "no record"
    }
  }
  double M = __builtin_huge_val();
  MTRACE(1, 1);
  if (keep) {
#pragma unroll
    for (int q = 0; q < kKeep; ++q) if (rk[q][6] > 0.0) M = fmin(M, rk[q][0]);
  } else if (valid) {
    for (int r = l; r < R; r += tpr) {
      const int g = r / S, sl = r - g * S;
      const double* rec = records + (((size_t)g * T + i) * S + sl) * TBNAV_MPPI_REC;
      if (field(rec, 6) > 0.0) M = fmin(M, field(rec, 0));
    }
  }
  if (tpr == kWave) M = tbnav::wave_min_dpp(M);  // a whole wave per time step: reductions on the DPP network
  else for (int off = tpr >> 1; off > 0; off >>= 1) M = fmin(M, __shfl_xor(M, off, kWave));
  double W = 0, NL = 0, NR = 0, SD = 0, SE = 0, SN = 0;
  MTRACE(1, 2);
  if (keep) {
#pragma unroll
    for (int q = 0; q < kKeep; ++q)
      if (rk[q][6] > 0.0) {
        const double sc = exp(div_lambda((rk[q][0] - M) * -1.0, lam));
        W += sc * rk[q][1]; NL += sc * rk[q][2]; NR += sc * rk[q][3];
        SD += rk[q][4]; SE += rk[q][5]; SN += rk[q][6];
      }
  } else if (valid) {
    for (int r = l; r < R; r += tpr) {
      const int g = r / S, sl = r - g * S;
      const double* rec = records + (((size_t)g * T + i) * S + sl) * TBNAV_MPPI_REC;
      const double rn = field(rec, 6);
      if (rn > 0.0) {
        const double sc = exp(div_lambda((field(rec, 0) - M) * -1.0, lam));
        W += sc * field(rec, 1); NL += sc * field(rec, 2); NR += sc * field(rec, 3);
        SD += field(rec, 4); SE += field(rec, 5); SN += rn;
      }
    }
  }
  if (tpr == kWave) {
    W = tbnav::wave_sum_dpp(W); NL = tbnav::wave_sum_dpp(NL); NR = tbnav::wave_sum_dpp(NR);
    SD = tbnav::wave_sum_dpp(SD); SE = tbnav::wave_sum_dpp(SE); SN = tbnav::wave_sum_dpp(SN);
  } else {
    for (int off = tpr >> 1; off > 0; off >>= 1) {
      W += __shfl_xor(W, off, kWave); NL += __shfl_xor(NL, off, kWave); NR += __shfl_xor(NR, off, kWave);
      SD += __shfl_xor(SD, off, kWave); SE += __shfl_xor(SE, off, kWave); SN += __shfl_xor(SN, off, kWave);
    }
  }
  if constexpr (DIRECT) {  // any lane of the time step's group
    int fl = failed ? 1 : 0;
    for (int off = tpr >> 1; off > 0; off >>= 1) fl |= __shfl_xor(fl, off, kWave);
    failed = fl != 0;
  }
  MTRACE(1, 3);
  if (valid && l == 0) {
    W += 1e-8 * SN;  // the reference adds 1e-8 to every weight before normalising (mppi.cpp:117)
    double ul = u_l + (NL + 1e-8 * SD) / W;
    double ur = u_r + (NR + 1e-8 * SE) / W;
    ul = fmin(fmax(ul, -umax), umax);  // std::clamp(u, -max, max), mppi.cpp:124-125
    ur = fmin(fmax(ur, -umax), umax);
    if (DIRECT && failed) { ul = u_l; ur = u_r; }
    u_out[i] = ul;
    u_out[T + i] = ur;
    if (i == 0) {
      out[0] = ul; out[1] = ur;
      if (out_host) {
        // synchronous ticks: the answer also lands in mapped pinned host memory, without a copy.  The tick number goes
        // last, behind a system-scope fence, so a host that sees it also sees the two values.  (Not done for
        // enqueue-only ticks: the fence and the write over the fabric sit on the kernel's critical path.)
        out_host[0] = ul; out_host[1] = ur;
        __threadfence_system();
        out_host[2] = seq;
      }
    }
  }
}

__global__ void mppi_debug_div_lambda(int n, const double* __restrict__ x, Lam lam, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = div_lambda(x[i], lam);
}
// raw[(k*T + i)*2 + c]  ->  duL[i*K + k], duR[i*K + k]
__global__ void mppi_debug_sincos(int n, const double* __restrict__ x, double* __restrict__ sn, double* __restrict__ cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { double a, b; fast_sincos(x[i], a, b); sn[i] = a; cs[i] = b; }
}

// raw[(k*T + i)*2 + c]  ->  duL[i*K + k], duR[i*K + k]
__global__ void mppi_unpack_noise(int T, int K, const double* __restrict__ raw,
                                  double* __restrict__ duL, double* __restrict__ duR) {
  const size_t n = (size_t)T * K;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx / T), i = (int)(idx % T);
    const double2 v = reinterpret_cast<const double2*>(raw)[idx];
    duL[(size_t)i * K + k] = v.x;
    duR[(size_t)i * K + k] = v.y;
  }
}

// ---- Philox4x32-10 (Salmon et al., SC'11) ------------------------------------------------------
// last node of a captured chunk of ticks: the next replay's first tick
__global__ void mppi_tick_advance(uint64_t* __restrict__ tick0, uint64_t n) { *tick0 += n; }
// the first tick of a replay that does not continue the previous one (the value rides in the launch arguments: no host buffer
// has to outlive the call)
__global__ void mppi_tick_set(uint64_t* __restrict__ tick0, uint64_t v) { *tick0 = v; }

// ---- direct exchange of the sharded tick's records between the ranks of ONE node (multi-process communicators) ---------------
// An RCCL all-gather of a few KB costs tens of microseconds per call on eight GPUs — several 9 us ticks.  Here every rank
// stores its records straight into every peer's gather buffer (mapped through hipIpcMemHandle, fine-grained memory, xGMI) as
// self-validating 8-byte words — (sequence number << 32) | 32 bits of payload, two words a double: a naturally aligned
// 8-byte store is atomic, so a word is either the old tick's or the new one's and no flag, fence or ordering between stores
// is needed — and the receiver polls its OWN buffer's words until they carry the tick's number (system-scope loads; bounded:
// a peer that never delivers raises an error word instead of hanging the device).  Two buffers take turns by the tick's
// parity: a rank can be at most one tick ahead of a peer still reading (it needs that peer's records to get further).
__global__ __launch_bounds__(256) void mppi_direct_publish(const double* __restrict__ mine, int n, unsigned long long* const* __restrict__ peers,
                                                            int me, int P, int parity, unsigned int seq, int only_self) {
  if (only_self && (int)blockIdx.y != me) return;  // (fault injection for the tests of the bound: the peers never see this tick's records)
  unsigned long long* dst = peers[blockIdx.y] + (size_t)(parity * P + me) * 2 * n;
  const unsigned long long tag = (unsigned long long)seq << 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(mine[i]);
    __hip_atomic_store(dst + 2 * i, tag | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 2 * i + 1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void mppi_direct_collect(unsigned long long* __restrict__ words, int n, int P, int parity, unsigned int seq,
                                                            double* __restrict__ out, int* __restrict__ err, unsigned long long budget_ticks) {
  const size_t total = (size_t)P * n;
  unsigned long long* w0 = words + (size_t)parity * P * 2 * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long lo = 0ull, hi = 0ull;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      lo = __hip_atomic_load(w0 + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      hi = __hip_atomic_load(w0 + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((unsigned int)(lo >> 32) == seq && (unsigned int)(hi >> 32) == seq) break;
      if (wall_clock64() - t0 > budget_ticks) {  // (100 MHz ticks) the records never came: report, deliver zeros
        __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lo = hi = 0ull;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    out[i] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
  }
}

__global__ void mppi_sample_noise(int T, int K, uint64_t seed, uint64_t base, double sig_l,
                                  double sig_r, double* __restrict__ duL, double* __restrict__ duR) {
  const size_t n = (size_t)T * K;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / K), k = (int)(idx % K);  // k fastest: coalesced stores
    const RngArgs g{seed, base, sig_l, sig_r};
    device_noise(g, T, i, k, duL[idx], duR[idx]);
  }
}

}  // namespace

// =================================================================================================
// Handle + C-ABI
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
  // tbnav_mppi_enqueue_rng_batch replays a captured hipGraph of kGraphTicks ticks (two launches each) instead of launching
  // them one by one: ~0.5 us less per tick of a 8-9 us tick
  bool graph_on = true;         // TBNAV_MPPI_OPT_BATCH_GRAPH; cleared for good if a capture ever fails
  hipGraph_t tg_graph = nullptr; hipGraphExec_t tg_exec = nullptr;
  uint64_t tg_seed = 0; double tg_x0[3] = {0, 0, 0}; hipStream_t tg_stream = nullptr; int tg_ucur = -1;
  // The graph's kernel nodes hold BY VALUE everything launch_fused / launch_combine read from the handle when it was captured
  // (waypoint, uinit, lambda, dynamics, trig, keep_j + the J pointer, the rng shard, fused_S and the record buffer).  Every
  // setter that changes one of those bumps cfg_epoch; a graph captured under another epoch is rebuilt, never replayed.
  uint64_t cfg_epoch = 0, tg_epoch = ~0ull;
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
  int* h_dx_err = nullptr; int* d_dx_err = nullptr;  // mapped pinned: raised by a combine that ran out of time waiting for a peer's words
  int* d_dx_dead = nullptr;                          // its device twin: later ticks see it without a trip over PCIe
  unsigned long long dx_budget = 200000000ull;       // 2 s of the 100 MHz clock (host-side skew between ranks is legitimate — a control loop's is milliseconds; longer: the peer has failed)
  bool dx_withhold = false;                          // fault injection (TBNAV_MPPI_OPT_DIRECT_EXCHANGE = 2, tests): this rank's records never reach its peers
  unsigned int dx_seq = 0;
  // set by the sharded tick round its call of the shard-partials entry point: if that ends in mppi_merge_records, the kernel
  // publishes the records itself (and clears this); otherwise the tick launches mppi_direct_publish
  bool pub_pending = false;
  DirectPub pub_next{nullptr, 0, 0, 0, 0u, 0};
};

namespace {

// the publication the sharded tick has asked for, taken over by the launch that can carry it out (see pub_pending)
DirectPub take_pub(tbnav_mppi* h) {
  if (!h->pub_pending) return DirectPub{nullptr, 0, 0, 0, 0u, 0};
  h->pub_pending = false;
  return h->pub_next;
}

Lam lam_of(const tbnav_mppi* h) {
  const double lambda = h->p.lambda, inv = 1.0 / lambda;
  unsigned long long bits; std::memcpy(&bits, &lambda, sizeof bits);
  const bool all_ones = (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;   // the one significand Markstein's theorem excludes
  return Lam{lambda, (std::isnormal(inv) && std::isnormal(lambda) && !all_ones) ? inv : 0.0};
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
  if (R == 8) { if (TL == 1) TBNAV_FUSED(TR, 8, 1, false); else TBNAV_FUSED(TR, 8, 2, false); }          \
  else if (R == 4) { if (TL == 1) TBNAV_FUSED(TR, 4, 1, false); else TBNAV_FUSED(TR, 4, 2, false); }     \
  else { if (TL == 1) TBNAV_FUSED(TR, 16, 1, false); else TBNAV_FUSED(TR, 16, 2, false); }
#define TBNAV_FUSED_RNG(TR)                                                                                  \
  if (R == 16) { if (TL == 1) TBNAV_FUSED(TR, 16, 1, true); else TBNAV_FUSED(TR, 16, 2, true); }               \
  else { if (TL == 1) TBNAV_FUSED(TR, 8, 1, true); else TBNAV_FUSED(TR, 8, 2, true); }
  if (rng) {  // in-kernel noise: instantiated for 8 and 16 rollouts per workgroup (the handle's own choices) — the caller checks
    if (h->dyn == 1) { TBNAV_FUSED_RNG(4); } else if (h->trig == 3) { TBNAV_FUSED_RNG(3); } else { TBNAV_FUSED_RNG(2); }
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

int launch_combine(tbnav_mppi* h, const double* d_records, int G, hipStream_t st, int S = -1, const DirectSrc* direct = nullptr) {
  if (S < 0) S = h->S;
  int tpr = 1;
  while (tpr < G * S && tpr < kWave) tpr <<= 1;
  const int wpb = TBNAV_COMBINE_WAVES;  // waves per workgroup
  const int steps_per_block = wpb * (kWave / tpr);
  const int blocks = (h->T + steps_per_block - 1) / steps_per_block;
  const USrc usrc{h->d_u[h->ucur], h->pending_shift ? 1 : 0, h->uinit[0], h->uinit[1]};
  const DirectSrc ds = direct ? *direct : DirectSrc{nullptr, 0ull, nullptr, nullptr, 0u};
#define TBNAV_COMBINE(KEEP, DIR) do { h->lk_combine[0] = KEEP; h->lk_combine[1] = DIR; hipLaunchKernelGGL((mppi_combine<KEEP, DIR>), dim3(blocks), dim3(wpb * kWave), 0, st, h->T, G, S, lam_of(h), h->p.max_wheel_vel, usrc, \
                                                    d_records, h->d_u[1 - h->ucur], h->d_out, h->publish_next ? h->d_out_host : nullptr, (double)(h->seq + 1), ds); } while (0)
  if (direct) {
    if (G * S > 4 * kWave && G * S <= 8 * kWave) TBNAV_COMBINE(8, true);
    else if (G * S > 2 * kWave && G * S <= 4 * kWave) TBNAV_COMBINE(4, true);
    else TBNAV_COMBINE(2, true);
  } else {
    if (G * S > 4 * kWave && G * S <= 8 * kWave) TBNAV_COMBINE(8, false);
    else if (G * S > 2 * kWave && G * S <= 4 * kWave) TBNAV_COMBINE(4, false);
    else TBNAV_COMBINE(2, false);
  }
#undef TBNAV_COMBINE
  TBNAV_HIP(hipGetLastError());
  ++h->seq;
  if (h->publish_next) h->published = h->seq;
  h->ucur = 1 - h->ucur;       // the freshly written vector is current ...
  h->pending_shift = true;     // ... and its shift is still owed
  return TBNAV_OK;
}

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

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
  }
  ~DeviceGuard() { if (ok && prev >= 0) (void)hipSetDevice(prev); }
};

// counter of rollout 0, step 0 of this handle at tick `tick`: counter(k, i) = tick*T*K_global + (k0 + k)*T + i, so the
// shards of one ensemble draw disjoint perturbations from one seed (and the same ones as the unsharded ensemble)
uint64_t rng_base(const tbnav_mppi* h, uint64_t tick) { return tick * (uint64_t)h->T * h->k_global + h->k0 * (uint64_t)h->T; }

bool pick_noise(tbnav_mppi* h, const double*& d_duL, const double*& d_duR) {
  if (!d_duL && !d_duR) { d_duL = h->d_duL; d_duR = h->d_duR; return true; }
  return d_duL && d_duR;
}

int sharded_tick(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, const uint64_t* seed, uint64_t tick, void* stream);

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

}  // extern "C"
namespace { void direct_teardown(tbnav_mppi* h); }
extern "C" {

void tbnav_mppi_destroy(tbnav_mppi* h) {
#ifdef TBNAV_PHASE_PROF
  {
    unsigned long long tr[2][8][8];
    if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_mtrace), sizeof(tr)) == hipSuccess && tr[0][0][0]) {
      const char* names[2] = {"mppi_rollout_fused, workgroup 5 (entry, noise + u requested, heading scan, x / y scans, J in LDS, barrier, records stored)",
                              "mppi_combine, workgroup 5 (entry, records requested, min, sums)"};
      for (int k = 0; k < 2; ++k) {
        std::fprintf(stderr, "[%s; us]\n", names[k]);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 8; ++w) if (tr[k][w][0] && tr[k][w][0] < t0) t0 = tr[k][w][0];
        for (int w = 0; w < 8; ++w) {
          if (!tr[k][w][0]) continue;
          std::fprintf(stderr, "  wave %d:", w);
          for (int i = 0; i < 7; ++i) std::fprintf(stderr, " %5.2f", tr[k][w][i] ? (double)(tr[k][w][i] - t0) * 0.01 : -1.0);
          std::fprintf(stderr, "\n");
        }
      }
    }
  }
#endif
  if (!h) return;
  DeviceGuard guard(h->device);
  (void)hipFree(h->d_u[0]); (void)hipFree(h->d_u[1]); (void)hipFree(h->d_J); (void)hipFree(h->d_duL); (void)hipFree(h->d_duR);
  (void)hipFree(h->d_total); (void)hipFree(h->d_records_all);
  direct_teardown(h);
  (void)hipFree(h->d_raw); (void)hipFree(h->d_records); (void)hipFree(h->d_records_f); (void)hipFree(h->d_out);
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->tg_exec) (void)hipGraphExecDestroy(h->tg_exec);
  if (h->tg_graph) (void)hipGraphDestroy(h->tg_graph);
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
  if (!(h->fused_rng && (h->fused_r == 8 || h->fused_r == 16) && kSlice % h->fused_r == 0)) {
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
  if (!(h->fused_dev && h->fused_rng && (h->fused_r == 8 || h->fused_r == 16))) {
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
      case 1: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_fused<%d, %d, %d, %s>", k[1], k[2], k[3], k[4] ? "true" : "false"); break;
      case 2: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_scan<%d, %d, %d>", k[1], k[2], k[3]); break;
      case 3: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_prefix<%d>", k[1]); break;
      case 4: snprintf(rollout, (size_t)rollout_cap, "mppi_rollout_cost<%d>", k[1]); break;
      default: rollout[0] = 0;
    }
  }
  if (combine && combine_cap > 0) {
    if (h->lk_combine[0]) snprintf(combine, (size_t)combine_cap, "mppi_combine<%d, %s>", h->lk_combine[0], h->lk_combine[1] ? "true" : "false");
    else combine[0] = 0;
  }
  return TBNAV_OK;
}

int tbnav_mppi_last_controls(tbnav_mppi* h, void* stream, double u_out[2]) {
  if (!h || !u_out) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto exchange_ok = [&]() -> int {  // (the stream has been waited for) a direct exchange that ran out of time left its mark in host memory
    if (!h->direct_on || !h->h_dx_err || !*h->h_dx_err) return TBNAV_OK;
    tbnav::last_hip_error_slot() = "direct exchange: a peer's records did not arrive in time";
    return TBNAV_ERR_HIP;
  };
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
                     h->T, h->K, seed, rng_base(h, tick), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var), h->d_duL,
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
  if (!(h->fused_rng && (h->fused_r == 8 || h->fused_r == 16))) {
    const int rc = tbnav_mppi_sample_noise(h, seed, tick, stream);
    return rc != TBNAV_OK ? rc : tbnav_mppi_enqueue_dev(h, x0, nullptr, nullptr, stream);
  }
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const RngArgs g{seed, rng_base(h, tick), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var)};
  const int rc = launch_fused(h, x0, h->d_duL, h->d_duR, st, &g);
  return rc != TBNAV_OK ? rc : launch_combine(h, h->d_records_f, 1, st, h->fused_S);
}

int tbnav_mppi_enqueue_rng_batch(tbnav_mppi* h, const double* x0s, int32_t x0_stride, uint64_t seed, uint64_t first_tick, int32_t n_ticks,
                                 void* stream) {
  if (!h || !x0s || n_ticks < 0 || (x0_stride != 0 && x0_stride < 3)) return TBNAV_ERR_INVALID_ARG;
  int32_t i = 0;
  constexpr int kGraphTicks = 100;  // (even: the controls' double buffer is back where it was after a chunk)
  hipStream_t st = static_cast<hipStream_t>(stream);
  // (only where the tick is short enough for the launches themselves to matter: K = 1024: 8.25 -> 8.15 us per tick on a fast host, 8.9 -> 8.3
  //  on a slower one; from K = 2048 up the device is the bound and the replay is 1-3 % slower than plain launches)
  if (h->graph_on && !h->comm && x0_stride == 0 && st != nullptr && h->fused_rng && h->fused_r == 8 && h->K <= 1536 && n_ticks >= 2) {
    DeviceGuard guard(h->device);
    // (the graph is built by the first batch call that could use one, however short — a warm-up call, typically — so that a
    //  later long call does not pay the ~1 ms of capture + instantiation)
    // the first tick after set_controls / set_initial_controls reads the vector unshifted: keep it out of the graph
    if (!h->pending_shift) { const int rc = tbnav_mppi_enqueue_rng(h, x0s, seed, first_tick, stream); if (rc != TBNAV_OK) return rc; ++i; }
    const bool usable = h->tg_exec && h->tg_epoch == h->cfg_epoch && h->tg_seed == seed && h->tg_stream == st &&
                        std::memcmp(h->tg_x0, x0s, sizeof h->tg_x0) == 0;
    // (the graph has the controls' double buffer baked in as it stood at capture: on the other parity ONE plain tick brings it
    //  back — a rebuild would cost ~0.5 ms inside the caller's batch.  Shorter chunks were measured and dropped: a 10-tick graph
    //  replays at 9.9-10.2 us per tick against 8.9 for plain launches — a replay's fixed cost needs ~100 ticks to amortise)
    if (usable && h->tg_ucur != h->ucur && n_ticks - i > kGraphTicks) {
      const int rc = tbnav_mppi_enqueue_rng(h, x0s, seed, first_tick + (uint64_t)i, stream);
      if (rc != TBNAV_OK) return rc;
      ++i;
    }
    const bool same = usable && h->tg_ucur == h->ucur;
    if (!same && (!usable || n_ticks - i >= kGraphTicks)) {
      // (another stream may not have run the previous replay's tick-advance node yet: what the device word holds is unknown)
      h->tg_dev_tick = ~0ull;
      if (h->tg_exec) { (void)hipGraphExecDestroy(h->tg_exec); h->tg_exec = nullptr; }
      if (h->tg_graph) { (void)hipGraphDestroy(h->tg_graph); h->tg_graph = nullptr; }
      if (!h->d_tick0 && hipMalloc((void**)&h->d_tick0, sizeof(uint64_t)) != hipSuccess) { h->d_tick0 = nullptr; h->graph_on = false; }
      if (h->graph_on && hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) == hipSuccess) {
        const int ucur0 = h->ucur; const uint64_t seq0 = h->seq;
        int rc = TBNAV_OK;
        for (int t = 0; t < kGraphTicks && rc == TBNAV_OK; ++t) {
          RngArgs g{seed, rng_base(h, (uint64_t)t), std::sqrt(h->p.ul_var), std::sqrt(h->p.ur_var)};
          g.tick0 = h->d_tick0; g.per_tick = (uint64_t)h->T * h->k_global;
          rc = launch_fused(h, x0s, h->d_duL, h->d_duR, st, &g);
          if (rc == TBNAV_OK) rc = launch_combine(h, h->d_records_f, 1, st, h->fused_S);
        }
        if (rc == TBNAV_OK) { hipLaunchKernelGGL(mppi_tick_advance, dim3(1), dim3(1), 0, st, h->d_tick0, (uint64_t)kGraphTicks); if (hipGetLastError() != hipSuccess) rc = TBNAV_ERR_HIP; }
        hipGraph_t gr = nullptr;
        const hipError_t e_end = hipStreamEndCapture(st, &gr);
        h->ucur = ucur0; h->seq = seq0;  // nothing ran: the host-side state goes back
        if (rc == TBNAV_OK && e_end == hipSuccess && gr && hipGraphInstantiate(&h->tg_exec, gr, nullptr, nullptr, 0) == hipSuccess) {
          h->tg_graph = gr; h->tg_seed = seed; h->tg_stream = st; h->tg_ucur = h->ucur; h->tg_epoch = h->cfg_epoch; std::memcpy(h->tg_x0, x0s, sizeof h->tg_x0);
        } else {
          if (gr) (void)hipGraphDestroy(gr);
          h->tg_exec = nullptr; h->graph_on = false; (void)hipGetLastError();  // plain launches from here on
        }
      } else h->graph_on = false;
    }
    while (h->tg_exec && n_ticks - i >= kGraphTicks) {
      const uint64_t t0 = first_tick + (uint64_t)i;
      // (consecutive chunks need no copy: the replay's last node has advanced the device word)
      if (t0 != h->tg_dev_tick) {
        h->tg_dev_tick = ~0ull;
        hipLaunchKernelGGL(mppi_tick_set, dim3(1), dim3(1), 0, st, h->d_tick0, t0);
        TBNAV_HIP(hipGetLastError());
      }
      { const hipError_t eg = hipGraphLaunch(h->tg_exec, st); if (eg != hipSuccess) { h->tg_dev_tick = ~0ull; TBNAV_HIP(eg); } }
      h->tg_dev_tick = t0 + (uint64_t)kGraphTicks;
      h->seq += kGraphTicks;  // ucur: unchanged after an even number of ticks; the shift stays owed
      h->graph_ticks += kGraphTicks;
      i += kGraphTicks;
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
  TBNAV_HIP(hipMalloc((void**)&dx, sizeof(double) * n));
  TBNAV_HIP(hipMalloc((void**)&ds, sizeof(double) * n));
  TBNAV_HIP(hipMalloc((void**)&dc, sizeof(double) * n));
  hipError_t e = hipMemcpy(dx, x_host, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) { hipLaunchKernelGGL(mppi_debug_sincos, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, dx, ds, dc); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpy(sin_host, ds, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(cos_host, dc, sizeof(double) * n, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(ds); (void)hipFree(dc);
  TBNAV_HIP(e);
  return TBNAV_OK;
}

int tbnav_mppi_debug_div_lambda(const double* x_host, int32_t n, double lambda, double* out_host, int32_t* used_reciprocal) {
  if (!x_host || !out_host || n <= 0) return TBNAV_ERR_INVALID_ARG;
  tbnav_mppi tmp;
  tmp.p.lambda = lambda;
  const Lam lam = lam_of(&tmp);
  if (used_reciprocal) *used_reciprocal = lam.inv != 0.0;
  double *dx = nullptr, *dy = nullptr;
  TBNAV_HIP(hipMalloc((void**)&dx, sizeof(double) * n));
  TBNAV_HIP(hipMalloc((void**)&dy, sizeof(double) * n));
  hipError_t e = hipMemcpy(dx, x_host, sizeof(double) * n, hipMemcpyHostToDevice);
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


// ---- sharded ensembles behind the same entry points ----------------------------------------------
}  // extern "C"
namespace { int direct_setup(tbnav_mppi* h); }
extern "C" {
int tbnav_mppi_attach_comm(tbnav_mppi* h, tbnav_comm* comm) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipDeviceSynchronize());
  if (h->comm && h->direct_on && tbnav::comm_is_multiprocess(h->comm)) {
    // detaching from a direct exchange is COLLECTIVE, like attaching: a peer's publish kernel may still be storing into this
    // rank's buffer (it can be one tick ahead) — every rank synchronises its device (above), then all meet here, and only then
    // are the peers' mappings closed and the buffer freed.  A communicator that fails here is reported; the teardown still runs.
    int mine = 1;
    std::vector<int> all((size_t)tbnav::comm_size(h->comm), 0);
    (void)tbnav::comm_all_gather_host(h->comm, &mine, all.data(), sizeof(int));
  }
  direct_teardown(h);
  (void)hipFree(h->d_records_all);
  h->d_records_all = nullptr;
  h->comm = nullptr;
  ++h->cfg_epoch;
  if (!comm) { h->k0 = 0; h->k_global = (uint64_t)h->K; return TBNAV_OK; }
  if (tbnav_comm_device(comm) != h->device) return TBNAV_ERR_INVALID_ARG;
  const int P = tbnav::comm_size(comm), r = tbnav::comm_rank(comm);
  TBNAV_HIP(hipMalloc((void**)&h->d_records_all, sizeof(double) * (size_t)P * h->T * h->S * TBNAV_MPPI_REC));
  h->comm = comm;
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
bool direct_alloc(tbnav_mppi* h) {
  const int P = tbnav::comm_size(h->comm), n = h->T * h->S * TBNAV_MPPI_REC;
  const size_t words = (size_t)2 * P * 2 * n;
  const bool ok = hipExtMallocWithFlags((void**)&h->d_dx, sizeof(unsigned long long) * words, hipDeviceMallocFinegrained) == hipSuccess &&
                  hipMemset(h->d_dx, 0, sizeof(unsigned long long) * words) == hipSuccess &&
                  hipHostMalloc((void**)&h->h_dx_err, sizeof(int), hipHostMallocMapped) == hipSuccess &&
                  hipHostGetDevicePointer((void**)&h->d_dx_err, h->h_dx_err, 0) == hipSuccess &&
                  hipMalloc((void**)&h->d_dx_peers, sizeof(unsigned long long*) * P) == hipSuccess &&
                  hipMalloc((void**)&h->d_dx_dead, sizeof(int)) == hipSuccess && hipMemset(h->d_dx_dead, 0, sizeof(int)) == hipSuccess;
  if (h->h_dx_err) *h->h_dx_err = 0;
  return ok;
}
double direct_pattern(int q, int it, int j) {  // the self-tests' records: every bit in play
  unsigned long long z = 0x9E3779B97F4A7C15ull * (unsigned long long)(q * 1000003 + it * 7919 + j + 1);
  z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
  double d; std::memcpy(&d, &z, sizeof d); return d;
}

void direct_teardown(tbnav_mppi* h) {
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
  (void)hipFree(h->d_dx_dead); h->d_dx_dead = nullptr;
  if (h->h_dx_err) (void)hipHostFree(h->h_dx_err);
  h->h_dx_err = nullptr; h->d_dx_err = nullptr;
}

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

int sharded_tick(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, const uint64_t* seed, uint64_t tick, void* stream) {
  if (h->direct_on) {
    if (h->h_dx_err && *h->h_dx_err) {  // an earlier tick's bound expired: say so now, not only at the next last_controls
      tbnav::last_hip_error_slot() = "direct exchange: a peer's records did not arrive in time (latched; re-attach the communicator)";
      return TBNAV_ERR_HIP;
    }
    const int rc = direct_partials_and_publish(h, x0, d_duL, d_duR, seed, tick, stream);
    if (rc != TBNAV_OK) return rc;
    DeviceGuard guard(h->device);
    return direct_combine(h, static_cast<hipStream_t>(stream));
  }
  const int rc_local = sharded_partials(h, x0, d_duL, d_duR, seed, tick, stream);
  DeviceGuard guard(h->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t block = sizeof(double) * (size_t)h->T * h->S * TBNAV_MPPI_REC;
  void* mine = reinterpret_cast<char*>(h->d_records_all) + (size_t)tbnav::comm_rank(h->comm) * block;
  if (rc_local != TBNAV_OK) {
    // This rank's rollouts failed.  Its peers are about to enter (or sit in) an all-gather that has no timeout: JOIN it — with
    // records that poison every rank's combine (cost NaN, count 1: the soft-min and with it the controls come out NaN on every
    // rank, which every caller checks) instead of leaving the others hanging or, worse, quietly combining without this shard —
    // and report the failure here.  (Round-3 advisor finding: the early return left the peers in ncclAllGather for good.)
    std::vector<double> bad((size_t)h->T * h->S * TBNAV_MPPI_REC, std::numeric_limits<double>::quiet_NaN());
    for (size_t q = 6; q < bad.size(); q += TBNAV_MPPI_REC) bad[q] = 1.0;
    (void)hipMemcpyAsync(mine, bad.data(), block, hipMemcpyHostToDevice, st);
    (void)hipStreamSynchronize(st);   // (`bad` is pageable host memory: do not let it go out of scope under the copy)
    (void)hipGetLastError();
  }
  const void* send = mine;
  void* recv = h->d_records_all;
  const int rc = tbnav::comm_all_gather(1, &h->comm, &send, &recv, block, &st);
  if (rc_local != TBNAV_OK) return rc_local;
  if (rc != TBNAV_OK) return rc;
  return launch_combine(h, h->d_records_all, tbnav::comm_size(h->comm), st);
}
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
  if (rc == TBNAV_OK && h->direct_on && *h->h_dx_err) { tbnav::last_hip_error_slot() = "direct exchange: a peer's records did not arrive in time"; rc = TBNAV_ERR_HIP; }
  *us_per_round = rc == TBNAV_OK ? (double)ms * 1e3 / rounds : 0.0;
  return rc;
}

}  // extern "C"

// One process driving several GPUs: the whole ensemble behind one object (what controller::MPPI built with n_gpus > 1 holds).
struct tbnav_mppi_group {
  int n = 0;
  std::vector<tbnav_mppi*> m;
  std::vector<tbnav_comm*> c;
  std::vector<hipStream_t> st;
  std::vector<double*> d_raw;  // per member: staging of its slice of host-order noise
  int K_global = 0;
};

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
    if (h->direct_on && h->h_dx_err && *h->h_dx_err) { tbnav::last_hip_error_slot() = "direct exchange: a member's records did not arrive in time"; return TBNAV_ERR_HIP; }
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
