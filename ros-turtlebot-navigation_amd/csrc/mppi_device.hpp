// mppi_device.hpp — what the MPPI kernel files (mppi_rollout.hip: the four rollout families; mppi_softmin.hip: soft-min records,
// combine, exchange words, noise) and the host side (mppi.hip: handle, launches, C-ABI; mppi_sharded.hip: exchange set-up,
// sharded tick, groups) share: the launch-argument structs, the rollout arithmetic (device trig, RK4 / exact-arc steps, losses),
// the Philox noise source and the declaration of every kernel.  Kernels are defined in their family's file and launched from the
// host files; the template kernels are explicitly instantiated where they are defined.  All four are compiled with
// -ffp-contract=fast-honor-pragmas (csrc/Makefile says why).
// Reference: controller/src/controller/mppi.cpp:72-140, rk4.cpp:49-115, controller/include/controller/mppi.hpp:41-105.
#ifndef TBNAV_MPPI_DEVICE_HPP
#define TBNAV_MPPI_DEVICE_HPP
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "common.hpp"
#include "tbnav_mppi.h"

namespace tbnav_mk {

constexpr int kWave = 64;
constexpr int kSliceThreads = 256;
constexpr int kSliceItems = 8;
constexpr int kSlice = kSliceThreads * kSliceItems;  // rollouts per K-slice record
constexpr int kMaxLdsBytes = 160 * 1024;
constexpr int kGroup = 4;        // steps per group of the one-lane-per-rollout kernels (independent trig chains)
constexpr int kRoundGroups = 3;  // groups per round of mppi_rollout_prefix
#ifndef TBNAV_COMBINE_WAVES
#define TBNAV_COMBINE_WAVES 1  // one wave per workgroup: the groups spread over as many CUs as there are time steps (K = 1024 tick 8.9 -> 8.4 us against four waves)
#endif

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
// The theorem assumes that nothing under- or overflows on the way (round-4 advisor finding): the statement holds for x = 0 and
// 2^-900 <= |x| <= 2^900 (lambda and 1 / lambda are normal, so q and the residual stay representable there).  Outside that range
// the result may differ from x / lambda in its last bit (|x| < 2^-900: the residual underflows) or be +-inf where the quotient is
// merely huge (x / lambda finite, x * (1 / lambda) not) — and the only consumer is exp(): of an argument that small it is exactly
// 1, of one that large and negative exactly 0, whichever way it was divided.  tests/test_mppi_gpu.py holds the bits inside the
// range (subnormal quotients, near-product arguments included) and exp() of the result outside it.  (A wave-voted fallback to
// the division outside the range was measured: +0.2 us on the 7.75 us tick, for no observable difference.)
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
// WIDE = false (the narrow sampler, TBNAV_MPPI_OPT_SAMPLER = 0; the default up to round 5): Box-Muller on the fp32 transcendental units (v_log_f32,
// v_sin_f32 / v_cos_f32 take their argument in turns): a handful of instructions instead of ~130 fp64 ones for log +
// sincospi + sqrt.  The perturbations are random numbers, not parity quantities: 24-bit uniforms give normals on a 2^-24 grid
// out to 5.9 sigma, which is all a sampling controller can use (the reference's own sampler is not reproducible run to run
// either, utilities.cpp:14).
// WIDE = true (TBNAV_MPPI_OPT_SAMPLER = 1, the default): what the reference's std::normal_distribution<double> is in width
// (utilities.cpp:20-24): the same Philox counter, all 128 bits of it — two uniforms of 52 random bits + the half-ulp centring
// ((n + 0.5) * 2^-52 is exact in a double: 53 significant bits, never 0 or 1), fp64 log / sqrt / sincospi: normals out to
// sqrt(2 * 53 * ln 2) = 8.57 sigma on a grid finer than 2^-52.
template <bool WIDE>
__device__ __forceinline__ void device_noise(const RngArgs& g, int T, int i, int k, double& dl, double& dr) {
  uint32_t r[4];
  philox4x32_10(g.base + (uint64_t)k * T + i, g.seed, r);
  if constexpr (WIDE) {
    const uint64_t b1 = ((uint64_t)r[0] << 20) | (uint64_t)(r[1] >> 12);   // 52 bits
    const uint64_t b2 = ((uint64_t)r[2] << 20) | (uint64_t)(r[3] >> 12);
    const double u1 = ((double)b1 + 0.5) * 0x1.0p-52;   // (0, 1)
    const double u2 = ((double)b2 + 0.5) * 0x1.0p-52;   // (0, 1) half turns x 2
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    dl = g.sig_l * (rad * cs);
    dr = g.sig_r * (rad * sn);
  } else {
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * 0x1.0p-24f;  // (0, 1)
    const float u2 = ((float)(r[1] >> 8) + 0.5f) * 0x1.0p-24f;  // [0, 1) turns
    const float rad = __builtin_sqrtf(-2.0f * 0.69314718056f * __builtin_amdgcn_logf(u1));  // v_log_f32 is log2
    dl = g.sig_l * (double)(rad * __builtin_amdgcn_cosf(u2));
    dr = g.sig_r * (double)(rad * __builtin_amdgcn_sinf(u2));
  }
}

// (the direct exchange's sending side, when the records are produced by mppi_merge_records: see mppi_direct_publish)
struct DirectPub { unsigned long long* const* peers; int me, P, parity; unsigned int seq; int only_self; };
// The direct exchange's receiving side (see mppi_direct_publish): a record field is two tagged 8-byte words in this rank's own
// fine-grained buffer; poll them until both carry the tick's sequence number (bounded: an error word and zeros after `budget`
// ticks of the 100 MHz clock).
constexpr double kPoison = -1.0;   // field 6 (the count) of a record that stands for "this rank's rollouts failed": see mppi_combine
struct DirectSrc { const unsigned long long* w0; unsigned long long budget; int* err; int* err_dev; unsigned int seq; };  // err: mapped host word (bit 0: a peer's words did not arrive in time; bit 1: a poisoned record was seen); err_dev: its device twin (what later ticks look at).  The all-gather exchange passes only the two error words.

// ---- kernels (defined in mppi_rollout.hip / mppi_softmin.hip) -----------------------------------------------------------------
// rollout families (mppi_rollout.hip)
template <int TRIG>
__global__ __launch_bounds__(kWave) void mppi_rollout_cost(RolloutArgs a, const double* __restrict__ duL, const double* __restrict__ duR, USrc u,
                                                           double* __restrict__ J);
template <int RG>
__global__ __launch_bounds__(kWave) void mppi_rollout_prefix(RolloutArgs a_in, const double* __restrict__ duL, const double* __restrict__ duR, USrc u,
                                                             double* __restrict__ J, double* __restrict__ total);
template <int TRIG, int TC, int MAXW>
__global__ __launch_bounds__(kWave * MAXW) void mppi_rollout_scan(RolloutArgs a, const double* __restrict__ duL, const double* __restrict__ duR, USrc u,
                                                                  double* __restrict__ J);
template <int TRIG, int R, int TL, int RNG>
__global__ __launch_bounds__(kWave * R) void mppi_rollout_fused(RolloutArgs a, const double* __restrict__ duL, const double* __restrict__ duR, USrc u, Lam lam,
                                                                double* __restrict__ J /* NULL: not kept */, double* __restrict__ records, int S,
                                                                RngArgs rng);
// soft-min, exchange words, noise (mppi_softmin.hip)
__global__ __launch_bounds__(kSliceThreads) void mppi_partials(int T, int K, int S, Lam lam, const double* __restrict__ J, const double* __restrict__ duL,
                                                               const double* __restrict__ duR, double* __restrict__ records, int prefix_rows,
                                                               const double* __restrict__ total);
__global__ __launch_bounds__(kWave) void mppi_merge_records(int T, int Sf, int per_slice, int S, Lam lam, const double* __restrict__ fine,
                                                            double* __restrict__ records, DirectPub pub);
template <int kKeep, int MODE>   // MODE 0: one group of records (every single-GPU tick); 1: records of several ranks through an all-gather; 2: the direct exchange
__global__ __launch_bounds__(256) void mppi_combine(int T, int G, int S, Lam lam, double umax, USrc u, const double* __restrict__ records,
                                                    double* __restrict__ u_out, double* __restrict__ out, double* __restrict__ out_host, double seq,
                                                    DirectSrc ds);
__global__ __launch_bounds__(256) void mppi_combine_wide(int T, int R, Lam lam, double umax, USrc u, const double* __restrict__ records, double* __restrict__ u_out,
                                                         double* __restrict__ out, double* __restrict__ out_host, double seq);
__global__ __launch_bounds__(256) void mppi_direct_publish(const double* __restrict__ mine, int n, unsigned long long* const* __restrict__ peers, int me, int P,
                                                           int parity, unsigned int seq, int only_self);
__global__ __launch_bounds__(256) void mppi_direct_collect(unsigned long long* __restrict__ words, int n, int P, int parity, unsigned int seq,
                                                           double* __restrict__ out, int* __restrict__ err, unsigned long long budget_ticks);
__global__ void mppi_sample_noise(int T, int K, uint64_t seed, uint64_t base, double sig_l, double sig_r, int wide, double* __restrict__ duL,
                                  double* __restrict__ duR);
__global__ void mppi_unpack_noise(int T, int K, const double* __restrict__ raw, double* __restrict__ duL, double* __restrict__ duR);
__global__ void mppi_tick_advance(uint64_t* __restrict__ tick0, uint64_t n);
__global__ void mppi_tick_set(uint64_t* __restrict__ tick0, uint64_t v);
__global__ void mppi_debug_div_lambda(int n, const double* __restrict__ x, Lam lam, double* __restrict__ out);
__global__ void mppi_debug_sincos(int n, const double* __restrict__ x, double* __restrict__ sn, double* __restrict__ cs);

}  // namespace tbnav_mk
#endif
