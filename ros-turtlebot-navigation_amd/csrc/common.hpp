// common.hpp — shared helpers for the HIP translation units (error capture, wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

#include "tbnav_status.h"

namespace tbnav {

// Thread-local text of the last failing HIP call; surfaced through tbnav_last_hip_error().
std::string& last_hip_error_slot();

inline int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
  last_hip_error_slot() = buf;
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
             ? TBNAV_ERR_NO_DEVICE
             : TBNAV_ERR_HIP;
}

#define TBNAV_HIP(call)                                                       \
  do {                                                                        \
    hipError_t tbnav_e_ = (call);                                             \
    if (tbnav_e_ != hipSuccess) return ::tbnav::hip_fail(tbnav_e_, #call, __FILE__, __LINE__); \
  } while (0)

// ---- wave64 reductions (gfx950 wavefront = 64 lanes) -------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// ---- wave64 scans on the DPP network (gfx9 row_shr / row_bcast; no LDS crossbar round trips) ---------
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl=false): a lane whose source is invalid or masked
// keeps `old`, which is the additive identity here.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_or_zero(double src) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the lanes of a wave (lane i gets v_0 + ... + v_i)
__device__ __forceinline__ double wave_scan_incl(double v, int /*lane*/) {
  double out = v;
  out += dpp_or_zero<0x111, 0xf, 0xf>(v);    // row_shr:1
  out += dpp_or_zero<0x112, 0xf, 0xf>(v);    // row_shr:2
  out += dpp_or_zero<0x113, 0xf, 0xf>(v);    // row_shr:3   -> sums of up to four consecutive lanes
  out += dpp_or_zero<0x114, 0xf, 0xe>(out);  // row_shr:4, banks 1-3
  out += dpp_or_zero<0x118, 0xf, 0xc>(out);  // row_shr:8, banks 2-3 -> inclusive within each row of 16
  out += dpp_or_zero<0x142, 0xa, 0xf>(out);  // row_bcast:15 into rows 1 and 3
  out += dpp_or_zero<0x143, 0xc, 0xf>(out);  // row_bcast:31 into rows 2 and 3
  return out;
}
// whole-wave sum / min on the same network; the result is broadcast from lane 63
__device__ __forceinline__ double lane63(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_sum_dpp(double v) { return lane63(wave_scan_incl(v, 0)); }
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_or_inf(double src) {
  const double inf = __builtin_huge_val();
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(inf), __double2loint(src), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(inf), __double2hiint(src), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min_dpp(double v) {
  double out = v;
  out = fmin(out, dpp_or_inf<0x111, 0xf, 0xf>(v));
  out = fmin(out, dpp_or_inf<0x112, 0xf, 0xf>(v));
  out = fmin(out, dpp_or_inf<0x113, 0xf, 0xf>(v));
  out = fmin(out, dpp_or_inf<0x114, 0xf, 0xe>(out));
  out = fmin(out, dpp_or_inf<0x118, 0xf, 0xc>(out));
  out = fmin(out, dpp_or_inf<0x142, 0xa, 0xf>(out));
  out = fmin(out, dpp_or_inf<0x143, 0xc, 0xf>(out));
  return lane63(out);
}
// The partner of a butterfly step inside groups of 2 / 4 / 8 / 16 consecutive lanes, on the DPP network: lane ^ 1 and
// lane ^ 2 are quad permutations; once a quad's lanes agree, the mirror image within 8 (row_half_mirror) or 16
// (row_mirror) lanes lies in the other quad / half, which is all a commutative reduction needs.
template <int STEP>
__device__ __forceinline__ double dpp_group_partner(double v) {
  constexpr int ctrl = STEP == 1 ? 0xB1 : STEP == 2 ? 0x4E : STEP == 4 ? 0x141 : 0x140;  // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// all-lanes reduction over groups of R consecutive lanes (R = 2, 4, 8, 16), steps in the order 1, 2, 4, 8
template <int R, class Op>
__device__ __forceinline__ double group_reduce_dpp(double v, Op op) {
  static_assert(R == 2 || R == 4 || R == 8 || R == 16, "groups of 2..16 lanes");
  v = op(v, dpp_group_partner<1>(v));
  if constexpr (R >= 4) v = op(v, dpp_group_partner<2>(v));
  if constexpr (R >= 8) v = op(v, dpp_group_partner<4>(v));
  if constexpr (R >= 16) v = op(v, dpp_group_partner<8>(v));
  return v;
}
// inclusive suffix sum (lane i gets v_i + ... + v_63): mirror the wave, scan, mirror back
__device__ __forceinline__ double readlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_scan_incl_rev(double v, int lane) {
  // the mirror image of wave_scan_incl inside each row of 16 lanes (row_shl), then the totals of the later rows — read
  // from their first lanes — are added; no LDS permute (two dependent ones before)
  double out = v;
  out += dpp_or_zero<0x101, 0xf, 0xf>(v);    // row_shl:1
  out += dpp_or_zero<0x102, 0xf, 0xf>(v);    // row_shl:2
  out += dpp_or_zero<0x103, 0xf, 0xf>(v);    // row_shl:3
  out += dpp_or_zero<0x104, 0xf, 0x7>(out);  // row_shl:4, banks 0-2
  out += dpp_or_zero<0x108, 0xf, 0x3>(out);  // row_shl:8, banks 0-1 -> inclusive suffix within each row of 16
  const double t1 = readlane_d(out, 16), t2 = readlane_d(out, 32), t3 = readlane_d(out, 48);
  const int row = lane >> 4;
  const double later = row == 0 ? (t1 + t2) + t3 : row == 1 ? t2 + t3 : row == 2 ? t3 : 0.0;
  return out + later;
}

}  // namespace tbnav
