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
// inclusive suffix sum (lane i gets v_i + ... + v_63): mirror the wave, scan, mirror back
__device__ __forceinline__ double wave_scan_incl_rev(double v, int lane) {
  const double m = __shfl(v, 63 - lane, 64);
  return __shfl(wave_scan_incl(m, lane), 63 - lane, 64);
}

}  // namespace tbnav
