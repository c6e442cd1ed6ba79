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

}  // namespace tbnav
