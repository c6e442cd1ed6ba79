// status.hip — status strings and HIP error capture shared by every C-ABI entry point
// (include/tbnav_status.h).
#include <hip/hip_runtime.h>

#include <string>

#include "common.hpp"
#include "tbnav_status.h"

namespace tbnav {
std::string& last_hip_error_slot() {
  static thread_local std::string slot;
  return slot;
}
}  // namespace tbnav

extern "C" {

const char* tbnav_status_string(int status) {
  switch (status) {
    case TBNAV_OK: return "ok";
    case TBNAV_ERR_INVALID_ARG: return "invalid argument";
    case TBNAV_ERR_NO_DEVICE: return "no HIP device available (the HIP path has no CPU fallback)";
    case TBNAV_ERR_HIP: return "HIP runtime error";
    // The four strings below are the reference's what() texts, verbatim, so that the C++ shims can
    // re-throw std::invalid_argument with the same message.
    case TBNAV_ERR_OUT_OF_WORLD: return "X position NOT in the bounds of the world";  // grid_mapper.cpp:856
    case TBNAV_ERR_ETA_ZERO: return "eta is 0";                                       // particle_filter.cpp:579
    case TBNAV_ERR_PDF_VARIANCE: return "Variance in pdfNormal is 0";                 // grid_mapper.cpp:22
    case TBNAV_ERR_BRESENHAM: return "Bresenham's Line Algorithm";                    // grid_mapper.cpp:701
    case TBNAV_ERR_UNSUPPORTED: return "configuration not supported by the device path";
    case TBNAV_ERR_POOL_EXHAUSTED: return "log-odds tile pool exhausted";
    default: return "unknown status";
  }
}

const char* tbnav_last_hip_error(void) { return tbnav::last_hip_error_slot().c_str(); }

int tbnav_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
