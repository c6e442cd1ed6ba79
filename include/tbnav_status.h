/* tbnav_status.h — status codes shared by the MPPI and RBPF C-ABI entry points.
 *
 * The reference reports errors by throwing (std::invalid_argument, grid_mapper.cpp:22,701,819,856;
 * particle_filter.cpp:579; diff_drive.cpp:72) or exit(EXIT_FAILURE) (rk4.cpp:51-55).  Nothing may
 * unwind across a C boundary, so every entry point returns one of these codes and the C++ class
 * shims (ros-turtlebot-navigation_amd/host/) turn them back into the reference's exception types.
 */
#ifndef TBNAV_STATUS_H
#define TBNAV_STATUS_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum tbnav_status {
  TBNAV_OK = 0,
  TBNAV_ERR_INVALID_ARG = 1,     /* null pointer, non-positive size, bad parameter                  */
  TBNAV_ERR_NO_DEVICE = 2,       /* no HIP device / HIP runtime call failed: never a CPU fallback   */
  TBNAV_ERR_HIP = 3,             /* a HIP API call failed after the device was opened               */
  TBNAV_ERR_OUT_OF_WORLD = 4,    /* "... position NOT in the bounds of the world" (grid_mapper.cpp:856,861) */
  TBNAV_ERR_ETA_ZERO = 5,        /* "eta is 0" (particle_filter.cpp:579)                            */
  TBNAV_ERR_PDF_VARIANCE = 6,    /* "Variance in pdfNormal is 0" (grid_mapper.cpp:22)               */
  TBNAV_ERR_BRESENHAM = 7,       /* "Bresenham's Line Algorithm" (grid_mapper.cpp:701)              */
  TBNAV_ERR_UNSUPPORTED = 8,     /* configuration outside what the device path implements           */
  TBNAV_ERR_POOL_EXHAUSTED = 9   /* RBPF: no free log-odds tile left in the handle's pool (tbnav_rbpf_create_pool) */
} tbnav_status;

/* Human-readable text for a status code; for OUT_OF_WORLD/ETA_ZERO/PDF_VARIANCE/BRESENHAM it is the
 * exact what() string the reference throws, so the shims can re-throw it verbatim. */
const char* tbnav_status_string(int status);

/* Last HIP error string recorded on this thread by a failing entry point ("" if none). */
const char* tbnav_last_hip_error(void);

/* Number of visible HIP devices (0 when there is no GPU or no driver); never fails. */
int tbnav_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* TBNAV_STATUS_H */
