/* tbnav_rbpf.h — C-ABI of the MI355X Rao-Blackwellized particle-filter scan update.
 *
 * Drop-in boundary for bmapping::ParticleFilter (reference
 * bmapping/include/bmapping/particle_filter.hpp:88-233, bmapping/src/bmapping/particle_filter.cpp:
 * 67-612) together with the per-particle bmapping::GridMapper it owns
 * (bmapping/include/bmapping/grid_mapper.hpp:117-246, bmapping/src/bmapping/grid_mapper.cpp).
 * Plain pointers and sizes; the handle owns all device memory; one handle per filter; not
 * thread-safe (neither is the reference: process-global RNG, particle_filter.cpp:17-22).
 *
 * Poses are (theta, x, y) everywhere, like rigid2d::Pose and the filter's Vector3d
 * (particle_filter.cpp:132-133).
 *
 * What stays on the host, exactly where the reference has it:
 *  - ICP (bmapping::ScanAlignment over PCL, cloud_alignment.cpp:37-223) runs ONCE per scan before
 *    the particle loop (particle_filter.cpp:146-153); its result (icp_ok, T_icp) is an argument.
 *  - the standard-normal draws (particle_filter.cpp:25-34, libstdc++ mt19937_64 +
 *    normal_distribution, a fresh distribution per draw) are an argument, in the reference's draw
 *    order, so results are reproducible against the CPU path.
 *
 * Device data layout (per handle; N particles, G = xsize*ysize cells, cell index
 * idx = i*xsize + j with i the x-cell, grid_mapper.cpp:890-898; square maps only, as the reference):
 *   pose, prev_pose : [N][3] f64        weight : [N] f64
 *   log_odds        : [N][G] f64        (Cell::log_odds, grid_mapper.hpp:67; beam-ordered adds)
 *   dist_code       : [N][G] u16        squared distance IN CELLS to the nearest occupied cell;
 *                     occ_dist = sqrt((double)code) * resolution is bit-identical to the reference's
 *                     distances_[di][dj] * resolution_ (grid_mapper.cpp:263,318); 0xFFFF = never
 *                     reached = max_occ_dist_ 10.0 (grid_mapper.cpp:49,58)
 *   occupancy       : [N][xsize][ceil(ysize/64)] u64 bitmap of cells with prob >= 0.90, decided in
 *                     log-odds space against a cut-off found on the host with glibc at create time
 *                     (SURVEY.md hard part 2: prob(log 9) == 0.9 exactly with glibc)
 */
#ifndef TBNAV_RBPF_H
#define TBNAV_RBPF_H

#include <stdint.h>
#include "tbnav_status.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Constructor arguments of bmapping::ParticleFilter (particle_filter.hpp:112-130), of the
 * bmapping::LaserProperties it is built with (sensor_model.hpp:63-66), of the prototype
 * bmapping::GridMapper (grid_mapper.hpp:121-122) and the robot->laser transform, flattened. */
typedef struct tbnav_rbpf_params {
  int32_t num_particles;
  int32_t num_samples_mode;            /* k                                               */
  double srr, srt, str_, stt;          /* odometry-model noise, particle_filter.cpp:388-391 */
  double motion_noise[3];              /* variances (theta, x, y)                         */
  double sample_range[3];              /* variances (theta, x, y) round the ICP mode      */
  double scan_likelihood_min, scan_likelihood_max;
  double pose_likelihood_min, pose_likelihood_max;
  float beam_min, beam_max, beam_delta, range_min, range_max; /* radians / metres, float as the reference */
  int32_t device;                      /* HIP device ordinal, -1 = current                */
  double z_hit, z_short, z_max, z_rand, sigma_hit;
  double Trs[3];                       /* robot -> laser (theta, x, y)                    */
  double resolution, xmin, xmax, ymin, ymax;
  double pose0[3];                     /* initial pose (theta, x, y)                      */
} tbnav_rbpf_params;

typedef struct tbnav_rbpf_stats {
  double sum_w;          /* sum of weights before normalisation (particle_filter.cpp:446-450) */
  double sq_sum;         /* normal_sqrd_sum_                                                  */
  int32_t neff;          /* (int)(1.0 / sq_sum) — the reference prints "Neff: <n>"            */
  int32_t resampled;     /* 1 if lowVarianceResampling ran — the reference prints "Resampling" */
  int32_t status;        /* TBNAV_OK or the status the reference would have thrown            */
  int32_t n_valid_beams; /* beams inside [range_min, range_max)                               */
} tbnav_rbpf_stats;

typedef struct tbnav_rbpf tbnav_rbpf; /* opaque */

/* ---- lifetime -------------------------------------------------------------------------------- */
/* ParticleFilter::ParticleFilter + initParticleSet (particle_filter.cpp:67-138): N particles at
 * pose0 with weight 1/N, empty maps (log-odds 0, distance = max_occ_dist). */
int tbnav_rbpf_create(const tbnav_rbpf_params* params, tbnav_rbpf** out);
void tbnav_rbpf_destroy(tbnav_rbpf* h);
int tbnav_rbpf_grid_size(const tbnav_rbpf* h, int32_t* xsize, int32_t* ysize);
/* Standard normals one SLAM call consumes, in draw order: N*(3k+3) (ICP ok) or N*3 (ICP failed),
 * plus 1 for the resampling offset (particle_filter.cpp:474), which is read only if resampling fires. */
int64_t tbnav_rbpf_num_normals(const tbnav_rbpf* h, int32_t icp_ok);
/* Seed of the device noise source (resets its scan counter).  Default seed 0x5EED. */
int tbnav_rbpf_set_seed(tbnav_rbpf* h, uint64_t seed);
/* The first n standard normals the LAST SLAM call consumed (host- or device-drawn) — for statistical tests. */
int tbnav_rbpf_get_normals(tbnav_rbpf* h, double* out, int64_t n);

/* ---- one scan: ParticleFilter::SLAM (particle_filter.cpp:141-251) --------------------------- */
/* scan: n_beams ranges (float, as sensor_msgs/LaserScan); u = body twist (w, vx, vy);
 * cur/prev_odom = (theta, x, y); (icp_ok, T_icp) = what ScanAlignment::pclICPWrapper returned.
 * normals: tbnav_rbpf_num_normals() standard normals in the reference's draw order (parity mode), or NULL:
 * they are then drawn ON the device (Philox4x32-10 + Box-Muller keyed by tbnav_rbpf_set_seed and the
 * scan count) — the production mode, no host RNG and no 1.2 MB/scan upload.
 * Synchronous.  Returns out->status (also when the reference would have thrown). */
int tbnav_rbpf_slam(tbnav_rbpf* h, const float* scan, int32_t n_beams, const double u[3],
                    const double cur_odom[3], const double prev_odom[3], int32_t icp_ok,
                    const double T_icp[3], const double* normals, tbnav_rbpf_stats* out);

/* OPTION (not something the reference does): per-particle scan-to-map matching.  The reference
 * aligns scan to scan once per call with PCL ICP (cloud_alignment.cpp:37-223) and every particle samples round
 * T(pose) * T_icp (particle_filter.cpp:146-153,181-188).  With this on, each particle first refines that pose
 * against its OWN map by hill climbing on the reference's scoring function (GridMapper::likelihoodFieldModel,
 * grid_mapper.cpp:69-133): six neighbours (+-x, +-y by lstep metres, +-theta by astep radians), move to the best
 * strictly better one, else halve the steps, `iterations` halvings; T_icp then only has to be a rough initial
 * guess (e.g. the odometry increment), which removes the third-party matcher from the loop.  Off by default.
 * tbnav_rbpf_get_scan_match returns the matched poses [N][3] (theta, x, y) and scores [N] of the last call. */
int tbnav_rbpf_set_scan_matching(tbnav_rbpf* h, int32_t enable, double lstep, double astep, int32_t iterations);
int tbnav_rbpf_get_scan_match(tbnav_rbpf* h, double* centers, double* scores);

/* ParticleFilter::getRobotState (particle_filter.cpp:255-274): pose of the arg-max-weight particle
 * (strict >, first wins, starting from 0.0).  best_index is optional. */
int tbnav_rbpf_best_state(tbnav_rbpf* h, double pose[3], int32_t* best_index);
/* ParticleFilter::newMap -> GridMapper::gridMap (particle_filter.cpp:277-291, grid_mapper.cpp:
 * 185-226): int8 {-1, 0, 100, (int8)(prob*100)}, transposed, of the arg-max-weight particle.  Computed on the
 * device: the log-odds at which the exported value changes were found on the host with glibc at create time,
 * so the result is the reference's bit for bit and only G bytes cross PCIe.  map holds G entries. */
int tbnav_rbpf_best_map(tbnav_rbpf* h, int8_t* map);

/* ---- multi-GPU building blocks (particles sharded across ranks) ------------------------------- */
/* SLAM without the normalise/resample tail: per-particle update only.  Weights are left
 * un-normalised; fetch them with tbnav_rbpf_get_particles, all-gather, then call
 * tbnav_rbpf_resample_global on every rank. */
int tbnav_rbpf_slam_local(tbnav_rbpf* h, const float* scan, int32_t n_beams, const double u[3],
                          const double cur_odom[3], const double prev_odom[3], int32_t icp_ok,
                          const double T_icp[3], const double* normals, tbnav_rbpf_stats* out);
/* normalizeWeights + effectiveParticles + lowVarianceResampling (particle_filter.cpp:442-500) over
 * the GLOBAL weight vector (n_global entries, identical on every rank), in the reference's
 * sequential order so Neff and the parent list are bit-exact; z = the one standard normal.
 * parents_out (n_global int32, host) receives the parent index of every slot (identity when no
 * resampling fires); weights_out (n_global, host) the normalised weights. */
int tbnav_rbpf_resample_global(const double* weights_all, int64_t n_global, double z,
                               int32_t* parents_out, double* weights_out, tbnav_rbpf_stats* out);
/* Re-populate this rank's slots from LOCAL parents (gather inside the handle); slots whose parent
 * lives on another rank are filled through get/set_particle_state below. */
int tbnav_rbpf_gather_local(tbnav_rbpf* h, const int32_t* local_parent /*[N], -1 = leave*/);

/* ---- state access: parity hooks and particle migration --------------------------------------- */
int tbnav_rbpf_get_particles(tbnav_rbpf* h, double* pose, double* prev_pose, double* weight);
int tbnav_rbpf_set_particles(tbnav_rbpf* h, const double* pose, const double* prev_pose,
                             const double* weight);
int tbnav_rbpf_get_log_odds(tbnav_rbpf* h, int32_t particle, double* out);
int tbnav_rbpf_set_log_odds(tbnav_rbpf* h, int32_t particle, const double* in);
/* Distance field of one particle as the reference stores it (Cell::occ_dist, metres).  set_ encodes
 * into the u16 code and fails with TBNAV_ERR_INVALID_ARG if a value is not one the reference can
 * produce (sqrt(di^2+dj^2)*resolution or max_occ_dist). */
int tbnav_rbpf_get_occ_dist(tbnav_rbpf* h, int32_t particle, double* out);
int tbnav_rbpf_set_occ_dist(tbnav_rbpf* h, int32_t particle, const double* in);
int tbnav_rbpf_get_dist_code(tbnav_rbpf* h, int32_t particle, uint16_t* out);
int tbnav_rbpf_get_occupied_count(tbnav_rbpf* h, int32_t* counts /*[N]*/);

/* Per-stage outputs of the LAST tbnav_rbpf_slam call (any pointer may be NULL):
 * sampled [N][k][3], p_scan [N][k] and p_pose [N][k] before the clamps, mu [N][3], sigma [N][9],
 * eta [N], new_pose [N][3], weight_raw [N] (after *= eta, before normalisation),
 * resample_parent [N] (parent slot of every particle; identity if no resampling). */
int tbnav_rbpf_get_trace(tbnav_rbpf* h, double* sampled, double* p_scan, double* p_pose, double* mu,
                         double* sigma, double* eta, double* new_pose, double* weight_raw,
                         int32_t* resample_parent);

/* ---- measurement hook --------------------------------------------------------------------------
 * Durations (ms, HIP events on the handle's stream) of the kernels of the LAST slam call:
 * [0] propose (sample+score+proposal), [1] raycast/log-odds, [2] occupancy bitmap,
 * [3] distance field, [4] normalise/select, [5] resample gather (0 if it did not run). */
#define TBNAV_RBPF_NKERNELS 6
int tbnav_rbpf_last_kernel_ms(tbnav_rbpf* h, float ms[TBNAV_RBPF_NKERNELS]);
/* The events that feed tbnav_rbpf_last_kernel_ms cost device time themselves (a few microseconds each, which is
 * not small next to the kernels), so they are recorded only after tbnav_rbpf_set_timing(h, 1); off by default,
 * in which case last_kernel_ms reports zeros. */
int tbnav_rbpf_set_timing(tbnav_rbpf* h, int32_t enable);

#ifdef __cplusplus
}
#endif
#endif /* TBNAV_RBPF_H */
