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
 *   log_odds        : f64 (Cell::log_odds, grid_mapper.hpp:67; beam-ordered adds), TILED and COPY-ON-WRITE: each
 *                     particle holds a table of ceil(xsize/32)^2 tile ids into one pool of 32x32-cell tiles
 *                     (8 KB each) shared by the handle's particles; id 0 = the shared all-zero tile (untouched
 *                     area).  The reference deep-copies a whole map per resampled particle
 *                     (particle_filter.cpp:495); here a resample copies tables and bumps reference counts, and
 *                     the next scan clones only the tiles it writes.  100 000 particles x 2000 x 2000 cells
 *                     (BASELINE configs[4]) would be 3.2 TB dense; tiled it is the scanned area per lineage.
 *   occupancy       : one bit per cell with prob >= 0.90, 32 x u32 per tile, stored, shared and copied WITH the
 *                     log-odds tile (+ occupied cells per tile row, [N][ceil(xsize/32)] i32); decided in
 *                     log-odds space against a cut-off found on the host with glibc at create time
 *                     (SURVEY.md hard part 2: prob(log 9) == 0.9 exactly with glibc)
 *   dist_code       : [N][G] u16, allocated only when something needs a STORED field (tbnav_rbpf_set_occ_dist,
 *                     get_occ_dist/get_dist_code, the stored-field modes of TBNAV_RBPF_OPT_DF_MODE): squared
 *                     distance IN CELLS to the nearest occupied cell; occ_dist = sqrt((double)code) * resolution
 *                     is bit-identical to the reference's distances_[di][dj] * resolution_
 *                     (grid_mapper.cpp:263,318); 0xFFFF = never reached = max_occ_dist_ 10.0
 *                     (grid_mapper.cpp:49,58)
 *
 * DISTANCE FIELD — what differs from the reference unless the REFERENCE mode is selected.  The reference's
 * GridMapper::euclideanSignedDistanceField (grid_mapper.cpp:333-435) is a priority-queue brushfire whose result
 * depends on std::unordered_set iteration order and heap tie-breaking and is not the exact Euclidean distance
 * transform (hard part 1).  By default a likelihood lookup computes the EXACT distance to the
 * nearest occupied cell within cell_radius_ (never larger than the reference's value, equal in most cells; a cell
 * with no obstacle within cell_radius_ reads max_occ_dist_, where the reference keeps whatever an earlier brushfire
 * left).  Scan likelihoods, eta and weights of a default-mode run therefore differ slightly from the reference's
 * (measured in DESIGN.md); TBNAV_RBPF_DF_REFERENCE reproduces the reference's field bit for bit for small
 * ensembles.
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
/* Same, with an explicit budget for the log-odds tile pool (bytes; 0 = the default: what every particle's map could
 * ever need if that fits in half of the device memory free at create time, else that half).  A scan that needs a
 * tile when none is free returns TBNAV_ERR_POOL_EXHAUSTED and leaves the maps of the particles concerned unchanged.
 * (Pools of 16 384 tiles — 128 MB — or more keep their free tiles in sixteen lists; a particle takes the tiles one map update
 * makes private from one of them, and what that list lacks from the next ones: a scan whose particles need no more tiles
 * than are free never fails, whatever the lists' lengths.) */
int tbnav_rbpf_create_pool(const tbnav_rbpf_params* params, uint64_t max_pool_bytes, tbnav_rbpf** out);
void tbnav_rbpf_destroy(tbnav_rbpf* h);
/* Tile pool occupancy: tiles the pool holds, tiles free now, bytes of log-odds per tile (any pointer may be NULL). */
int tbnav_rbpf_pool_stats(tbnav_rbpf* h, uint64_t* capacity_tiles, uint64_t* free_tiles, uint64_t* tile_bytes);
int tbnav_rbpf_grid_size(const tbnav_rbpf* h, int32_t* xsize, int32_t* ysize);
/* Standard normals one SLAM call consumes, in draw order: N*(3k+3) (ICP ok) or N*3 (ICP failed),
 * plus 1 for the resampling offset (particle_filter.cpp:474), which is read only if resampling fires. */
int64_t tbnav_rbpf_num_normals(const tbnav_rbpf* h, int32_t icp_ok);
/* Seed of the device noise source (resets its scan counter).  Default seed 0x5EED. */
int tbnav_rbpf_set_seed(tbnav_rbpf* h, uint64_t seed);
/* Sharded filters: this handle's particles are particles [first_particle, first_particle + N) of an ensemble of
 * particles_global.  The device noise source then draws element first_particle * stride + j of the ENSEMBLE's stream for the
 * handle's normal j (stride = 3k+3 or 3, tbnav_rbpf_num_normals) and the ensemble's resampling offset, so ranks that share a
 * seed draw DISJOINT normals — exactly the ones the unsharded filter of particles_global particles draws from that seed.  Without
 * this call every handle numbers its particles from 0 and ranks sharing a seed would all draw the same normals (their copies of a
 * migrated particle would then evolve identically).  particles_global = 0 switches back to the unsharded numbering. */
int tbnav_rbpf_set_rng_shard(tbnav_rbpf* h, uint64_t first_particle, uint64_t particles_global);
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
/* Replay of a LOGGED run (what turtle_mapping_node.cpp:459-494 does per laser message, for n_scans messages in a row):
 * scans [n_scans][n_beams]; u [n_scans][3]; odom [n_scans + 1][3] with odom[s] = prev, odom[s + 1] = cur of scan s;
 * icp_ok [n_scans] or NULL (all 1); T_icp [n_scans][3]; noise drawn on the device; out [n_scans].  The results of
 * exactly n_scans tbnav_rbpf_slam calls, bit for bit, without a trip through the caller's language per scan — and, in the
 * default configuration, without the device waiting for the host between scans: scan s + 1 is enqueued before the host
 * has seen whether scan s resamples; its kernels check that decision on the device and do nothing if it does, the host
 * then runs the copies and enqueues scan s + 1 again (TBNAV_RBPF_OPT_BATCH_PIPELINE 0 = one synchronous call per scan).
 * Stops at the first scan whose status is not TBNAV_OK and returns it; the filter's state after an error is unspecified
 * (the reference throws there and the node dies). */
int tbnav_rbpf_slam_batch(tbnav_rbpf* h, const float* scans, int32_t n_beams, int32_t n_scans, const double* u,
                          const double* odom, const int32_t* icp_ok, const double* T_icp, tbnav_rbpf_stats* out);

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

/* ---- multi-GPU building blocks (particles sharded across ranks) -------------------------------
 * One handle per rank holds N/P particles.  Per scan: tbnav_rbpf_slam_local (no normalise/resample tail) ->
 * tbnav_rbpf_copy_weights_dev into the rank's slice of a device buffer -> ONE all-gather of the raw weights (RCCL)
 * -> tbnav_rbpf_resample_global_dev on every rank: the reference's sequential normalise / Neff / low-variance
 * selection (particle_filter.cpp:442-500) over the identical global vector, so Neff and the parent list are
 * bit-exact and the same everywhere; the local slice of the normalised weights lands in the handle.  Only when
 * resampling fires do particles move: tbnav_rbpf_gather_local for parents that live on this rank,
 * tbnav_rbpf_export_particle_dev / tbnav_rbpf_import_particle_dev (device buffers, sent point to point) for the
 * others, tbnav_rbpf_set_weights_from_global_dev for the slots' weights.  Nothing passes through host memory
 * except the parent list, which the host needs to plan the sends. */
int tbnav_rbpf_slam_local(tbnav_rbpf* h, const float* scan, int32_t n_beams, const double u[3],
                          const double cur_odom[3], const double prev_odom[3], int32_t icp_ok,
                          const double T_icp[3], const double* normals, tbnav_rbpf_stats* out);
/* This handle's weights [N] -> device buffer (synchronous on the handle's stream). */
int tbnav_rbpf_copy_weights_dev(tbnav_rbpf* h, double* d_dst);
/* d_weights_all: n_global raw weights on the device (identical on every rank); offset: global index of this
 * handle's slot 0; z: the one standard normal of lowVarianceResampling (particle_filter.cpp:474) — NaN = the one the
 * handle's last scan carries (device noise with tbnav_rbpf_set_rng_shard: the ensemble's offset, identical on every rank).  parents_out
 * (n_global int32, host, may be NULL) is filled only if out->resampled. */
int tbnav_rbpf_resample_global_dev(tbnav_rbpf* h, const double* d_weights_all, int64_t n_global, int64_t offset, double z,
                                   int32_t* parents_out, tbnav_rbpf_stats* out);
/* After a resample: slot m takes the normalised weight of its GLOBAL parent (weights are not reset, :495). */
int tbnav_rbpf_set_weights_from_global_dev(tbnav_rbpf* h, const int32_t* global_parent_of_slot /*[N]*/);
/* Same selection on host buffers (n_global entries, identical on every rank), for callers without device tensors.
 * parents_out (n_global int32, host) receives the parent index of every slot (identity when no
 * resampling fires); weights_out (n_global, host) the normalised weights. */
int tbnav_rbpf_resample_global(const double* weights_all, int64_t n_global, double z,
                               int32_t* parents_out, double* weights_out, tbnav_rbpf_stats* out);
/* Parity-test hook: out[i] = x[i] after n[i] times x = fl(x + d[i]) as the map update evaluates a cell that n beams cross
 * (grid_mapper.cpp:438-477 adds the same log-odds once per beam) — on the device, WITHOUT the chain of dependent adds (integer steps
 * inside a binade, plain adds across its ends); must equal the plain loop bit for bit.  Host buffers; uses the current device. */
int tbnav_rbpf_add_repeated(const double* x, const double* d, const int32_t* n, double* out, int64_t count);
/* Test hook for the tile pool's free lists, on a pool of cap_tiles tiles that belongs to no handle (see tbnav_rbpf_create_pool:
 * sixteen lists from 16 384 tiles, one below): `rounds` times, `callers` callers pop tiles_each (<= 64) tiles at once — every
 * caller starting at the list its number names, or all at list 0 (same_hint != 0) — and the tiles are pushed back.  ids_out
 * [callers * tiles_each]: the ids the LAST round's callers got (zeros where no list could supply a caller); the free counts after
 * that round's pops and pushes.  Host buffers; uses the current device. */
int tbnav_rbpf_pool_selftest(uint32_t cap_tiles, int32_t rounds, int32_t callers, int32_t tiles_each, int32_t same_hint,
                             uint32_t* ids_out, uint64_t* free_after_pop, uint64_t* free_after_push);
/* Re-populate this rank's slots from LOCAL parents (tables and state move on the device); -1 = keep. */
int tbnav_rbpf_gather_local(tbnav_rbpf* h, const int32_t* local_parent /*[N], -1 = leave*/);
/* A particle as one device buffer: header, state (pose, prev_pose, weight), the indices and 8 KB payloads of the
 * tiles (log-odds + occupancy bits) it does not share with the empty map, its tile-row counts, and its stored distance field if
 * that is authoritative (injected).  export_size: bytes the export of `slot` needs now. */
int tbnav_rbpf_export_size(tbnav_rbpf* h, int32_t slot, uint64_t* bytes);
int tbnav_rbpf_export_particle_dev(tbnav_rbpf* h, int32_t slot, void* d_buf, uint64_t capacity, uint64_t* bytes);
int tbnav_rbpf_import_particle_dev(tbnav_rbpf* h, int32_t slot, const void* d_buf, uint64_t bytes);
/* The same for many particles per call (a cross-rank resample moves hundreds per rank; slots / sizes / offsets are host
 * arrays).  export: the blobs of slots[0..n) back to back in d_buf (a slot may be listed more than once), offsets_out[i] =
 * where blob i starts, offsets_out[n] = bytes written; sizes first (export_batch_sizes) to size the buffer and tell the
 * receivers.  import: slot slots[i] (each at most once) takes the blob at d_buf + offsets[i] (several slots may name the
 * same blob).  Two launches each, whatever n.  import returns TBNAV_ERR_POOL_EXHAUSTED with NOTHING touched when the incoming
 * tiles exceed the free tiles plus every tile the destination slots name; inside that bound the slots are released first and an
 * exhaustion found while unpacking leaves those slots with empty maps (a sharded driver must then stop on every rank). */
int tbnav_rbpf_export_batch_sizes(tbnav_rbpf* h, int32_t n, const int32_t* slots, uint64_t* sizes_out /*[n]*/);
int tbnav_rbpf_export_batch_dev(tbnav_rbpf* h, int32_t n, const int32_t* slots, void* d_buf, uint64_t capacity,
                                uint64_t* offsets_out /*[n + 1]*/);
int tbnav_rbpf_import_batch_dev(tbnav_rbpf* h, int32_t n, const int32_t* slots, const void* d_buf, uint64_t bytes,
                                const uint64_t* offsets /*[n]*/);
/* Particle src_slot of `src` -> dst_slot of `dst` (same device, same grid, same distance-field mode): the deep copy
 * behind bmapping::GridMapper's value semantics.  In the REFERENCE mode the occupied set travels with its history. */
int tbnav_rbpf_copy_particle(tbnav_rbpf* dst, int32_t dst_slot, tbnav_rbpf* src, int32_t src_slot);

/* ---- the sharded filter behind the ordinary entry points (include/tbnav_comm.h) -----------------------------------
 * One process per GPU: after tbnav_rbpf_attach_comm(h, comm) — comm on the handle's device, params.num_particles = this
 * rank's share, every rank the same, weights initialised to 1 / (nranks * N) by the caller (tbnav_rbpf_set_particles) —
 * tbnav_rbpf_slam (and _slam_batch, scan by scan) IS the sharded scan, issued by the library on the handle's own streams:
 *   main stream    noise -> propose -> map update ............................... -> (when resampling fires: migration)
 *   second stream           wait for the proposal kernel -> ONE ncclAllGather of the raw weights -> the reference's
 *                           sequential normalise / Neff / low-variance selection over the GLOBAL vector
 * i.e. the global normalise (a chain of dependent adds as long as the ENSEMBLE) runs beside the local map update; the host
 * waits once per scan.  A resample that moves particles: one all-gather of blob sizes, one batched export, one ncclSend /
 * ncclRecv per (source, destination) pair inside one group, one batched import, then an all-gather of the ranks' statuses
 * (a rank whose tile pool is exhausted stops every rank with TBNAV_ERR_POOL_EXHAUSTED instead of leaving them in the next
 * collective).  Every scan also ends with an all-gather of the ranks' statuses (4 bytes each): what the reference reports by
 * throwing — a particle out of the world, eta is 0 — is returned by EVERY rank at the same scan (the lowest failing rank's code).  `normals` is this rank's slice of the ensemble's stream + the resampling offset (as for _slam_local), or
 * NULL: device noise — the handle is given its place in the ensemble's counter space (tbnav_rbpf_set_rng_shard), so the
 * sharded filter draws what the unsharded one would.  Not available in the REFERENCE distance-field mode.  comm = NULL
 * detaches.  The communicator must outlive the handle's last scan; the handle does not own it. */
struct tbnav_comm;
int tbnav_rbpf_attach_comm(tbnav_rbpf* h, struct tbnav_comm* comm);

/* One process driving n_gpus devices (what bmapping::ParticleFilter(..., n_gpus) holds): params->num_particles (a multiple
 * of n_gpus) split evenly over devices[0..n_gpus) (NULL: 0, 1, ...; a device may repeat: those members exchange by copies
 * instead of RCCL, see tbnav_comm.h); max_pool_bytes_per_member as in tbnav_rbpf_create_pool.  Results equal the unsharded
 * filter's bit for bit (same weights -> same global selection; a particle is its state and its tiles wherever it lives). */
typedef struct tbnav_rbpf_group tbnav_rbpf_group;
int tbnav_rbpf_group_create(const tbnav_rbpf_params* params, int32_t n_gpus, const int32_t* devices, uint64_t max_pool_bytes_per_member,
                            tbnav_rbpf_group** out);
void tbnav_rbpf_group_destroy(tbnav_rbpf_group* g);
int tbnav_rbpf_group_size(const tbnav_rbpf_group* g);
int tbnav_rbpf_group_member(tbnav_rbpf_group* g, int32_t rank, tbnav_rbpf** out);  /* borrowed: parity hooks of one shard */
int tbnav_rbpf_group_set_seed(tbnav_rbpf_group* g, uint64_t seed);
int tbnav_rbpf_group_set_option(tbnav_rbpf_group* g, int32_t option, int32_t value);
/* Standard normals one scan of the WHOLE filter consumes: N_global * (3k+3 | 3) + 1. */
int64_t tbnav_rbpf_group_num_normals(const tbnav_rbpf_group* g, int32_t icp_ok);
/* ParticleFilter::SLAM over the whole filter.  normals: the ensemble's draw stream in the reference's order, or NULL
 * (device noise).  out: the ensemble's sum_w / sq_sum / neff / resampled. */
int tbnav_rbpf_group_slam(tbnav_rbpf_group* g, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                          const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals, tbnav_rbpf_stats* out);
/* getRobotState / newMap over the ensemble (strict >, first wins; best_index is the GLOBAL slot). */
int tbnav_rbpf_group_best_state(tbnav_rbpf_group* g, double pose[3], int32_t* best_index);
int tbnav_rbpf_group_best_map(tbnav_rbpf_group* g, int8_t* map);

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

/* ---- one particle's map on its own: the methods of bmapping::GridMapper (grid_mapper.hpp:128-140) ------------------
 * The host class bmapping::GridMapper is a one-particle handle driven through these three calls (same kernels as
 * the filter's per-particle update).  pose = (theta, x, y) of the ROBOT in the map frame.
 *  integrate_scan : GridMapper::integrateScan (grid_mapper.cpp:140-182) — moves `particle` to `pose`, ray-casts the
 *                   scan into its map; in the stored-field / REFERENCE modes the distance field is refreshed too.
 *  likelihood     : GridMapper::likelihoodFieldModel (grid_mapper.cpp:69-133) of the particle's map at `pose`.
 *  particle_map   : GridMapper::gridMap (grid_mapper.cpp:185-226) of that particle (int8, transposed, G entries). */
int tbnav_rbpf_integrate_scan(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, const double pose[3]);
int tbnav_rbpf_likelihood(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, const double pose[3], double* out);
int tbnav_rbpf_particle_map(tbnav_rbpf* h, int32_t particle, int8_t* map);

/* ---- options (explicit setters; nothing in the library reads the environment) --------------------------------
 * TBNAV_RBPF_OPT_DF_MODE — where a likelihood lookup gets its distance from.  Must be chosen before the first scan.
 *   TBNAV_RBPF_DF_QUERY     (default) exact nearest-obstacle query on the occupancy bitmap at lookup time; no
 *                           transform and no stored field in the SLAM path.
 *   TBNAV_RBPF_DF_WINDOW    exact transform of a window round each particle, stored, before each update.
 *   TBNAV_RBPF_DF_FULL      exact transform of every whole map after every update (the reference's data flow).
 *                           QUERY / WINDOW / FULL give bit-identical results while every looked-up cell has an
 *                           obstacle within cell_radius_.
 *   TBNAV_RBPF_DF_REFERENCE the reference's own brushfire (grid_mapper.cpp:272-435), reproduced on the host per
 *                           particle with the same libstdc++ containers fed the same occupied-set insert / erase
 *                           sequence (the beam-ordered raycast kernel logs it) and copied on resampling the way
 *                           particle_filter.cpp:495-499 copies particles; its result becomes the stored field the
 *                           next scan reads.  Serial host work per particle: N <= 4096, meant for the reference's
 *                           own launch configuration.  After tbnav_rbpf_set_log_odds the set of that particle is
 *                           rebuilt in ascending cell order (its history is unknown).
 * TBNAV_RBPF_OPT_RAYCAST_ORDERED 1 = always use the beam-ordered raycast kernel (development / A-B runs).
 * TBNAV_RBPF_OPT_RAYCAST_THREADS 256 | 512 | 1024 threads per workgroup of the tile raycast (default 0: chosen per launch, see _RAYCAST_ADAPT).
 * TBNAV_RBPF_OPT_COUNT_CELLS     1 = the tile raycast counts the cells it updates (tbnav_rbpf_scan_counts).
 * TBNAV_RBPF_OPT_RAYCAST_FORM    retired.  0 is accepted (the box-counter kernel rbpf_raycast_box, the only form); 1 — round 2's first tile kernel,
 *                                removed in round 4 — is TBNAV_ERR_INVALID_ARG: the beam-ordered kernel is selected by _RAYCAST_ORDERED.
 * TBNAV_RBPF_OPT_NOISE_IN_KERNEL 0 (default since round 6) = with device noise (normals == NULL) rbpf_sample_normals stores the standard normals first
 *                                and carries the scan's beam table over: three launches per scan, no hand-over inside a launch;
 *                                1 = they are drawn INSIDE rbpf_propose and never stored, and the beam table reaches the device through
 *                                that launch's leading workgroup (two launches; the proposal kernel is ~5 us slower per 1000 particles
 *                                and a synchronous scan takes the same wall time either way, which is why it is not the default; the
 *                                other workgroups wait for the table with a bound — a table that never arrives is TBNAV_ERR_HIP and
 *                                nothing of the scan is applied).  Same Philox counters, same values either way
 *                                (particle_filter.cpp:25-34, :504-519 are what both replace).
 * TBNAV_RBPF_OPT_RAYCAST_BAND_ROWS n > 0 = rbpf_raycast_box keeps at most about n rows of a scan's bounding box in LDS at a
 *                                time (0 = as many as fit): drives its band loop on small maps (tests).
 * TBNAV_RBPF_OPT_RAYCAST_ADAPT   1 = rbpf_raycast_box's LDS array is sized by what the particles' boxes needed in the last scans (default;
 *                                the kernel reports it through mapped memory: less LDS per workgroup = three — or, when need + 256 words fits a
 *                                quarter of a CU's LDS, FOUR — workgroups per CU instead of two; a box that outgrows the guess takes a second
 *                                band; the four-per-CU form lists 4 events per end-point cell instead of 8 before it replays the cell against
 *                                every beam — and only where a box does not fit four per CU with 8); 2 = as 1 but never the four-per-CU form (A-B
 *                                runs); 3 = as 1 with the 4-event lists wherever four fit (tests); 0 = by the scan's longest beam in every direction.
 *                                TBNAV_RBPF_OPT_RAYCAST_THREADS 0 = 512 threads when three workgroups fit a CU's LDS, else 1024 (default).
 * TBNAV_RBPF_OPT_RAYCAST_CELL16  rbpf_raycast_box's 16-bit cell form (half the LDS per cell of the box, slots by table look-up): 1 = where it lets
 *                                more workgroups share a CU than the 32-bit form (default), 0 = never, 2 = wherever it can run (tests, A-B).
 *                                Bit-identical maps in either form.
 * TBNAV_RBPF_OPT_BATCH_PIPELINE  1 = tbnav_rbpf_slam_batch keeps two scans in the stream (default); 0 = n synchronous calls.
 * TBNAV_RBPF_OPT_HOST_THREADS    host threads the REFERENCE distance-field mode spreads its per-particle brushfires over (particles are
 *                                independent; the order of operations inside one particle is the reference's).  0 = the default,
 *                                automatic: a scan's passes are a burst of CPU time (60 ms of it in 4 ms at configs[2], then nothing
 *                                until the next scan), so the count follows the CPU TIME the process is granted — the cgroup's
 *                                cpu.max quota, or the affinity mask without one: as many threads as keep the average over a scan
 *                                period under 85 % of the quota, at least the quota's own count, at most four times it (and the
 *                                affinity mask, and 128).  n > 0: exactly n.
 * TBNAV_RBPF_OPT_REF_REACH       REFERENCE mode: how many cells out from the occupied cells a scan's brushfire runs before it stops
 *                                (default 1: the occupied cells and their neighbours' neighbours; 0 = to the end, the round-3..5 behaviour).  The pass writes every cell once, in a
 *                                deterministic order, so a stopped pass equals the finished one on every cell it has written and can be
 *                                resumed; the proposal kernel reports a lookup that lands on an unwritten cell, exactly that state's pass
 *                                is resumed and the proposal run again (tbnav_rbpf_reference_field_stats counts both).  Results are
 *                                bit-identical for every value; whole-field exports finish the pass (and replay the lineage where a
 *                                stale cell's value depends on the unfinished part of an earlier pass). */
enum { TBNAV_RBPF_OPT_DF_MODE = 1, TBNAV_RBPF_OPT_RAYCAST_ORDERED = 2, TBNAV_RBPF_OPT_RAYCAST_THREADS = 3, TBNAV_RBPF_OPT_COUNT_CELLS = 4,
       TBNAV_RBPF_OPT_RAYCAST_FORM = 5, TBNAV_RBPF_OPT_RAYCAST_BAND_ROWS = 6, TBNAV_RBPF_OPT_BATCH_PIPELINE = 7, TBNAV_RBPF_OPT_HOST_THREADS = 8, TBNAV_RBPF_OPT_RAYCAST_ADAPT = 9, TBNAV_RBPF_OPT_RAYCAST_CELL16 = 10,
       TBNAV_RBPF_OPT_NOISE_IN_KERNEL = 11, TBNAV_RBPF_OPT_REF_REACH = 12 };
enum { TBNAV_RBPF_DF_FULL = 0, TBNAV_RBPF_DF_WINDOW = 1, TBNAV_RBPF_DF_QUERY = 2, TBNAV_RBPF_DF_REFERENCE = 3 };
int tbnav_rbpf_set_option(tbnav_rbpf* h, int32_t option, int32_t value);
/* Since the last reset, summed over particles and scans (TBNAV_RBPF_OPT_COUNT_CELLS on): cell_updates = log-odds
 * adds the reference performs (free cells of every ray + end points, grid_mapper.cpp:153-177), distinct_cells =
 * cells actually read-modified-written (a cell touched by several beams of one scan counts once per scan). */
int tbnav_rbpf_scan_counts(tbnav_rbpf* h, uint64_t* cell_updates, uint64_t* distinct_cells, int32_t reset);
/* Reference-field mode (TBNAV_RBPF_DF_REFERENCE) only: how many DISTINCT (occupied set with its history, field) states the particles
 * hold now, how many brushfires the last scan ran and how many all scans so far — one per distinct (state, insert / erase sequence)
 * of the scan, not one per particle: particles that are copies of one another and saw the same cells change share the result
 * (csrc/ref_field.hpp).  Any pointer may be NULL. */
int tbnav_rbpf_reference_field_counts(tbnav_rbpf* h, int32_t* distinct_states, int32_t* last_brushfires, int64_t* total_brushfires);
/* Reference-field mode only, since the mode was chosen — the lazy brushfire's bookkeeping (csrc/ref_field.hpp; round 6):
 *   out[0] passes started (one per distinct (state, insert / erase sequence) of a scan)
 *   out[1] iterations of grid_mapper.cpp:399-433 run, all passes (a whole 400 x 400 pass is about 157 000)
 *   out[2] states whose pass was RESUMED because a likelihood lookup landed on a cell it had not written yet
 *   out[3] passes that ran to their end          out[4] lineages REPLAYED from their last exact ancestor (stale cells wanted)
 *   out[5] generations re-run by those replays   out[6] bytes of history (event lists) alive
 *   out[7] proposals run a second time (tbnav_rbpf_slam: the scans in which out[2] grew)
 *   out[8..13] host microseconds spent, summed over the scans: fetching the scans' insert / erase logs | the grouped brushfires
 *              (RefField::step) | a resampling's copies | bringing the device's field slots up to date | before the proposal
 *              (kept particle state, flags) | looking for pending lookups after it;  out[14], out[15]: of out[9], the part spent grouping the
 *              particles by (state, event sequence) | releasing the scan's old states
 *   out[16] microseconds the host threads spent in the passes, summed over the threads (the CPU time the fields cost)
 *   out[17] host threads the last scan's passes ran on (TBNAV_RBPF_OPT_HOST_THREADS 0: chosen per scan, see there) */
int tbnav_rbpf_reference_field_stats(tbnav_rbpf* h, int64_t out[18]);

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
/* rbpf_raycast_box's LDS array at the last map update: the cells the particles' bounding boxes needed lately (0: not known yet or
 * TBNAV_RBPF_OPT_RAYCAST_ADAPT 0) and the cells the launch's array held (what decides two / three / four workgroups per CU). */
int tbnav_rbpf_raycast_box_cells(const tbnav_rbpf* h, int32_t* need_cells, int32_t* array_cells);
/* Names of the instantiations the handle's LAST proposal and map-update launches were, as a profiler prints them
 * ("rbpf_propose<256>"; "rbpf_raycast_box<512, 8, false, 4>" = threads, waves per SIMD, 16-bit cells, events per end-point slot; "rbpf_raycast"),
 * and the map update's workgroup count (particles + 1 when the normalise / select workgroup rode in its launch): lets a benchmark
 * line point at one row of a rocprofv3 --stats summary. */
int tbnav_rbpf_last_kernel_names(const tbnav_rbpf* h, char* propose, int32_t propose_cap, char* raycast, int32_t raycast_cap,
                                 int32_t* raycast_workgroups);

#ifdef __cplusplus
}
#endif
#endif /* TBNAV_RBPF_H */
