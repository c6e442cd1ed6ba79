// bmapping/particle_filter.hpp — bmapping::ParticleFilter with the reference's class surface
// (reference bmapping/include/bmapping/particle_filter.hpp:88-233): the 19-argument constructor,
// SLAM(scan, u, cur_odom, prev_odom), getRobotState(), newMap(map).  bmapping/src/turtle_mapping_node.cpp
// (:404-410, :474, :479, :494) compiles unchanged against it.  All particle state lives on the GPU
// behind include/tbnav_rbpf.h.
//
// DISTANCE FIELD.  The reference's likelihood field is a priority-queue brushfire (grid_mapper.cpp:333-435) whose result
// depends on the occupied set's hash order and keeps stale cells.  This class reproduces it bit for bit BY DEFAULT for the
// ensembles the reference can run (num_particles <= 4096, one GPU): the device logs every change of the occupied set in the
// reference's order and the per-particle brushfires run on the host's cores with the same libstdc++ containers — never slower
// than the reference's own loop.  useExactDistanceField() switches to the fast mode (exact nearest-obstacle distance computed
// at lookup, no transform at all: include/tbnav_rbpf.h "DISTANCE FIELD"; measured effect on the weights: DESIGN.md section 4),
// which is also what larger ensembles and n_gpus > 1 always use.
#ifndef TBNAV_BMAPPING_PARTICLE_FILTER_HPP
#define TBNAV_BMAPPING_PARTICLE_FILTER_HPP

#include <cstdint>
#include <random>
#include <vector>

#include "bmapping/cloud_alignment.hpp"
#include "bmapping/grid_mapper.hpp"
#include "bmapping/sensor_model.hpp"
#include "rigid2d/diff_drive.hpp"
#include "rigid2d/rigid2d.hpp"

struct tbnav_rbpf;        // C-ABI handle
struct tbnav_rbpf_group;  // the same over several GPUs (include/tbnav_rbpf.h)

namespace bmapping {

using rigid2d::Pose;
using rigid2d::Transform2D;
using rigid2d::Twist2D;

/// the filter's own process-global engine (particle_filter.cpp:17-22); reseed with getTwister().seed(s)
std::mt19937_64& getTwister();

class ParticleFilter {
 public:
  ParticleFilter(int num_particles, int k, double srr, double srt, double str, double stt, double motion_noise_theta,
                 double motion_noise_x, double motion_noise_y, double sample_range_theta, double sample_range_x,
                 double sample_range_y, double scan_likelihood_min, double scan_likelihood_max,
                 double pose_likelihood_min, double pose_likelihood_max, ScanAlignment& scan_matcher,
                 const Transform2D& pose, const GridMapper& mapper, int n_gpus = 1, const std::vector<int>& devices = {});
  // (the reference's 19 arguments, particle_filter.hpp:112-130.  n_gpus — not in the reference, SURVEY.md section 8-b — > 1
  //  splits num_particles, a multiple of n_gpus, evenly over that many MI355X of this process: per scan ONE RCCL all-gather of
  //  the weights, the same global selection on every device, particles migrate as tile blobs when resampling fires.)
  ~ParticleFilter();
  ParticleFilter(const ParticleFilter&) = delete;
  ParticleFilter& operator=(const ParticleFilter&) = delete;

  /// One scan update.  Prints "Neff: <n>" and, when it resamples, "Resampling", like the reference.
  /// Throws std::invalid_argument with the reference's messages ("eta is 0", "... NOT in the bounds of
  /// the world", "Variance in pdfNormal is 0").
  void SLAM(const std::vector<float>& scan, const Twist2D& u, const Pose& cur_odom, const Pose& prev_odom);
  Transform2D getRobotState();
  void newMap(std::vector<int8_t>& map);

  // ---- additions (not in the reference) ----
  void useDeviceNoise(std::uint64_t seed);  ///< draw the standard normals on the GPU instead of from getTwister()
  /// Not in the reference: every particle refines T(pose) * T_icp against its own map (hill climbing on the
  /// likelihood field) before sampling, so T_icp may be a rough guess such as the odometry increment.
  void useScanMatching(bool on = true, double lstep = 0.05, double astep = 0.05, int iterations = 5);
  /// The reference's brushfire distance field, bit for bit (the default up to 4096 particles on one GPU; see the header note).
  /// Call before the first SLAM(): the brushfire's result depends on the whole history of the occupied set.
  void useReferenceDistanceField(bool on = true);
  /// The fast mode: exact nearest-obstacle distances computed at lookup (not the reference's field).  Before the first SLAM().
  void useExactDistanceField() { useReferenceDistanceField(false); }
  bool referenceDistanceField() const { return reference_field_; }
  int gpus() const;
  int effectiveParticles() const { return last_neff_; }
  bool resampledLastScan() const { return last_resampled_; }

 private:
  tbnav_rbpf* h_ = nullptr;         // n_gpus == 1
  tbnav_rbpf_group* g_ = nullptr;   // n_gpus > 1
  bool reference_field_ = false;
  ScanAlignment scan_matcher_;  // copied, as the reference does (particle_filter.hpp:222)
  int num_particles_ = 0, k_ = 0, last_neff_ = 0;
  bool last_resampled_ = false, device_noise_ = false;
  std::vector<double> normals_;
};

}  // namespace bmapping
#endif
