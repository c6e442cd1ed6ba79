// bmapping/sensor_model.hpp — LaserProperties and LaserScanner with the reference's surface
// (reference bmapping/include/bmapping/sensor_model.hpp:20-117, bmapping/src/bmapping/sensor_model.cpp:9-131).
// LaserScanner's two public methods are small host loops and stay on the host (src/grid_mapper_shim.cpp): the device
// kernels transform beams themselves (csrc/rbpf_propose.hip, rbpf_raycast.hip), from a per-scan table built exactly like laserEndPoints does.
#ifndef TBNAV_BMAPPING_SENSOR_MODEL_HPP
#define TBNAV_BMAPPING_SENSOR_MODEL_HPP

#include <vector>

#include "rigid2d/rigid2d.hpp"

namespace bmapping {

using rigid2d::Transform2D;
using rigid2d::Vector2D;

struct LaserProperties {
  float beam_min = 0.0f, beam_max = 0.0f, beam_delta = 0.0f, range_min = 0.0f, range_max = 0.0f;
  double z_hit = 0.25, z_short = 0.25, z_max = 0.25, z_rand = 0.25, sigma_hit = 1.0;

  LaserProperties() = default;
  LaserProperties(float beam_min, float beam_max, float beam_delta, float range_min, float range_max, double z_hit,
                  double z_short, double z_max, double z_rand, double sigma_hit)
      : beam_min(beam_min), beam_max(beam_max), beam_delta(beam_delta), range_min(range_min), range_max(range_max),
        z_hit(z_hit), z_short(z_short), z_max(z_max), z_rand(z_rand), sigma_hit(sigma_hit) {}
};

/// Models a 2D laser range finder (sensor_model.hpp:81-117)
class LaserScanner {
 public:
  LaserScanner(const LaserProperties& props, const Transform2D& Trs)
      : z_hit_(props.z_hit), z_short_(props.z_short), z_max_(props.z_max), z_rand_(props.z_rand), sigma_hit_(props.sigma_hit),
        Trs_(Trs), beam_min_(props.beam_min), beam_max_(props.beam_max), beam_delta_(props.beam_delta),
        range_min_(props.range_min), range_max_(props.range_max) {}

  /// cartesian end points of the valid beams in the map frame (sensor_model.cpp:43-112)
  void laserEndPoints(std::vector<Vector2D>& end_points, const std::vector<float>& beam_length, const Transform2D& pose) const;
  /// number of range measurements inside [range_min, range_max) (sensor_model.cpp:116-131)
  unsigned int numberValidMeasurements(const std::vector<float>& beam_length) const;

  double z_hit_, z_short_, z_max_, z_rand_;  // public in the reference too (sensor_model.hpp:108-109)
  double sigma_hit_;

  // (additions: read access for the shims that flatten a scanner into the C-ABI's parameter struct)
  LaserProperties properties() const {
    return LaserProperties(beam_min_, beam_max_, beam_delta_, range_min_, range_max_, z_hit_, z_short_, z_max_, z_rand_, sigma_hit_);
  }
  const Transform2D& robotToLaser() const { return Trs_; }

 private:
  Transform2D Trs_;                         // robot to laser scanner
  float beam_min_, beam_max_, beam_delta_;  // start, end, increment scan angles
  float range_min_, range_max_;             // min and max range limit for laser
};

}  // namespace bmapping
#endif
