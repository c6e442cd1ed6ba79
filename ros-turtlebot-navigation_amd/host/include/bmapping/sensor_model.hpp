// bmapping/sensor_model.hpp — LaserProperties with the reference's layout and ctor
// (reference bmapping/include/bmapping/sensor_model.hpp:20-79).  The beam -> end-point transform
// (LaserScanner::laserEndPoints, sensor_model.cpp:43-112) runs inside the device kernels; the host
// only builds the per-scan table of valid beams in the sensor frame (csrc/rbpf.hip build_scan_consts).
#ifndef TBNAV_BMAPPING_SENSOR_MODEL_HPP
#define TBNAV_BMAPPING_SENSOR_MODEL_HPP

namespace bmapping {

struct LaserProperties {
  float beam_min = 0.0f, beam_max = 0.0f, beam_delta = 0.0f, range_min = 0.0f, range_max = 0.0f;
  double z_hit = 0.25, z_short = 0.25, z_max = 0.25, z_rand = 0.25, sigma_hit = 1.0;

  LaserProperties() = default;
  LaserProperties(float beam_min, float beam_max, float beam_delta, float range_min, float range_max, double z_hit,
                  double z_short, double z_max, double z_rand, double sigma_hit)
      : beam_min(beam_min), beam_max(beam_max), beam_delta(beam_delta), range_min(range_min), range_max(range_max),
        z_hit(z_hit), z_short(z_short), z_max(z_max), z_rand(z_rand), sigma_hit(sigma_hit) {}
};

}  // namespace bmapping
#endif
