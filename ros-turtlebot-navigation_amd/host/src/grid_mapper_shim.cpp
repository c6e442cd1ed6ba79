// grid_mapper_shim.cpp — bmapping::LaserScanner (host) and bmapping::GridMapper (one-particle handle of the RBPF
// C-ABI, include/tbnav_rbpf.h).  Reference: bmapping/src/bmapping/sensor_model.cpp, grid_mapper.cpp.
#include <cmath>
#include <stdexcept>
#include <string>

#include "bmapping/grid_mapper.hpp"
#include "tbnav_rbpf.h"

namespace bmapping {

// ---- LaserScanner ---------------------------------------------------------------------------------------------
void LaserScanner::laserEndPoints(std::vector<Vector2D>& end_points, const std::vector<float>& beam_length, const Transform2D& pose) const {
  end_points.reserve(numberValidMeasurements(beam_length));
  const Transform2D Tms = pose * Trs_;  // map -> sensor
  double beam_angle = beam_min_;
  for (unsigned int i = 0; i < beam_length.size(); i++) {
    const double range = beam_length.at(i);
    if (range >= range_min_ && range < range_max_) {
      const Vector2D in_sensor(range * std::cos(beam_angle), range * std::sin(beam_angle));  // range2Cartesian, :9-16
      end_points.push_back(Tms(in_sensor));
    }
    beam_angle += beam_delta_;  // float -> double accumulation, wrap at beam_max (:96-108)
    if (beam_max_ < 0.0 && beam_angle <= beam_max_) beam_angle = beam_min_;
    else if (beam_max_ >= 0.0 && beam_angle >= beam_max_) beam_angle = beam_min_;
  }
}

unsigned int LaserScanner::numberValidMeasurements(const std::vector<float>& beam_length) const {
  unsigned int valid = 0;
  for (const float r : beam_length)
    if (r >= range_min_ && r < range_max_) valid++;
  return valid;
}

// ---- GridMapper -----------------------------------------------------------------------------------------------
namespace {
void check(int rc, const char* where) {
  if (rc == TBNAV_OK) return;
  const std::string text = tbnav_status_string(rc);
  switch (rc) {  // what the reference throws (grid_mapper.cpp:22,701,856)
    case TBNAV_ERR_OUT_OF_WORLD:
    case TBNAV_ERR_PDF_VARIANCE:
    case TBNAV_ERR_BRESENHAM:
      throw std::invalid_argument(text);
    case TBNAV_ERR_INVALID_ARG:
      throw std::invalid_argument(std::string(where) + ": " + text);
    default: {
      std::string msg = std::string(where) + ": " + text;
      const char* hip = tbnav_last_hip_error();
      if (hip && *hip) msg += std::string(" [") + hip + "]";
      throw std::runtime_error(msg);
    }
  }
}
}  // namespace

GridMapper::GridMapper(double resolution, double xmin, double xmax, double ymin, double ymax, const LaserProperties& props,
                       const Transform2D& Trs)
    : LaserScanner(props, Trs), resolution_(resolution), xmin_(xmin), xmax_(xmax), ymin_(ymin), ymax_(ymax) {}

GridMapper::~GridMapper() { tbnav_rbpf_destroy(h_); }

tbnav_rbpf* GridMapper::handle() const {
  if (h_) return h_;
  tbnav_rbpf_params p{};
  p.num_particles = 1; p.num_samples_mode = 1;
  p.srr = p.srt = p.str_ = p.stt = 0.1;   // (the filter's own parameters are not used by the map methods)
  for (int q = 0; q < 3; ++q) { p.motion_noise[q] = 1e-10; p.sample_range[q] = 1e-10; }
  p.scan_likelihood_min = 0.0; p.scan_likelihood_max = 1e300; p.pose_likelihood_min = 0.0; p.pose_likelihood_max = 1e300;
  const LaserProperties L = properties();
  p.beam_min = L.beam_min; p.beam_max = L.beam_max; p.beam_delta = L.beam_delta; p.range_min = L.range_min; p.range_max = L.range_max;
  p.device = -1;
  p.z_hit = L.z_hit; p.z_short = L.z_short; p.z_max = L.z_max; p.z_rand = L.z_rand; p.sigma_hit = L.sigma_hit;
  const auto trs = robotToLaser().displacement();
  p.Trs[0] = trs.theta; p.Trs[1] = trs.x; p.Trs[2] = trs.y;
  p.resolution = resolution_; p.xmin = xmin_; p.xmax = xmax_; p.ymin = ymin_; p.ymax = ymax_;
  check(tbnav_rbpf_create(&p, &h_), "bmapping::GridMapper");
  if (reference_field_) check(tbnav_rbpf_set_option(h_, TBNAV_RBPF_OPT_DF_MODE, TBNAV_RBPF_DF_REFERENCE), "GridMapper::useReferenceDistanceField");
  return h_;
}

GridMapper::GridMapper(const GridMapper& o)
    : LaserScanner(o), resolution_(o.resolution_), xmin_(o.xmin_), xmax_(o.xmax_), ymin_(o.ymin_), ymax_(o.ymax_),
      reference_field_(o.reference_field_) {
  if (!o.h_) return;  // a prototype that never mapped: nothing on the device to copy
  check(tbnav_rbpf_copy_particle(handle(), 0, o.h_, 0), "GridMapper copy");
}

GridMapper& GridMapper::operator=(const GridMapper& o) {
  if (this == &o) return *this;
  GridMapper tmp(o);
  LaserScanner::operator=(tmp);
  resolution_ = tmp.resolution_; xmin_ = tmp.xmin_; xmax_ = tmp.xmax_; ymin_ = tmp.ymin_; ymax_ = tmp.ymax_;
  reference_field_ = tmp.reference_field_;
  tbnav_rbpf_destroy(h_);
  h_ = tmp.h_;
  tmp.h_ = nullptr;
  return *this;
}

void GridMapper::useReferenceDistanceField(bool on) {
  reference_field_ = on;
  if (h_) check(tbnav_rbpf_set_option(h_, TBNAV_RBPF_OPT_DF_MODE, on ? TBNAV_RBPF_DF_REFERENCE : TBNAV_RBPF_DF_QUERY), "GridMapper::useReferenceDistanceField");
}

double GridMapper::likelihoodFieldModel(const std::vector<float>& beam_length, const Transform2D& pose) const {
  const auto d = pose.displacement();
  const double ps[3] = {d.theta, d.x, d.y};
  double out = 1.0;
  check(tbnav_rbpf_likelihood(handle(), 0, beam_length.data(), (int32_t)beam_length.size(), ps, &out), "GridMapper::likelihoodFieldModel");
  return out;
}

void GridMapper::integrateScan(const std::vector<float>& beam_length, const Transform2D& pose) {
  const auto d = pose.displacement();
  const double ps[3] = {d.theta, d.x, d.y};
  check(tbnav_rbpf_integrate_scan(handle(), 0, beam_length.data(), (int32_t)beam_length.size(), ps), "GridMapper::integrateScan");
}

void GridMapper::gridMap(std::vector<int8_t>& map) const {
  int32_t xs = 0, ys = 0;
  check(tbnav_rbpf_grid_size(handle(), &xs, &ys), "GridMapper::gridMap");
  map.resize((size_t)xs * ys, 0);
  check(tbnav_rbpf_particle_map(h_, 0, map.data()), "GridMapper::gridMap");
}

}  // namespace bmapping
