// particle_filter_shim.cpp — bmapping::ParticleFilter over the C-ABI (include/tbnav_rbpf.h).
#include <iostream>
#include <stdexcept>
#include <string>

#include "bmapping/particle_filter.hpp"
#include "tbnav_rbpf.h"

namespace bmapping {

std::mt19937_64& getTwister() {
  static std::random_device rd;
  static std::mt19937_64 gen(rd());
  return gen;
}

namespace {
void check(int rc, const char* where) {
  if (rc == TBNAV_OK) return;
  const std::string text = tbnav_status_string(rc);
  switch (rc) {  // what the reference throws (grid_mapper.cpp:22,701,856; particle_filter.cpp:579)
    case TBNAV_ERR_OUT_OF_WORLD:
    case TBNAV_ERR_ETA_ZERO:
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

ParticleFilter::ParticleFilter(int num_particles, int k, double srr, double srt, double str, double stt,
                               double motion_noise_theta, double motion_noise_x, double motion_noise_y,
                               double sample_range_theta, double sample_range_x, double sample_range_y,
                               double scan_likelihood_min, double scan_likelihood_max, double pose_likelihood_min,
                               double pose_likelihood_max, ScanAlignment& scan_matcher, const Transform2D& pose,
                               const GridMapper& mapper, int n_gpus, const std::vector<int>& devices)
    : scan_matcher_(scan_matcher), num_particles_(num_particles), k_(k) {
  tbnav_rbpf_params p{};
  p.num_particles = num_particles; p.num_samples_mode = k;
  p.srr = srr; p.srt = srt; p.str_ = str; p.stt = stt;
  p.motion_noise[0] = motion_noise_theta; p.motion_noise[1] = motion_noise_x; p.motion_noise[2] = motion_noise_y;
  p.sample_range[0] = sample_range_theta; p.sample_range[1] = sample_range_x; p.sample_range[2] = sample_range_y;
  p.scan_likelihood_min = scan_likelihood_min; p.scan_likelihood_max = scan_likelihood_max;
  p.pose_likelihood_min = pose_likelihood_min; p.pose_likelihood_max = pose_likelihood_max;
  const LaserProperties& L = mapper.laser();
  p.beam_min = L.beam_min; p.beam_max = L.beam_max; p.beam_delta = L.beam_delta; p.range_min = L.range_min; p.range_max = L.range_max;
  p.device = -1;
  p.z_hit = L.z_hit; p.z_short = L.z_short; p.z_max = L.z_max; p.z_rand = L.z_rand; p.sigma_hit = L.sigma_hit;
  const auto trs = mapper.robotToLaser().displacement();
  p.Trs[0] = trs.theta; p.Trs[1] = trs.x; p.Trs[2] = trs.y;
  p.resolution = mapper.resolution(); p.xmin = mapper.xmin(); p.xmax = mapper.xmax(); p.ymin = mapper.ymin(); p.ymax = mapper.ymax();
  const auto p0 = pose.displacement();  // initParticleSet, particle_filter.cpp:132-133
  p.pose0[0] = p0.theta; p.pose0[1] = p0.x; p.pose0[2] = p0.y;
  if (n_gpus > 1) {
    if (!devices.empty() && (int)devices.size() != n_gpus) throw std::invalid_argument("bmapping::ParticleFilter: devices.size() != n_gpus");
    check(tbnav_rbpf_group_create(&p, n_gpus, devices.empty() ? nullptr : devices.data(), 0, &g_), "bmapping::ParticleFilter (n_gpus)");
  } else {
    check(tbnav_rbpf_create(&p, &h_), "bmapping::ParticleFilter");
    // the reference's own distance field is the default wherever it can run (header note)
    if (num_particles <= 4096) useReferenceDistanceField(true);
  }
}

ParticleFilter::~ParticleFilter() { tbnav_rbpf_destroy(h_); tbnav_rbpf_group_destroy(g_); }
int ParticleFilter::gpus() const { return g_ ? tbnav_rbpf_group_size(g_) : 1; }

void ParticleFilter::useScanMatching(bool on, double lstep, double astep, int iterations) {
  if (g_) {
    for (int r = 0; r < tbnav_rbpf_group_size(g_); ++r) {
      tbnav_rbpf* m = nullptr;
      check(tbnav_rbpf_group_member(g_, r, &m), "useScanMatching");
      check(tbnav_rbpf_set_scan_matching(m, on ? 1 : 0, lstep, astep, iterations), "useScanMatching");
    }
    return;
  }
  check(tbnav_rbpf_set_scan_matching(h_, on ? 1 : 0, lstep, astep, iterations), "useScanMatching");
}

void ParticleFilter::useReferenceDistanceField(bool on) {
  if (g_) {  // the brushfire reproduction is host work of ONE handle: not available across GPUs
    if (on) throw std::invalid_argument("useReferenceDistanceField: not available with n_gpus > 1");
    return;
  }
  check(tbnav_rbpf_set_option(h_, TBNAV_RBPF_OPT_DF_MODE, on ? TBNAV_RBPF_DF_REFERENCE : TBNAV_RBPF_DF_QUERY), "useReferenceDistanceField");
  reference_field_ = on;
}

void ParticleFilter::useDeviceNoise(std::uint64_t seed) {
  device_noise_ = true;
  check(g_ ? tbnav_rbpf_group_set_seed(g_, seed) : tbnav_rbpf_set_seed(h_, seed), "useDeviceNoise");
}

void ParticleFilter::SLAM(const std::vector<float>& scan, const Twist2D& u, const Pose& cur_odom, const Pose& prev_odom) {
  // icpInitGuess (particle_filter.cpp:602-612): the raw world-frame odometry delta
  const double dth = rigid2d::normalize_angle_PI(rigid2d::normalize_angle_PI(cur_odom.theta) - rigid2d::normalize_angle_PI(prev_odom.theta));
  const Transform2D Tinit(rigid2d::Vector2D(cur_odom.x - prev_odom.x, cur_odom.y - prev_odom.y), dth);
  Transform2D Ticp;
  const bool ok = scan_matcher_.pclICPWrapper(Ticp, Tinit, scan);  // :153, once per scan, host
  const auto t = Ticp.displacement();

  // the standard normals this call consumes, drawn in the reference's order from the filter's own
  // engine: per particle 3k (sampleMode) + 3 (new pose), or 3 (motion model); the resampling offset
  // is drawn only if resampling fires, so the engine is rewound when it does not.
  const int64_t n = g_ ? tbnav_rbpf_group_num_normals(g_, ok ? 1 : 0) : tbnav_rbpf_num_normals(h_, ok ? 1 : 0);
  std::mt19937_64& gen = getTwister();
  std::mt19937_64 before_resample_draw = gen;
  if (!device_noise_) {
    normals_.resize((size_t)n);
    for (int64_t i = 0; i + 1 < n; ++i) {
      std::normal_distribution<double> dis(0, 1);
      normals_[(size_t)i] = dis(gen);
    }
    before_resample_draw = gen;
    std::normal_distribution<double> dis(0, 1);
    normals_[(size_t)n - 1] = dis(gen);
  }
  const double uu[3] = {u.w, u.vx, u.vy};
  const double cur[3] = {cur_odom.theta, cur_odom.x, cur_odom.y};
  const double prev[3] = {prev_odom.theta, prev_odom.x, prev_odom.y};
  const double ticp[3] = {t.theta, t.x, t.y};
  tbnav_rbpf_stats st{};
  const double* nz = device_noise_ ? nullptr : normals_.data();
  const int rc = g_ ? tbnav_rbpf_group_slam(g_, scan.data(), (int32_t)scan.size(), uu, cur, prev, ok ? 1 : 0, ticp, nz, &st)
                    : tbnav_rbpf_slam(h_, scan.data(), (int32_t)scan.size(), uu, cur, prev, ok ? 1 : 0, ticp, nz, &st);
  if (rc != TBNAV_OK || !st.resampled) gen = before_resample_draw;
  check(rc, "ParticleFilter::SLAM");
  last_neff_ = st.neff;
  last_resampled_ = st.resampled != 0;
  std::cout << "Neff: " << st.neff << std::endl;          // particle_filter.cpp:463
  if (st.resampled) std::cout << "Resampling" << std::endl;  // :247
}

Transform2D ParticleFilter::getRobotState() {
  double pose[3];
  check(g_ ? tbnav_rbpf_group_best_state(g_, pose, nullptr) : tbnav_rbpf_best_state(h_, pose, nullptr), "getRobotState");
  return Transform2D(rigid2d::Vector2D(pose[1], pose[2]), pose[0]);
}

void ParticleFilter::newMap(std::vector<int8_t>& map) {
  int32_t xs = 0, ys = 0;
  tbnav_rbpf* any = h_;
  if (g_) check(tbnav_rbpf_group_member(g_, 0, &any), "newMap");
  check(tbnav_rbpf_grid_size(any, &xs, &ys), "newMap");
  map.resize((size_t)xs * ys, 0);
  check(g_ ? tbnav_rbpf_group_best_map(g_, map.data()) : tbnav_rbpf_best_map(h_, map.data()), "newMap");
}

}  // namespace bmapping
