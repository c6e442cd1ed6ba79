// ref_harness_pf.cpp — extern "C" access to the REAL bmapping::ParticleFilter, compiled together with the reference's own
// unmodified bmapping/src/bmapping/{particle_filter,grid_mapper,sensor_model}.cpp and rigid2d sources (oracle/Makefile, target
// `ref_pf`).  Needs Eigen 3.3 AND the PCL HEADERS (bmapping/include/bmapping/cloud_alignment.hpp:11-13 includes them): built ONLY
// when the image really has both — no stand-in headers.  What is NOT compiled is the reference's cloud_alignment.cpp (it needs
// the PCL libraries and is third-party arithmetic, an INPUT of the path: SURVEY.md section 8-c): this file defines
// bmapping::ScanAlignment's constructor and pclICPWrapper itself and returns the (ok, T_icp) the test injects, exactly where
// particle_filter.cpp:153 calls it.  TEST INFRASTRUCTURE ONLY; -fno-access-control; cannot be compiled in an image without Eigen / PCL.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "bmapping/particle_filter.hpp"

namespace {
thread_local std::string g_err;
bool g_icp_ok = true;
rigid2d::Transform2D g_icp_T;
}

namespace bmapping {
// the injected scan matcher (replaces cloud_alignment.cpp:21-72; member layout is the header's)
ScanAlignment::ScanAlignment(const LaserProperties& props, const Transform2D& Trs)
    : max_iter_(0), max_correspondence_dist_(0), transform_epsilon_(0), fitness_epsilon_(0), Trs_(Trs), beam_min_(props.beam_min),
      beam_max_(props.beam_max), beam_delta_(props.beam_delta), range_min_(props.range_min), range_max_(props.range_max),
      first_scan_recieved(false) {}
bool ScanAlignment::pclICPWrapper(Transform2D& T, const Transform2D&, const std::vector<float>&) {
  T = g_icp_T;
  return g_icp_ok;
}
}  // namespace bmapping

extern "C" {

const char* refp_last_error() { return g_err.c_str(); }
void refp_seed(uint64_t s) { bmapping::getTwister().seed(s); }   // particle_filter.hpp:39
void refp_set_icp(int ok, const double T_icp[3] /*theta, x, y*/) {
  g_icp_ok = ok != 0;
  g_icp_T = rigid2d::Transform2D(rigid2d::Vector2D(T_icp[1], T_icp[2]), T_icp[0]);
}
// pf = N, k, srr, srt, str, stt, motion_noise[3], sample_range[3], scan_min, scan_max, pose_min, pose_max (16 doubles);
// laser = 5 floats; mix = (z_hit, z_short, z_max, z_rand, sigma_hit); grid = (res, xmin, xmax, ymin, ymax); trs, pose0 = (theta, x, y)
void* refp_create(const double* pf, const float laser[5], const double mix[5], const double grid[5], const double trs[3], const double pose0[3]) {
  try {
    bmapping::LaserProperties props(laser[0], laser[1], laser[2], laser[3], laser[4], mix[0], mix[1], mix[2], mix[3], mix[4]);
    const rigid2d::Transform2D Trs(rigid2d::Vector2D(trs[1], trs[2]), trs[0]);
    bmapping::GridMapper mapper(grid[0], grid[1], grid[2], grid[3], grid[4], props, Trs);
    bmapping::ScanAlignment aligner(props, Trs);
    const rigid2d::Transform2D start(rigid2d::Vector2D(pose0[1], pose0[2]), pose0[0]);
    return new bmapping::ParticleFilter((int)pf[0], (int)pf[1], pf[2], pf[3], pf[4], pf[5], pf[6], pf[7], pf[8], pf[9], pf[10], pf[11],
                                        pf[12], pf[13], pf[14], pf[15], aligner, start, mapper);
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void refp_destroy(void* h) { delete static_cast<bmapping::ParticleFilter*>(h); }
// u = (w, vx, vy); odometry (theta, x, y).  Returns 0, or 1 with the exception text in refp_last_error.
int refp_slam(void* h, const float* scan, int n_beams, const double u[3], const double cur[3], const double prev[3]) {
  try {
    rigid2d::Twist2D tw; tw.w = u[0]; tw.vx = u[1]; tw.vy = u[2];
    rigid2d::Pose c, p; c.theta = cur[0]; c.x = cur[1]; c.y = cur[2]; p.theta = prev[0]; p.x = prev[1]; p.y = prev[2];
    static_cast<bmapping::ParticleFilter*>(h)->SLAM(std::vector<float>(scan, scan + n_beams), tw, c, p);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// pose, prev_pose [N][3] (theta, x, y as the filter stores them, particle_filter.cpp:132-133), weight [N], normal_sqrd_sum_
void refp_get(void* h, double* pose, double* prev_pose, double* weight, double* sq_sum) {
  auto* f = static_cast<bmapping::ParticleFilter*>(h);
  for (size_t i = 0; i < f->particle_set_.size(); ++i) {
    for (int q = 0; q < 3; ++q) { pose[3 * i + q] = f->particle_set_[i].pose(q); prev_pose[3 * i + q] = f->particle_set_[i].prev_pose(q); }
    weight[i] = f->particle_set_[i].weight;
  }
  *sq_sum = f->normal_sqrd_sum_;
}
void refp_set_weights(void* h, const double* weight) {
  auto* f = static_cast<bmapping::ParticleFilter*>(h);
  for (size_t i = 0; i < f->particle_set_.size(); ++i) f->particle_set_[i].weight = weight[i];
}
// log-odds [G] of particle i's map (GridMapper::map_, grid_mapper.hpp)
void refp_log_odds(void* h, int i, double* out) {
  auto* f = static_cast<bmapping::ParticleFilter*>(h);
  const auto& m = f->particle_set_[(size_t)i].grid.map_;
  for (size_t c = 0; c < m.size(); ++c) out[c] = m[c].log_odds;
}
void refp_best_state(void* h, double pose[3]) {
  const auto d = static_cast<bmapping::ParticleFilter*>(h)->getRobotState().displacement();
  pose[0] = d.theta; pose[1] = d.x; pose[2] = d.y;
}
int refp_new_map(void* h, int8_t* out, int cap) {
  std::vector<int8_t> m;
  static_cast<bmapping::ParticleFilter*>(h)->newMap(m);
  if ((int)m.size() > cap) return -1;
  std::memcpy(out, m.data(), m.size());
  return (int)m.size();
}

}  // extern "C"
