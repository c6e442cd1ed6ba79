// ref_harness.cpp — extern "C" access to the REAL reference classes, compiled together with the
// reference's own unmodified sources where they lie under /root/reference (see oracle/Makefile,
// target `ref`).  TEST INFRASTRUCTURE ONLY; output goes to oracle/_ref/ (git-ignored).
//
// Built with -fno-access-control so the harness can read GridMapper's private map_/occ_cells_ and
// call its private helpers without touching the reference headers.
//
// Covered (compile as-is with g++): rigid2d::{Transform2D, DiffDrive, normalize_angle_PI},
// bmapping::{LaserScanner, GridMapper, pdfNormal}.  NOT covered (need Eigen / PCL, absent here):
// controller::MPPI, controller::RK4, bmapping::ParticleFilter, bmapping::ScanAlignment.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "bmapping/grid_mapper.hpp"
#include "bmapping/sensor_model.hpp"
#include "rigid2d/diff_drive.hpp"
#include "rigid2d/rigid2d.hpp"

using bmapping::GridMapper;
using bmapping::LaserProperties;
using rigid2d::Transform2D;
using rigid2d::Twist2D;
using rigid2d::Vector2D;

namespace {
thread_local std::string g_err;
Transform2D make_T(const double pose[3]) {  // pose = (theta, x, y)
  return Transform2D(Vector2D(pose[1], pose[2]), pose[0]);
}
}  // namespace

extern "C" {

const char* ref_last_error() { return g_err.c_str(); }

// ---- rigid2d -------------------------------------------------------------------------------------
double ref_normalize_angle_PI(double rad) { return rigid2d::normalize_angle_PI(rad); }
int ref_almost_equal(double a, double b, double eps) { return rigid2d::almost_equal(a, b, eps) ? 1 : 0; }

// out = (theta, x, y, ctheta, stheta)
static void dump_T(const Transform2D& T, double out[5]) {
  out[0] = T.theta; out[1] = T.x; out[2] = T.y; out[3] = T.ctheta; out[4] = T.stheta;
}
void ref_transform_make(const double pose[3], double out[5]) { dump_T(make_T(pose), out); }
void ref_transform_compose(const double a[3], const double b[3], double out[5]) {
  Transform2D T = make_T(a);
  T *= make_T(b);
  dump_T(T, out);
}
void ref_transform_apply(const double a[3], const double v[2], double out[2]) {
  Vector2D r = make_T(a)(Vector2D(v[0], v[1]));
  out[0] = r.x; out[1] = r.y;
}
void ref_transform_inv(const double a[3], double out[5]) { dump_T(make_T(a).inv(), out); }
void ref_transform_integrate_twist(const double a[3], const double tw[3], double out[5]) {
  Twist2D t; t.w = tw[0]; t.vx = tw[1]; t.vy = tw[2];
  dump_T(make_T(a).integrateTwist(t), out);
}

// ---- DiffDrive -----------------------------------------------------------------------------------
void* ref_dd_create(const double pose[3], double wheel_base, double wheel_radius) {
  rigid2d::Pose p; p.theta = pose[0]; p.x = pose[1]; p.y = pose[2];
  return new rigid2d::DiffDrive(p, wheel_base, wheel_radius);
}
void ref_dd_destroy(void* d) { delete static_cast<rigid2d::DiffDrive*>(d); }
int ref_dd_twist_to_wheels(void* d, const double tw[3], double out[2]) {
  try {
    Twist2D t; t.w = tw[0]; t.vx = tw[1]; t.vy = tw[2];
    auto v = static_cast<rigid2d::DiffDrive*>(d)->twistToWheels(t);
    out[0] = v.ul; out[1] = v.ur;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void ref_dd_wheels_to_twist(void* d, const double w[2], double out[3]) {
  rigid2d::WheelVelocities v; v.ul = w[0]; v.ur = w[1];
  auto t = static_cast<rigid2d::DiffDrive*>(d)->wheelsToTwist(v);
  out[0] = t.w; out[1] = t.vx; out[2] = t.vy;
}
void ref_dd_update_odometry(void* d, double left, double right, double out[2]) {
  auto v = static_cast<rigid2d::DiffDrive*>(d)->updateOdometry(left, right);
  out[0] = v.ul; out[1] = v.ur;
}
int ref_dd_feedforward(void* d, const double tw[3]) {
  try {
    Twist2D t; t.w = tw[0]; t.vx = tw[1]; t.vy = tw[2];
    static_cast<rigid2d::DiffDrive*>(d)->feedforward(t);
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// One plant step with the reference's own DiffDrive (same contract as orc_dd_arc_step): pose is (x, y, theta) in/out.
int ref_dd_arc_step(double wheel_base, double wheel_radius, double dt, double pose_xyt[3], const double wheels[2]) {
  try {
    rigid2d::Pose p; p.theta = pose_xyt[2]; p.x = pose_xyt[0]; p.y = pose_xyt[1];
    rigid2d::DiffDrive d(p, wheel_base, wheel_radius);
    rigid2d::WheelVelocities v; v.ul = wheels[0]; v.ur = wheels[1];
    Twist2D t = d.wheelsToTwist(v);
    t.w = t.w * dt; t.vx = t.vx * dt; t.vy = t.vy * dt;
    d.feedforward(t);
    const auto q = d.pose();
    pose_xyt[0] = q.x; pose_xyt[1] = q.y; pose_xyt[2] = q.theta;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// out = (theta, x, y, enc_left, enc_right, ul, ur)
void ref_dd_state(void* d, double out[7]) {
  auto* dd = static_cast<rigid2d::DiffDrive*>(d);
  auto p = dd->pose(); auto e = dd->getEncoders(); auto w = dd->wheelVelocities();
  out[0] = p.theta; out[1] = p.x; out[2] = p.y; out[3] = e.left; out[4] = e.right; out[5] = w.ul; out[6] = w.ur;
}

// ---- GridMapper / LaserScanner ---------------------------------------------------------------------
double ref_pdf_normal(double a, double b, int* err) {
  try { *err = 0; return bmapping::pdfNormal(a, b); }
  catch (const std::exception& e) { g_err = e.what(); *err = 1; return 0.0; }
}
double ref_log_odds_to_prob(double l) { return bmapping::logOdds2Prob(l); }
double ref_prob_to_log_odds(double p) { return bmapping::prob2LogOdds(p); }

// laser = (beam_min, beam_max, beam_delta, range_min, range_max) floats; mix = (z_hit, z_short,
// z_max, z_rand, sigma_hit); grid = (resolution, xmin, xmax, ymin, ymax); trs = (theta, x, y)
void* ref_gm_create(const double grid[5], const float laser[5], const double mix[5], const double trs[3]) {
  LaserProperties props(laser[0], laser[1], laser[2], laser[3], laser[4], mix[0], mix[1], mix[2], mix[3], mix[4]);
  return new GridMapper(grid[0], grid[1], grid[2], grid[3], grid[4], props, make_T(trs));
}
void* ref_gm_clone(void* g) { return new GridMapper(*static_cast<GridMapper*>(g)); }
void ref_gm_destroy(void* g) { delete static_cast<GridMapper*>(g); }
void ref_gm_size(void* g, int* xsize, int* ysize) {
  auto* gm = static_cast<GridMapper*>(g);
  *xsize = gm->xsize_; *ysize = gm->ysize_;
}
// constants derived in the ctor: (l_prior, l_occ, l_free, cell_radius)
void ref_gm_constants(void* g, double out[4]) {
  auto* gm = static_cast<GridMapper*>(g);
  out[0] = gm->log_odds_prior_; out[1] = gm->log_odds_occ_; out[2] = gm->log_odds_free_; out[3] = gm->cell_radius_;
}
int ref_gm_integrate_scan(void* g, const float* scan, int n, const double pose[3]) {
  try {
    std::vector<float> s(scan, scan + n);
    static_cast<GridMapper*>(g)->integrateScan(s, make_T(pose));
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
double ref_gm_likelihood(void* g, const float* scan, int n, const double pose[3], int* err) {
  try {
    std::vector<float> s(scan, scan + n);
    *err = 0;
    return static_cast<GridMapper*>(g)->likelihoodFieldModel(s, make_T(pose));
  } catch (const std::exception& e) { g_err = e.what(); *err = 1; return 0.0; }
}
// per-cell dump; any pointer may be null
void ref_gm_dump(void* g, double* log_odds, double* prob, double* occ_dist, int32_t* state) {
  auto* gm = static_cast<GridMapper*>(g);
  const size_t G = gm->map_.size();
  for (size_t c = 0; c < G; ++c) {
    if (log_odds) log_odds[c] = gm->map_[c].log_odds;
    if (prob) prob[c] = gm->map_[c].prob;
    if (occ_dist) occ_dist[c] = gm->map_[c].occ_dist;
    if (state) state[c] = gm->map_[c].state;
  }
}
// occupied set in the unordered_set's iteration order; returns its size
int ref_gm_occ_cells(void* g, int32_t* out, int cap) {
  auto* gm = static_cast<GridMapper*>(g);
  int n = 0;
  for (int key : gm->occ_cells_) { if (n < cap && out) out[n] = key; ++n; }
  return n;
}
void ref_gm_grid_map(void* g, int8_t* out) {
  std::vector<int8_t> m;
  static_cast<GridMapper*>(g)->gridMap(m);
  std::memcpy(out, m.data(), m.size());
}
// laserEndPoints: returns the number of valid beams; xy = [n][2]
int ref_gm_end_points(void* g, const float* scan, int n, const double pose[3], double* xy) {
  std::vector<float> s(scan, scan + n);
  std::vector<Vector2D> pts;
  static_cast<GridMapper*>(g)->laserEndPoints(pts, s, make_T(pose));
  for (size_t i = 0; i < pts.size(); ++i) { xy[2 * i] = pts[i].x; xy[2 * i + 1] = pts[i].y; }
  return (int)pts.size();
}
int64_t ref_gm_world2rowmajor(void* g, double x, double y) {
  try { return (int64_t) static_cast<GridMapper*>(g)->world2RowMajor(x, y); }
  catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// freeGridIndex for one beam end point; returns count or -1 on throw
int ref_gm_free_index(void* g, const double point[2], const double pose[3], int32_t* out, int cap) {
  try {
    std::vector<int> idx;
    static_cast<GridMapper*>(g)->freeGridIndex(idx, Vector2D(point[0], point[1]), make_T(pose));
    for (size_t i = 0; i < idx.size() && (int)i < cap; ++i) out[i] = idx[i];
    return (int)idx.size();
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// Bresenham pieces on raw grid coordinates (private helpers), for octant-by-octant fixtures
int ref_gm_line_cells(void* g, int which, int x0, int y0, int x1, int y1, int32_t* out, int cap) {
  std::vector<int> idx;
  auto* gm = static_cast<GridMapper*>(g);
  if (which == 0) gm->lineLow(idx, x0, y0, x1, y1);
  else if (which == 1) gm->lineHigh(idx, x0, y0, x1, y1);
  else gm->lineDiag(idx, x0, y0, x1, y1);
  for (size_t i = 0; i < idx.size() && (int)i < cap; ++i) out[i] = idx[i];
  return (int)idx.size();
}

}  // extern "C"
