// test_hooks.cpp — extern "C" access to the C++ class shims so that tests/ can drive them through
// ctypes: the rigid2d layer is compared bit-exactly with the reference build, and the two class
// surfaces (controller::MPPI, bmapping::ParticleFilter) are exercised end to end on the GPU exactly
// the way the ROS nodes call them.  Same argument conventions as oracle/ref_harness.cpp.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "bmapping/grid_mapper.hpp"
#include "bmapping/particle_filter.hpp"
#include "controller/mppi.hpp"
#include "rigid2d/diff_drive.hpp"
#include "rigid2d/rigid2d.hpp"
#include "rigid2d/utilities.hpp"

using rigid2d::Transform2D;
using rigid2d::Twist2D;
using rigid2d::Vector2D;

namespace {
thread_local std::string g_err;
Transform2D make_T(const double p[3]) { return Transform2D(Vector2D(p[1], p[2]), p[0]); }
void dump_T(const Transform2D& T, double out[5]) {
  const auto d = T.displacement();
  out[0] = d.theta; out[1] = d.x; out[2] = d.y;
  const Vector2D ex = T(Vector2D(1.0, 0.0)) - T(Vector2D(0.0, 0.0));  // (cos, sin) as the transform applies them
  out[3] = ex.x; out[4] = ex.y;
}
}  // namespace

extern "C" {

const char* hst_last_error() { return g_err.c_str(); }
double hst_normalize_angle_PI(double r) { return rigid2d::normalize_angle_PI(r); }
void hst_transform_compose(const double a[3], const double b[3], double out[5]) { Transform2D T = make_T(a); T *= make_T(b); dump_T(T, out); }
void hst_transform_apply(const double a[3], const double v[2], double out[2]) { const Vector2D r = make_T(a)(Vector2D(v[0], v[1])); out[0] = r.x; out[1] = r.y; }
void hst_transform_inv(const double a[3], double out[5]) { dump_T(make_T(a).inv(), out); }
void hst_transform_integrate_twist(const double a[3], const double tw[3], double out[5]) {
  Twist2D t; t.w = tw[0]; t.vx = tw[1]; t.vy = tw[2];
  dump_T(make_T(a).integrateTwist(t), out);
}
int hst_transform_print(const double a[3], char* buf, int cap) {
  std::ostringstream os; os << make_T(a);
  const std::string s = os.str();
  std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0;
  return (int)s.size();
}

void* hst_dd_create(const double pose[3], double wheel_base, double wheel_radius) {
  rigid2d::Pose p; p.theta = pose[0]; p.x = pose[1]; p.y = pose[2];
  return new rigid2d::DiffDrive(p, wheel_base, wheel_radius);
}
void hst_dd_destroy(void* d) { delete static_cast<rigid2d::DiffDrive*>(d); }
int hst_dd_twist_to_wheels(void* d, const double tw[3], double out[2]) {
  try {
    Twist2D t; t.w = tw[0]; t.vx = tw[1]; t.vy = tw[2];
    const auto v = static_cast<rigid2d::DiffDrive*>(d)->twistToWheels(t);
    out[0] = v.ul; out[1] = v.ur; return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void hst_dd_wheels_to_twist(void* d, const double w[2], double out[3]) {
  rigid2d::WheelVelocities v; v.ul = w[0]; v.ur = w[1];
  const auto t = static_cast<rigid2d::DiffDrive*>(d)->wheelsToTwist(v);
  out[0] = t.w; out[1] = t.vx; out[2] = t.vy;
}
void hst_dd_update_odometry(void* d, double left, double right, double out[2]) {
  const auto v = static_cast<rigid2d::DiffDrive*>(d)->updateOdometry(left, right);
  out[0] = v.ul; out[1] = v.ur;
}
int hst_dd_feedforward(void* d, const double tw[3]) {
  try { Twist2D t; t.w = tw[0]; t.vx = tw[1]; t.vy = tw[2]; static_cast<rigid2d::DiffDrive*>(d)->feedforward(t); return 0; }
  catch (const std::exception& e) { g_err = e.what(); return 1; }
}
void hst_dd_state(void* d, double out[7]) {
  auto* dd = static_cast<rigid2d::DiffDrive*>(d);
  const auto p = dd->pose(); const auto e = dd->getEncoders(); const auto w = dd->wheelVelocities();
  out[0] = p.theta; out[1] = p.x; out[2] = p.y; out[3] = e.left; out[4] = e.right; out[5] = w.ul; out[6] = w.ur;
}

// the host twister draws exactly like oracle/orc_normal_stream (fresh distribution per draw)
void hst_twister_stream(uint64_t seed, int64_t n, double mu, double sigma, double* out) {
  rigid2d::getTwister().seed(seed);
  for (int64_t i = 0; i < n; ++i) out[i] = rigid2d::sampleNormalDistribution(mu, sigma);
}

// ---- controller::MPPI through its class surface (needs a GPU) --------------------------------------
// params = (wheel_radius, wheel_base, lambda, max_wheel_vel, ul_var, ur_var, horizon, dt, Q0..2, R0..1, P0..2)
// Closed loop of nuturtle_robot/src/mppi_waypoints_node.cpp:226-305 without ROS: newControls ->
// DiffDrive::wheelsToTwist -> plant DiffDrive::feedforward(twist / rate) -> pose -> waypoint switch at
// goal_thresh.  traj_out: [n_ticks][5] = (x, y, theta, ul, ur).  Returns ticks run or -1.
// odometry_mode 0: the controller sees the plant's pose directly.
// odometry_mode 1: the full chain of the demo: the plant's wheel encoders, wrapped to
//   [-pi, pi) as fake_diff_encoders publishes them (fake_diff_encoders_node.cpp:100-144), feed a second DiffDrive
//   through updateOdometry (odometry_node.cpp:169-253); the controller sees THAT pose.
// odometry_mode bit 1 (value 2): the controller integrates its rollouts with exact arcs (useExactArcDynamics, N4).
int hst_mppi_closed_loop(const double* params, int rollouts, uint64_t seed, const double* waypoints /*[n][3] x,y,theta*/,
                         int n_wpts, double goal_thresh, double rate, int max_ticks, double* traj_out, int* wpts_reached,
                         int odometry_mode, double* max_odom_dev) {
  double dev = 0.0;
  try {
    controller::CartModel cart(params[0], params[1]);
    controller::LossFunc loss({params[8], params[9], params[10]}, {params[11], params[12]}, {params[13], params[14], params[15]});
    controller::MPPI mppi(cart, loss, params[2], params[3], params[4], params[5], params[6], params[7], rollouts);
    if (odometry_mode & 2) mppi.useExactArcDynamics(true);
    odometry_mode &= 1;
    rigid2d::getTwister().seed(seed);
    rigid2d::Pose start; start.x = waypoints[0]; start.y = waypoints[1]; start.theta = waypoints[2];
    rigid2d::DiffDrive plant(start, params[1], params[0]);
    rigid2d::DiffDrive odometer(start, params[1], params[0]);
    const rigid2d::DiffDrive model(start, params[1], params[0]);
    mppi.setInitialControls(0.0, 0.0);
    int target = 1 % n_wpts, reached = 0, tick = 0;
    auto set_wpt = [&](int i) { rigid2d::Pose w; w.x = waypoints[3 * i]; w.y = waypoints[3 * i + 1]; w.theta = waypoints[3 * i + 2]; mppi.setWaypoint(w); };
    set_wpt(target);
    for (; tick < max_ticks; ++tick) {
      const rigid2d::Pose ps = odometry_mode ? odometer.pose() : plant.pose();
      if (rigid2d::euclideanDistance(ps.x, ps.y, waypoints[3 * target], waypoints[3 * target + 1]) < goal_thresh) {
        ++reached;
        target = (target + 1) % n_wpts;
        set_wpt(target);
        if (reached >= n_wpts) break;
      }
      const rigid2d::WheelVelocities u = mppi.newControls(ps);
      Twist2D cmd = model.wheelsToTwist(u);
      cmd.w /= rate; cmd.vx /= rate; cmd.vy = 0.0;
      plant.feedforward(cmd);
      if (odometry_mode) {
        const rigid2d::WheelEncoders enc = plant.getEncoders();  // already wrapped by feedforward (diff_drive.cpp:166-167)
        odometer.updateOdometry(enc.left, enc.right);
        const rigid2d::Pose po = odometer.pose(), pp = plant.pose();
        dev = std::max(dev, std::max(std::fabs(po.x - pp.x), std::max(std::fabs(po.y - pp.y),
                                     std::fabs(rigid2d::normalize_angle_PI(po.theta - pp.theta)))));
      }
      const rigid2d::Pose np = plant.pose();
      double* row = traj_out + (size_t)tick * 5;
      row[0] = np.x; row[1] = np.y; row[2] = np.theta; row[3] = u.ul; row[4] = u.ur;
    }
    *wpts_reached = reached;
    if (max_odom_dev) *max_odom_dev = dev;
    return tick;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// next hst_mppi_tick: controller::MPPI built with n_gpus = n, every member on device 0 (a one-GPU box: the members then
// exchange by copies instead of RCCL — same records, same layout, same stream order); 0 / 1 = the reference's nine arguments
static int g_mppi_gpus = 1;
void hst_mppi_gpus(int n) { g_mppi_gpus = n < 1 ? 1 : n; }
// on: the members of the next n_gpus > 1 object go on devices 0, 1, ... n-1 (a multi-GPU box: RCCL / peer stores between them)
// instead of device 0 n times
static int g_distinct_devices = 0;
void hst_distinct_devices(int on) { g_distinct_devices = on ? 1 : 0; }
static std::vector<int> member_devices(int n) {
  std::vector<int> d(n > 1 ? n : 0, 0);
  if (g_distinct_devices) for (int i = 0; i < (int)d.size(); ++i) d[i] = i;
  return d;
}
// One MPPI tick through the class with the host twister seeded: returns (ul, ur) and u[2][T].
int hst_mppi_tick(const double* params, int rollouts, uint64_t seed, const double wpt[3] /*x,y,theta*/, const double pose[3] /*theta,x,y*/,
                  int n_ticks, double* out_ul_ur, double* u_out) {
  try {
    controller::CartModel cart(params[0], params[1]);
    controller::LossFunc loss({params[8], params[9], params[10]}, {params[11], params[12]}, {params[13], params[14], params[15]});
    controller::MPPI mppi(cart, loss, params[2], params[3], params[4], params[5], params[6], params[7], rollouts, g_mppi_gpus,
                          member_devices(g_mppi_gpus));
    rigid2d::getTwister().seed(seed);
    rigid2d::Pose w; w.x = wpt[0]; w.y = wpt[1]; w.theta = wpt[2];
    mppi.setWaypoint(w);
    rigid2d::Pose ps; ps.theta = pose[0]; ps.x = pose[1]; ps.y = pose[2];
    for (int t = 0; t < n_ticks; ++t) {
      const auto u = mppi.newControls(ps);
      out_ul_ur[2 * t] = u.ul; out_ul_ur[2 * t + 1] = u.ur;
    }
    const auto uu = mppi.controls();
    std::memcpy(u_out, uu.data(), sizeof(double) * uu.size());
    return mppi.steps();
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// ---- bmapping::ParticleFilter through its class surface (needs a GPU) -------------------------------
// Shipped parameters (slam.launch) with N, k, map half-size given; runs n_scans of SLAM() on the given
// scans/odometry with the filter's twister seeded; out_pose [n_scans][3] = getRobotState per scan,
// out_neff [n_scans]; map_out = newMap() after the last scan.  Returns xsize or -1 (message in hst_last_error).
static int g_pf_reference_field = 1, g_pf_gpus = 1;
void hst_pf_reference_field(int on) { g_pf_reference_field = on; }  // next hst_pf_run: 1 = the class's default (the reference's field), 0 = useExactDistanceField()
void hst_pf_gpus(int n) { g_pf_gpus = n < 1 ? 1 : n; }              // next hst_pf_run: ParticleFilter(..., n_gpus = n), every member on device 0

int hst_pf_run(int N, int k, double map_half, uint64_t seed, const float* scans, int n_beams, int n_scans,
               const double* odom /*[n_scans+1][3] theta,x,y: odom[s] = prev, odom[s+1] = cur*/, double* out_pose,
               int* out_neff, int8_t* map_out) {
  try {
    const double d2r = rigid2d::PI / 180.0;
    bmapping::LaserProperties props((float)(0.0 * d2r), (float)(360.0 * d2r), (float)(1.0 * d2r), 0.12f, 3.5f, 0.95, 0.0, 0.04, 0.01, 0.5);
    Transform2D Trs;
    bmapping::GridMapper grid(0.05, -map_half, map_half, -map_half, map_half, props, Trs);
    bmapping::ScanAlignment aligner(props, Trs);
    // a converged ICP on this world returns the body-frame odometry increment inv(T_prev) * T_cur
    Transform2D icp_result;
    aligner.setMatcher([&](Transform2D& T, const Transform2D&, const std::vector<float>&, const std::vector<float>&) { T = icp_result; return true; });
    Transform2D start(Vector2D(odom[1], odom[2]), odom[0]);
    bmapping::ParticleFilter pf(N, k, 0.1, 0.2, 0.1, 0.2, 1e-10, 1e-10, 1e-10, 1e-10, 1e-8, 1e-8, 1.0, 20.0, 1.0, 10.0, aligner, start, grid,
                                g_pf_gpus, member_devices(g_pf_gpus));
    if (g_pf_gpus == 1 && pf.referenceDistanceField() != true) throw std::runtime_error("the class's default is the reference's distance field");
    if (!g_pf_reference_field) pf.useExactDistanceField();
    bmapping::getTwister().seed(seed);
    for (int s = 0; s < n_scans; ++s) {
      std::vector<float> scan(scans + (size_t)s * n_beams, scans + (size_t)(s + 1) * n_beams);
      rigid2d::Pose prev, cur;
      prev.theta = odom[3 * s]; prev.x = odom[3 * s + 1]; prev.y = odom[3 * s + 2];
      cur.theta = odom[3 * (s + 1)]; cur.x = odom[3 * (s + 1) + 1]; cur.y = odom[3 * (s + 1) + 2];
      Twist2D u;  // unused on the ICP-ok branch
      icp_result = Transform2D(Vector2D(prev.x, prev.y), prev.theta).inv() * Transform2D(Vector2D(cur.x, cur.y), cur.theta);
      pf.SLAM(scan, u, cur, prev);
      const auto d = pf.getRobotState().displacement();
      out_pose[3 * s] = d.theta; out_pose[3 * s + 1] = d.x; out_pose[3 * s + 2] = d.y;
      out_neff[s] = pf.effectiveParticles();
    }
    std::vector<int8_t> map;
    pf.newMap(map);
    std::memcpy(map_out, map.data(), map.size());
    int xs = 0; while ((size_t)xs * xs < map.size()) ++xs;
    return xs;
  } catch (const std::exception& e) { g_err = e.what(); return -1; }
}


// ---- bmapping::GridMapper / LaserScanner (host class over a one-particle handle) ---------------------------------
// grid = (res, xmin, xmax, ymin, ymax); laser = 5 floats (beam_min, beam_max, beam_delta, range_min, range_max);
// mix = (z_hit, z_short, z_max, z_rand, sigma_hit); trs / pose = (theta, x, y)
void* hst_gm_create(const double grid[5], const float laser[5], const double mix[5], const double trs[3], int reference_field) {
  try {
    bmapping::LaserProperties props(laser[0], laser[1], laser[2], laser[3], laser[4], mix[0], mix[1], mix[2], mix[3], mix[4]);
    auto* g = new bmapping::GridMapper(grid[0], grid[1], grid[2], grid[3], grid[4], props, make_T(trs));
    g->useReferenceDistanceField(reference_field != 0);  // (the class's default is true)
    return g;
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void hst_gm_destroy(void* g) { delete static_cast<bmapping::GridMapper*>(g); }
void* hst_gm_clone(void* g) {
  try { return new bmapping::GridMapper(*static_cast<bmapping::GridMapper*>(g)); }
  catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
int hst_gm_integrate_scan(void* g, const float* scan, int n, const double pose[3]) {
  try { static_cast<bmapping::GridMapper*>(g)->integrateScan(std::vector<float>(scan, scan + n), make_T(pose)); return 0; }
  catch (const std::invalid_argument& e) { g_err = e.what(); return 1; }
  catch (const std::exception& e) { g_err = e.what(); return 2; }
}
int hst_gm_likelihood(void* g, const float* scan, int n, const double pose[3], double* out) {
  try { *out = static_cast<const bmapping::GridMapper*>(g)->likelihoodFieldModel(std::vector<float>(scan, scan + n), make_T(pose)); return 0; }
  catch (const std::invalid_argument& e) { g_err = e.what(); return 1; }
  catch (const std::exception& e) { g_err = e.what(); return 2; }
}
int hst_gm_grid_map(void* g, int8_t* out, int cap) {
  try {
    std::vector<int8_t> m;
    static_cast<const bmapping::GridMapper*>(g)->gridMap(m);
    if ((int)m.size() > cap) return -1;
    std::memcpy(out, m.data(), m.size());
    return (int)m.size();
  } catch (const std::exception& e) { g_err = e.what(); return -2; }
}
int hst_gm_end_points(void* g, const float* scan, int n, const double pose[3], double* xy /*[n][2]*/) {
  std::vector<Vector2D> pts;
  const bmapping::LaserScanner* s = static_cast<const bmapping::GridMapper*>(g);
  s->laserEndPoints(pts, std::vector<float>(scan, scan + n), make_T(pose));
  for (size_t i = 0; i < pts.size(); ++i) { xy[2 * i] = pts[i].x; xy[2 * i + 1] = pts[i].y; }
  return (int)pts.size() == (int)s->numberValidMeasurements(std::vector<float>(scan, scan + n)) ? (int)pts.size() : -1;
}

}  // extern "C"
