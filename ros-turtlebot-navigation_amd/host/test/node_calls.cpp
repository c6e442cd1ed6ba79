// node_calls.cpp — COMPILE-ONLY translation unit: every call the reference's two ROS nodes make on the class
// surfaces of the hot paths, spelled the way the nodes spell them, against THIS build's headers.  If this compiles
// (host/Makefile target `node_calls`, run by the CPU test suite), the nodes' source compiles unchanged.
//   bmapping/src/turtle_mapping_node.cpp:389-410 (construction), :459-479 (odometry + SLAM + newMap), :494 (getRobotState)
//   nuturtle_robot/src/mppi_waypoints_node.cpp:186-199 (construction), :216 / :257 (setWaypoint), :238 (distance),
//   :265 (newControls), :276 (wheelsToTwist)
// Nothing here runs; ROS types are replaced by plain locals.
#include <cstdint>
#include <vector>

#include "bmapping/cloud_alignment.hpp"
#include "bmapping/grid_mapper.hpp"
#include "bmapping/particle_filter.hpp"
#include "bmapping/sensor_model.hpp"
#include "controller/mppi.hpp"
#include "rigid2d/diff_drive.hpp"
#include "rigid2d/rigid2d.hpp"
#include "rigid2d/utilities.hpp"

using bmapping::GridMapper;
using bmapping::LaserProperties;
using bmapping::LaserScanner;
using bmapping::ParticleFilter;
using bmapping::ScanAlignment;
using rigid2d::Transform2D;
using rigid2d::Vector2D;

void turtle_mapping_node_calls() {
  float beam_min = 0, beam_max = 6.28f, beam_delta = 0.0174f, range_min = 0.12f, range_max = 3.5f;
  double z_hit = 0.95, z_short = 0.0, z_max = 0.04, z_rand = 0.01, sigma_hit = 0.5;
  double map_resolution = 0.05, map_min = -2.0, map_max = 2.0;
  int num_particles = 40, num_samples_mode = 50;
  double srr = 0.1, srt = 0.2, str = 0.1, stt = 0.2, motion_noise_theta = 1e-10, motion_noise_x = 1e-10, motion_noise_y = 1e-10;
  double sample_range_theta = 1e-10, sample_range_x = 1e-8, sample_range_y = 1e-8;
  double scan_likelihood_min = 1, scan_likelihood_max = 20, pose_likelihood_min = 1, pose_likelihood_max = 10;
  double wheel_base = 0.16, wheel_radius = 0.033, left = 0.0, right = 0.0;
  std::vector<float> scan(360, 1.0f);
  std::vector<int8_t> map;
  Transform2D robot_pose;

  Transform2D Trs;                                                                           // :389
  LaserProperties props(beam_min, beam_max, beam_delta, range_min, range_max, z_hit, z_short, z_max, z_rand, sigma_hit);  // :392-393
  GridMapper grid(map_resolution, map_min, map_max, map_min, map_max, props, Trs);            // :397
  ScanAlignment aligner(props, Trs);                                                          // :400
  ParticleFilter pf(num_particles, num_samples_mode, srr, srt, str, stt, motion_noise_theta, motion_noise_x, motion_noise_y,
                    sample_range_theta, sample_range_x, sample_range_y, scan_likelihood_min, scan_likelihood_max,
                    pose_likelihood_min, pose_likelihood_max, aligner, robot_pose, grid);     // :404-410

  rigid2d::Pose pose, cur_odom, prev_odom;
  rigid2d::DiffDrive drive(pose, wheel_base, wheel_radius), pf_drive(pose, wheel_base, wheel_radius);
  drive.updateOdometry(left, right);                                                          // :459
  pose = drive.pose();
  pf_drive.updateOdometry(left, right);                                                       // :468
  cur_odom = pf_drive.pose();
  rigid2d::WheelVelocities vel = pf_drive.wheelVelocities();
  rigid2d::Twist2D vb = pf_drive.wheelsToTwist(vel);                                          // :472
  pf.SLAM(scan, vb, cur_odom, prev_odom);                                                     // :474
  pf.newMap(map);                                                                             // :479
  prev_odom = cur_odom;
  Transform2D Tmr = pf.getRobotState();                                                       // :494
  Vector2D vor(pose.x, pose.y);
  Transform2D Tor(vor, pose.theta);
  Transform2D Tmo = Tmr * Tor.inv();
  (void)Tmo.displacement();

  // the rest of GridMapper's / LaserScanner's public surface (grid_mapper.hpp:128-140, sensor_model.hpp:93-109)
  const GridMapper& cgrid = grid;
  double p = cgrid.likelihoodFieldModel(scan, robot_pose);
  grid.integrateScan(scan, robot_pose);
  cgrid.gridMap(map);
  std::vector<Vector2D> end_points;
  const LaserScanner& scanner = grid;
  scanner.laserEndPoints(end_points, scan, robot_pose);
  unsigned int n = scanner.numberValidMeasurements(scan);
  (void)p; (void)n; (void)grid.z_hit_; (void)grid.sigma_hit_;
  GridMapper copy = grid;  // value semantics (particle_filter.cpp:125-138)
  (void)copy;
}

void mppi_waypoints_node_calls() {
  double wheel_radius = 0.033, wheel_base = 0.16, lambda = 0.01, max_rot_motor = 6.35495, ul_var = 0.9, ur_var = 0.9;
  double horizon = 1.0, time_step = 0.01, ul_init = 0.0, ur_init = 0.0, goal_thresh = 0.05;
  int rollouts = 5;
  std::vector<double> Q{1e4, 1e4, 1.0}, R{0.1, 0.1}, P1{1e3, 1e3, 1e3};
  rigid2d::Pose pose;

  controller::CartModel cart_model(wheel_radius, wheel_base);                                 // :186
  controller::LossFunc loss_func(Q, R, P1);                                                   // :187
  controller::MPPI mppi(cart_model, loss_func, lambda, max_rot_motor, ul_var, ur_var, horizon, time_step, rollouts);  // :188-196
  mppi.setInitialControls(ul_init, ur_init);                                                  // :199
  rigid2d::DiffDrive diff_drive(pose, wheel_base, wheel_radius);                              // :205
  rigid2d::Pose wpt;
  wpt.x = 1.0; wpt.y = 0.0; wpt.theta = 1.5707;
  mppi.setWaypoint(wpt);                                                                      // :216, :257
  const auto d2g = rigid2d::euclideanDistance(wpt.x, wpt.y, pose.x, pose.y);                  // :238
  if (d2g < goal_thresh) mppi.setWaypoint(wpt);
  rigid2d::WheelVelocities wheel_vel = mppi.newControls(pose);                                // :265
  rigid2d::Twist2D cmd = diff_drive.wheelsToTwist(wheel_vel);                                 // :276
  (void)cmd.vx; (void)cmd.w;
}
