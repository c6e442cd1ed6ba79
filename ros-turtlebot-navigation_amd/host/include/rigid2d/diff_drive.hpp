// rigid2d/diff_drive.hpp — differential-drive kinematics / odometry, reference surface
// rigid2d/include/rigid2d/diff_drive.hpp:16-104 (Pose, WheelVelocities, WheelEncoders, DiffDrive).
// Host-only: it is the step AFTER the MPPI tick (wheel velocities -> body twist,
// nuturtle_robot/src/mppi_waypoints_node.cpp:276) and BEFORE the SLAM call (odometry,
// bmapping/src/turtle_mapping_node.cpp:459-472); ~30 flops per node tick, not a kernel.
#ifndef TBNAV_DIFF_DRIVE_HPP
#define TBNAV_DIFF_DRIVE_HPP

#include "rigid2d/rigid2d.hpp"

namespace rigid2d {

struct Pose { double theta = 0.0, x = 0.0, y = 0.0; };
struct WheelVelocities { double ul = 0.0, ur = 0.0; };
struct WheelEncoders { double left = 0.0, right = 0.0; };

class DiffDrive {
 public:
  DiffDrive() = default;  ///< robot at the origin, wheel_base 0.1, wheel_radius 0.02
  DiffDrive(const Pose& pose, double wheel_base, double wheel_radius);

  WheelVelocities twistToWheels(const Twist2D& twist) const;  ///< throws std::invalid_argument if twist.vy != 0
  Twist2D wheelsToTwist(const WheelVelocities& vel) const;
  WheelVelocities updateOdometry(double left, double right);  ///< absolute encoder angles in, wheel deltas out
  void feedforward(const Twist2D& cmd);                       ///< follow a body twist for one time unit
  Pose pose() const;
  WheelVelocities wheelVelocities() const { return {ul_, ur_}; }
  void reset(Pose ps) { theta_ = ps.theta; x_ = ps.x; y_ = ps.y; }
  WheelEncoders getEncoders() const { return {left_, right_}; }

 private:
  void advance(const Twist2D& body_twist);
  double theta_ = 0.0, x_ = 0.0, y_ = 0.0, wheel_base_ = 0.1, wheel_radius_ = 0.02;
  double left_ = 0.0, right_ = 0.0, ul_ = 0.0, ur_ = 0.0;
};

}  // namespace rigid2d
#endif
