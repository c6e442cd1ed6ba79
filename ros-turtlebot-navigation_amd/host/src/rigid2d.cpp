// rigid2d.cpp — Transform2D, DiffDrive and the RNG helpers (host side of the drop-in boundary).
// Arithmetic follows rigid2d/src/rigid2d/{rigid2d,diff_drive,utilities}.cpp of the reference
// expression by expression, because the RBPF oracle and the ROS nodes depend on the exact poses.
#include <iostream>
#include <stdexcept>

#include "rigid2d/diff_drive.hpp"
#include "rigid2d/rigid2d.hpp"
#include "rigid2d/utilities.hpp"

namespace rigid2d {

double angle(const Vector2D& a, const Vector2D& b) {
  const double dot = a.x * b.x + a.y * b.y;
  return std::acos(dot / (length(a) * length(b)));
}

NormalVec2D normalize(const Vector2D& v) {
  const double mag = std::sqrt(v.x * v.x + v.y * v.y);
  return {v.x / mag, v.y / mag};
}

Vector2D Transform2D::operator()(Vector2D v) const {
  return {ctheta * v.x - stheta * v.y + x, stheta * v.x + ctheta * v.y + y};
}

Twist2D Transform2D::operator()(Twist2D t) const {
  Twist2D out;
  out.w = t.w;
  out.vx = t.vx * ctheta - t.vy * stheta + t.w * y;
  out.vy = t.vy * ctheta + t.vx * stheta - t.w * x;
  return out;
}

Transform2D Transform2D::inv() const {
  Transform2D r(theta, ctheta, stheta, x, y);
  r.stheta = -1.0 * stheta;
  r.theta = std::atan2(r.stheta, r.ctheta);
  r.x = -(r.ctheta * x - r.stheta * y);
  r.y = -(r.stheta * x + r.ctheta * y);
  return r;
}

Transform2D& Transform2D::operator*=(const Transform2D& rhs) {
  const double nx = ctheta * rhs.x - stheta * rhs.y + x;
  const double ny = stheta * rhs.x + ctheta * rhs.y + y;
  x = nx;
  y = ny;
  theta += rhs.theta;
  ctheta = std::cos(theta);
  stheta = std::sin(theta);
  return *this;
}

Transform2D Transform2D::integrateTwist(const Twist2D& twist) const {
  Screw2D S;
  double beta = 0.0;
  if (!almost_equal(twist.w, 0.0)) {
    beta = std::abs(twist.w);
    S.w = twist.w / beta;
    S.vx = twist.vx / beta;
    S.vy = twist.vy / beta;
  } else if (almost_equal(twist.vx, 0.0) && almost_equal(twist.vy, 0.0)) {
    return *this;
  } else {
    beta = std::sqrt(twist.vx * twist.vx + twist.vy * twist.vy);
    S.vx = twist.vx / beta;
    S.vy = twist.vy / beta;
  }
  const double cb = std::cos(beta), sb = std::sin(beta);
  const double w2 = S.w * S.w;
  const double th_new = std::atan2(sb * S.w, 1 + (1 - cb) * (-1.0 * w2));
  const double x_new = S.vx * (beta + (beta - sb) * (-1.0 * w2)) + S.vy * ((1 - cb) * (-1.0 * S.w));
  const double y_new = S.vx * ((1 - cb) * S.w) + S.vy * (beta + (beta - sb) * (-1.0 * w2));
  // the increment carries cos/sin of THIS transform's angle (reference quirk, rigid2d.cpp:286-287);
  // harmless because operator*= only reads rhs.theta/x/y.
  const Transform2D step(th_new, std::cos(theta), std::sin(theta), x_new, y_new);
  Transform2D self(theta, ctheta, stheta, x, y);
  return self * step;
}

std::ostream& operator<<(std::ostream& os, const Vector2D& v) { return os << "[" << v.x << " " << v.y << "]\n"; }
std::ostream& operator<<(std::ostream& os, const Twist2D& t) { return os << "[" << t.w << " " << t.vx << " " << t.vy << "]\n"; }
std::ostream& operator<<(std::ostream& os, const Transform2D& tf) {
  return os << "theta (degrees): " << rad2deg(tf.theta) << " x: " << tf.x << " y: " << tf.y << "\n";  // rigid2d.cpp:309
}
std::istream& operator>>(std::istream& is, Vector2D& v) {
  is >> std::ws;
  if (is.peek() == '[') is.get();
  is >> v.x >> v.y;
  is >> std::ws;
  if (is.peek() == ']') is.get();
  return is;
}
std::istream& operator>>(std::istream& is, Twist2D& t) {
  is >> std::ws;
  if (is.peek() == '[') is.get();
  is >> t.w >> t.vx >> t.vy;
  is >> std::ws;
  if (is.peek() == ']') is.get();
  return is;
}
std::istream& operator>>(std::istream& is, Transform2D& tf) {
  double deg = 0.0;
  Vector2D v;
  is >> deg >> v.x >> v.y;
  tf = Transform2D(v, deg2rad(deg));
  return is;
}

// ---- DiffDrive -------------------------------------------------------------------------------------
DiffDrive::DiffDrive(const Pose& pose, double wheel_base, double wheel_radius)
    : theta_(pose.theta), x_(pose.x), y_(pose.y), wheel_base_(wheel_base), wheel_radius_(wheel_radius) {}

WheelVelocities DiffDrive::twistToWheels(const Twist2D& twist) const {
  const double d = wheel_base_ / 2;
  WheelVelocities v;
  v.ul = (1 / wheel_radius_) * (-d * twist.w + twist.vx);
  v.ur = (1 / wheel_radius_) * (d * twist.w + twist.vx);
  if (twist.vy != 0) throw std::invalid_argument("Twist cannot have y velocity component");
  return v;
}

Twist2D DiffDrive::wheelsToTwist(const WheelVelocities& vel) const {
  const double d = 1 / wheel_base_;
  Twist2D t;
  t.w = wheel_radius_ * d * (vel.ur - vel.ul);
  t.vx = wheel_radius_ * 0.5 * (vel.ul + vel.ur);
  t.vy = 0.0;
  return t;
}

void DiffDrive::advance(const Twist2D& body_twist) {
  const Transform2D step = Transform2D().integrateTwist(body_twist);
  const Transform2D world = Transform2D(Vector2D(x_, y_), theta_) * step;
  const TransformData2D d = world.displacement();
  theta_ = normalize_angle_PI(d.theta);
  x_ = d.x;
  y_ = d.y;
}

WheelVelocities DiffDrive::updateOdometry(double left, double right) {
  WheelVelocities v;
  v.ul = normalize_angle_PI(left - left_);
  v.ur = normalize_angle_PI(right - right_);
  ul_ = v.ul;
  ur_ = v.ur;
  left_ = normalize_angle_PI(left);
  right_ = normalize_angle_PI(right);
  advance(wheelsToTwist(v));
  return v;
}

void DiffDrive::feedforward(const Twist2D& cmd) {
  const WheelVelocities v = twistToWheels(cmd);
  ul_ = normalize_angle_PI(v.ul);
  ur_ = normalize_angle_PI(v.ur);
  left_ = normalize_angle_PI(left_ + v.ul);
  right_ = normalize_angle_PI(right_ + v.ur);
  advance(cmd);
}

Pose DiffDrive::pose() const { return {normalize_angle_PI(theta_), x_, y_}; }

// ---- RNG helpers -----------------------------------------------------------------------------------
std::mt19937_64& getTwister() {
  static std::random_device rd;
  static std::mt19937_64 gen(rd());
  return gen;
}
double sampleNormalDistribution(double mu, double sigma) {
  std::normal_distribution<double> dis(mu, sigma);
  return dis(getTwister());
}
double sampleUniformDistribution(double min, double max) {
  std::uniform_real_distribution<double> dis(min, max);
  return dis(getTwister());
}
std::vector<double> sampleStandardNormal(int n) {
  std::vector<double> out(n > 0 ? n : 0);
  for (auto& v : out) {
    std::normal_distribution<double> dis(0, 1);
    v = dis(getTwister());
  }
  return out;
}
double euclideanDistance(double x0, double y0, double x1, double y1) {
  const double dx = x0 - x1, dy = y0 - y1;
  return std::sqrt(dx * dx + dy * dy);
}

}  // namespace rigid2d
