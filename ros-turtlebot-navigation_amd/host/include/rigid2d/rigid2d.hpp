// rigid2d/rigid2d.hpp — host-side SE(2) types with the reference's public surface
// (reference rigid2d/include/rigid2d/rigid2d.hpp:11-372, rigid2d/src/rigid2d/rigid2d.cpp).
//
// Written for this build (Eigen-free, compiles with g++ AND hipcc/clang: the angle helpers are plain
// inline functions, not constexpr-with-libm as in the reference, which only GCC accepts).  The ROS
// nodes use: Vector2D, Twist2D, TransformData2D, Transform2D{ctor(Vector2D,double), operator(),
// operator*=, displacement, integrateTwist, inv}, operator*, normalize_angle_PI, almost_equal,
// deg2rad, rad2deg — all kept with identical signatures and identical arithmetic (checked bit-exactly
// against the reference build in tests/test_host_shims.py).
#ifndef TBNAV_RIGID2D_HPP
#define TBNAV_RIGID2D_HPP

#include <cmath>
#include <iosfwd>

namespace rigid2d {

constexpr double PI = 3.14159265358979323846;

inline bool almost_equal(double d1, double d2, double epsilon = 1.0e-12) { return std::fabs(d1 - d2) < epsilon; }
constexpr double deg2rad(double deg) { return deg * (PI / 180.0); }
constexpr double rad2deg(double rad) { return rad * (180.0 / PI); }

/// wrap to [-pi, pi)
inline double normalize_angle_PI(double rad) {
  const double turns = std::floor((rad + PI) / (2.0 * PI));
  rad = (rad + PI) - turns * 2.0 * PI;
  if (rad < 0) rad += 2.0 * PI;
  return rad - PI;
}
/// wrap to [0, 2pi)
inline double normalize_angle_2PI(double rad) {
  const double turns = std::floor(rad / (2.0 * PI));
  rad = rad - turns * 2.0 * PI;
  if (rad < 0) rad += 2.0 * PI;
  return rad;
}

struct Vector2D {
  double x = 0.0, y = 0.0;
  Vector2D() = default;
  Vector2D(double vx, double vy) : x(vx), y(vy) {}
  Vector2D& operator+=(const Vector2D& o) { x += o.x; y += o.y; return *this; }
  Vector2D& operator-=(const Vector2D& o) { x -= o.x; y -= o.y; return *this; }
  Vector2D& operator*=(double k) { x *= k; y *= k; return *this; }
};
inline Vector2D operator+(Vector2D a, const Vector2D& b) { return a += b; }
inline Vector2D operator-(Vector2D a, const Vector2D& b) { return a -= b; }
inline Vector2D operator*(Vector2D a, double k) { return a *= k; }
inline Vector2D operator*(double k, Vector2D a) { return a *= k; }
inline double length(const Vector2D& v) { return std::sqrt(v.x * v.x + v.y * v.y); }
inline double distance(const Vector2D& a, const Vector2D& b) { return length(a - b); }
double angle(const Vector2D& a, const Vector2D& b);

struct NormalVec2D { double nx = 0.0, ny = 0.0; };
NormalVec2D normalize(const Vector2D& v);

struct Twist2D { double w = 0.0, vx = 0.0, vy = 0.0; };
struct TransformData2D { double theta = 0.0, x = 0.0, y = 0.0; };
struct Screw2D { double w = 0.0, vx = 0.0, vy = 0.0; };

class Transform2D {
 public:
  Transform2D() = default;
  explicit Transform2D(const Vector2D& trans) : x(trans.x), y(trans.y) {}
  explicit Transform2D(double radians) : theta(radians), ctheta(std::cos(radians)), stheta(std::sin(radians)) {}
  Transform2D(const Vector2D& trans, double radians)
      : theta(radians), ctheta(std::cos(radians)), stheta(std::sin(radians)), x(trans.x), y(trans.y) {}

  Vector2D operator()(Vector2D v) const;   ///< apply to a point
  Twist2D operator()(Twist2D t) const;     ///< adjoint on a twist
  Transform2D inv() const;
  Transform2D& operator*=(const Transform2D& rhs);
  TransformData2D displacement() const { return {theta, x, y}; }
  Transform2D integrateTwist(const Twist2D& twist) const;

  friend std::ostream& operator<<(std::ostream& os, const Transform2D& tf);

 private:
  Transform2D(double th, double c, double s, double px, double py) : theta(th), ctheta(c), stheta(s), x(px), y(py) {}
  double theta = 0.0, ctheta = 1.0, stheta = 0.0, x = 0.0, y = 0.0;
};
inline Transform2D operator*(Transform2D lhs, const Transform2D& rhs) { return lhs *= rhs; }

std::ostream& operator<<(std::ostream& os, const Vector2D& v);
std::ostream& operator<<(std::ostream& os, const Twist2D& t);
std::ostream& operator<<(std::ostream& os, const Transform2D& tf);
std::istream& operator>>(std::istream& is, Vector2D& v);
std::istream& operator>>(std::istream& is, Twist2D& t);
std::istream& operator>>(std::istream& is, Transform2D& tf);

}  // namespace rigid2d
#endif
