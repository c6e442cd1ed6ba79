// ref_harness_mppi.cpp — extern "C" access to the REAL controller::MPPI, compiled together with the reference's own unmodified
// controller/src/controller/{mppi,rk4}.cpp and rigid2d/src/rigid2d/{utilities,rigid2d,diff_drive}.cpp where they lie under
// /root/reference (oracle/Makefile, target `ref_mppi`).  Needs Eigen 3.3 (controller/CMakeLists.txt:18): the target is built
// ONLY when the image really has <eigen3/Eigen/Dense> — no stand-in headers.  TEST INFRASTRUCTURE ONLY; output oracle/_ref/.
// Built with -fno-access-control so the harness can read the controller's private u / J / duL / duR and `steps`.
//
// This file cannot be compiled in an image without Eigen: it is written against controller/include/controller/mppi.hpp:119-185
// and rigid2d/include/rigid2d/utilities.hpp:18-40 as they read, and is exercised by tests/test_oracle_vs_reference.py the day
// Eigen is there (the tests skip until then).
#include <cstdint>
#include <string>
#include <vector>

#include "controller/mppi.hpp"
#include "rigid2d/utilities.hpp"

namespace {
thread_local std::string g_err;
}

extern "C" {

const char* refm_last_error() { return g_err.c_str(); }

// params = wheel_radius, wheel_base, lambda, max_wheel_vel, ul_var, ur_var, horizon, dt, Q[3], R[2], P1[3]
void* refm_create(const double* p, int rollouts) {
  try {
    controller::CartModel cart(p[0], p[1]);
    controller::LossFunc loss({p[8], p[9], p[10]}, {p[11], p[12]}, {p[13], p[14], p[15]});
    return new controller::MPPI(cart, loss, p[2], p[3], p[4], p[5], p[6], p[7], rollouts);
  } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void refm_destroy(void* h) { delete static_cast<controller::MPPI*>(h); }
void refm_seed(uint64_t s) { rigid2d::getTwister().seed(s); }   // utilities.hpp:18: the public process-global engine
int refm_steps(void* h) { return static_cast<controller::MPPI*>(h)->steps; }
void refm_set_waypoint(void* h, double x, double y, double theta) {
  rigid2d::Pose w; w.x = x; w.y = y; w.theta = theta;
  static_cast<controller::MPPI*>(h)->setWaypoint(w);
}
void refm_set_initial_controls(void* h, double ul, double ur) { static_cast<controller::MPPI*>(h)->setInitialControls(ul, ur); }
// u_in [2][T] row-major
void refm_set_controls(void* h, const double* u_in) {
  auto* m = static_cast<controller::MPPI*>(h);
  for (int r = 0; r < 2; ++r) for (int i = 0; i < m->steps; ++i) m->u(r, i) = u_in[r * m->steps + i];
}
// pose = (x, y, theta); out = (ul, ur).  Draws 2*T*K normals from rigid2d::getTwister() (mppi.cpp:81-89,173-184).
int refm_new_controls(void* h, const double pose[3], double out[2]) {
  try {
    rigid2d::Pose ps; ps.x = pose[0]; ps.y = pose[1]; ps.theta = pose[2];
    const rigid2d::WheelVelocities w = static_cast<controller::MPPI*>(h)->newControls(ps);
    out[0] = w.ul; out[1] = w.ur;
    return 0;
  } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
// u [2][T]; J, duL, duR [T][K] row-major (J is the matrix AFTER the per-row min subtraction of mppi.cpp:115: add the row
// minimum back from the oracle's side, or compare differences — the test does the latter)
void refm_get(void* h, double* u, double* J, double* duL, double* duR) {
  auto* m = static_cast<controller::MPPI*>(h);
  const int T = m->steps, K = m->rollouts;
  for (int r = 0; r < 2; ++r) for (int i = 0; i < T; ++i) u[r * T + i] = m->u(r, i);
  for (int i = 0; i < T; ++i) for (int k = 0; k < K; ++k) {
    J[(size_t)i * K + k] = m->J(i, k);
    duL[(size_t)i * K + k] = m->duL(i, k);
    duR[(size_t)i * K + k] = m->duR(i, k);
  }
}

}  // extern "C"
