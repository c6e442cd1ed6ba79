// mppi_shim.cpp — controller::MPPI over the C-ABI.  Status codes become the exceptions a caller of
// the reference would have seen (std::invalid_argument / std::runtime_error); there is no CPU path.
#include <cmath>
#include <stdexcept>
#include <string>

#include "controller/mppi.hpp"
#include "rigid2d/utilities.hpp"
#include "tbnav_mppi.h"

namespace controller {
namespace {
void check(int rc, const char* where) {
  if (rc == TBNAV_OK) return;
  std::string msg = std::string(where) + ": " + tbnav_status_string(rc);
  const char* hip = tbnav_last_hip_error();
  if (hip && *hip) msg += std::string(" [") + hip + "]";
  if (rc == TBNAV_ERR_INVALID_ARG) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
}  // namespace

MPPI::MPPI(const CartModel& cart, const LossFunc& loss, double lambda, double max_wheel_vel, double ul_var, double ur_var,
           double horizon, double dt, int rollouts, int n_gpus, const std::vector<int>& devices) {
  tbnav_mppi_params p{};
  p.wheel_radius = cart.wheel_radius; p.wheel_base = cart.wheel_base;
  p.lambda = lambda; p.max_wheel_vel = max_wheel_vel; p.ul_var = ul_var; p.ur_var = ur_var;
  p.horizon = horizon; p.dt = dt;
  for (int i = 0; i < 3; ++i) { p.Q[i] = loss.Q[i]; p.P1[i] = loss.P1[i]; }
  p.R[0] = loss.R[0]; p.R[1] = loss.R[1];
  p.rollouts = rollouts; p.device = -1;
  if (n_gpus > 1) {
    if (!devices.empty() && (int)devices.size() != n_gpus) throw std::invalid_argument("controller::MPPI: devices.size() != n_gpus");
    check(tbnav_mppi_group_create(&p, n_gpus, devices.empty() ? nullptr : devices.data(), &g_), "controller::MPPI (n_gpus)");
    tbnav_mppi* m0 = nullptr;
    check(tbnav_mppi_group_member(g_, 0, &m0), "controller::MPPI (n_gpus)");
    steps_ = tbnav_mppi_steps(m0);
  } else {
    check(tbnav_mppi_create(&p, &h_), "controller::MPPI");
    steps_ = tbnav_mppi_steps(h_);
  }
  rollouts_ = rollouts;
  ul_sig_ = std::sqrt(ul_var);  // mppi.cpp:176-177: the sampler takes a standard deviation
  ur_sig_ = std::sqrt(ur_var);
}

MPPI::~MPPI() { tbnav_mppi_destroy(h_); tbnav_mppi_group_destroy(g_); }
int MPPI::gpus() const { return g_ ? tbnav_mppi_group_size(g_) : 1; }
MPPI::MPPI(MPPI&& o) noexcept
    : h_(o.h_), g_(o.g_), steps_(o.steps_), rollouts_(o.rollouts_), ul_sig_(o.ul_sig_), ur_sig_(o.ur_sig_), device_noise_(o.device_noise_),
      seed_(o.seed_), tick_(o.tick_), noise_(std::move(o.noise_)) { o.h_ = nullptr; o.g_ = nullptr; }

void MPPI::setInitialControls(double uL, double uR) {
  check(g_ ? tbnav_mppi_group_set_initial_controls(g_, uL, uR) : tbnav_mppi_set_initial_controls(h_, uL, uR), "setInitialControls");
}
void MPPI::setWaypoint(const Pose& w) {
  check(g_ ? tbnav_mppi_group_set_waypoint(g_, w.x, w.y, w.theta) : tbnav_mppi_set_waypoint(h_, w.x, w.y, w.theta), "setWaypoint");
}
void MPPI::useDeviceNoise(std::uint64_t seed) { device_noise_ = true; seed_ = seed; tick_ = 0; }
void MPPI::useExactArcDynamics(bool on) {
  const int model = on ? TBNAV_MPPI_DYN_ARC : TBNAV_MPPI_DYN_RK4;
  check(g_ ? tbnav_mppi_group_set_dynamics(g_, model) : tbnav_mppi_set_dynamics(h_, model), "useExactArcDynamics");
}

WheelVelocities MPPI::newControls(const Pose& ps) {
  const double x0[3] = {ps.x, ps.y, ps.theta};  // mppi.cpp:75-76: state order (x, y, theta)
  double out[2] = {0.0, 0.0};
  if (device_noise_) {
    // the perturbations of (seed, tick) are generated inside the rollout kernel where the configuration has the fused one
    if (g_) check(tbnav_mppi_group_new_controls_rng(g_, x0, seed_, tick_++, out), "newControls");
    else check(tbnav_mppi_new_controls_rng(h_, x0, seed_, tick_++, nullptr, out), "newControls");
  } else {
    noise_.resize((size_t)2 * steps_ * rollouts_);
    size_t n = 0;
    for (int k = 0; k < rollouts_; ++k)        // rollout-major, left then right per step (mppi.cpp:81-89,179-183)
      for (int i = 0; i < steps_; ++i) {
        noise_[n++] = rigid2d::sampleNormalDistribution(0.0, ul_sig_);
        noise_[n++] = rigid2d::sampleNormalDistribution(0.0, ur_sig_);
      }
    // (n_gpus > 1: the ENSEMBLE's stream, rollout k of the reference = rollout k of the ensemble; the devices take consecutive slices)
    check(g_ ? tbnav_mppi_group_new_controls(g_, x0, noise_.data(), out) : tbnav_mppi_new_controls(h_, x0, noise_.data(), out), "newControls");
  }
  WheelVelocities w;
  w.ul = out[0];
  w.ur = out[1];
  return w;
}

std::vector<double> MPPI::controls() const {
  std::vector<double> u((size_t)2 * steps_);
  check(g_ ? tbnav_mppi_group_get_controls(g_, u.data()) : tbnav_mppi_get_controls(h_, u.data()), "controls");
  return u;
}

}  // namespace controller
