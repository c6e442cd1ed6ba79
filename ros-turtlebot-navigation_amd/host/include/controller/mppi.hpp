// controller/mppi.hpp — controller::MPPI with the reference's class surface
// (reference controller/include/controller/mppi.hpp:31-185), Eigen-free, backed by the HIP path
// through the C-ABI of include/tbnav_mppi.h.  A node that did
//     controller::CartModel cart(r, b); controller::LossFunc loss(Q, R, P1);
//     controller::MPPI mppi(cart, loss, lambda, umax, ul_var, ur_var, horizon, dt, K);
//     mppi.setInitialControls(ul0, ur0); mppi.setWaypoint(wpt); auto u = mppi.newControls(pose);
// (nuturtle_robot/src/mppi_waypoints_node.cpp:186-199,216,257,265) compiles unchanged against this
// header and links libtbnav_host.so + libtbnav_hip.so instead of libcontroller.
#ifndef TBNAV_CONTROLLER_MPPI_HPP
#define TBNAV_CONTROLLER_MPPI_HPP

#include <cstdint>
#include <vector>

#include "rigid2d/diff_drive.hpp"
#include "rigid2d/rigid2d.hpp"

struct tbnav_mppi;        // C-ABI handle
struct tbnav_mppi_group;  // the same over several GPUs (include/tbnav_mppi.h)

namespace controller {

using rigid2d::Pose;
using rigid2d::WheelVelocities;

/// kinematic cart: only its geometry crosses the boundary (the ODE lives in the rollout kernel)
struct CartModel {
  CartModel(double wheel_radius, double wheel_base) : wheel_radius(wheel_radius), wheel_base(wheel_base) {}
  double wheel_radius, wheel_base;
};

/// diagonal LQR weights; .at() throws std::out_of_range on short vectors like the reference ctor
struct LossFunc {
  LossFunc(std::vector<double> Qdiag, std::vector<double> Rdiag, std::vector<double> P1diag)
      : Q{Qdiag.at(0), Qdiag.at(1), Qdiag.at(2)}, R{Rdiag.at(0), Rdiag.at(1)}, P1{P1diag.at(0), P1diag.at(1), P1diag.at(2)} {}
  double Q[3], R[2], P1[3];
};

class MPPI {
 public:
  /// The reference's nine arguments (mppi.hpp:133-141).  n_gpus (not in the reference): > 1 splits the
  /// `rollouts` (a multiple of n_gpus) evenly over that many MI355X of this process — per tick every device rolls its share
  /// out, ONE RCCL all-gather of the per-time-step soft-min records, the same combine on every device.  devices: HIP ordinals
  /// (empty: 0 .. n_gpus-1).
  MPPI(const CartModel& cart_model, const LossFunc& loss_func, double lambda, double max_wheel_vel, double ul_var,
       double ur_var, double horizon, double dt, int rollouts, int n_gpus = 1, const std::vector<int>& devices = {});
  ~MPPI();
  MPPI(const MPPI&) = delete;  // the handle owns device memory (the reference's copy is latently broken anyway: SURVEY.md section 5)
  MPPI& operator=(const MPPI&) = delete;
  MPPI(MPPI&& o) noexcept;

  void setInitialControls(double uL, double uR);
  void setWaypoint(const Pose& wpt);
  /// One control tick.  Draws 2*T*K perturbations from rigid2d::getTwister() in the reference's
  /// order (mppi.cpp:81-89,173-184) unless useDeviceNoise() was called.
  WheelVelocities newControls(const Pose& ps);

  // ---- additions (not in the reference) ----
  void useDeviceNoise(std::uint64_t seed);   ///< draw the perturbations on the GPU (Philox) instead of the host twister
  /// Not in the reference: integrate the rollouts with the plant's own step, rigid2d::DiffDrive::feedforward of
  /// wheelsToTwist(u) * dt (exact arcs), instead of CartModel + RK4.  Off by default (tbnav_mppi_set_dynamics).
  void useExactArcDynamics(bool on = true);
  int steps() const { return steps_; }
  int rollouts() const { return rollouts_; }
  int gpus() const;
  std::vector<double> controls() const;      ///< warm-start matrix u, [2][T]

 private:
  tbnav_mppi* h_ = nullptr;         // n_gpus == 1
  tbnav_mppi_group* g_ = nullptr;   // n_gpus > 1
  int steps_ = 0, rollouts_ = 0;
  double ul_sig_ = 0.0, ur_sig_ = 0.0;
  bool device_noise_ = false;
  std::uint64_t seed_ = 0, tick_ = 0;
  std::vector<double> noise_;
};

}  // namespace controller
#endif
