// bmapping/cloud_alignment.hpp — ScanAlignment with the reference's surface
// (reference bmapping/include/bmapping/cloud_alignment.hpp:28-80, cloud_alignment.cpp:37-72).
//
// The reference wraps pcl::IterativeClosestPoint (PCL, third party, version unpinned, not in this
// image).  ICP is serial, runs once per scan BEFORE the particle loop (particle_filter.cpp:146-153)
// and is out of the hot path: here the matcher is a pluggable host
// callable.  The default reproduces what the reference does when PCL converges onto the initial
// guess: it returns the guess.  The reference's bookkeeping is kept: the first call returns
// (true, identity) and stores the scan (cloud_alignment.cpp:43-50); a failed match does not
// refresh the stored scan (:62-71).
#ifndef TBNAV_BMAPPING_CLOUD_ALIGNMENT_HPP
#define TBNAV_BMAPPING_CLOUD_ALIGNMENT_HPP

#include <functional>
#include <iostream>
#include <vector>

#include "bmapping/sensor_model.hpp"
#include "rigid2d/rigid2d.hpp"

namespace bmapping {

using rigid2d::Transform2D;

class ScanAlignment {
 public:
  /// matcher(T_out, T_init, previous_scan, current_scan) -> converged
  using Matcher = std::function<bool(Transform2D&, const Transform2D&, const std::vector<float>&, const std::vector<float>&)>;

  ScanAlignment(const LaserProperties& props, const Transform2D& Trs) : props_(props), Trs_(Trs) {}

  /// plug in a real scan matcher (e.g. a PCL ICP wrapper in a catkin workspace that has PCL)
  void setMatcher(Matcher m) { matcher_ = std::move(m); }

  bool pclICPWrapper(Transform2D& T, const Transform2D& T_init, const std::vector<float>& scan) {
    if (!have_prev_) {
      prev_scan_ = scan;
      have_prev_ = true;
      T = Transform2D();
      return true;
    }
    bool ok = true;
    if (matcher_) ok = matcher_(T, T_init, prev_scan_, scan);
    else {
      // No scan matcher plugged in: the initial guess goes back unchanged (what a PCL ICP that converges onto its guess returns).
      // A node linked against this header WITHOUT setMatcher() therefore runs on odometry alone — say so, once.
      if (!warned_) {
        std::cerr << "bmapping::ScanAlignment: no scan matcher set (setMatcher); pclICPWrapper returns its initial guess" << std::endl;
        warned_ = true;
      }
      T = T_init;
    }
    if (ok) prev_scan_ = scan;
    return ok;
  }

 private:
  LaserProperties props_;
  Transform2D Trs_;
  Matcher matcher_;
  std::vector<float> prev_scan_;
  bool have_prev_ = false;
  bool warned_ = false;
};

}  // namespace bmapping
#endif
