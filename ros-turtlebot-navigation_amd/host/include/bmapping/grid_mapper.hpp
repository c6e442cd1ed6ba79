// bmapping/grid_mapper.hpp — GridMapper as the nodes use it: a value that carries the map geometry,
// the laser model and the robot->laser transform into ParticleFilter's ctor
// (reference grid_mapper.hpp:121-122, call site bmapping/src/turtle_mapping_node.cpp:396).
// The per-particle maps themselves (log-odds, distance codes, occupancy bitmaps) live in HBM inside
// the filter handle; likelihoodFieldModel / integrateScan / gridMap are device kernels there
// (include/tbnav_rbpf.h), not host methods.
#ifndef TBNAV_BMAPPING_GRID_MAPPER_HPP
#define TBNAV_BMAPPING_GRID_MAPPER_HPP

#include "bmapping/sensor_model.hpp"
#include "rigid2d/rigid2d.hpp"

namespace bmapping {

using rigid2d::Transform2D;

class GridMapper {
 public:
  GridMapper(double resolution, double xmin, double xmax, double ymin, double ymax, const LaserProperties& props,
             const Transform2D& Trs)
      : resolution_(resolution), xmin_(xmin), xmax_(xmax), ymin_(ymin), ymax_(ymax), props_(props), Trs_(Trs) {}

  double resolution() const { return resolution_; }
  double xmin() const { return xmin_; }
  double xmax() const { return xmax_; }
  double ymin() const { return ymin_; }
  double ymax() const { return ymax_; }
  const LaserProperties& laser() const { return props_; }
  const Transform2D& robotToLaser() const { return Trs_; }

 private:
  double resolution_, xmin_, xmax_, ymin_, ymax_;
  LaserProperties props_;
  Transform2D Trs_;
};

}  // namespace bmapping
#endif
