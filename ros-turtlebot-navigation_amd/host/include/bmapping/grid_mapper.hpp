// bmapping/grid_mapper.hpp — bmapping::GridMapper with the reference's public surface
// (reference bmapping/include/bmapping/grid_mapper.hpp:117-140): the 7-argument constructor, likelihoodFieldModel,
// integrateScan, gridMap; derives from LaserScanner like the reference.  bmapping/src/turtle_mapping_node.cpp:392-400
// builds one and hands it to ParticleFilter's ctor, which reads its geometry and laser model.
//
// The map itself lives in HBM: the first call of a map method opens a ONE-PARTICLE handle of the RBPF C-ABI
// (include/tbnav_rbpf.h: tbnav_rbpf_integrate_scan / tbnav_rbpf_likelihood / tbnav_rbpf_particle_map — the same
// kernels the filter runs per particle).  A GridMapper that is only constructed and passed on never touches the GPU.
// Copies are deep, like the reference's value semantics (particle_filter.cpp:125-138 copies the prototype per
// particle): the copy gets its own handle with the same log-odds.
#ifndef TBNAV_BMAPPING_GRID_MAPPER_HPP
#define TBNAV_BMAPPING_GRID_MAPPER_HPP

#include <cstdint>
#include <vector>

#include "bmapping/sensor_model.hpp"
#include "rigid2d/rigid2d.hpp"

struct tbnav_rbpf;  // C-ABI handle

namespace bmapping {

using rigid2d::Transform2D;

class GridMapper : public LaserScanner {
 public:
  GridMapper(double resolution, double xmin, double xmax, double ymin, double ymax, const LaserProperties& props,
             const Transform2D& Trs);
  GridMapper(const GridMapper& other);
  GridMapper& operator=(const GridMapper& other);
  ~GridMapper();

  /// scan likelihood P(z | m, x) of the likelihood-field model (grid_mapper.cpp:69-133).
  /// Throws std::invalid_argument("... NOT in the bounds of the world") like the reference.
  double likelihoodFieldModel(const std::vector<float>& beam_length, const Transform2D& pose) const;
  /// ray-cast the scan into the map from `pose` and refresh the distance field (grid_mapper.cpp:140-182)
  void integrateScan(const std::vector<float>& beam_length, const Transform2D& pose);
  /// int8 map for rviz: -1 unknown, 0 free, 100 occupied, else prob * 100; transposed (grid_mapper.cpp:185-226)
  void gridMap(std::vector<int8_t>& map) const;

  // ---- additions (not in the reference) ----
  /// true (the default: a GridMapper used on its own behaves as the reference's): the reference's own priority-queue
  /// brushfire, bit for bit (host work per scan); false: the exact nearest-obstacle distance, computed at lookup
  /// (include/tbnav_rbpf.h "DISTANCE FIELD").  Before the first scan.
  void useReferenceDistanceField(bool on = true);
  void useExactDistanceField() { useReferenceDistanceField(false); }
  double resolution() const { return resolution_; }
  double xmin() const { return xmin_; }
  double xmax() const { return xmax_; }
  double ymin() const { return ymin_; }
  double ymax() const { return ymax_; }
  LaserProperties laser() const { return properties(); }

 private:
  tbnav_rbpf* handle() const;  // opens the one-particle handle on first use
  double resolution_, xmin_, xmax_, ymin_, ymax_;
  bool reference_field_ = true;
  mutable tbnav_rbpf* h_ = nullptr;
};

}  // namespace bmapping
#endif
