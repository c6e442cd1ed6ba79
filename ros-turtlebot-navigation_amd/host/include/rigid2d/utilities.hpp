// rigid2d/utilities.hpp — the process-global RNG and scalar samplers the MPPI controller draws
// from (reference rigid2d/include/rigid2d/utilities.hpp:18-40, src/rigid2d/utilities.cpp:12-32).
// The Eigen-returning helpers of the reference (sampleStandardNormal, sampleMultivariate...) are
// replaced by std::vector / pointer forms; nothing on the two hot paths used them with Eigen types
// across the class boundary.
#ifndef TBNAV_RIGID2D_UTILITIES_HPP
#define TBNAV_RIGID2D_UTILITIES_HPP

#include <random>
#include <vector>

namespace rigid2d {

std::mt19937_64& getTwister();                                   ///< static engine seeded from random_device; reseed with getTwister().seed(s)
double sampleNormalDistribution(double mu, double sigma);        ///< a fresh normal_distribution per call, like the reference
double sampleUniformDistribution(double min, double max);
std::vector<double> sampleStandardNormal(int n);
double euclideanDistance(double x0, double y0, double x1, double y1);

}  // namespace rigid2d
#endif
