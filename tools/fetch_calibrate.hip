// Calibration (development): what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the map update's access pattern — 16-byte
// pair loads / stores by the lanes of a wave, a FRACTION of the pairs of every 128-byte line touched — against known byte counts.
// The guide (MI355X_MICROARCH.md, HBM) calibrates wide dense streaming reads only (FETCH_SIZE = half the bytes) and calls everything
// else uncalibrated.   hipcc --offload-arch=gfx950 -O2 tools/fetch_calibrate.hip -o tools/_fetch_calibrate
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/_fetch_calibrate      (and again with WRITE_SIZE)
// Kernels (each moves `touched` bytes of a 1 GiB buffer, far beyond the 256 MiB Infinity Cache):
//   rd_dense / wr_dense        every pair                       (8 of 8 pairs of a line)
//   rd_half / wr_half          pairs 0-3 of every line          (the left half: 64 of 128 bytes)
//   rd_alt / wr_alt            every other pair                 (4 of 8, spread over the line)
//   rd_one / wr_one            one pair per line                (16 of 128 bytes)
//   rmw_alt                    read + write every other pair    (the update's read-modify-write)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE> __device__ __forceinline__ bool pick(size_t pair) {   // pair index within the buffer; 8 pairs per 128-byte line
  const int q = (int)(pair & 7);
  return MODE == 0 ? true : MODE == 1 ? q < 4 : MODE == 2 ? (q & 1) == 0 : q == 3;
}
template <int MODE> __global__ void rd(const double2* __restrict__ a, size_t n, double* sink) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (pick<MODE>(i)) { const double2 v = a[i]; acc += v.x + v.y; }
  if (acc == 12345.678) *sink = acc;
}
template <int MODE> __global__ void wr(double2* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (pick<MODE>(i)) a[i] = double2{1.0, 2.0};
}
__global__ void rmw_alt(double2* __restrict__ a, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (pick<2>(i)) { double2 v = a[i]; v.x += 1.0; v.y += 1.0; a[i] = v; }
}
int main() {
  const size_t bytes = (size_t)1 << 30, n = bytes / 16;
  double2* a; double* sink; hipMalloc(&a, bytes); hipMalloc(&sink, 8); hipMemset(a, 0, bytes); hipDeviceSynchronize();
  const dim3 g(256 * 16), b(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(rd<0>, g, b, 0, 0, a, n, sink); hipLaunchKernelGGL(rd<1>, g, b, 0, 0, a, n, sink);
    hipLaunchKernelGGL(rd<2>, g, b, 0, 0, a, n, sink); hipLaunchKernelGGL(rd<3>, g, b, 0, 0, a, n, sink);
    hipLaunchKernelGGL(wr<0>, g, b, 0, 0, a, n); hipLaunchKernelGGL(wr<1>, g, b, 0, 0, a, n);
    hipLaunchKernelGGL(wr<2>, g, b, 0, 0, a, n); hipLaunchKernelGGL(wr<3>, g, b, 0, 0, a, n);
    hipLaunchKernelGGL(rmw_alt, g, b, 0, 0, a, n);
    hipDeviceSynchronize();
  }
  std::printf("buffer %zu bytes; touched per launch: dense %zu, half %zu, alt %zu, one %zu\n", bytes, bytes, bytes / 2, bytes / 2, bytes / 8);
  return 0;
}
