// Probe (development): does hipExtAnyOrderLaunch let a kernel start before its predecessor in the SAME stream has finished on gfx950?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this measures it.  hipcc --offload-arch=gfx950 tools/any_order_probe.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long* t, unsigned long long cycles) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
__global__ void stamp(unsigned long long* t) { if (threadIdx.x == 0 && blockIdx.x == 0) t[2] = wall_clock64(); }
int main() {
  unsigned long long* d; hipMalloc(&d, 64); hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (int flag = 0; flag < 2; ++flag) for (int rep = 0; rep < 3; ++rep) {
    hipMemset(d, 0, 64); hipDeviceSynchronize();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 5000ull);   // 100 MHz clock: 50 us
    hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, d);
    hipStreamSynchronize(s);
    unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    std::printf("flag %d: spin ran %.1f us; the second kernel's stamp is %.1f us after the spin's START (%s)\n", flag, (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0,
                h[2] < h[1] ? "BEFORE its end: overlapped" : "after its end: in order");
  }
  return 0;
}
