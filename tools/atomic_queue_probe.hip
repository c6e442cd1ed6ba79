// How long do 1000 co-resident workgroups wait for ONE returning atomic each — on one word, or on S words (stride bytes apart)?
// (the tile pool's free list: csrc/rbpf_device.hpp tile_pop_n).  hipcc --offload-arch=gfx950 -O2 tools/atomic_queue_probe.hip -o /tmp/aqp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(512) void probe(unsigned long long* ctr, int S, int stride_words, unsigned long long* out, long long* wait_ticks) {
  __shared__ unsigned long long got;
  extern __shared__ int pad[];
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) got = atomicAdd(&ctr[(size_t)(blockIdx.x % S) * stride_words], 14ull);
  __syncthreads();
  if (threadIdx.x == 0) { out[blockIdx.x] = got; wait_ticks[blockIdx.x] = wall_clock64() - t0; }
  if (pad[threadIdx.x] == 12345) out[0] = 1;
}
int main() {
  const int N = 1000;
  unsigned long long *ctr, *out; long long* wt;
  hipMalloc(&ctr, 64 << 20); hipMemset(ctr, 0, 64 << 20);
  hipMalloc(&out, N * 8); hipMalloc(&wt, N * 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int shards[] = {1, 2, 4, 8, 16, 32, 64};
  const int strides[] = {16, 512, 1024 + 16, 8192 + 48};   // words of 8 bytes: 128 B, 4 KB, 8 KB + 128 B, 64 KB + 384 B
  for (int st : strides) for (int S : shards) {
    float best = 1e9f; std::vector<long long> h(N); double mean_wait = 0, max_wait = 0;
    for (int r = 0; r < 12; ++r) {
      hipEventRecord(a); hipLaunchKernelGGL(probe, dim3(N), dim3(512), 36 << 10, 0, ctr, S, st, out, wt); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (r >= 2 && ms < best) { best = ms; hipMemcpy(h.data(), wt, N * 8, hipMemcpyDeviceToHost); mean_wait = 0; max_wait = 0; for (auto v : h) { mean_wait += v * 0.01 / N; max_wait = std::max(max_wait, v * 0.01); } }
    }
    std::printf("stride %6d B  shards %2d: launch %.1f us, wait for the atomic: mean %.2f us, max %.2f us\n", st * 8, S, best * 1e3, mean_wait, max_wait);
    if (S == 1 && st != strides[0]) {}
  }
  return 0;
}
