// rbpf_pool.hip — the tile pool's free lists on their own (rbpf_device.hpp: tile_grab / tile_take / tile_push): a self-test the
// GPU tests drive through the C-ABI, on a pool that belongs to no handle (rings and counters only, no tiles behind them).
#include "rbpf_host.hpp"

namespace tbnav_rk {
// caller c pops `each` tiles at once (hint: its own number, or 0 for everybody) and writes their ids, or zeros if no list could
// supply them
__global__ __launch_bounds__(64) void rbpf_pool_test_pop(TilePool P, int callers, int each, int same_hint, unsigned int* __restrict__ ids,
                                                         unsigned int* __restrict__ parked, int* __restrict__ n_parked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= callers) return;
  const unsigned int hint = same_hint ? 0u : (unsigned int)c;
  unsigned int* const mine = ids + (size_t)c * each;
  // one request, the way the map update makes it: the first list's grant, the next lists' for what is missing — all of them or
  // none: what an unfinished request took goes to `parked` (pushed back with the rest; the ids array reports zeros for this caller)
  TileTaker tk = tile_taker(hint);
  int got = 0;
  for (; got < each; ++got) { const unsigned int id = tile_take(P, tk, (unsigned int)(each - got)); if (!id) break; mine[got] = id; }
  if (got < each) for (int i = 0; i < each; ++i) { if (i < got) parked[atomicAdd(n_parked, 1)] = mine[i]; mine[i] = 0u; }
}
__global__ __launch_bounds__(256) void rbpf_pool_test_push(TilePool P, size_t n, const unsigned int* __restrict__ ids) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ids[i]) tile_push(P, ids[i]);
}
}  // namespace tbnav_rk

int tbnav_rbpf_pool_selftest(uint32_t cap_tiles, int32_t rounds, int32_t callers, int32_t tiles_each, int32_t same_hint,
                             uint32_t* ids_out, uint64_t* free_after_pop, uint64_t* free_after_push) {
  using namespace tbnav_rk;
  if (cap_tiles < 2 || rounds < 1 || callers < 1 || tiles_each < 1 || tiles_each > 64 || !ids_out || !free_after_pop || !free_after_push)
    return TBNAV_ERR_INVALID_ARG;
  TilePool P{};
  P.cap = cap_tiles;
  P.shards = tbnav_rh::pool_lists_for(cap_tiles);
  P.shard_cap = (cap_tiles + P.shards - 1) / P.shards;
  unsigned int *d_ids = nullptr, *d_parked = nullptr;
  int* d_n_parked = nullptr;
  const size_t n = (size_t)callers * tiles_each;
  auto free_tiles = [&](uint64_t* out) -> int {
    unsigned long long ctr[kPoolCtrWords];
    TBNAV_HIP(hipMemcpy(ctr, P.ctr, sizeof ctr, hipMemcpyDeviceToHost));
    *out = 0;
    for (unsigned int s = 0; s < P.shards; ++s) *out += ctr[s * kPoolCtrStride + 1] - ctr[s * kPoolCtrStride];
    return TBNAV_OK;
  };
  auto body = [&]() -> int {
    TBNAV_HIP(hipMalloc((void**)&P.ring, sizeof(unsigned int) * (size_t)P.shard_cap * P.shards));
    TBNAV_HIP(hipMalloc((void**)&P.ctr, sizeof(unsigned long long) * kPoolCtrWords));
    TBNAV_HIP(hipMalloc((void**)&P.ref, sizeof(int)));   // (the init kernel pins tile 0's count)
    TBNAV_HIP(hipMalloc((void**)&d_ids, sizeof(unsigned int) * n));
    TBNAV_HIP(hipMalloc((void**)&d_parked, sizeof(unsigned int) * n));
    TBNAV_HIP(hipMalloc((void**)&d_n_parked, sizeof(int)));
    hipLaunchKernelGGL(rbpf_pool_init, dim3(256), dim3(256), 0, 0, P);
    for (int r = 0; r < rounds; ++r) {
      TBNAV_HIP(hipMemset(d_parked, 0, sizeof(unsigned int) * n));
      TBNAV_HIP(hipMemset(d_n_parked, 0, sizeof(int)));
      hipLaunchKernelGGL(rbpf_pool_test_pop, dim3((unsigned int)((callers + 63) / 64)), dim3(64), 0, 0, P, callers, tiles_each, same_hint, d_ids, d_parked, d_n_parked);
      hipLaunchKernelGGL(rbpf_pool_test_push, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, 0, P, n, d_parked);   // what unfinished requests held
      TBNAV_HIP(hipDeviceSynchronize());
      { const int rc = free_tiles(free_after_pop); if (rc != TBNAV_OK) return rc; }
      TBNAV_HIP(hipMemcpy(ids_out, d_ids, sizeof(unsigned int) * n, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL(rbpf_pool_test_push, dim3((unsigned int)((n + 255) / 256)), dim3(256), 0, 0, P, n, d_ids);
      TBNAV_HIP(hipDeviceSynchronize());
      { const int rc = free_tiles(free_after_push); if (rc != TBNAV_OK) return rc; }
    }
    return TBNAV_OK;
  };
  const int rc = body();
  (void)hipFree(P.ring); (void)hipFree(P.ctr); (void)hipFree(P.ref); (void)hipFree(d_ids); (void)hipFree(d_parked); (void)hipFree(d_n_parked);
  return rc;
}
