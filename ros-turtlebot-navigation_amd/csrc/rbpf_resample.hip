// rbpf_resample.hip — normalizeWeights / effectiveParticles / lowVarianceResampling (particle_filter.cpp:442-500) as a
// kernel of its own, the resampling copies as tile-table copies + reference counts (:495), the tile pool, dense views of one
// particle's map, and getRobotState / newMap on the device (:255-291, grid_mapper.cpp:185-226).
#include "rbpf_device.hpp"
#include "rbpf_normalize.hpp"

namespace tbnav_rk {

__global__ __launch_bounds__(256) void rbpf_normalize(int N, const double* __restrict__ zp, const double* weight, double* weight_out,
                                                      double* __restrict__ cs, int* __restrict__ parent, NormOut* __restrict__ out,
                                                      int* __restrict__ gate, const int* __restrict__ gate_prev,
                                                      unsigned int* seq, unsigned int seq_val,
                                                      int* __restrict__ children) {
  __shared__ __attribute__((aligned(16))) double w[kNormChunk], cl[kNormChunk];
  if (gate_prev && *gate_prev) return;
  normalize_body<256, true>(N, zp, weight, weight_out, cs, parent, out, w, cl, gate, seq, seq_val, children);
}
// free lists = every tile but tile 0 (the shared zero tile, pinned): tile id sits in slot id / shards of list id % shards
// (list 0's first slot holds tile 0 and is skipped: its head starts at 1)
__global__ void rbpf_pool_init(TilePool P) {
  const unsigned int sh = P.shards == 1u ? 0u : (unsigned int)kPoolShardsLog2;
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.cap; i += gridDim.x * blockDim.x)
    P.ring[(size_t)(i & (P.shards - 1u)) * P.shard_cap + (i >> sh)] = i;
  if (blockIdx.x == 0 && threadIdx.x < kPoolShards) {   // every line of the counters (an unused list: empty)
    const unsigned int s = threadIdx.x;
    const bool list = s < P.shards;
    P.ctr[(size_t)s * kPoolCtrStride] = list && s == 0u ? 1ull : 0ull;
    P.ctr[(size_t)s * kPoolCtrStride + 1] = list && P.cap > s ? (unsigned long long)((P.cap - s + P.shards - 1u) >> sh) : 0ull;
    if (s == 0u) P.ref[0] = 1 << 30;
  }
}
__global__ __launch_bounds__(kResampleThreads) void rbpf_resample_apply(int N, int TT, const int* __restrict__ parent, const int* __restrict__ children,
                                                           const unsigned int* __restrict__ tab_old, unsigned int* __restrict__ tab_new,
                                                           unsigned int* __restrict__ shed, TilePool P, int table_blocks, int chunks,
                                                           GatherArgs ga) {
  const int b = blockIdx.x;
  if (b < table_blocks) { resample_tables_body(N, TT, parent, children, tab_old, tab_new, shed, P, b, table_blocks); return; }
  const int g = b - table_blocks;
  gather_body(N, parent, ga, g / chunks, g % chunks, chunks);
}
// ---- dense views of one particle's tiled log-odds (tbnav_rbpf_get/set_log_odds, parity hooks) ------------------
__global__ __launch_bounds__(256) void rbpf_tiles_to_dense(int xs, size_t G, TilePool P, MapT M, int p, double* __restrict__ out) {
  const unsigned int* tab = M.table + (size_t)p * M.TT;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i / xs), cj = (int)(i - (size_t)ci * xs);
    out[i] = P.lo[(size_t)tab[tile_of(M, ci, cj)] * kTileCells + in_tile(ci, cj)];
  }
}
// grid = TT workgroups of one wave: tile t of particle p takes the values of `in` and the occupancy bits they imply
// (the caller has zeroed the particle's occupied counts); a tile that is all zero in `in` and still the shared zero
// tile stays shared.
__global__ __launch_bounds__(kWave) void rbpf_dense_to_tiles(int xs, double cut_occ, TilePool P, MapT M, int p, const double* __restrict__ in,
                                                             int* __restrict__ trow_occ, int* __restrict__ n_occ, int* __restrict__ err) {
  const int t = blockIdx.x, lane = threadIdx.x, ti = t / M.TW, tj = t - ti * M.TW;
  unsigned int* tab = M.table + (size_t)p * M.TT;
  unsigned int* shed = M.shed + (size_t)p * M.TT;
  bool nz = false;
  for (int q = lane; q < kTileCells; q += kWave) {
    const int ci = ti * kTS + (q >> kTSh), cj = tj * kTS + (q & (kTS - 1));
    if (ci < xs && cj < xs && in[(size_t)ci * xs + cj] != 0.0) nz = true;
  }
  if (__ballot(nz) == 0ull && tab[t] == 0u) return;
  const unsigned int id = tile_make_private(P, tab, shed, t, lane);
  if (id == 0u) { if (lane == 0) atomicOr(&err[3], 8); return; }
  int n_occ_tile = 0;
  for (int q0 = 0; q0 < kTileCells; q0 += kWave) {  // two tile rows per trip: lanes 0-31 row 2i, 32-63 row 2i+1
    const int q = q0 + lane;
    const int ci = ti * kTS + (q >> kTSh), cj = tj * kTS + (q & (kTS - 1));
    const double v = (ci < xs && cj < xs) ? in[(size_t)ci * xs + cj] : 0.0;
    P.lo[(size_t)id * kTileCells + q] = v;
    const unsigned long long m = __ballot(v >= cut_occ);
    if (lane == 0) { P.bm[(size_t)id * kTS + (q0 >> kTSh)] = (unsigned int)m; P.bm[(size_t)id * kTS + (q0 >> kTSh) + 1] = (unsigned int)(m >> 32); }
    n_occ_tile += __popcll(m);
  }
  if (lane == 0 && n_occ_tile) { atomicAdd(&trow_occ[(size_t)p * M.TW + ti], n_occ_tile); atomicAdd(&n_occ[p], n_occ_tile); }
}
// Drop every tile reference of slot p (table and shed) and leave it with the empty map: the slot is about to receive
// an imported particle (tbnav_rbpf_import_particle_dev).
__global__ __launch_bounds__(256) void rbpf_release_slot(TilePool P, MapT M, int p) {
  unsigned int* tab = M.table + (size_t)p * M.TT;
  unsigned int* shed = M.shed + (size_t)p * M.TT;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < M.TT; t += gridDim.x * blockDim.x) {
    const unsigned int id = tab[t], sh = shed[t];
    if (id && atomicSub(&P.ref[id], 1) == 1) tile_push(P, id);
    if (sh && atomicSub(&P.ref[sh], 1) == 1) tile_push(P, sh);
    tab[t] = 0u; shed[t] = 0u;
  }
}
__global__ __launch_bounds__(256) void rbpf_gather_weights(int N, const double* __restrict__ gw, const int* __restrict__ parent, double* __restrict__ weight) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < N) weight[m] = gw[parent[m]];
}
// ---- getRobotState / newMap on the device ------------------------------------------
// arg-max weight with the reference's tie rule (strict '>', first wins, starting from 0.0:
// particle_filter.cpp:260-267): the smallest index among the maxima, 0 if no weight is positive.
__global__ __launch_bounds__(256) void rbpf_argmax(int N, const double* __restrict__ weight, const double* __restrict__ pose,
                                                   int* __restrict__ best_idx, double* __restrict__ best_pose) {
  __shared__ double sv[256];
  __shared__ int si[256];
  double bv = 0.0;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const double w = weight[i];
    if (w > bv) { bv = w; bi = i; }  // strided scan keeps the lowest index per thread for equal values
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const double ov = sv[threadIdx.x + off];
      const int oi = si[threadIdx.x + off];
      if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int idx = (sv[0] > 0.0 && si[0] != 0x7fffffff) ? si[0] : 0;
    *best_idx = idx;
    best_pose[0] = pose[idx * 3 + 0]; best_pose[1] = pose[idx * 3 + 1]; best_pose[2] = pose[idx * 3 + 2];
  }
}
__global__ __launch_bounds__(256) void rbpf_export_map(int xs, size_t G, ExportCuts cuts, const int* __restrict__ best_idx,
                                                       TilePool P, MapT M, int8_t* __restrict__ out) {
  const unsigned int* tab = M.table + (size_t)(*best_idx) * M.TT;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < G; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i / xs), cj = (int)(i - (size_t)ci * xs);
    const double l = P.lo[(size_t)tab[tile_of(M, ci, cj)] * kTileCells + in_tile(ci, cj)];
    int v;
    if (l >= cuts.half_lo && l <= cuts.half_hi) v = -1;
    else if (l >= cuts.occ_cut) v = 100;
    else if (l <= cuts.free_cut) v = 0;
    else {
      int a = 0, b = cuts.n_steps;  // number of steps <= l
      while (a < b) { const int m = (a + b) >> 1; if (cuts.step[m] <= l) a = m + 1; else b = m; }
      v = 35 + a;
    }
    const size_t row = i / xs, col = i % xs;
    out[col * xs + row] = (int8_t)v;
  }
}

}  // namespace tbnav_rk
