// rbpf_field.hip — the stored u16 distance field (modes window / full, injected and on-demand fields): dense bitmap rows
// from the tiles, the per-particle refresh window, and the exact Euclidean distance transform in its three tiers — what
// replaces the reference's whole-map priority-queue brushfire (grid_mapper.cpp:333-435) when a stored field is wanted.
#include "rbpf_device.hpp"

namespace tbnav_rk {

// The exact-transform kernels below (stored-field modes, on-demand fields) work on dense bitmap rows and per-row
// counts; this rebuilds them from the tiles for particles [p0, p0 + gridDim.y).  grid (rows/4, count), 256 threads:
// one wave per map row, lane w assembles the row's u64 word w.
__global__ __launch_bounds__(256) void rbpf_densify(GridC g, int p0, TilePool P, MapT M, const int* __restrict__ trow_occ,
                                                    unsigned long long* __restrict__ bitmap, int* __restrict__ row_count) {
  const int p = p0 + blockIdx.y;
  const int row = blockIdx.x * 4 + threadIdx.x / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  if (row >= g.xsize) return;
  const OccT occ = occ_of(P, M, trow_occ, p);
  unsigned long long* bm = bitmap + ((size_t)p * g.xsize + row) * g.words;
  int cnt = 0;
  for (int w = lane; w < g.words; w += kWave) {
    const unsigned long long v = occ.row_any(row) ? occ.word(row, w) : 0ull;
    bm[w] = v;
    cnt += __popcll(v);
  }
  cnt = wave_sum_i(cnt);
  if (lane == 0) row_count[(size_t)p * g.xsize + row] = cnt;
}
// ---- exact distance transform ------------------------------------------------------------------------
// ---- windowed refresh --------------------------------------------------------------------------------
// The distance field is recomputed from the occupancy bitmap from scratch (it has no state of its own apart
// from "cells out of reach keep their value"), and the only reader between two scans is the next scan's
// likelihood: beam end points within range_max of poses near the particle's predicted pose.  So the refresh
// runs at the START of the next SLAM call, for a window round the particle that provably contains every
// lookup of that call (checked in the likelihood: a miss is reported, never read stale); the whole field of a
// particle is produced on demand (tbnav_rbpf_get_occ_dist / get_dist_code, particle export).
// state[p]: 0 = bitmap changed since the last transform, 1 = window fresh, 2 = whole field fresh (or injected).
__global__ void rbpf_window(GridC g, int N, int half_cells, int mark_fresh, const double* __restrict__ pose, int* __restrict__ state,
                            int* __restrict__ skip, int4* __restrict__ win) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const int stt = state[p];
  skip[p] = (stt == 2) ? 1 : 0;
  int4 w = make_int4(0, g.xsize - 1, 0, g.ysize - 1);
  if (stt != 2) {
    int ci, cj;
    if (world2cell(g, pose[p * 3 + 1], pose[p * 3 + 2], ci, cj)) {
      w.x = max(0, ci - half_cells); w.y = min(g.xsize - 1, ci + half_cells);
      w.z = max(0, cj - half_cells); w.w = min(g.ysize - 1, cj + half_cells);
    }
    if (mark_fresh) state[p] = 1;
  }
  win[p] = w;
}
// grid (column tiles, N), C threads (one per column of the tile; C = 64 or 32).  LDS: the particle's
// bitmap rows, f[xsize][C] u8 (row-pass distance, 255 = none), v[xsize][C] u16 and z[xsize][C] i16
// (lower-envelope stack).  Integer arithmetic only: d2 = min_i' (i-i')^2 + f(i',j)^2 exactly.
template <int C>
__global__ __launch_bounds__(C) void rbpf_edt(GridC g, int radius, const unsigned long long* __restrict__ bitmap,
                                              uint16_t* __restrict__ codes, const int* __restrict__ tier, int my_tier, EdtJob job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int p = job.p0 + blockIdx.y;
  if (job.skip[p] || tier[p] != my_tier) return;  // fresh already / handled by a compact-row kernel
  const int4 wn = job.win[p];
  const int tile = wn.z / C + blockIdx.x;
  if (tile * C > wn.w) return;
  const int xs = g.xsize, words = g.words;
  unsigned long long* rows = reinterpret_cast<unsigned long long*>(lds_raw);            // [xs][words]
  uint16_t* v = reinterpret_cast<uint16_t*>(lds_raw + (size_t)xs * words * 8);          // [xs][C]
  int16_t* z = reinterpret_cast<int16_t*>(lds_raw + (size_t)xs * words * 8 + (size_t)xs * C * 2);  // [xs][C]
  uint8_t* f = lds_raw + (size_t)xs * words * 8 + (size_t)xs * C * 4;                   // [xs][C]
  const int lane = threadIdx.x;
  const int j = tile * C + lane;
  const unsigned long long* bm = bitmap + (size_t)p * xs * words;
  for (int t = lane; t < xs * words; t += C) rows[t] = bm[t];
  __syncthreads();
  if (j >= g.ysize) return;
  // row pass
  for (int i = 0; i < xs; ++i) f[i * C + lane] = (uint8_t)row_nearest(rows + (size_t)i * words, words, j, radius);
  // lower envelope of the parabolas (i - q)^2 + f(q)^2 over rows q with f(q) finite
  int top = -1;
  for (int q = 0; q < xs; ++q) {
    const int fq = f[q * C + lane];
    if (fq == 255) continue;
    const int hq = fq * fq + q * q;
    int s = -32768;
    while (top >= 0) {
      const int vq = v[top * C + lane];
      const int fv = f[vq * C + lane];
      s = floor_div(hq - (fv * fv + vq * vq), 2 * (q - vq));
      if (s <= z[top * C + lane]) --top; else break;
    }
    ++top;
    v[top * C + lane] = (uint16_t)q;
    if (top == 0) s = -32768;
    z[top * C + lane] = (int16_t)(s < -32768 ? -32768 : (s > 32767 ? 32767 : s));
  }
  uint16_t* out = codes + (size_t)p * xs * g.ysize;
  if (top < 0) return;  // nothing within reach of this column: every cell keeps its previous value
  const int r2 = radius * radius;
  int kk = 0;
  for (int i = wn.x; i <= wn.y; ++i) {
    while (kk < top && z[(kk + 1) * C + lane] < i) ++kk;
    const int vq = v[kk * C + lane];
    const int fv = f[vq * C + lane];
    const int d2 = (i - vq) * (i - vq) + fv * fv;
    // farther than cell_radius_: the reference never writes such a cell (grid_mapper.cpp:310-313)
    if (d2 <= r2) out[(size_t)i * g.ysize + j] = (uint16_t)d2;
  }
}
template __global__ void rbpf_edt<64>(GridC, int, const unsigned long long* __restrict__, uint16_t* __restrict__, const int* __restrict__, int, EdtJob);
template __global__ void rbpf_edt<32>(GridC, int, const unsigned long long* __restrict__, uint16_t* __restrict__, const int* __restrict__, int, EdtJob);
template <int SMAX>
__global__ __launch_bounds__(kWave) void rbpf_edt_compact(GridC g, int radius, const unsigned long long* __restrict__ bitmap,
                                                          const int* __restrict__ row_count,
                                                          uint16_t* __restrict__ codes, int* __restrict__ tier, int my_tier, EdtJob job) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int p = job.p0 + blockIdx.y, lane = threadIdx.x;
  if (job.skip[p] || tier[p] != my_tier) return;
  const int4 wn = job.win[p];
  const int tw = (wn.z >> 6) + blockIdx.x;          // the tile is exactly bitmap word `tw` of every row
  if (tw > (wn.w >> 6)) return;
  uint32_t* ent = reinterpret_cast<uint32_t*>(lds_raw);                                  // [SMAX][64] packed stack entries
  unsigned long long* roww = reinterpret_cast<unsigned long long*>(lds_raw + (size_t)SMAX * kWave * 4);  // [SMAX] tile word of the row
  uint16_t* rowlist = reinterpret_cast<uint16_t*>(lds_raw + (size_t)SMAX * kWave * 4 + (size_t)SMAX * 8);  // [SMAX]
  uint16_t* rowdl = rowlist + SMAX;   // [SMAX] distance from the tile's first column to the nearest occupied cell left of the tile
  uint16_t* rowdr = rowdl + SMAX;     // [SMAX] distance from the tile's last column to the nearest one right of it
  const int xs = g.xsize, words = g.words;
  const int j = tw * kWave + lane;
  const unsigned long long* bm = bitmap + (size_t)p * xs * words;
  const int* rc = row_count + (size_t)p * xs;
  // compact list of non-empty rows (ascending)
  int S = 0;
  for (int base = 0; base < xs; base += kWave) {
    const int row = base + lane;
    const bool ne = (row < xs) && (rc[row] != 0);
    const unsigned long long m = __ballot(ne);
    if (ne) {
      const int pos = S + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < SMAX) rowlist[pos] = (uint16_t)row;
    }
    S += __popcll(m);
  }
  if (S == 0) return;  // empty map: nothing to write
  if (S > SMAX || xs > kEdtCompactMaxRows) { if (lane == 0 && blockIdx.x == 0) tier[p] = my_tier + 1; return; }
  __syncthreads();
  // per (row, tile): the tile's own word and the distances to the nearest set bits outside the tile
  for (int s = lane; s < S; s += kWave) {
    const unsigned long long* r = bm + (size_t)rowlist[s] * words;
    roww[s] = r[tw];
    int dl = 0xFFFF, dr = 0xFFFF;
    for (int w = tw - 1; w >= 0 && (tw - w - 1) * 64 < radius; --w) {
      const unsigned long long m = r[w];
      if (m) { dl = tw * 64 - (w * 64 + 63 - __clzll((long long)m)); break; }
    }
    for (int w = tw + 1; w < words && (w - tw - 1) * 64 < radius; ++w) {
      const unsigned long long m = r[w];
      if (m) { dr = (w * 64 + (__ffsll((long long)m) - 1)) - (tw * 64 + 63); break; }
    }
    rowdl[s] = (uint16_t)dl; rowdr[s] = (uint16_t)dr;
  }
  __syncthreads();
  if (j >= g.ysize) return;
  const unsigned long long le_mask = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);  // bits <= lane
  const unsigned long long ge_mask = ~((1ull << lane) - 1ull);                        // bits >= lane
  // lower envelope over the non-empty rows; top-of-stack (v_t, f_t, z_t) lives in registers, and the
  // (wave-uniform) row record of the NEXT iteration is fetched from LDS before this one is processed
  int top = -1, v_t = 0, f_t = 0, z_t = -1;
  int nq = rowlist[0];
  unsigned long long nword = roww[0];
  int ndl = rowdl[0], ndr = rowdr[0];
  for (int s = 0; s < S; ++s) {
    const int q = nq;
    const unsigned long long word = nword;
    const int dl = ndl, dr = ndr;
    if (s + 1 < S) { nq = rowlist[s + 1]; nword = roww[s + 1]; ndl = rowdl[s + 1]; ndr = rowdr[s + 1]; }
    int fq = min(lane + dl, (63 - lane) + dr);
    const unsigned long long ml = word & le_mask, mr = word & ge_mask;
    if (ml) fq = min(fq, lane - (63 - __clzll((long long)ml)));
    if (mr) fq = min(fq, (__ffsll((long long)mr) - 1) - lane);
    if (fq > radius) continue;
    const int hq = fq * fq + q * q;
    // pop while the newcomer's intersection with the top is at or left of the top's own start:
    // floor(num/den) <= z  <=>  num < (z+1)*den  (den > 0) — no division needed to decide
    while (top >= 0) {
      const int num = hq - (f_t * f_t + v_t * v_t), den = 2 * (q - v_t);
      if (num >= (z_t + 1) * den) break;
      --top;
      if (top >= 0) unpack(ent[top * kWave + lane], v_t, f_t, z_t);
    }
    int sd = -1;
    if (top >= 0) sd = floor_div_small(hq - (f_t * f_t + v_t * v_t), 2 * (q - v_t));
    ++top;
    // z only ever meets row indices 0..xs-1: clamping it to [-1, kZMax] changes no decision that matters
    sd = sd < -1 ? -1 : (sd > kZMax ? kZMax : sd);
    v_t = q; f_t = fq; z_t = sd;
    ent[top * kWave + lane] = pack(q, fq, sd);
  }
  if (top < 0) return;  // nothing within reach of this column: every cell keeps its previous value
  uint16_t* out = codes + (size_t)p * xs * g.ysize + j;
  const int r2 = radius * radius;
  // walk the envelope; the NEXT entry is already in registers when the walk reaches its start row
  int kk = 0, vq, fv, zz, vn = 0, fn = 0, zn = 0x7fffffff;
  unpack(ent[lane], vq, fv, zz);
  if (top >= 1) unpack(ent[kWave + lane], vn, fn, zn);
  for (int i = wn.x; i <= wn.y; ++i) {
    while (zn < i) {
      ++kk;
      vq = vn; fv = fn;
      if (kk < top) unpack(ent[(kk + 1) * kWave + lane], vn, fn, zn); else zn = 0x7fffffff;
    }
    const int d2 = (i - vq) * (i - vq) + fv * fv;
    if (d2 <= r2) out[(size_t)i * g.ysize] = (uint16_t)d2;
  }
}
template __global__ void rbpf_edt_compact<kEdtRowsA>(GridC, int, const unsigned long long* __restrict__, const int* __restrict__, uint16_t* __restrict__, int* __restrict__, int, EdtJob);
template __global__ void rbpf_edt_compact<kEdtRowsB>(GridC, int, const unsigned long long* __restrict__, const int* __restrict__, uint16_t* __restrict__, int* __restrict__, int, EdtJob);

}  // namespace tbnav_rk
