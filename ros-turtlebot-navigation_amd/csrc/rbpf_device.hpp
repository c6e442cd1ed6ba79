// rbpf_device.hpp — what the RBPF kernel files (rbpf_propose.hip, rbpf_raycast.hip, rbpf_field.hip, rbpf_resample.hip,
// rbpf_migrate.hip) and the host side (rbpf.hip: handle, launches, C-ABI) share: the launch-argument structs, the tiled
// copy-on-write map's accessors, world -> cell, the distance lookups, wave reductions, the exact normalise / selection body
// (it runs in a kernel of its own AND as workgroup 0 of the map update), and the declarations of every kernel.  Kernels are
// defined in their family's file and launched from rbpf.hip; the template kernels are explicitly instantiated where they are
// defined.  Everything is compiled -ffp-contract=off (csrc/Makefile): grid indices, log-odds, Neff and parent lists are
// bit-exact targets.
#ifndef TBNAV_RBPF_DEVICE_HPP
#define TBNAV_RBPF_DEVICE_HPP
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.hpp"
#include "tbnav_rbpf.h"

namespace tbnav_rk {

constexpr double kPI = 3.14159265358979323846;  // rigid2d.hpp:13
constexpr int kWave = 64;
#ifndef TBNAV_PROPOSE_THREADS
#define TBNAV_PROPOSE_THREADS 256
#endif
#ifndef TBNAV_PROPOSE_WAVES
#define TBNAV_PROPOSE_WAVES 4  // four workgroups of four waves per CU (what its LDS allows): 128 VGPRs
#endif
#ifndef TBNAV_PROPOSE_KSB
#define TBNAV_PROPOSE_KSB 1
#endif
constexpr int kProposeThreads = TBNAV_PROPOSE_THREADS;
constexpr int kUnCap = 16;  // unstable beams handled by the per-pair path of the proposal kernel
static_assert(kProposeThreads >= 128 && kProposeThreads % 64 == 0, "wave 0 samples, the other waves look the beams up");
constexpr uint16_t kCodeUnreached = 0xFFFF;
constexpr int kMaxLds = 160 * 1024;

// ---- small math shared by host and device ----------------------------------------------------------
__host__ __device__ inline bool almost_equal(double a, double b, double eps = 1.0e-12) { return fabs(a - b) < eps; }
__host__ __device__ inline double normalize_angle_PI(double rad) {  // rigid2d.hpp:52-64
  const double q = floor((rad + kPI) / (2.0 * kPI));
  rad = (rad + kPI) - q * 2.0 * kPI;
  if (rad < 0) rad += 2.0 * kPI;
  return (rad - kPI);
}

__device__ __forceinline__ int floor_div_small(int num, int den);  // exact floor(num/den), |num| < 2^24, 0 < den < 2^13

// Development build (-DTBNAV_PHASE_PROF): per-phase wall-clock stamps inside the proposal and raycast kernels, summed
// over workgroups and printed by tbnav_rbpf_destroy.  The stamps add barriers and global atomics — the kernels
// run measurably slower with them; the numbers are for comparing phases, not for the bench.
#ifdef TBNAV_PHASE_PROF
static __device__ unsigned long long g_trace_p[2][4][16];  // [which][wave][stamp] of TWO proposal workgroups (blockIdx.x == 96, 100: XCCs 0 and 4)
#define TRACE_P(i) do { if ((blockIdx.x == 96 || blockIdx.x == 100) && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 4) g_trace_p[blockIdx.x == 100][threadIdx.x >> 6][i] = wall_clock64(); } while (0)
static __device__ unsigned long long g_wgp[4096][3];   // [workgroup] entry, exit (10 ns ticks), XCC_ID << 32 | HW_ID of the LAST proposal launch
#define WGP_IN() do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_wgp[blockIdx.x][0] = wall_clock64(); \
  g_wgp[blockIdx.x][2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned int)__builtin_amdgcn_s_getreg(63492); } } while (0)
#define WGP_OUT() do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_wgp[blockIdx.x][1] = wall_clock64(); } while (0)
#else
#define TRACE_P(i)
#define WGP_IN()
#define WGP_OUT()
#endif
struct GridC {
  double xmin, xmax, ymin, ymax, res;
  int xsize, ysize, words;  // words = ceil(ysize / 64) u64 per bitmap row
  double max_occ_dist;
  double inv_res;  // fl(1/res), for the guarded fast path of world2cell
};

struct ScanC {  // everything constant during one SLAM call
  GridC g;
  int N, k, Bv, icp_ok;
  double Trs[3];                 // theta, x, y
  double z_hit, var_hit, sqrt_inv_hit, rand_term;  // mixture: z_hit * N(z;0,var) + z_rand/z_max
  double Ld[3], Lm[3];           // sqrt of sample_range / motion_noise diagonals (LLT of a diagonal)
  double scan_min, scan_max, pose_min, pose_max;
  double a1, a2, a3, a4;
  double rot1, trans, rot2;      // odometry deltas, particle-independent (particle_filter.cpp:393-403)
  double Ticp[3];
  double u[3];                   // w, vx, vy
  double d_free, d_occ, cut_occ; // log-odds increments and the host-derived occupied cut-off
  int stride_normals;            // 3k+3 or 3
  int p0;                        // first particle of the launch (0 for a whole-filter update)
  double rmax;                   // longest valid beam of this scan
};

// ---- tiled copy-on-write log-odds maps -------------------------------------------------------------------
// The reference gives every particle its own dense map and deep-copies it when a particle is resampled
// (particle_filter.cpp:125-138, :495).  Here a particle's log-odds are a TABLE of kTS x kTS-cell tiles drawn from
// one pool shared by all particles of the handle:
//   table[p][ti * TW + tj] = id of the tile holding cells (32*ti .. 32*ti+31, 32*tj .. 32*tj+31); id 0 = the shared
//   all-zero tile (a cell nobody has touched has log-odds 0 = log_odds_prior_, grid_mapper.cpp:42-58);
//   ref[id] = how many table (and shed) entries name the tile.
// A resample copies tables and adjusts counts (rbpf_resample_apply) instead of copying maps;
// the raycast makes a tile private on first write (tile_make_private): it takes a fresh tile from the free ring,
// copies (or zero-fills) 8 KB, and notes the tile it left in shed[p][t].  Counts of shared tiles are NOT touched
// while a scan runs (every sharer sees a stable count > 1 and copies); the shed notes are settled at the next
// resample, which is also the only time tiles return to the ring.  Pops (scan) and pushes (resample) therefore
// never run concurrently and the ring needs no ABA protection.
constexpr int kTS = 32, kTSh = 5, kTileCells = kTS * kTS;
struct TilePool {
  double* lo;               // [cap][kTileCells], in-tile index = (i & 31) * 32 + (j & 31)
  unsigned int* bm;         // [cap][kTS] occupancy bits of the tile's cells (prob >= 0.90): row i & 31, bit j & 31
  int* ref;                 // [cap]
  unsigned int* ring;       // [cap] free tile ids
  unsigned long long* ctr;  // [0] head: tiles popped, [1] tail: tiles pushed (free = tail - head)
  unsigned int cap;
};
struct MapT {
  unsigned int* table;  // [N][TT] of the current buffer
  unsigned int* shed;   // [N][TT] tile this slot stopped using since the last resample (0 = none)
  int TW, TT;           // tiles per side, tiles per map
};
__device__ __forceinline__ int tile_of(const MapT& M, int ci, int cj) { return (ci >> kTSh) * M.TW + (cj >> kTSh); }
__device__ __forceinline__ int in_tile(int ci, int cj) { return ((ci & (kTS - 1)) << kTSh) | (cj & (kTS - 1)); }
__device__ __forceinline__ unsigned int tile_pop(const TilePool& P) {  // 0 = pool exhausted
  const unsigned long long pos = atomicAdd(P.ctr, 1ull);
  if (pos >= P.ctr[1]) { atomicAdd(P.ctr, ~0ull); return 0u; }  // (no push can be in flight: see above)
  return P.ring[pos % P.cap];
}
__device__ __forceinline__ void tile_push(const TilePool& P, unsigned int id) {
  const unsigned long long pos = atomicAdd(P.ctr + 1, 1ull);
  P.ring[pos % P.cap] = id;
}
// n tiles at once: ONE atomic on the ring's head per caller (a workgroup that clones 15 tiles after a resample would
// otherwise queue 15 times on a word every other workgroup is queueing on — a single address retires ~90 atomics
// per microsecond).  Returns the position of the first tile in the ring, ~0 if fewer than n are free.
__device__ __forceinline__ unsigned long long tile_pop_n(const TilePool& P, unsigned int n) {
  const unsigned long long pos = atomicAdd(P.ctr, (unsigned long long)n);
  if (pos + n > P.ctr[1]) { atomicAdd(P.ctr, ~(unsigned long long)n + 1ull); return ~0ull; }
  return pos;
}
__device__ __forceinline__ unsigned int tile_at(const TilePool& P, unsigned long long pos) { return P.ring[pos % P.cap]; }
__device__ __forceinline__ bool tile_is_private(const TilePool& P, const unsigned int* __restrict__ table_p, int t) {
  const unsigned int id = table_p[t];
  return id != 0u && P.ref[id] == 1;
}
// Tile t of one particle becomes the fresh tile nid, filled from the tile it named so far (all 64 lanes of a wave).
__device__ __forceinline__ void tile_clone_into(const TilePool& P, unsigned int* __restrict__ table_p, unsigned int* __restrict__ shed_p,
                                                int t, unsigned int nid, int lane) {
  const unsigned int id = table_p[t];
  double2* dst = reinterpret_cast<double2*>(P.lo + (size_t)nid * kTileCells);
  const double2* src = reinterpret_cast<const double2*>(P.lo + (size_t)id * kTileCells);  // id 0 = the zero tile
#pragma unroll
  for (int i = 0; i < kTileCells / 2 / kWave; ++i) dst[i * kWave + lane] = src[i * kWave + lane];
  if (lane < kTS) P.bm[(size_t)nid * kTS + lane] = P.bm[(size_t)id * kTS + lane];
  if (lane == 0) {
    P.ref[nid] = 1;
    table_p[t] = nid;
    if (id != 0u) shed_p[t] = id;  // a (p, t) entry leaves a shared tile at most once between two resamples
  }
}
// Make tile t of one particle private to it (called by all 64 lanes of a wave, wave-uniform arguments).
// Returns the tile's id, 0 if the pool is exhausted.
__device__ __forceinline__ unsigned int tile_make_private(const TilePool& P, unsigned int* __restrict__ table_p,
                                                          unsigned int* __restrict__ shed_p, int t, int lane) {
  const unsigned int id = table_p[t];
  if (id != 0u && P.ref[id] == 1) return id;
  unsigned int nid = 0u;
  if (lane == 0) nid = tile_pop(P);
  nid = __shfl(nid, 0, kWave);
  if (nid == 0u) return 0u;
  double2* dst = reinterpret_cast<double2*>(P.lo + (size_t)nid * kTileCells);
  const double2* src = reinterpret_cast<const double2*>(P.lo + (size_t)id * kTileCells);  // id 0 = the zero tile
#pragma unroll
  for (int i = 0; i < kTileCells / 2 / kWave; ++i) dst[i * kWave + lane] = src[i * kWave + lane];
  if (lane < kTS) P.bm[(size_t)nid * kTS + lane] = P.bm[(size_t)id * kTS + lane];
  if (lane == 0) {
    P.ref[nid] = 1;
    table_p[t] = nid;
    if (id != 0u) shed_p[t] = id;  // a (p, t) entry leaves a shared tile at most once between two resamples
  }
  return nid;
}
// One particle's occupancy bits, read through its tile table.  trow[ti] = occupied cells in tile row ti (cells
// 32*ti .. 32*ti+31 of the x axis): lets a search skip 32 map rows at a time.
struct OccT {
  const unsigned int* bm;   // pool.bm
  const unsigned int* tab;  // the particle's table
  const int* trow;          // [TW]
  int TW;
  // columns 64w .. 64w+63 of map row r, as the dense bitmap's u64 word was: two tiles side by side
  __device__ __forceinline__ unsigned long long word(int r, int w) const {
    const unsigned int* t = tab + (r >> kTSh) * TW + 2 * w;
    const unsigned int lo = bm[(size_t)t[0] * kTS + (r & (kTS - 1))];
    const unsigned int hi = (2 * w + 1 < TW) ? bm[(size_t)t[1] * kTS + (r & (kTS - 1))] : 0u;
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
  }
  __device__ __forceinline__ bool row_any(int r) const { return trow[r >> kTSh] != 0; }
};
__device__ __forceinline__ OccT occ_of(const TilePool& P, const MapT& M, const int* trow_occ, int p) {
  return OccT{P.bm, M.table + (size_t)p * M.TT, trow_occ + (size_t)p * M.TW, M.TW};
}

// world -> cell, grid_mapper.cpp:810-887.  false = outside the world (the reference throws).
// The reference's cell is floor(fl(fl(x - xmin) / res)).  An f64 division costs ~25 instructions, and this runs
// once per (sample, beam): so the quotient is first formed with the reciprocal (q~ = fl(d * fl(1/res)), off the
// exact quotient by < 4 ulp, i.e. < 2e-11 cells for maps up to 2^15 cells a side) and used when it is further
// than 1e-9 from a cell border — then floor(q~) IS the reference's floor; only a point that close to a border
// takes the division.  Bit-identical by construction (and checked against the oracle's division).
__device__ __forceinline__ double cell_floor(double d, const GridC& g) {
  const double q = d * g.inv_res;
  double f = floor(q);
  const double fr = q - f;
  if (!(fr > 1e-9 && fr < 1.0 - 1e-9)) f = floor(d / g.res);
  return f;
}
__device__ __forceinline__ bool world2cell(const GridC& g, double x, double y, int& ci, int& cj) {
  if (!(x >= g.xmin && x <= g.xmax)) return false;
  if (!(y >= g.ymin && y <= g.ymax)) return false;
  double fi = cell_floor(x - g.xmin, g);
  if (fi == g.xsize) fi -= 1.0;
  double fj = cell_floor(y - g.ymin, g);
  if (fj == g.ysize) fj -= 1.0;
  ci = (int)fi;
  cj = (int)fj;
  return true;
}

__device__ __forceinline__ double code_to_dist(const GridC& g, uint16_t code) {
  return code == kCodeUnreached ? g.max_occ_dist : sqrt((double)code) * g.res;
}

// grid_mapper.cpp:18-28 with the variance check hoisted (err set by the caller)
__device__ __forceinline__ double pdf_normal(double a, double b) {
  const double sqrt_inv = 1.0 / sqrt(2.0 * kPI * b);
  const double var = -0.5 * (a * a) / b;
  return sqrt_inv * exp(var);
}

// Wave-wide reductions of doubles without LDS round trips (__shfl_xor is ds_bpermute: six dependent LDS-latency steps per
// reduction, two permutes each for a double): an inclusive scan inside each row of 16 lanes by DPP shifts, then the row
// totals carried down the rows (row_bcast:15 / :31); lane 63 holds the result, which is handed to every lane.  A fixed
// order of operations, the same on every call (the proposal kernel's sums and products are compared with the oracle at
// 1e-9, not bit for bit).
#define TBNAV_DPP_D(v, ident, ctrl, rmask)                                                                                      \
  __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), ctrl, rmask, 0xf, false),              \
                   __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), ctrl, rmask, 0xf, false))
template <class Op> __device__ __forceinline__ double wave_reduce_dpp_d(double v, double ident, Op op) {
  v = op(v, TBNAV_DPP_D(v, ident, 0x111, 0xf));  // row_shr:1
  v = op(v, TBNAV_DPP_D(v, ident, 0x112, 0xf));  // row_shr:2
  v = op(v, TBNAV_DPP_D(v, ident, 0x114, 0xf));  // row_shr:4
  v = op(v, TBNAV_DPP_D(v, ident, 0x118, 0xf));  // row_shr:8
  v = op(v, TBNAV_DPP_D(v, ident, 0x142, 0xa));  // row_bcast:15 into rows 1, 3
  v = op(v, TBNAV_DPP_D(v, ident, 0x143, 0xc));  // row_bcast:31 into rows 2, 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_max_d(double v) { return wave_reduce_dpp_d(v, -1.0e300, [](double a, double b) { return fmax(a, b); }); }
__device__ __forceinline__ double wave_sum_d(double v) { return wave_reduce_dpp_d(v, 0.0, [](double a, double b) { return a + b; }); }
__device__ __forceinline__ double wave_prod(double v) { return wave_reduce_dpp_d(v, 1.0, [](double a, double b) { return a * b; }); }
// A value every lane of the workgroup holds alike (the particle's pose, what is derived from it): into scalar registers — the
// proposal kernel lives at its 128-VGPR ceiling, and these are a dozen doubles that stay live across its phases.
__device__ __forceinline__ double uniform_d(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// GridMapper::likelihoodFieldModel for ONE pose, evaluated by one wave (lanes stride the valid beams).
// beams[b] = (r*cos a_b, r*sin a_b) in the sensor frame, built on the host exactly as
// sensor_model.cpp:73-108 does.  Returns the product in every lane; *oob is set if a beam leaves
// the world (the reference throws from world2RowMajor).
template <class Word> __device__ __forceinline__ int row_nearest_f(Word word, int words, int j, int cap);
__device__ __forceinline__ int row_nearest(const unsigned long long* row, int words, int j, int cap);

// Where a lookup gets its distance code from.
//  field  : the particle's u16 field is authoritative (injected, or whole-field fresh) -> read it
//  window : the field was refreshed inside `win` for this call -> read it, report a lookup outside the window
//  query  : no field refresh at all — the squared distance to the nearest occupied cell is computed from the
//           occupancy bitmap at the looked-up cell: rows i, i+-1, i+-2, ... each contribute (dr^2 + nearest set
//           bit in that row)^2 and the walk stops once dr^2 >= best.  A beam ends on or next to a wall, so this
//           is a handful of rows; the result is the exact transform's value (same integer arithmetic), and a
//           cell with no obstacle within cell_radius keeps its stored code, like the transform.
struct DistSrc {
  const uint16_t* code;             // [G] of the particle; NULL when the handle keeps no stored field (query mode only)
  OccT occ;                         // the particle's occupancy bits (tiled)
  int4 win;
  int mode;                         // 0 field, 1 window, 2 query
  // query mode, optional: the part of the bitmap round the particle held in LDS (rows R0..R1, 64-cell word
  // columns W0..W0+nW-1; any[r] = row r has a set bit inside those columns).  nW == 0: no tile.
  const unsigned long long* tbm;
  const int* tany;
  int R0, R1, W0, nW;
  // optional, with the LDS tile: lut7[m] = least (c - 3)^2 over the set bits c of the 7-bit pattern m (100: none) — lets a
  // lookup read the 7 x 7 cells round it as seven table look-ups instead of seven 64-column bit scans
  const unsigned char* lut7;
};
// Walk rows i, i+-1, i+-2, ... of an occupancy bitmap (stride `words` u64 per row, rows row_lo..row_hi present,
// cell columns [0, words*64) relative to the bitmap) and return the least squared distance found (INT_MAX: none
// within `radius`).  row_any(r) says whether row r can hold a set bit.
template <class RowWord, class RowAny>
__device__ __forceinline__ int nearest_d2_rows(RowWord row_word, int words, int row_lo, int row_hi, int radius,
                                               int ci, int cj, RowAny row_any) {
  int best = 0x7fffffff;
  for (int dr = 0; dr <= radius; ++dr) {
    if (dr * dr >= best) break;
    if (ci + dr > row_hi && ci - dr < row_lo) break;
    for (int sg = 0; sg < (dr ? 2 : 1); ++sg) {
      const int r = sg ? ci - dr : ci + dr;
      if (r < row_lo || r > row_hi || !row_any(r)) continue;
      int cap = radius;
      if (best != 0x7fffffff) { cap = (int)sqrtf((float)(best - dr * dr)) + 1; cap = cap < radius ? cap : radius; }
      const int f = row_nearest_f([&](int w) { return row_word(r, w); }, words, cj, cap);
      if (f != 255) { const int cand = dr * dr + f * f; best = cand < best ? cand : best; }
    }
  }
  return best;
}
// The whole search.  Inlined by the scan matcher (~100 poses x Bv lookups per particle, many of them beyond the 7 x 7 look);
// the proposal kernel inlines a lookup at four places, and with both row walks in each of them it was ~100 KB of code against
// a 64 KB instruction cache shared by two CUs: there only the 7 x 7 look on the LDS tile is inline (it decides nearly
// every lookup of a beam that ends on or next to a wall) and the rest is ONE out-of-line copy.
__device__ __forceinline__ uint16_t nearest_code_query_body(const GridC& g, const DistSrc& d, int radius, int ci, int cj);
__device__ __attribute__((noinline)) uint16_t nearest_code_query_full(const GridC g, const DistSrc d, int radius, int ci, int cj) {
  return nearest_code_query_body(g, d, radius, ci, cj);
}
template <bool OUTLINE = true>
__device__ __forceinline__ uint16_t nearest_code_query(const GridC& g, const DistSrc& d, int radius, int ci, int cj) {
  if constexpr (!OUTLINE) return nearest_code_query_body(g, d, radius, ci, cj);
  if (d.nW > 0 && d.lut7) {
    const int C0 = d.W0 * 64, C1 = (d.W0 + d.nW) * 64 - 1;
    const int p0 = cj - C0 - 3, wi = p0 >> 5;
    if (ci - 3 >= d.R0 && ci + 3 <= d.R1 && p0 >= 0 && wi + 1 < 2 * d.nW && cj <= C1) {
      int clear = radius + 1;
      if (d.R0 > 0) clear = min(clear, ci - d.R0 + 1);
      if (d.R1 < g.xsize - 1) clear = min(clear, d.R1 - ci + 1);
      if (C0 > 0) clear = min(clear, cj - C0 + 1);
      if (C1 < g.ysize - 1) clear = min(clear, C1 - cj + 1);
      const unsigned int* t32 = reinterpret_cast<const unsigned int*>(d.tbm) + wi;
      const int sh = p0 & 31, stride = 2 * d.nW;
      int bw = 0x7fffffff;
#pragma unroll
      for (int dr = -3; dr <= 3; ++dr) {
        const unsigned int* rp = t32 + (ci + dr - d.R0) * stride;
        const unsigned int pat = __builtin_amdgcn_alignbit(rp[1], rp[0], sh) & 0x7Fu;
        bw = min(bw, dr * dr + (int)d.lut7[pat]);
      }
      if (bw <= 9 && bw <= clear * clear) return (uint16_t)bw;
    }
  }
  return nearest_code_query_full(g, d, radius, ci, cj);
}
__device__ __forceinline__ uint16_t nearest_code_query_body(const GridC& g, const DistSrc& d, int radius, int ci, int cj) {
  if (d.nW > 0) {
    // LDS tile first.  Its answer is the map's answer when no cell outside the tile can be nearer: a side of the
    // tile that is not the map's own border is (distance to that side + 1) cells away at least.
    const int C0 = d.W0 * 64, C1 = (d.W0 + d.nW) * 64 - 1;
    if (ci >= d.R0 && ci <= d.R1 && cj >= C0 && cj <= C1) {
      int clear = radius + 1;  // nothing beyond the radius matters
      if (d.R0 > 0) clear = min(clear, ci - d.R0 + 1);
      if (d.R1 < g.xsize - 1) clear = min(clear, d.R1 - ci + 1);
      if (C0 > 0) clear = min(clear, cj - C0 + 1);
      if (C1 < g.ysize - 1) clear = min(clear, C1 - cj + 1);
      bool looked7 = false;
      {
        // A beam ends on or next to a wall: the 7 x 7 cells round the looked-up cell first.  Every cell outside them is
        // >= 4 cells away, so a result <= 9 (and <= clear^2) is the map's answer.  Row by row: the seven bits round the
        // column (one v_alignbit on two adjacent dwords of the LDS tile) index a 128-entry table of least column offsets.
        const int p0 = cj - C0 - 3, wi = p0 >> 5;
        if (d.lut7 && ci - 3 >= d.R0 && ci + 3 <= d.R1 && p0 >= 0 && wi + 1 < 2 * d.nW) {
          const unsigned int* t32 = reinterpret_cast<const unsigned int*>(d.tbm) + wi;
          const int sh = p0 & 31, stride = 2 * d.nW;
          int bw = 0x7fffffff;
#pragma unroll
          for (int dr = -3; dr <= 3; ++dr) {
            const unsigned int* rp = t32 + (ci + dr - d.R0) * stride;
            const unsigned int pat = __builtin_amdgcn_alignbit(rp[1], rp[0], sh) & 0x7Fu;
            bw = min(bw, dr * dr + (int)d.lut7[pat]);
          }
          if (bw <= 9 && bw <= clear * clear) return (uint16_t)bw;
          looked7 = true;
        }
      }
      if (!looked7) {
        // (no table, or the 7 x 7 window sticks out of the tile) the same 7 rows, 64 columns each, by bit scans, branch-free
        const int cjr = cj - C0, s0 = cjr - 32, w = s0 >> 6, sh = s0 & 63;
        int bw = 0x7fffffff;
#pragma unroll
        for (int dr = -3; dr <= 3; ++dr) {
          const int r = ci + dr;
          if (r < d.R0 || r > d.R1) continue;
          const unsigned long long* row = d.tbm + (size_t)(r - d.R0) * d.nW;
          const unsigned long long lo64 = (w >= 0 && w < d.nW) ? row[w] : 0ull, hi64 = (w + 1 >= 0 && w + 1 < d.nW) ? row[w + 1] : 0ull;
          const unsigned long long W = sh ? ((lo64 >> sh) | (hi64 << (64 - sh))) : lo64;  // bit i = column s0 + i, the cell at bit 32
          const unsigned long long L = W & 0x1FFFFFFFFull, Rr = W >> 33;
          int f = 1 << 12;
          if (L) f = __clzll((long long)L) - 31;
          if (Rr) f = min(f, __ffsll((long long)Rr));
          bw = min(bw, dr * dr + f * f);
        }
        if (bw <= 9 && bw <= clear * clear) return (uint16_t)bw;
      }
      const int* any = d.tany;
      const int R0 = d.R0;
      const unsigned long long* tbm = d.tbm;
      const int nW = d.nW;
      const int best = nearest_d2_rows([tbm, nW, R0](int r, int w) { return tbm[(size_t)(r - R0) * nW + w]; }, d.nW, d.R0, d.R1, radius, ci, cj - C0,
                                       [any, R0](int r) { return any[r - R0] != 0; });
      if (best != 0x7fffffff && best <= clear * clear && best <= radius * radius) return (uint16_t)best;
      if (best == 0x7fffffff && clear > radius) return d.code ? d.code[(size_t)ci * g.xsize + cj] : kCodeUnreached;
    }
  }
  const OccT occ = d.occ;
  const int best = nearest_d2_rows([&occ](int r, int w) { return occ.word(r, w); }, g.words, 0, g.xsize - 1, radius, ci, cj,
                                   [&occ](int r) { return occ.row_any(r); });
  // nothing within cell_radius_: the stored code if the handle keeps a stored field (injected / materialised), else
  // "never reached" (the reference keeps whatever an earlier brushfire left there, grid_mapper.cpp:310-313)
  return (best <= radius * radius) ? (uint16_t)best : (d.code ? d.code[(size_t)ci * g.xsize + cj] : kCodeUnreached);
}
// Distance code of cell (ci, cj), or -1 when a windowed lookup falls outside the refreshed window.
template <bool OUTLINE = true>
__device__ __forceinline__ int lookup_code(const GridC& g, const DistSrc& d, int radius, int ci, int cj) {
  if (d.mode == 2) return nearest_code_query<OUTLINE>(g, d, radius, ci, cj);
  if (d.mode == 1 && (ci < d.win.x || ci > d.win.y || cj < d.win.z || cj > d.win.w)) return -1;
  return d.code[(size_t)ci * g.xsize + cj];
}

// Mixture term of one beam as a function of the distance code it lands on (grid_mapper.cpp:119-121).
__device__ __forceinline__ double beam_mixture(const ScanC& c, uint16_t code) {
  const double z = code_to_dist(c.g, code);
  double pz = 0.0;
  pz += c.z_hit * (c.sqrt_inv_hit * exp(-0.5 * (z * z) / c.var_hit));
  pz += c.rand_term;
  return pz;
}

// ctag/ccell/cpz (nullable): per-beam cache filled once per particle for the centre of its k samples — the
// samples lie within ~1e-4 m of it, so nearly every (sample, beam) lands on the same cell (no lookup at all) or at
// least the same code, and takes its mixture term from LDS instead of re-evaluating sqrt + exp.  Read-only here;
// a miss computes the term afresh.
// Tms = T(pose) * Trs  (rigid2d.cpp:214-224) as (X, Y, sin, cos); Trs.theta == 0 (the shipped robot) needs one sincos
__device__ __forceinline__ void sensor_transform(const ScanC& c, double th, double x, double y, double out[4]) {
  double s0, c0;
  sincos(th, &s0, &c0);
  out[0] = c0 * c.Trs[1] - s0 * c.Trs[2] + x;
  out[1] = s0 * c.Trs[1] + c0 * c.Trs[2] + y;
  if (c.Trs[0] == 0.0) { out[2] = s0; out[3] = c0; }  // th + 0.0 == th: same bits
  else sincos(th + c.Trs[0], &out[2], &out[3]);
}
// Mixture term of one beam seen from one sensor pose (grid_mapper.cpp:100-121).  (cc, tg, pzc) is the beam's cache
// entry — cell / code / term at the centre of the particle's samples (0xFFFFFFFF: none): the samples lie within
// ~1e-4 m of the centre, so nearly every (sample, beam) lands on the same cell (no lookup at all) or at least the
// same code, and takes its term from the cache instead of re-evaluating sqrt + exp.  A beam that leaves the world
// sets *oob (the reference throws from world2RowMajor) and contributes 1.
// The mixture term depends on the distance code and on constants fixed at create (z_hit, sigma_hit, z_rand / z_max,
// resolution, max_occ_dist): the handle tabulates it ONCE for the codes below kMixLut (rbpf_mix_lut, same device code
// as beam_mixture -> same bits) and the kernels read the table — its first kMixLds entries from LDS, the rest from
// global memory — instead of a square root, a division and an exponential per beam.
constexpr int kMixLut = 1024, kMixLds = 128;
struct MixLut { const double* lds; const double* glob; };  // either may be NULL
__device__ __forceinline__ double mix_term(const ScanC& c, const MixLut& L, int cd) {
  if (L.lds && cd < kMixLds) return L.lds[cd];
  if (L.glob && cd < kMixLut) return L.glob[cd];
  return beam_mixture(c, (uint16_t)cd);
}
__device__ __forceinline__ double beam_factor(const ScanC& c, const DistSrc& ds, int radius, const double2 pt, double X, double Y,
                                              double st, double ct, unsigned int cc, unsigned int tg, double pzc, int* oob,
                                              const MixLut& L = MixLut{nullptr, nullptr}) {
  const double ex = ct * pt.x - st * pt.y + X;
  const double ey = st * pt.x + ct * pt.y + Y;
  int ci, cj;
  if (!world2cell(c.g, ex, ey, ci, cj)) { *oob |= 1; return 1.0; }
  if (cc == (unsigned int)(ci * c.g.xsize + cj)) return pzc;  // same cell -> same code -> same term
  // (window mode: the window is sized so that a miss cannot happen — if it ever does it is reported, never read stale)
  const int cd = lookup_code(c.g, ds, radius, ci, cj);
  if (cd < 0) { *oob |= 2; return 1.0; }
  return (tg == (unsigned int)cd) ? pzc : mix_term(c, L, cd);
}
// GridMapper::likelihoodFieldModel for ONE pose, evaluated by one wave (lanes stride the valid beams).
__device__ __forceinline__ double wave_scan_likelihood_t(const ScanC& c, const double2* __restrict__ beams,
                                                         const DistSrc& ds, int radius, int n_occ,
                                                         double X, double Y, double st, double ct, int lane, int* oob,
                                                         const MixLut& L = MixLut{nullptr, nullptr}) {
  if (n_occ == 0) return 1.0;  // grid_mapper.cpp:94-98
  double p = 1.0;
  for (int b = lane; b < c.Bv; b += kWave) p *= beam_factor(c, ds, radius, beams[b], X, Y, st, ct, 0xFFFFFFFFu, 0xFFFFFFFFu, 0.0, oob, L);
  return wave_prod(p);
}
__device__ __forceinline__ double wave_scan_likelihood(const ScanC& c, const double2* __restrict__ beams,
                                                       const DistSrc& ds, int radius, int n_occ,
                                                       double th, double x, double y, int lane, int* oob,
                                                       const MixLut& L = MixLut{nullptr, nullptr}) {
  if (n_occ == 0) return 1.0;
  double T[4];
  sensor_transform(c, th, x, y, T);
  return wave_scan_likelihood_t(c, beams, ds, radius, n_occ, T[0], T[1], T[2], T[3], lane, oob, L);
}

// particle_filter.cpp:383-437 (odometry part precomputed on the host: rot1, trans, rot2)
// nrot1 / nrot2: normalize_angle_PI(c.rot1) / (c.rot2), particle- and sample-independent (the caller keeps them in scalar registers)
__device__ __forceinline__ double pose_likelihood_odom(const ScanC& c, const double* cur, const double* prev, int* var_err, double nrot1, double nrot2) {
  const double rot1_hat = atan2(cur[2] - prev[2], cur[1] - prev[1]) - prev[0];
  const double dx = cur[1] - prev[1], dy = cur[2] - prev[2];
  const double trans_hat = sqrt(dx * dx + dy * dy);
  const double rot2_hat = normalize_angle_PI(normalize_angle_PI(cur[0]) - normalize_angle_PI(prev[0]) - rot1_hat);
  const double temp1 = c.a1 * rot1_hat * rot1_hat + c.a2 * trans_hat * trans_hat;
  const double temp2 = c.a3 * trans_hat * trans_hat + c.a4 * rot1_hat * rot1_hat + c.a4 * rot2_hat * rot2_hat;
  const double temp3 = c.a1 * rot2_hat * rot2_hat + c.a2 * trans_hat * trans_hat;
  if (almost_equal(temp1, 0.0) || almost_equal(temp2, 0.0) || almost_equal(temp3, 0.0)) { *var_err = 1; return 0.0; }
  const double p1 = pdf_normal(normalize_angle_PI(nrot1 - normalize_angle_PI(rot1_hat)), temp1);
  const double p2 = pdf_normal(c.trans - trans_hat, temp2);
  const double p3 = pdf_normal(normalize_angle_PI(nrot2 - normalize_angle_PI(rot2_hat)), temp3);
  return p1 * p2 * p3;
}

// Eigen 3.3 unblocked lower LLT of a 3x3 (stops at a non-positive pivot, like llt_inplace)
__device__ inline void llt3(const double A[3][3], double L[3][3]) {
  double M[3][3];
  for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) M[r][q] = A[r][q];
  for (int kk = 0; kk < 3; ++kk) {
    double x = M[kk][kk];
    if (kk > 0) { double sq = 0.0; for (int q = 0; q < kk; ++q) sq += M[kk][q] * M[kk][q]; x -= sq; }
    if (x <= 0.0) break;
    x = sqrt(x);
    M[kk][kk] = x;
    for (int r = kk + 1; r < 3; ++r) {
      if (kk > 0) { double dot = 0.0; for (int q = 0; q < kk; ++q) dot += M[r][q] * M[kk][q]; M[r][kk] -= dot; }
      M[r][kk] /= x;
    }
  }
  for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) L[r][q] = (q <= r) ? M[r][q] : 0.0;
}

// ---- production noise source: standard normals drawn on the device (normals == NULL) --------------------
// Philox4x32-10 keyed by the handle's seed, counter = scan_index * 2^40 + pair index; each counter value
// yields one Box-Muller pair.  Replaces the host's mt19937_64 draws (particle_filter.cpp:25-34) when
// reproducibility against the CPU path is not needed; same layout as the host stream.
__device__ __forceinline__ void philox4x32_10(unsigned long long ctr, unsigned long long key, unsigned int (&out)[4]) {
  unsigned int c0 = (unsigned int)ctr, c1 = (unsigned int)(ctr >> 32), c2 = 0u, c3 = 0u;
  unsigned int k0 = (unsigned int)key, k1 = (unsigned int)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned int)p1;
    const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned int)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct Trace {
  double *sampled, *p_scan, *p_pose, *mu, *sigma, *eta, *new_pose, *weight_raw;
};





// ---- per-particle scan matcher (SURVEY.md 8-f N1 — an OPTION, not the reference) -----------------------------
// The reference matches scan to scan ONCE per call with PCL ICP (cloud_alignment.cpp:37-223) and every particle
// samples round T(pose) * T_icp (particle_filter.cpp:146-153,181-188).  With scan matching on, each particle
// refines that pose against ITS OWN map before sampling, gmapping-style: hill climbing on the likelihood field
// (GridMapper::likelihoodFieldModel, grid_mapper.cpp:69-133 — the reference's own scoring function).  From the
// current pose evaluate the six neighbours +x, -x, +y, -y, +theta, -theta (world frame); move to the best of them if it
// is better by a factor > 1 + 1e-9 (the likelihood only sees cells, so neighbouring poses often carry the same
// factors on different beams: a bare > would follow rounding noise); otherwise halve both steps; stop after
// `iters` halvings (or max_moves rounds).
// Workgroup = particle, 6 waves: wave m scores neighbour m (lanes over the beams, lookups on the LDS slice of the
// bitmap), thread 0 applies the rule.  Same rule, same order of comparisons as oracle/rbpf_oracle.cpp::scan_match.
struct ScanMatchC { double lstep, astep; int iters, max_moves; };
constexpr int kMatchThreads = 6 * kWave;

// The 7 x 7 look of the query mode (or a read of the stored field) and nothing else: the code (>= 0), -1 = a windowed lookup
// outside the refreshed window, kNeedSearch = the query mode's answer needs the row walks (nearest_code_query_body).  The
// proposal kernel defers those to a phase of their own — ONE inlined copy of the search per phase, run by all threads over the
// marked entries — instead of calling an out-of-line copy from inside its lookup loops (round 3: seven call sites, 224 B of
// scratch per lane for the saves and restores round them).
constexpr int kNeedSearch = -2;
__device__ __forceinline__ int lookup_code_fast(const GridC& g, const DistSrc& d, int radius, int ci, int cj) {
  if (d.mode == 2) {
    if (d.nW > 0 && d.lut7) {
      const int C0 = d.W0 * 64, C1 = (d.W0 + d.nW) * 64 - 1;
      const int p0 = cj - C0 - 3, wi = p0 >> 5;
      if (ci - 3 >= d.R0 && ci + 3 <= d.R1 && p0 >= 0 && wi + 1 < 2 * d.nW && cj <= C1) {
        int clear = radius + 1;
        if (d.R0 > 0) clear = min(clear, ci - d.R0 + 1);
        if (d.R1 < g.xsize - 1) clear = min(clear, d.R1 - ci + 1);
        if (C0 > 0) clear = min(clear, cj - C0 + 1);
        if (C1 < g.ysize - 1) clear = min(clear, C1 - cj + 1);
        const unsigned int* t32 = reinterpret_cast<const unsigned int*>(d.tbm) + wi;
        const int sh = p0 & 31, stride = 2 * d.nW;
        int bw = 0x7fffffff;
#pragma unroll
        for (int dr = -3; dr <= 3; ++dr) {
          const unsigned int* rp = t32 + (ci + dr - d.R0) * stride;
          const unsigned int pat = __builtin_amdgcn_alignbit(rp[1], rp[0], sh) & 0x7Fu;
          bw = min(bw, dr * dr + (int)d.lut7[pat]);
        }
        if (bw <= 9 && bw <= clear * clear) return bw;
      }
    }
    return kNeedSearch;
  }
  if (d.mode == 1 && (ci < d.win.x || ci > d.win.y || cj < d.win.z || cj > d.win.w)) return -1;
  return d.code[(size_t)ci * g.xsize + cj];
}


// ---- raycast ---------------------------------------------------------------------------------------
// n-th free cell of the ray robot(x0,y0) -> endpoint(x1,y1), grid_mapper.cpp:549-807, in closed form:
// Bresenham's error recurrence D > 0 <=> c_t < (2*dmin*t - dmaj)/(2*dmaj) gives the minor-axis offset
// after t major steps  c_t = max(0, ceil((2*dmin*t - dmaj) / (2*dmaj)))  (checked against the
// reference's loops for every octant in tests).  Reversed octants start from the endpoint side.
struct Ray {
  int kind, count;   // 0 vertical, 1 horizontal, 2 low, 3 high, 4 diagonal
  int x0, y0, xa, ya, dmaj, dmin, sgn, sx, sy;
};
__device__ __forceinline__ Ray make_ray(int x0, int y0, int x1, int y1) {
  Ray r;
  r.x0 = x0; r.y0 = y0; r.xa = x0; r.ya = y0; r.dmaj = 0; r.dmin = 0; r.sgn = 1; r.sx = 1; r.sy = 1;
  const int dx = x1 - x0, dy = y1 - y0;
  const int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
  if (dx == 0) { r.kind = 0; r.count = ady; r.sy = dy < 0 ? -1 : 1; }
  else if (dy == 0) { r.kind = 1; r.count = adx; r.sx = dx < 0 ? -1 : 1; }
  else if (ady < adx) {
    r.kind = 2; r.count = adx;
    int xb, yb;
    if (x0 > x1) { r.xa = x1; r.ya = y1; xb = x0; yb = y0; } else { xb = x1; yb = y1; }
    r.dmaj = xb - r.xa;
    const int d = yb - r.ya;
    r.sgn = d < 0 ? -1 : 1;
    r.dmin = d < 0 ? -d : d;
  } else if (ady > adx) {
    r.kind = 3; r.count = ady;
    int xb, yb;
    if (y0 > y1) { r.xa = x1; r.ya = y1; xb = x0; yb = y0; } else { xb = x1; yb = y1; }
    r.dmaj = yb - r.ya;
    const int d = xb - r.xa;
    r.sgn = d < 0 ? -1 : 1;
    r.dmin = d < 0 ? -d : d;
  } else { r.kind = 4; r.count = adx; r.sx = dx < 0 ? -1 : 1; r.sy = dy < 0 ? -1 : 1; }
  return r;
}
__device__ __forceinline__ void ray_cell(const Ray& r, int n, int& cx, int& cy) {
  switch (r.kind) {
    case 0: cx = r.x0; cy = r.y0 + r.sy * n; break;
    case 1: cx = r.x0 + r.sx * n; cy = r.y0; break;
    case 4: cx = r.x0 + r.sx * n; cy = r.y0 + r.sy * n; break;
    default: {
      if (n == 0) { cx = r.x0; cy = r.y0; break; }
      const int a = 2 * r.dmin * n - r.dmaj;
      const int ct = a > 0 ? floor_div_small(a + 2 * r.dmaj - 1, 2 * r.dmaj) : 0;  // operands < 2^24
      if (r.kind == 2) { cx = r.xa + n; cy = r.ya + r.sgn * ct; }
      else { cx = r.xa + r.sgn * ct; cy = r.ya + n; }
    }
  }
}

// One wave per particle.  Beams are applied IN ORDER (the per-cell floating-point add order is the
// reference's); the cells of one ray are distinct, so the lanes of the wave update them in parallel
// without atomics.  Endpoints are staged in LDS first.
// The occupancy bits (one u32 per tile row, copy-on-write with the tile) / per-tile-row counts / occupied count of
// the particle are kept up to date here: a log-odds add that crosses the occupied cut-off toggles the cell's bit
// (rare: a few hundred cells per scan), so no pass over the whole map is needed to find the nearest-obstacle
// query's rows.
__device__ __forceinline__ bool add_log_odds(const TilePool& P, unsigned int id, double d, double cut, int cx, int cy,
                                             int* __restrict__ trow, int* __restrict__ nocc) {
  double* cell = P.lo + (size_t)id * kTileCells + in_tile(cx, cy);
  const double old = *cell;
  const double nw = old + d;
  *cell = nw;
  const bool was = old >= cut, now = nw >= cut;
  if (was != now) {
    atomicXor(&P.bm[(size_t)id * kTS + (cx & (kTS - 1))], 1u << (cy & (kTS - 1)));
    const int delta = now ? 1 : -1;
    atomicAdd(&trow[cx >> kTSh], delta);
    atomicAdd(nocc, delta);
  }
  return was != now;
}

// Ordered log of the occupied-set changes of one scan, per particle (reference distance-field mode only): entry =
// cell index, bit 31 set = the cell LEFT the set.  Same order as the reference's occ_cells_ insert / erase calls
// (grid_mapper.cpp:153-177 -> updateCellState/updateCellHash :438-546): beam by beam, the ray's free cells in
// free_index order, then the end point.  ev == NULL: no log.
struct OccLog { int* ev; int* count; int cap; };


// Is map cell (cx, cy) one of the FREE cells of ray r (i.e. some n in [0, count) has ray_cell(r, n) == it)?
__device__ __forceinline__ bool on_ray(const Ray& r, int cx, int cy) {
  switch (r.kind) {
    case 0: { const int n = (cy - r.y0) * r.sy; return cx == r.x0 && n >= 0 && n < r.count; }
    case 1: { const int n = (cx - r.x0) * r.sx; return cy == r.y0 && n >= 0 && n < r.count; }
    case 4: { const int n = (cx - r.x0) * r.sx; return n >= 0 && n < r.count && cy == r.y0 + r.sy * n; }
    default: {
      if (cx == r.x0 && cy == r.y0) return r.count > 0;
      const int n = (r.kind == 2) ? cx - r.xa : cy - r.ya;      // steps along the major axis
      if (n < 1 || n > r.dmaj - 1) return false;
      const int t = ((r.kind == 2) ? cy - r.ya : cx - r.xa) * r.sgn;  // offset along the minor axis
      // ray_cell gives offset c = max(0, ceil(a / (2*dmaj))) with a = 2*dmin*n - dmaj; test t == c without dividing
      const int a = 2 * r.dmin * n - r.dmaj, d2 = 2 * r.dmaj;
      return (a <= 0) ? (t == 0) : (t >= 1 && d2 * (t - 1) < a && a <= d2 * t);
    }
  }
}

// Tile version of the raycast (the default): no per-beam barrier.
//  F. every distinct END-POINT cell (<= Bv of them; the only cells that see both kinds of update in one scan,
//     and there the floating-point add order matters) is flagged in an LDS tile covering the scan's bounding
//     box (<= (2*range_max/res + 3)^2 cells) and gets a slot: a short list of (beam, kind) events;
//  1. every (beam, step) pair looks at its cell in the tile: a plain cell bumps its 15-bit counter (order-free
//     LDS atomic), a flagged cell records the event "beam b, free" in the cell's slot; every beam also records
//     "beam b, occupied" in its own end point's slot;
//  2. one LANE per end-point cell replays its slot in beam order ("+= l_free" / "+= l_occ": exactly the
//     reference's sequence of adds for that cell).  A slot that overflowed (kEvCap events; e.g. the robot's
//     own cell) is replayed by a whole wave instead, which tests the cell against every beam;
//  3. every other touched cell gets its count of "+= l_free" (same addend each time, so the order among
//     them is immaterial) — bit-identical to the beam-ordered loop, checked against it and the oracle.
// LDS (ints): ex ey own rk rxy rdd ecnt [Bv each] | ev u16[Bv][kEvCap] | tile u32[(cap+1)/2] (two 16-bit
// halves per word: bit 15 = end-point flag, low 15 bits = free-add count, or the slot index when flagged).
constexpr int kMapTilesMax = 64;  // map tiles a scan's bounding box can span: (ceil(175 / 32) + 1)^2 = 49 for tile_cap 30000
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// The same reductions without LDS round trips: an inclusive scan inside each row of 16 lanes by DPP shifts, then the row
// totals broadcast down the rows (row_bcast:15 / :31); lane 63 holds the result.  (__shfl_xor is ds_bpermute: six
// dependent LDS-latency steps per reduction.)
template <class Op> __device__ __forceinline__ int wave_reduce_dpp(int v, int ident, Op op) {
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));  // row_shr:8
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1, 3
  v = op(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_dpp(int v) { return wave_reduce_dpp(v, 0x7FFFFFFF, [](int a, int b) { return a < b ? a : b; }); }
__device__ __forceinline__ int wave_max_dpp(int v) { return wave_reduce_dpp(v, (int)0x80000000, [](int a, int b) { return a > b ? a : b; }); }
__device__ __forceinline__ int wave_sum_dpp(int v) { return wave_reduce_dpp(v, 0, [](int a, int b) { return a + b; }); }
constexpr int kBoxSideMax = 176;  // rows a scan's bounding box can have (tile_cap <= 30000 -> side <= 173)
// ---- dense view of the occupancy bits -----------------------------------------------------------------------
// The packed form of a ray straight from its two ends, selects only (what pack_ray(make_ray(..)) returns; the Ray struct's
// case analysis turns into a private array the compiler indexes at run time).  Along the major axis the ray starts at
// its LOW end (xa, ya) — the robot's cell or, for a reversed ray, the end point — takes dmaj steps and moves c_t =
// max(0, ceil((2 dmin t - dmaj) / (2 dmaj))) cells sideways (negated if neg); its free cells are the robot's cell and
// the cells strictly between the ends.
// n times  x = fl(x + d)  — the updates one cell takes from n beams (grid_mapper.cpp:438-477 adds the same log-odds once per beam) —
// bit for bit WITHOUT the chain of n dependent adds (13 ns each for one lane: the robot's own cell takes one per beam).  While x
// stays in one binade it is m * u (u = ulp(x), m a 53-bit integer) and d = kd * ud with ud = u / 2^sh: x + d = (m + q) u + rem ud
// (q = kd >> sh, rem = the bits shifted out), which rounds to (m + q) u or (m + q + 1) u by rem against half a u — the SAME integer
// step s every time, so j steps are m + j s (exact in 64-bit integers) as long as m + j s < 2^53.  What does not fit the pattern is
// done with a plain add: a step that leaves the binade (the sum is then rounded to the coarser grid), a tie (rem == u / 2: round to
// even alternates), opposite signs, x within a factor 4 of d, zeros, subnormals, infinities and NaNs.  (chain_exact is the same idea
// for a sum of different addends.)
__device__ __forceinline__ double add_repeated(double x, const double d, int n) {
  constexpr unsigned long long kMant = (1ull << 52) - 1ull;
  const unsigned long long bd = (unsigned long long)__double_as_longlong(d);
  const int ed = (int)((bd >> 52) & 0x7FFull);
  const unsigned long long kd = (bd & kMant) | (1ull << 52);
  while (n > 0) {
    const unsigned long long bx = (unsigned long long)__double_as_longlong(x);
    const int ex = (int)((bx >> 52) & 0x7FFull), sh = ex - ed;
    if (n < 4 || ((bx ^ bd) >> 63) != 0ull || sh < 2 || ex == 0x7FF || ed == 0 || ed == 0x7FF) { x += d; --n; continue; }
    if (sh > 54) return x;  // |d| < ulp(x) / 4: no add changes x
    const unsigned long long rem = kd & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
    if (rem == half) { x += d; --n; continue; }
    const unsigned long long s = (kd >> sh) + (rem > half ? 1ull : 0ull);
    if (s == 0ull) return x;  // d is less than half an ulp of x: no add changes it
    const unsigned long long m = (bx & kMant) | (1ull << 52);
    const unsigned long long room = (1ull << 53) - 1ull - m;  // the steps that stay in the binade: m + j s <= 2^53 - 1
    unsigned long long j = (unsigned long long)n;
    if (__umul64hi(j, s) != 0ull || j * s > room) {
      j = (unsigned long long)((double)room / (double)s);     // both exact in fp64 and the division is correctly rounded: floor or floor + 1
      if (j * s > room) --j;
    }
    const unsigned long long mj = m + j * s;
    x = __longlong_as_double((long long)((bx & (1ull << 63)) | ((unsigned long long)ex << 52) | (mj & kMant)));
    n -= (int)j;
    if (n > 0) { x += d; --n; }  // the step across the binade's end
  }
  return x;
}
struct RayP { int xa, ya, dmaj, dmin; bool ymajor, neg; };
__device__ __forceinline__ RayP ray_packed(int x0, int y0, int x1, int y1) {
  const int dx = x1 - x0, dy = y1 - y0, adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
  RayP r;
  r.ymajor = ady > adx;
  const bool rev = r.ymajor ? (y0 > y1) : (x0 > x1);
  r.xa = rev ? x1 : x0; r.ya = rev ? y1 : y0;
  const int d = r.ymajor ? (rev ? x0 : x1) - r.xa : (rev ? y0 : y1) - r.ya;
  r.neg = d < 0;
  r.dmaj = r.ymajor ? ady : adx; r.dmin = r.ymajor ? adx : ady;
  return r;
}
// Is (cx, cy) a free cell of the ray (x0, y0) -> (x1, y1)?  Same set as on_ray(make_ray(..)).
__device__ __forceinline__ bool on_ray_packed(int x0, int y0, int x1, int y1, int cx, int cy) {
  const RayP r = ray_packed(x0, y0, x1, y1);
  if (r.dmaj == 0) return false;  // the beam ends in the robot's cell: no free cell
  if (cx == x0 && cy == y0) return true;
  const int n = r.ymajor ? cy - r.ya : cx - r.xa;  // steps along the major axis
  const int tm = r.ymajor ? cx - r.xa : cy - r.ya, t = r.neg ? -tm : tm;  // offset along the minor axis, in the ray's sense
  const int a = 2 * r.dmin * n - r.dmaj, d2 = 2 * r.dmaj;
  const bool side = (a <= 0) ? (t == 0) : (t >= 1 && d2 * (t - 1) < a && a <= d2 * t);
  return n >= 1 && n <= r.dmaj - 1 && side;
}

// ---- normalise / Neff / low-variance selection (sequential order = the reference's) ---------------
struct NormOut { double sum_w, sq_sum; int neff, resampled; };
// One workgroup.  The three reductions that decide integers (sum, sum of squares -> Neff, the comb's
// running sum c) are done by ONE lane in index order — the reference's association — over an LDS copy of
// the weights (the only serial part: 3N dependent fp64 adds).  Everything else is parallel: the
// divisions, and the selection itself — with the sequential prefix c[] in hand, slot m's parent is the
// first i with U_m <= c[i] (the reference's while-loop, particle_filter.cpp:485-493, advances to exactly
// that i because U_m and c[] are both non-decreasing), found by binary search, clamped to N-1.
// Any N: the weights pass through LDS in chunks of kNormChunk (parallel loads / divisions, the one lane carries its
// running sums from chunk to chunk); the prefix c[] lives in LDS when one chunk holds it, else in a global scratch.
constexpr int kNormChunk = 2048;
constexpr int kScanSlots = 4;  // per-scan host-visible results (error flags, normalisation result, staged beams): a ring
// Left-to-right sum (of squares) of an LDS array by ONE thread, continuing from `acc` — the reference's order
// (particle_filter.cpp:446-450, 458-461), which Neff and the resampling decision depend on.  The chain of adds is
// inherent; the loads are not part of it: the next eight values are fetched while the current eight are added.
template <bool SQ, int BLK = 32>
__device__ __forceinline__ double seq_sum(double acc, const double* w, int N) {
  // 32 values per trip: sixteen 16-byte LDS reads issued together, then the 32 dependent adds and nothing else — a lone wave
  // issues an instruction every four to five cycles, so every instruction that is not an add stretches the chain (the
  // first version's register shuffling made it 13 ns per add)
  const double2* w2 = reinterpret_cast<const double2*>(w);  // (w is 16-byte aligned LDS)
  int i = 0;
  // (BLK values per trip: 32 in the kernel of its own; 16 where the body rides in rbpf_raycast_box, whose 64-register budget made
  //  a block of 32 spill three values per trip INTO the chain of adds — scratch loads with a full wait each)
  for (; i + BLK <= N; i += BLK) {
    double2 a[BLK / 2];
#pragma unroll
    for (int q = 0; q < BLK / 2; ++q) a[q] = w2[(i >> 1) + q];
#pragma unroll
    for (int q = 0; q < BLK / 2; ++q) { acc += SQ ? a[q].x * a[q].x : a[q].x; acc += SQ ? a[q].y * a[q].y : a[q].y; }
  }
  for (; i < N; ++i) acc += SQ ? w[i] * w[i] : w[i];
  return acc;
}
// ---- the reference's left-to-right sums, bit for bit, WITHOUT the chain of dependent adds (round 3) ---------------------------
// s_{j+1} = fl(s_j + a_j) looks inherently serial (10 ns per dependent fp64 add on one wave: 2-3 ms for the 100 000 weights of
// BASELINE configs[4], on every rank of the sharded filter).  It is not, binade by binade: while the running sum stays in one
// binade [2^e, 2^(e+1)) it is a multiple of u = 2^(e-52), so fl(s + a) = s + RN_u(a) — the addend rounded to the grid, to nearest,
// and that is an INTEGER increment q_j = floor(a_j / u) + (frac > 1/2), exact in fp64 arithmetic (scaling by a power of two, floor
// and the difference are all exact).  Integer sums are associative: the whole chunk is one parallel prefix sum of the q_j.  Only
// two things break the pattern, and both are detected exactly and in parallel: a TIE (frac == 1/2: round-half-even needs the
// parity of the sum so far) and a CROSSING (the integer sum reaches 2^53: the result leaves the binade and rounds on a coarser
// grid).  The first such element m is found by a block-wide min; everything before it is applied in bulk, element m itself is
// ONE plain fp64 add (which does the right thing by definition), and the scan resumes behind it on the new grid.  Non-negative
// finite addends only (weights and their squares); anything else, or a sum below 2^-900, takes plain sequential adds.
// A chunk of 2048 costs a block scan or two instead of 2048 dependent adds (measured: tools/normalize_time.py).
// PREFIX: also writes the running sum after every element (the comb's c[], particle_filter.cpp:478,492).
template <bool SQ, bool PREFIX, int IPT>
__device__ __forceinline__ double chain_exact(double s, const double* w, double* cl, int n, int head = 0) {
  __shared__ unsigned long long sh_wtot[16];
  __shared__ int sh_first[16];
  __shared__ double sh_s;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave, nw = blockDim.x / kWave;
  const int j0 = tid * IPT;  // this thread's elements: [j0, j0 + IPT), in index order across the block
  const double inf = __builtin_huge_val();
  double a[IPT];
  bool bad = false;
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const int j = j0 + q;
    const double v = j < n ? w[j] : 0.0;
    a[q] = SQ ? v * v : v;
    bad |= !(a[q] >= 0.0 && a[q] < inf);
  }
  if (__syncthreads_or(bad ? 1 : 0)) {  // (never for weights: negative / NaN / Inf addends take the plain chain)
    if (tid == 0) {
      double c = s;
      for (int j = 0; j < n; ++j) { c += SQ ? w[j] * w[j] : w[j]; if (PREFIX) cl[j] = c; }
      sh_s = c;
    }
    __syncthreads();
    const double r = sh_s;
    __syncthreads();
    return r;
  }
  int i0 = 0;  // elements below i0 are in the sum (everything here is workgroup-uniform)
  if (head > 0) {
    // the first elements of a vector by the plain chain on one lane (register-blocked: 10 ns an add) — the sum doubles after 1, 2,
    // 4, ... addends of similar size, i.e. a binade crossing (one trip of the loop below: a block scan and three barriers) every
    // few elements until it has grown
    const int hn = head < n ? head : n;
    if (tid == 0) {
      double c = s;
      if (PREFIX) { for (int j = 0; j < hn; ++j) { c += SQ ? w[j] * w[j] : w[j]; cl[j] = c; } }
      else c = seq_sum<SQ, 16>(c, w, hn);
      sh_s = c;
    }
    __syncthreads();
    s = sh_s;
    i0 = hn;
    __syncthreads();
  }
  while (i0 < n) {
    if (!(s >= 0x1p-900)) {  // no binade to work in yet (the sum is still zero or tiny, or NaN): one plain add
      const double v = w[i0];
      s = s + (SQ ? v * v : v);
      if (PREFIX && tid == 0) cl[i0] = s;
      ++i0;
      continue;
    }
    const int e = (int)((__double_as_longlong(s) >> 52) & 0x7FF) - 1023;             // s in [2^e, 2^(e+1))
    const double inv_u = __longlong_as_double((long long)(1023 + 52 - e) << 52);     // 1 / ulp of that binade
    const double u = __longlong_as_double((long long)(1023 - 52 + e) << 52);
    const unsigned long long B = (unsigned long long)(s * inv_u);                    // s on the grid: in [2^52, 2^53)
    unsigned long long pre[IPT], run = 0ull;
    unsigned int tie = 0u;
#pragma unroll
    for (int q = 0; q < IPT; ++q) {
      const int j = j0 + q;
      unsigned long long inc = 0ull;
      if (j >= i0 && j < n) {
        const double x = a[q] * inv_u;            // exact (a power of two)
        if (x >= 0x1p53) inc = 1ull << 53;        // by itself beyond the binade: a crossing at this element
        else {
          const double fl = floor(x), fr = x - fl;  // both exact
          inc = (unsigned long long)fl + (fr > 0.5 ? 1ull : 0ull);
          if (fr == 0.5) tie |= 1u << q;
        }
      }
      run += inc;
      pre[q] = run;
    }
    // block-wide exclusive offset of `run` (wave scan by shuffles, wave totals through LDS)
    unsigned long long incl = run;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const unsigned long long o = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += o;
    }
    if (lane == kWave - 1) sh_wtot[wid] = incl;
    __syncthreads();
    unsigned long long offset = incl - run, all = 0ull;
    for (int q = 0; q < nw; ++q) { const unsigned long long t = sh_wtot[q]; if (q < wid) offset += t; all += t; }
    // the first element that is a tie or takes the sum out of the binade
    int first = 0x7FFFFFFF;
#pragma unroll
    for (int q = IPT - 1; q >= 0; --q) {
      const int j = j0 + q;
      if (j >= i0 && j < n && (((tie >> q) & 1u) || B + offset + pre[q] >= (1ull << 53))) first = j;
    }
    first = wave_min_i(first);
    if (lane == 0) sh_first[wid] = first;
    __syncthreads();
    int m = 0x7FFFFFFF;
    for (int q = 0; q < nw; ++q) m = min(m, sh_first[q]);
    // everything before m: in bulk (integers below 2^53 convert exactly, times a power of two)
    if (PREFIX) {
#pragma unroll
      for (int q = 0; q < IPT; ++q) {
        const int j = j0 + q;
        if (j >= i0 && j < n && j < m) cl[j] = (double)(B + offset + pre[q]) * u;
      }
    }
    if (m == 0x7FFFFFFF) { s = (double)(B + all) * u; i0 = n; break; }
    if (m >= j0 && m < j0 + IPT) {  // the thread that owns element m: the sum just before it, then ONE plain add
      const int q = m - j0;
      const double before = (double)(B + offset + (q > 0 ? pre[q - 1] : 0ull)) * u;
      const double after = before + a[q];
      if (PREFIX) cl[m] = after;
      sh_s = after;
    }
    __syncthreads();
    s = sh_s;
    i0 = m + 1;
    __syncthreads();  // (sh_s / sh_wtot / sh_first are rewritten in the next trip)
  }
  return s;
}

// weight_out: where the normalised weights go ([N]; may alias weight).  cs: [N] scratch for the prefix (N > kNormChunk).
// The body, for one workgroup of any size; w, cl: two LDS arrays of kNormChunk doubles (16-byte aligned).
// gate (optional, device memory): 1 if this scan resamples, else 0 — what a scan enqueued BEHIND this one, before the host has
// seen the decision, checks before it touches anything (gate_prev; see tbnav_rbpf_slam_batch).
// seq (optional, mapped host memory): set to seq_val once `out` is written and visible to the host — what the host polls
// instead of waiting for the whole launch.
struct NormArgs { int N; const double* zp; const double* weight; double* weight_out; double* cs; int* parent; NormOut* out;
                  int* gate; const int* gate_prev; unsigned int* seq; unsigned int seq_val; int* children; };
template <int NTHR, bool PAR>
__device__ __forceinline__ void normalize_body(int N, const double* __restrict__ zp, const double* weight, double* weight_out,
                                               double* __restrict__ cs, int* __restrict__ parent, NormOut* __restrict__ out,
                                               double* w, double* cl, int* __restrict__ gate = nullptr,
                                               unsigned int* seq = nullptr, unsigned int seq_val = 0, int* __restrict__ children = nullptr) {
  const double z = *zp;  // the one standard normal of lowVarianceResampling (particle_filter.cpp:474)
  __shared__ int s_res;
  const int tid = threadIdx.x, nthr = NTHR;
  constexpr int kIpt = kNormChunk / NTHR;  // elements of a chunk per thread in the exact parallel chains (chain_exact)
  static_assert(kNormChunk % NTHR == 0, "the chunk splits evenly over the workgroup");
  // One chunk (N <= 2048: BASELINE configs[2], the reference's launch file): the plain chain on one lane — 10 ns an add, 20 us at
  // N = 1000, hidden beside the map update; the parallel form's ~2 us per binade crossing (log2 N of them) would cost more.
  // More than one chunk (the sharded filter's global vector, 100 000 for configs[4]): chain_exact.
  // PAR = false (the copy that rides in rbpf_raycast_box's launch as workgroup 0): always the plain chain — it runs beside that
  // launch's other workgroups anyway, and the parallel form inlined there cost the map update 3 % (registers, code size).
  const bool one_chunk = N <= kNormChunk;
  const bool plain = one_chunk || !PAR;
  constexpr int kSeqBlk = NTHR == 256 ? 32 : 16;  // (register block of the plain chain: 16 under rbpf_raycast_box's 64-register budget)
  constexpr int kHead = 128;
  __shared__ double s_acc;
  double run = 0.0;  // (workgroup-uniform)
  for (int base = 0; base < N; base += kNormChunk) {
    const int n = min(kNormChunk, N - base);
    __syncthreads();
    for (int i = tid; i < n; i += nthr) w[i] = weight[base + i];
    __syncthreads();
    if (plain) { if (tid == 0) s_acc = seq_sum<false, kSeqBlk>(run, w, n); __syncthreads(); run = s_acc; }
    else if constexpr (PAR) run = chain_exact<false, false, kIpt>(run, w, nullptr, n, base == 0 ? kHead : 0);   // sum += weight(i), particle_filter.cpp:446-450
  }
  __syncthreads();
  const double sum = run;
  run = 0.0;
  for (int base = 0; base < N; base += kNormChunk) {
    const int n = min(kNormChunk, N - base);
    __syncthreads();
    for (int i = tid; i < n; i += nthr) { const double v = weight[base + i] / sum; w[i] = v; weight_out[base + i] = v; }
    __syncthreads();
    if (plain) { if (tid == 0) s_acc = seq_sum<true, kSeqBlk>(run, w, n); __syncthreads(); run = s_acc; }
    else if constexpr (PAR) run = chain_exact<true, false, kIpt>(run, w, nullptr, n, base == 0 ? kHead : 0);    // normal_sqrd_sum_ += w * w, :458-461
  }
  __syncthreads();
  if (tid == 0) {
    const double sq = run;
    const int neff = (int)(1.0 / sq);
    const int res = (neff < (N / 2)) ? 1 : 0;
    out->sum_w = sum; out->sq_sum = sq; out->neff = neff; out->resampled = res;
    if (gate) *gate = res;
    if (seq) {
      __threadfence_system();  // the four stores above reach the host before the flag does
      __hip_atomic_store(seq, seq_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    s_res = res;
  }
  __syncthreads();
  if (!s_res) { for (int m = tid; m < N; m += nthr) parent[m] = m; return; }
  run = 0.0;
  for (int base = 0; base < N; base += kNormChunk) {
    const int n = min(kNormChunk, N - base);
    __syncthreads();
    if (!one_chunk) for (int i = tid; i < n; i += nthr) w[i] = weight_out[base + i];  // (one chunk: w[] still holds them)
    __syncthreads();
    // c = weight(0); c += weight(i), particle_filter.cpp:478,492 — every c[i] kept
    if (plain) {
      if (tid == 0) {
        double c = run;
        const double2* w2 = reinterpret_cast<const double2*>(w);
        double2* c2 = reinterpret_cast<double2*>(cl);
        int i = 0;
        for (; i + 16 <= n; i += 16) {
          double2 a[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) a[q] = w2[(i >> 1) + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) { double2 o; c += a[q].x; o.x = c; c += a[q].y; o.y = c; c2[(i >> 1) + q] = o; }
        }
        for (; i < n; ++i) { c += w[i]; cl[i] = c; }
        s_acc = c;
      }
      __syncthreads();
      run = s_acc;
    } else if constexpr (PAR) run = chain_exact<false, true, kIpt>(run, w, cl, n, base == 0 ? kHead : 0);
    __syncthreads();
    if (!one_chunk) for (int i = tid; i < n; i += nthr) cs[base + i] = cl[i];
  }
  __syncthreads();
  const double* csr = one_chunk ? cl : cs;
  const double r = z / (double)N;
  for (int m = tid; m < N; m += nthr) {
    const double U = r + (double)(m * (1.0 / (N - 1)));
    int lo = 0, hi = N - 1;  // first index with U <= cs[i]; N-1 if none (the reference clamps there)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (U > csr[mid]) lo = mid + 1; else hi = mid;
    }
    parent[m] = lo;
  }
  if (!children) return;
  // children[i] = how many slots chose parent i (optional): what the table / reference-count kernel needs per OLD particle.
  // parent[] is non-decreasing, so a parent's children are one run: its first slot finds the run's end by bisection.
  __threadfence_block();
  __syncthreads();
  const int* par = parent;
  if (one_chunk) {  // (w[] is free by now: an LDS copy of parent[] for the bisections)
    int* pl = reinterpret_cast<int*>(w);
    for (int m = tid; m < N; m += nthr) pl[m] = parent[m];
    par = pl;
  }
  for (int m = tid; m < N; m += nthr) children[m] = 0;
  __threadfence_block();
  __syncthreads();
  for (int m = tid; m < N; m += nthr) {
    const int me = par[m];
    if (m > 0 && par[m - 1] == me) continue;
    int lo = m, hi = N;  // first index > m whose parent is not `me`
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (par[mid] == me) lo = mid; else hi = mid;
    }
    children[me] = hi - m;
  }
}

// ---- the default map update: box counters ------------------------------------------------------------------------
// Same contract as rbpf_raycast_tile (bit-identical maps) with fewer, cheaper phases:
//  F. the beams' end-point cells — the only cells that see both l_free and l_occ in one scan, i.e. where the floating-
//     point add order matters — are flagged in an LDS array with one 32-bit word per cell of the scan's bounding box
//     (bit 31; bits 16-30 = the cell's slot in the list of distinct end-point cells);
//  1. every ray segment walks its cells with ONE returning LDS add per cell (low 16 bits = free adds) and never waits
//     for it: the value that comes back is looked at one step later, and only if it carries the flag does the lane
//     record "beam b, free" in that cell's slot (a few percent of the steps); every beam records "beam b, occupied" in
//     its own end point's slot;
//  2. one pass over the box, a PAIR of cells (16 bytes of a map tile's row) per lane and consecutive pairs in consecutive
//     lanes — whole cache lines per wave: a counted or flagged pair marks its map tile as written and requests its log-odds
//     from whichever tile the particle's table names now (shared, private or the zero tile hold the same values); the
//     written tiles are then made private to the particle (usually they already are) while the loads are in flight;
//  3. one lane per end-point cell replays its slot in beam order — bit (beam - own beam + 32) of a 64-bit mask per kind
//     orders the events without sorting; an overflowed slot: a whole wave tests the cell against every beam — and the
//     cells round the robot, tens to hundreds of DEPENDENT adds each because every ray starts there, get a lane of their
//     own in the last wave, which walks no ray (the robot's own cell, one add per beam, is started right after the end
//     points are known and worked off in pieces between the barriers); both hand their result over through LDS;
//  4. the pairs: a plain cell adds its count of l_free (same addend each time, so the order among the adds is
//     immaterial), an end-point or hot cell takes the value worked out for it; the pair goes back as one 16-byte store.
// The LDS array holds as many rows of the box as fit (tile_cap words: the host keeps a workgroup under half of the CU's
// 160 KB so that two are resident); a box with more rows (a long-range scan seen from a rotated pose) is worked through in
// bands of rows, every phase once per band with the rays clipped to the band.
// What bounds it (per-wave trace): instruction issue — ~28 k wave-instructions per particle through
// 16 waves on 4 SIMDs between 9 barriers; memory traffic is the distinct cells once each way.
// LDS: tile u32[tile_cap] (rows padded to an even number of columns: pair i = words 2i, 2i+1) |
// ev u16[Bv][kBoxEv] (slot o's first 8 bytes double as its replayed value) | hot-cell values f64[64] | exy i32[Bv] | ecnt u16[Bv]
#ifdef TBNAV_PHASE_PROF
static __device__ unsigned long long g_phase_w[16];
#endif
#if defined(TBNAV_PHASE_PROF) && !defined(TBNAV_TRACE_ONLY)
#define PHASE_STAMP_W(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase_w[i], now_ - t_prev_); t_prev_ = now_; } } while (0)
#else
#define PHASE_STAMP_W(i)
#endif
#ifdef TBNAV_PHASE_PROF
static __device__ unsigned long long g_trace[2][16][16];  // [which][wave][stamp] of TWO workgroups (blockIdx.x == 100: first round of residents; 900: second): 10 ns ticks
#define TRACE_W(i) do { if ((blockIdx.x == 100 || blockIdx.x == 900) && (threadIdx.x & 63) == 0) g_trace[blockIdx.x == 900][threadIdx.x >> 6][i] = wall_clock64(); } while (0)
static __device__ unsigned long long g_wg[4096][3];    // [workgroup] entry, exit (10 ns ticks), XCC_ID << 32 | HW_ID — of the LAST launch
#define WG_IN() do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_wg[blockIdx.x][0] = wall_clock64(); \
  g_wg[blockIdx.x][2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned int)__builtin_amdgcn_s_getreg(63492); } } while (0)
#define WG_OUT() do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_wg[blockIdx.x][1] = wall_clock64(); } while (0)
#else
#define WG_IN()
#define WG_OUT()
#define TRACE_W(i)
#endif
constexpr int kBoxEv = 8;     // events a slot holds before it is replayed exhaustively
constexpr int kHotSide = 7;   // the kHotSide x kHotSide cells round the robot are candidates for a lane of their own ...
constexpr int kHotMin = 16;   // ... when they collect at least this many free adds
constexpr int kVeryHot = 80;  // ... and from this many on they are worked out without the chain of adds (add_repeated)
// tile u32[cap] | ev u16[bv][kBoxEv] (a slot's first 8 bytes become its replayed value once its events are read) | hot values f64[64] | exy, ecnt i32[bv]
__host__ __device__ constexpr size_t box_lds_bytes(size_t cap, size_t bv) { return 4 * cap + 2 * kBoxEv * bv + 8 * 64 + 4 * bv + 2 * ((bv + 1) & ~(size_t)1); }
constexpr size_t kBoxStaticLds = 768;  // the kernel's __shared__ variables (tools/kernel_resources.py rbpf_raycast: 7xx B), rounded up
// (512 threads: three workgroups = 24 waves per CU when the LDS array is sized by what the boxes need, see launch_raycast —
//  6 waves per SIMD leave 80 registers a lane: the kernel needs 77 and spills nothing; 1024 threads: two workgroups = 32 waves, 64)



struct EdtJob { const int4* win; const int* skip; int p0; };

// distance (cells) from column j to the nearest set bit of a bitmap row (word(w) = its u64 word w), capped at `cap`
// (255 = none)
template <class Word>
__device__ __forceinline__ int row_nearest_f(Word word, int words, int j, int cap) {
  const int w = j >> 6, b = j & 63;
  int best = 1 << 20;
  const unsigned long long here = word(w);
  // at or left of j
  unsigned long long m = here & (b == 63 ? ~0ull : ((1ull << (b + 1)) - 1ull));
  int ww = w;
  while (true) {
    if (m) { best = j - (ww * 64 + 63 - __clzll((long long)m)); break; }
    if (--ww < 0 || (j - (ww * 64 + 63)) > cap) break;
    m = word(ww);
  }
  // right of j
  m = here & ~(b == 63 ? ~0ull : ((1ull << (b + 1)) - 1ull));
  ww = w;
  while (true) {
    if (m) { const int d = (ww * 64 + (__ffsll((long long)m) - 1)) - j; best = d < best ? d : best; break; }
    if (++ww >= words || (ww * 64 - j) > cap) break;
    m = word(ww);
  }
  return best <= cap ? best : 255;
}
__device__ __forceinline__ int row_nearest(const unsigned long long* row, int words, int j, int cap) {
  return row_nearest_f([row](int w) { return row[w]; }, words, j, cap);
}

__device__ __forceinline__ int floor_div(int num, int den) {  // den > 0
  int q = num / den;
  if ((num % den != 0) && (num < 0)) --q;
  return q;
}
// Exact floor(num/den) for |num| < 2^24 and 0 < den < 2^13 (the envelope's operands: |num| <= 255^2 +
// 2047^2, den <= 2*2047): both convert to float exactly, the float quotient is within 1 of the true
// one, and an integer remainder check fixes it — ~12 instructions instead of the ~40 of an int division.
// (the quotient comes from v_rcp_f32 — one instruction, 1 ulp — not from a float division, which without fast-math is a
//  twelve-instruction sequence: the estimate may then be off by two, hence two correction steps each way)
__device__ __forceinline__ int floor_div_small(int num, int den) {
  int q = (int)floorf((float)num * __builtin_amdgcn_rcpf((float)den));
  int r = num - q * den;
  if (r < 0) { --q; r += den; }
  if (r < 0) { --q; r += den; }
  if (r >= den) { ++q; r -= den; }
  if (r >= den) ++q;
  return q;
}


// Fast path of the distance transform.  Only map rows that hold at least one occupied cell can
// contribute a parabola to a column's lower envelope, and in a room-sized world that is ~100 of the
// 400 rows: the envelope stack is sized by SMAX compacted rows instead of xsize, which cuts LDS per
// wave from 150 KB to <= 40 KB (4 waves per CU instead of 1), the row pass only visits those rows, and
// the stack top is kept in registers.  grid (column tiles, N), 64 threads.  A particle with more than
// SMAX non-empty rows raises its tier and is left to the next kernel (SMAX doubled, finally the
// general kernel above).  LDS: rowlist u16[SMAX] | vz u32[SMAX][64] (row | (z+32768)<<16) | f u8[SMAX][64].
// packed envelope entry (maps up to 2047 rows): row v in bits 0-10, row distance f in bits 11-18, z+1 in bits 19-31
constexpr int kZMax = 8190;
constexpr int kEdtCompactMaxRows = 2047;  // packed entry: 11 bits of row index
__device__ __forceinline__ uint32_t pack(int v, int f, int z) { return (uint32_t)v | ((uint32_t)f << 11) | ((uint32_t)(z + 1) << 19); }
__device__ __forceinline__ void unpack(uint32_t e, int& v, int& f, int& z) { v = (int)(e & 0x7FFu); f = (int)((e >> 11) & 0xFFu); z = (int)(e >> 19) - 1; }
constexpr size_t edt_compact_lds(int smax) { return (size_t)smax * kWave * 4 + (size_t)smax * (8 + 6) + 16; }
constexpr int kEdtRowsA = 144;  // 38.9 KB -> 4 waves per CU
constexpr int kEdtRowsB = 288;  // 77.8 KB -> 2 waves per CU


// ---- resampling: slot m <- parent[m] (particle_filter.cpp:495 deep copies) -----------------------------------
// Maps: the new slot takes a COPY OF ITS PARENT'S TILE TABLE and every named tile gains a reference (pass 1); then the
// old generation's references — table entries and the shed notes of tiles left since the last resample — are
// dropped and tiles nobody names any more go back to the free ring (pass 2, a separate launch: no count may reach
// zero before every new reference is in).  16 KB of table per particle at 2000 x 2000 instead of a 32 MB map.
// One pass over the [N][TT] table entries does both halves of a resample's bookkeeping (children[i] = how many slots chose
// particle i; the new tables go to the alternate buffer, so the two halves do not see each other):
//  A. slot m's new table is its parent's old one;
//  B. old particle i held one reference on each tile its table named: its children hold children[i] now.  A tile nobody
//     else referenced (count 1 — nobody else can be touching it) gets the new count with a plain store, or goes back to the
//     pool when the particle died; a shared tile takes ONE atomic add of the difference.  While some holder has not been
//     through yet the count stays above zero (every holder still counts 1), so the add that lands on zero is the last
//     word on that tile.  Tiles a slot stopped using since the last resample (shed) are released likewise.
// Freed tiles go back with one atomic on the ring's tail per WORKGROUP and round (lane-private pushes queue on that word:
// ~90 atomics per microsecond on one address, and a resample that kills 900 of 1000 particles frees 13 000 tiles).  Instead of one atomic per child and tile plus one per old entry, in two launches.
__device__ __forceinline__ void resample_tables_body(int N, int TT, const int* __restrict__ parent, const int* __restrict__ children,
                                                     const unsigned int* __restrict__ tab_old, unsigned int* __restrict__ tab_new,
                                                     unsigned int* __restrict__ shed, const TilePool& P, int block, int nblocks) {
  const size_t n = (size_t)N * TT;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = blockDim.x / kWave;
  const unsigned long long below = (1ull << lane) - 1ull;
  __shared__ int s_wave_total[16];
  __shared__ unsigned long long s_base;
  for (size_t e0 = (size_t)block * blockDim.x; e0 < n; e0 += (size_t)nblocks * blockDim.x) {
    const size_t e = e0 + threadIdx.x;
    unsigned int freed[2] = {0u, 0u};
    if (e < n) {
      const int m = (int)(e / TT), t = (int)(e - (size_t)m * TT);
      tab_new[e] = tab_old[(size_t)parent[m] * TT + t];
      const unsigned int id = tab_old[e], sh = shed[e];
      const int c = children[m];
      if (id && c != 1) {
        if (P.ref[id] == 1) { P.ref[id] = c; if (c == 0) freed[0] = id; }
        else if (atomicAdd(&P.ref[id], c - 1) + (c - 1) == 0) freed[0] = id;
      }
      if (sh) { if (atomicSub(&P.ref[sh], 1) == 1) freed[1] = sh; shed[e] = 0u; }
    }
    // (the trip count is the same for the whole workgroup: barriers inside the loop are safe)
    const unsigned long long m0 = __ballot(freed[0] != 0u), m1 = __ballot(freed[1] != 0u);
    const int total = __popcll(m0) + __popcll(m1);
    if (lane == 0) s_wave_total[wid] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
      int sum = 0;
      for (int q = 0; q < nw; ++q) { const int v = s_wave_total[q]; s_wave_total[q] = sum; sum += v; }  // -> exclusive prefix
      s_base = sum ? atomicAdd(P.ctr + 1, (unsigned long long)sum) : 0ull;
    }
    __syncthreads();
    unsigned long long at = s_base + s_wave_total[wid] + __popcll(m0 & below) + __popcll(m1 & below);
    if (freed[0]) P.ring[at++ % P.cap] = freed[0];
    if (freed[1]) P.ring[at % P.cap] = freed[1];
    __syncthreads();  // (s_wave_total is rewritten by the next round)
  }
}
// Everything else a particle owns: pose / prev_pose / weight (weights are NOT reset, :495), its occupied counts (per tile
// row and total; the occupancy BITS live in the tiles and follow the tables), the state of its stored distance field and — only where that field is authoritative (injected or
// materialised, state 2; always in the stored-field modes) — the field itself.  grid (N, chunks).
struct GatherArgs {
  size_t G; int TW;
  const double* st_src; double* st_dst;
  const int* rc_src; int* rc_dst;
  const int* nocc_src; int* nocc_dst;
  const int* fs_src; int* fs_dst;
  const uint16_t* cd_src; uint16_t* cd_dst; int copy_all_codes;
};
__device__ __forceinline__ void gather_body(int N, const int* __restrict__ parent, const GatherArgs& a, int m, int chunk, int chunks) {
  const size_t G = a.G; const int TW = a.TW;
  const double* __restrict__ st_src = a.st_src; double* __restrict__ st_dst = a.st_dst;
  const int* __restrict__ rc_src = a.rc_src; int* __restrict__ rc_dst = a.rc_dst;
  const int* __restrict__ nocc_src = a.nocc_src; int* __restrict__ nocc_dst = a.nocc_dst;
  const int* __restrict__ fs_src = a.fs_src; int* __restrict__ fs_dst = a.fs_dst;
  const uint16_t* __restrict__ cd_src = a.cd_src; uint16_t* __restrict__ cd_dst = a.cd_dst; const int copy_all_codes = a.copy_all_codes;
  const int src = parent[m];
  const size_t t0 = (size_t)chunk * blockDim.x + threadIdx.x, stride = (size_t)chunks * blockDim.x;
  for (size_t t = t0; t < (size_t)TW; t += stride) rc_dst[(size_t)m * TW + t] = rc_src[(size_t)src * TW + t];
  const int fs = fs_src[src];
  if (cd_src && (copy_all_codes || fs == 2)) {
    const uint2* ca = reinterpret_cast<const uint2*>(cd_src + (size_t)src * G);
    uint2* cb = reinterpret_cast<uint2*>(cd_dst + (size_t)m * G);
    for (size_t t = t0; t < G / 4; t += stride) cb[t] = ca[t];
  }
  if (t0 == 0) {
    nocc_dst[m] = nocc_src[src];
    fs_dst[m] = fs;
    for (int q = 0; q < 3; ++q) {
      st_dst[(size_t)m * 3 + q] = st_src[(size_t)src * 3 + q];
      st_dst[(size_t)3 * N + (size_t)m * 3 + q] = st_src[(size_t)3 * N + (size_t)src * 3 + q];
    }
    st_dst[(size_t)6 * N + m] = st_src[(size_t)6 * N + src];
  }
}
// lowVarianceResampling's copies (particle_filter.cpp:495) in ONE launch: workgroups [0, table_blocks) do the tables and the
// reference counts, the next N * chunks gather slot m's state from its parent.
constexpr int kResampleThreads = 1024;



// ---- the same, many particles per launch (a cross-rank resample moves hundreds of particles per rank: one call per particle
//      is a host round trip each).  The buffer is the per-particle blobs of tbnav_rbpf_export_particle_dev back to back.
struct BlobHeader { uint64_t magic; uint32_t n_tiles, has_codes; int32_t nocc, fstate; uint32_t xsize, TT; };
constexpr uint64_t kBlobMagic = 0x54424e4156504631ull;  // "TBNAVPF1"
struct BlobLayout { size_t state, tidx, tiles, tile_bm, trow, codes, total; };
__host__ __device__ inline BlobLayout blob_layout_hd(int TW, size_t G, uint32_t n_tiles, bool has_codes) {
  auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
  BlobLayout L{};
  size_t o = sizeof(BlobHeader);
  L.state = o; o += sizeof(double) * 7;
  L.tidx = o; o = up8(o + sizeof(uint32_t) * n_tiles);
  L.tiles = o; o += sizeof(double) * kTileCells * n_tiles;
  L.tile_bm = o; o = up8(o + sizeof(unsigned int) * kTS * n_tiles);
  L.trow = o; o = up8(o + sizeof(int) * TW);
  L.codes = o; if (has_codes) o = up8(o + sizeof(uint16_t) * G);
  L.total = o;
  return L;
}
struct BatchItem { int slot; unsigned int n_tiles; int has_codes; int pad; unsigned long long off; };


// GridMapper::gridMap (grid_mapper.cpp:185-226) of the best particle: int8 {-1, 0, 100, (int8)(prob*100)},
// transposed.  prob is never evaluated here: the host found, with glibc, the log-odds at which the exported
// value changes (ExportCuts), so the device output is the reference's bit for bit.
struct ExportCuts {
  double occ_cut;     // smallest l exported as 100 (prob >= 0.90)
  double free_cut;    // largest l exported as 0    (prob <= 0.35)
  double half_lo, half_hi;  // [lo, hi]: prob == 0.5 exactly -> -1 (unknown)
  double step[64];    // step[m] = smallest l exported as >= 36 + m   (values 35..89 in between)
  int n_steps;
};

// ---- the kernels (defined in rbpf_<family>.hip) ----------------------------------------------------------------------
// rbpf_propose.hip
__global__ void rbpf_mix_lut(ScanC c, double* __restrict__ out);
__global__ void rbpf_sample_normals(size_t n, unsigned long long seed, unsigned long long scan, double* __restrict__ out,
                                    const double2* __restrict__ host_beams, double2* __restrict__ dev_beams, int n_copy,
                                    size_t out_stride = 0, size_t beam_stride = 0, size_t base = 0, size_t z_index = ~(size_t)0,
                                    size_t z_slot = 0);
__global__ __launch_bounds__(256) void rbpf_field_by_query(GridC g, int radius, int particle, TilePool P, MapT M,
                                                           const int* __restrict__ trow_occ, uint16_t* __restrict__ codes);
__global__ __launch_bounds__(kWave) void rbpf_likelihood_one(ScanC c, const double2* __restrict__ beams, const uint16_t* __restrict__ codes,
                                                            TilePool P, MapT M, const int* __restrict__ trow_occ,
                                                            const int* __restrict__ fstate, int radius, const int* __restrict__ n_occ,
                                                            double th, double x, double y, double* __restrict__ out, int* __restrict__ err,
                                                            const double* __restrict__ mixlut);
__global__ __launch_bounds__(kMatchThreads) void rbpf_scanmatch(ScanC c, ScanMatchC sm, const double2* __restrict__ beams,
                                                                const uint16_t* __restrict__ codes,
                                                                TilePool P, MapT M,
                                                                const int* __restrict__ trow_occ, const int* __restrict__ skip,
                                                                int skip_eq, int df_mode, int radius, int occ_half,
                                                                const int* __restrict__ n_occ, const int4* __restrict__ win,
                                                                const double* __restrict__ pose, double* __restrict__ center,
                                                                double* __restrict__ score, int* __restrict__ err,
                                                                const int* __restrict__ gate_prev, const double* __restrict__ mixlut);
template <int NT>
__global__ __launch_bounds__(NT, TBNAV_PROPOSE_WAVES) void rbpf_propose(ScanC c, const double2* __restrict__ beams,
                                                                const uint16_t* __restrict__ codes,
                                                                TilePool P, MapT M,
                                                                const int* __restrict__ trow_occ, const int* __restrict__ skip,
                                                                int skip_eq, int df_mode, int radius, int occ_half,
                                                                const int* __restrict__ n_occ, const int4* __restrict__ win,
                                                                const double* __restrict__ normals, const double* __restrict__ center,
                                                                double* __restrict__ pose, double* __restrict__ prev_pose,
                                                                double* __restrict__ weight, Trace tr, double* __restrict__ sens,
                                                                int* __restrict__ err, const int* __restrict__ gate_prev,
                                                                const double* __restrict__ mixlut);
// rbpf_raycast.hip
__global__ __launch_bounds__(kWave) void rbpf_raycast(ScanC c, TilePool P, MapT M, const double2* __restrict__ beams,
                                                     const double* __restrict__ pose, int* __restrict__ trow_occ,
                                                     int* __restrict__ n_occ, int* __restrict__ err, OccLog log,
                                                     const int* __restrict__ gate_prev = nullptr);
__global__ void rbpf_add_repeated_test(const double* __restrict__ x, const double* __restrict__ d, const int* __restrict__ n, double* __restrict__ out, int count);
// rbpf_resample.hip
__global__ __launch_bounds__(256) void rbpf_normalize(int N, const double* __restrict__ zp, const double* weight, double* weight_out,
                                                      double* __restrict__ cs, int* __restrict__ parent, NormOut* __restrict__ out,
                                                      int* __restrict__ gate = nullptr, const int* __restrict__ gate_prev = nullptr,
                                                      unsigned int* seq = nullptr, unsigned int seq_val = 0,
                                                      int* __restrict__ children = nullptr);
// rbpf_raycast.hip
template <int NT, int WPS>
__global__ __launch_bounds__(NT, WPS) void rbpf_raycast_box(ScanC c, TilePool P, MapT M, const double2* __restrict__ beams,
                                                          const double* __restrict__ pose, const double* __restrict__ sens,
                                                          int* __restrict__ trow_occ, int* __restrict__ n_occ, int* __restrict__ err,
                                                          int tile_cap, unsigned long long* __restrict__ touched, NormArgs nz,
                                                          int* __restrict__ box_need, int* __restrict__ box_need_host, int need_slot);
// rbpf_field.hip
__global__ __launch_bounds__(256) void rbpf_densify(GridC g, int p0, TilePool P, MapT M, const int* __restrict__ trow_occ,
                                                    unsigned long long* __restrict__ bitmap, int* __restrict__ row_count);
__global__ void rbpf_window(GridC g, int N, int half_cells, int mark_fresh, const double* __restrict__ pose, int* __restrict__ state,
                            int* __restrict__ skip, int4* __restrict__ win);
template <int C>
__global__ __launch_bounds__(C) void rbpf_edt(GridC g, int radius, const unsigned long long* __restrict__ bitmap,
                                              uint16_t* __restrict__ codes, const int* __restrict__ tier, int my_tier, EdtJob job);
template <int SMAX>
__global__ __launch_bounds__(kWave) void rbpf_edt_compact(GridC g, int radius, const unsigned long long* __restrict__ bitmap,
                                                          const int* __restrict__ row_count,
                                                          uint16_t* __restrict__ codes, int* __restrict__ tier, int my_tier, EdtJob job);
// rbpf_resample.hip
__global__ void rbpf_pool_init(TilePool P);
__global__ __launch_bounds__(kResampleThreads) void rbpf_resample_apply(int N, int TT, const int* __restrict__ parent, const int* __restrict__ children,
                                                           const unsigned int* __restrict__ tab_old, unsigned int* __restrict__ tab_new,
                                                           unsigned int* __restrict__ shed, TilePool P, int table_blocks, int chunks,
                                                           GatherArgs ga);
__global__ __launch_bounds__(256) void rbpf_tiles_to_dense(int xs, size_t G, TilePool P, MapT M, int p, double* __restrict__ out);
__global__ __launch_bounds__(kWave) void rbpf_dense_to_tiles(int xs, double cut_occ, TilePool P, MapT M, int p, const double* __restrict__ in,
                                                             int* __restrict__ trow_occ, int* __restrict__ n_occ, int* __restrict__ err);
__global__ __launch_bounds__(256) void rbpf_release_slot(TilePool P, MapT M, int p);
// rbpf_migrate.hip
__global__ __launch_bounds__(256) void rbpf_pack_tiles(TilePool P, const unsigned int* __restrict__ ids, double* __restrict__ out,
                                                       unsigned int* __restrict__ out_bm);
__global__ __launch_bounds__(256) void rbpf_unpack_tiles(TilePool P, MapT M, int p, const unsigned int* __restrict__ tidx,
                                                         const double* __restrict__ in, const unsigned int* __restrict__ in_bm,
                                                         int* __restrict__ err);
__global__ __launch_bounds__(256) void rbpf_count_tiles(MapT M, const int* __restrict__ slots, const int* __restrict__ fstate, int2* __restrict__ out);
__global__ __launch_bounds__(256) void rbpf_pack_batch(TilePool P, MapT M, const double* __restrict__ pose, const double* __restrict__ prev,
                                                       const double* __restrict__ weight, const int* __restrict__ trow, const int* __restrict__ nocc,
                                                       const int* __restrict__ fstate, const uint16_t* __restrict__ codes, size_t G, int xsize,
                                                       const BatchItem* __restrict__ items, char* __restrict__ buf);
__global__ __launch_bounds__(256) void rbpf_blob_headers(const BatchItem* __restrict__ items, const char* __restrict__ buf, BlobHeader* __restrict__ out, int n);
__global__ __launch_bounds__(256) void rbpf_release_slots(TilePool P, MapT M, const BatchItem* __restrict__ items);
__global__ __launch_bounds__(256) void rbpf_unpack_batch(TilePool P, MapT M, double* __restrict__ pose, double* __restrict__ prev,
                                                         double* __restrict__ weight, int* __restrict__ trow, int* __restrict__ nocc,
                                                         int* __restrict__ fstate, uint16_t* __restrict__ codes, size_t G,
                                                         const BatchItem* __restrict__ items, const char* __restrict__ buf, int* __restrict__ err);
// rbpf_resample.hip
__global__ __launch_bounds__(256) void rbpf_gather_weights(int N, const double* __restrict__ gw, const int* __restrict__ parent, double* __restrict__ weight);
__global__ __launch_bounds__(256) void rbpf_argmax(int N, const double* __restrict__ weight, const double* __restrict__ pose,
                                                   int* __restrict__ best_idx, double* __restrict__ best_pose);
__global__ __launch_bounds__(256) void rbpf_export_map(int xs, size_t G, ExportCuts cuts, const int* __restrict__ best_idx,
                                                       TilePool P, MapT M, int8_t* __restrict__ out);

#ifdef TBNAV_PHASE_PROF
// development builds: each kernel file prints the stamps its kernels left (tbnav_rbpf_destroy calls both)
void rbpf_prof_print_propose();
void rbpf_prof_print_raycast();
#endif

}  // namespace tbnav_rk
#endif  // TBNAV_RBPF_DEVICE_HPP
