// rbpf.hip — MI355X (gfx950) implementation of bmapping::ParticleFilter::SLAM behind the C-ABI of
// include/tbnav_rbpf.h.  Reference (paths relative to the reference tree):
//   bmapping/src/bmapping/particle_filter.cpp:141-251 (SLAM), :295-322, :383-437, :442-500, :504-599
//   bmapping/src/bmapping/grid_mapper.cpp:69-182 (likelihood field, integrateScan), :549-898
//   bmapping/src/bmapping/sensor_model.cpp:43-112 (laserEndPoints)
//
// Kernels (fp64 / integer; compiled with -ffp-contract=off; DESIGN.md section 4 has the reasoning and the numbers):
//   rbpf_sample_normals   production noise source (Philox + Box-Muller) when the caller passes no normals
//   rbpf_propose          workgroup per particle: k sampled poses, ONE likelihood lookup per beam at their centre,
//                         stable-beam collapse of the k x Bv evaluations, Gaussian proposal, 3x3 Cholesky, new pose,
//                         weight *= eta; lookups by exact nearest-obstacle query on an LDS slice of the bitmap
//                         (particle_filter.cpp:158-231, grid_mapper.cpp:69-133)
//   rbpf_scanmatch        option (N1): per-particle hill climbing on the likelihood field before sampling
//   rbpf_raycast_tile     workgroup per particle: LDS tile of 16-bit counters over the scan's bounding box, integer DDA
//                         walk per ray segment, end-point cells replayed in beam order, log-odds += l_free / l_occ,
//                         the tiles' occupancy bits kept current         (grid_mapper.cpp:140-178, :549-807)
//   rbpf_raycast          fallback when the tile cannot hold the scan: one wave per particle, beams in order
//   rbpf_normalize(_seq)  sequential-order normalise / Neff / low-variance selection (particle_filter.cpp:442-500)
//   rbpf_resample_apply   tables, reference counts and parents' state into the alternate buffers after a resample (:495 deep copies)
//   rbpf_argmax, rbpf_export_map   getRobotState / newMap on the device (:255-291, grid_mapper.cpp:185-226)
//   rbpf_densify          dense bitmap rows of a range of particles from their tiles, for the exact-transform kernels
//   rbpf_window, rbpf_edt_compact<R>, rbpf_edt<C>, rbpf_field_by_query
//                         the stored u16 distance field: windowed / whole-map exact EDT (modes TBNAV_RBPF_DF=window|full,
//                         and every on-demand field), cell-by-cell query for maps too large for the LDS transform
//                         (replaces the whole-map priority-queue brushfire, grid_mapper.cpp:333-435)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <atomic>
#include <map>
#include <thread>
#include <vector>
#include <sched.h>

#include "comm.hpp"
#include "common.hpp"
#include "ref_field.hpp"
#include "tbnav_rbpf.h"

namespace {

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
__device__ unsigned long long g_trace_p[2][4][16];  // [which][wave][stamp] of TWO proposal workgroups (blockIdx.x == 96, 100: XCCs 0 and 4)
#define TRACE_P(i) do { if ((blockIdx.x == 96 || blockIdx.x == 100) && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 4) g_trace_p[blockIdx.x == 100][threadIdx.x >> 6][i] = wall_clock64(); } while (0)
__device__ unsigned long long g_wgp[4096][3];   // [workgroup] entry, exit (10 ns ticks), XCC_ID << 32 | HW_ID of the LAST proposal launch
#define WGP_IN() do { if (threadIdx.x == 0 && blockIdx.x < 4096) { g_wgp[blockIdx.x][0] = wall_clock64(); \
  g_wgp[blockIdx.x][2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned int)__builtin_amdgcn_s_getreg(63492); } } while (0)
#define WGP_OUT() do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_wgp[blockIdx.x][1] = wall_clock64(); } while (0)
#else
#define TRACE_P(i)
#define WGP_IN()
#define WGP_OUT()
#endif
#ifdef TBNAV_PHASE_PROF
__device__ unsigned long long g_phase[8];
__device__ unsigned long long g_phase_p[8];
#define PHASE_STAMP(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase[i], now_ - t_prev_); t_prev_ = now_; } } while (0)
#define PHASE_STAMP_P(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase_p[i], now_ - t_prev_); t_prev_ = now_; } } while (0)
#ifdef TBNAV_TRACE_ONLY
#undef PHASE_STAMP
#undef PHASE_STAMP_P
#define PHASE_STAMP(i)
#define PHASE_STAMP_P(i)
#endif
#else
#define PHASE_STAMP(i)
#define PHASE_STAMP_P(i)
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
__global__ void rbpf_mix_lut(ScanC c, double* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < kMixLut) out[q] = beam_mixture(c, (uint16_t)q);
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
// (also carries the scan's beam table from pinned host memory to the device — n_copy entries, 0 = none: one launch and
//  one dependent boundary fewer per scan than a separate copy)
// blockIdx.y: scan within a chunk of consecutive scans (tbnav_rbpf_slam_batch draws a few scans ahead in one launch) — scan
// number scan + y, normals at out + y * out_stride, beam tables at + y * beam_stride.
// Sharded filters (tbnav_rbpf_set_rng_shard): the handle's local normal j is element base + j of the ENSEMBLE's stream and the
// resampling offset (slot z_slot of `out`) is element z_index of it, so ranks that share a seed draw disjoint normals — the ones
// the unsharded filter of all the particles would draw.  base = 0 / z_index = ~0: one contiguous stream of n values (unsharded).
__global__ void rbpf_sample_normals(size_t n, unsigned long long seed, unsigned long long scan, double* __restrict__ out,
                                    const double2* __restrict__ host_beams, double2* __restrict__ dev_beams, int n_copy,
                                    size_t out_stride = 0, size_t beam_stride = 0, size_t base = 0, size_t z_index = ~(size_t)0,
                                    size_t z_slot = 0) {
  scan += blockIdx.y; out += blockIdx.y * out_stride; host_beams += blockIdx.y * beam_stride; dev_beams += blockIdx.y * beam_stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_copy; i += gridDim.x * blockDim.x) dev_beams[i] = host_beams[i];
  auto pair = [&](size_t P, double& a_out, double& b_out) {
    unsigned int r[4];
    philox4x32_10((scan << 40) + P, seed, r);
    const unsigned long long a = ((unsigned long long)r[0] << 32) | r[1], b = ((unsigned long long)r[2] << 32) | r[3];
    const double u1 = ((double)(a >> 11) + 0.5) * 0x1.0p-53, u2 = ((double)(b >> 11) + 0.5) * 0x1.0p-53;
    const double rad = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    a_out = rad * cs; b_out = rad * sn;
  };
  const size_t p0 = base >> 1, pairs = n ? ((base + n - 1) >> 1) - p0 + 1 : 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x) {
    double va, vb;
    pair(p0 + i, va, vb);
    const size_t g0 = 2 * (p0 + i);
    if (g0 >= base && g0 < base + n) out[g0 - base] = va;
    if (g0 + 1 >= base && g0 + 1 < base + n) out[g0 + 1 - base] = vb;
  }
  if (z_index != ~(size_t)0 && blockIdx.x == 0 && threadIdx.x == 0) {
    double va, vb;
    pair(z_index >> 1, va, vb);
    out[z_slot] = (z_index & 1) ? vb : va;
  }
}

struct Trace {
  double *sampled, *p_scan, *p_pose, *mu, *sigma, *eta, *new_pose, *weight_raw;
};



// Whole field of ONE particle for maps whose column envelope does not fit a workgroup's LDS (xsize > ~640): every
// cell asks the same exact query the likelihood uses (rows i, i+-1, ... on the global bitmap).  On-demand path only
// (get_occ_dist / get_dist_code / export) — the SLAM path of such maps runs in query mode and never needs it.
__global__ __launch_bounds__(256) void rbpf_field_by_query(GridC g, int radius, int particle, TilePool P, MapT M,
                                                           const int* __restrict__ trow_occ, uint16_t* __restrict__ codes) {
  const size_t G = (size_t)g.xsize * g.ysize;
  const size_t cell = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= G) return;
  const int ci = (int)(cell / g.xsize), cj = (int)(cell - (size_t)ci * g.xsize);
  uint16_t* code = codes + (size_t)particle * G;
  const DistSrc ds{code, occ_of(P, M, trow_occ, particle), make_int4(0, 0, 0, 0), 2, nullptr, nullptr, 0, 0, 0, 0};
  code[cell] = nearest_code_query(g, ds, radius, ci, cj);  // a cell out of reach keeps its stored code, like the transform
}

// GridMapper::likelihoodFieldModel (grid_mapper.cpp:69-133) of ONE particle's map at an arbitrary pose — the host
// class bmapping::GridMapper's method of that name (tbnav_rbpf_likelihood).  One wave; product in beam order per lane,
// closed by the wave's butterfly.
__global__ __launch_bounds__(kWave) void rbpf_likelihood_one(ScanC c, const double2* __restrict__ beams, const uint16_t* __restrict__ codes,
                                                            TilePool P, MapT M, const int* __restrict__ trow_occ,
                                                            const int* __restrict__ fstate, int radius, const int* __restrict__ n_occ,
                                                            double th, double x, double y, double* __restrict__ out, int* __restrict__ err,
                                                            const double* __restrict__ mixlut) {
  const int p = c.p0, lane = threadIdx.x;
  const DistSrc ds{codes ? codes + (size_t)p * c.g.xsize * c.g.ysize : nullptr, occ_of(P, M, trow_occ, p),
                   make_int4(0, c.g.xsize - 1, 0, c.g.ysize - 1), (codes && fstate[p] == 2) ? 0 : 2, nullptr, nullptr, 0, 0, 0, 0};
  int oob = 0;
  const double v = wave_scan_likelihood(c, beams, ds, radius, n_occ[p], th, x, y, lane, &oob, MixLut{nullptr, mixlut});
  if (oob & 1) atomicOr(&err[0], 1);
  if (lane == 0) *out = v;
}

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
__global__ __launch_bounds__(kMatchThreads) void rbpf_scanmatch(ScanC c, ScanMatchC sm, const double2* __restrict__ beams,
                                                                const uint16_t* __restrict__ codes,
                                                                TilePool P, MapT M,
                                                                const int* __restrict__ trow_occ, const int* __restrict__ skip,
                                                                int skip_eq, int df_mode, int radius, int occ_half,
                                                                const int* __restrict__ n_occ, const int4* __restrict__ win,
                                                                const double* __restrict__ pose, double* __restrict__ center,
                                                                double* __restrict__ score, int* __restrict__ err,
                                                                const int* __restrict__ gate_prev, const double* __restrict__ mixlut) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (gate_prev && *gate_prev) return;  // the scan before this one resamples: see rbpf_raycast_box
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  double2* lbeams = reinterpret_cast<double2*>(lds);                       // [Bv]
  double* lut = reinterpret_cast<double*>(lbeams + c.Bv);                 // [kMixLut] mixture term per distance code: the
  //   matcher scores ~100 poses x Bv beams, nearly all of them a few cells from a wall (sqrt + exp each otherwise)
  // per-beam cache of looked-up cells, shared by the six waves: [Bv][4] words, slot = parity of (ci, cj) — the four
  // cells of any 2 x 2 neighbourhood never collide, and the matcher's poses move a beam's end point by a cell or two.
  // One u64 per entry ((cell + 1) << 16 | code) so that concurrent writers leave a consistent entry either way.
  unsigned long long* ccache = reinterpret_cast<unsigned long long*>(lut + kMixLut);
  unsigned long long* const tile_bm = ccache + (size_t)4 * c.Bv;
  __shared__ double cur[3], best, steps[2], cand[6];
  __shared__ int refinements, done;
  const double th0 = pose[p * 3 + 0], x0 = pose[p * 3 + 1], y0 = pose[p * 3 + 2];
  double s0, c0;
  sincos(th0, &s0, &c0);
  const double mu0[3] = {th0 + c.Ticp[0], c0 * c.Ticp[1] - s0 * c.Ticp[2] + x0, s0 * c.Ticp[1] + c0 * c.Ticp[2] + y0};
  const int nocc = n_occ[p];
  if (nocc == 0) {  // empty map: the likelihood is 1 everywhere (grid_mapper.cpp:94-98), nothing can improve
    if (tid == 0) { center[p * 3 + 0] = mu0[0]; center[p * 3 + 1] = mu0[1]; center[p * 3 + 2] = mu0[2]; score[p] = 1.0; }
    return;
  }
  DistSrc ds{codes ? codes + (size_t)p * c.g.xsize * c.g.ysize : nullptr, occ_of(P, M, trow_occ, p),
             win[p], skip[p] == skip_eq ? 0 : df_mode, tile_bm, reinterpret_cast<const int*>(tile_bm), 0, 0, 0, 0};
  for (int b = tid; b < c.Bv; b += kMatchThreads) lbeams[b] = beams[b];
  for (int q = tid; q < kMixLut; q += kMatchThreads) lut[q] = mixlut[q];  // (the handle's table: same values, no sqrt / exp here)
  for (int q = tid; q < 4 * c.Bv; q += kMatchThreads) ccache[q] = 0ull;
  if (ds.mode == 2 && occ_half > 0) {  // the same LDS slice of the bitmap as the proposal kernel, round the first guess
    double Tc[4];
    sensor_transform(c, mu0[0], mu0[1], mu0[2], Tc);
    int sci, scj;
    if (world2cell(c.g, Tc[0], Tc[1], sci, scj)) {
      const int R0 = max(0, sci - occ_half), R1 = min(c.g.xsize - 1, sci + occ_half);
      const int W0 = max(0, scj - occ_half) >> 6, W1 = min(c.g.ysize - 1, scj + occ_half) >> 6, nW = W1 - W0 + 1;
      int* ta = reinterpret_cast<int*>(tile_bm + (size_t)(R1 - R0 + 1) * nW);
      for (int r = tid; r <= R1 - R0; r += kMatchThreads) {
        unsigned long long acc = 0ull;
        for (int w = 0; w < nW; ++w) {
          const unsigned long long v = ds.occ.word(R0 + r, W0 + w);
          tile_bm[r * nW + w] = v;
          acc |= v;
        }
        ta[r] = acc != 0ull;
      }
      __shared__ unsigned char sm_lut7[128];  // nearest_code_query's 7 x 7 look (visible after the barrier below)
      if (tid < 128) {
        int best = 100;
        for (int cbit = 0; cbit < 7; ++cbit) if ((tid >> cbit) & 1) { const int dc = cbit - 3; best = min(best, dc * dc); }
        sm_lut7[tid] = (unsigned char)best;
      }
      ds.tany = ta; ds.R0 = R0; ds.R1 = R1; ds.W0 = W0; ds.nW = nW; ds.lut7 = sm_lut7;
    }
  }
  if (tid == 0) { cur[0] = mu0[0]; cur[1] = mu0[1]; cur[2] = mu0[2]; steps[0] = sm.lstep; steps[1] = sm.astep; refinements = 0; done = 0; }
  __syncthreads();
  int oob = 0;
  auto likelihood = [&](double th, double x, double y) {
    double T[4];
    sensor_transform(c, th, x, y, T);
    double pr = 1.0;
    for (int b = lane; b < c.Bv; b += kWave) {
      const double2 pt = lbeams[b];
      int ci, cj;
      if (!world2cell(c.g, T[3] * pt.x - T[2] * pt.y + T[0], T[2] * pt.x + T[3] * pt.y + T[1], ci, cj)) { oob |= 1; continue; }
      const unsigned long long cell1 = (unsigned long long)(ci * c.g.xsize + cj) + 1ull;
      unsigned long long* slot = ccache + 4 * b + ((ci & 1) | ((cj & 1) << 1));
      const unsigned long long e = *slot;
      int cd;
      if ((e >> 16) == cell1) cd = (int)(e & 0xFFFFull);
      else {
        cd = lookup_code<false>(c.g, ds, radius, ci, cj);
        if (cd < 0) { oob |= 2; continue; }
        *slot = (cell1 << 16) | (unsigned long long)cd;
      }
      pr *= cd < kMixLut ? lut[cd] : beam_mixture(c, (uint16_t)cd);
    }
    return wave_prod(pr);
  };
  if (wid == 0) {
    const double l0 = likelihood(cur[0], cur[1], cur[2]);
    if (lane == 0) best = l0;
  }
  __syncthreads();
  for (int round = 0; round < sm.max_moves; ++round) {
    {
      const double sgn = (wid & 1) ? -1.0 : 1.0;
      double q[3] = {cur[0], cur[1], cur[2]};
      if (wid < 2) q[1] = cur[1] + sgn * steps[0];
      else if (wid < 4) q[2] = cur[2] + sgn * steps[0];
      else q[0] = normalize_angle_PI(cur[0] + sgn * steps[1]);
      const double sc = likelihood(q[0], q[1], q[2]);
      if (lane == 0) cand[wid] = sc;
    }
    __syncthreads();
    if (tid == 0) {
      double cb = best;
      int arg = -1;
      for (int m = 0; m < 6; ++m) if (cand[m] > cb * (1.0 + 1e-9)) { cb = cand[m]; arg = m; }
      if (arg >= 0) {
        const double sgn = (arg & 1) ? -1.0 : 1.0;
        if (arg < 2) cur[1] = cur[1] + sgn * steps[0];
        else if (arg < 4) cur[2] = cur[2] + sgn * steps[0];
        else cur[0] = normalize_angle_PI(cur[0] + sgn * steps[1]);
        best = cb;
      } else {
        steps[0] *= 0.5; steps[1] *= 0.5;
        if (++refinements >= sm.iters) done = 1;
      }
    }
    __syncthreads();
    if (done) break;
  }
  if (oob & 1) atomicOr(&err[0], 1);
  if (oob & 2) atomicOr(&err[3], 4);
  if (tid == 0) { center[p * 3 + 0] = cur[0]; center[p * 3 + 1] = cur[1]; center[p * 3 + 2] = cur[2]; score[p] = best; }
}

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

// err[0] = out of world, err[1] = eta zero, err[2] = pdf variance zero, err[3] = bresenham
//
// One workgroup per particle (particle_filter.cpp:158-231).  Round 4's schedule — six barriers on the usual path, nine before:
//   0. every thread: pose / table / beams requested together; the sensor transform at the centre of the samples, T(pose) * T_icp;
//      the slice of the occupancy bitmap within reach of the sensor staged in LDS (two round trips: table entries, then rows)
//   1. wave 0: the k samples, their sensor transforms, how far any of them is from the centre, and — same lanes, no barrier in
//      between — the odometry likelihood of every sample (:542);
//      the OTHER waves, beside it: ONE lookup per beam at the centre (cell, code, mixture term) and the distance of the centre's
//      end point from the nearest border of its cell.  (Round 3 ran the samples first, a barrier, then the two side by side.)
//   2. [only if a lookup could not be settled by the 7 x 7 look] every thread: the full nearest-obstacle search for those beams
//   3. wave 0: a beam is STABLE if that distance exceeds what the samples' spread can move an end point
//          |e_j - e_c|_inf <= max_j |T_j - T_c|_inf + |beam| * max_j |theta_j - theta_c|   (chord <= arc)  + 1e-9 m:
//      every sample then sees the beam in the centre's cell, i.e. with the centre's term — the k x Bv evaluations of the
//      reference (grid_mapper.cpp:100-121 from particle_filter.cpp:541) collapse to Bv + (k x the few unstable beams); the
//      product over the stable beams and the list of the unstable ones, in beam order
//   4. every thread: one (sample, unstable beam) pair each, kUnCap unstable beams at a time (any number of them: chunks);
//      [rarely: the full search for pairs that need it]; each sample's thread multiplies its terms in beam order, clamps,
//      forms likelihoods.at(j) and writes the trace
//   5. wave 0 alone: the weighted sums, the 3 x 3 LLT, the new pose, weight *= eta (:545-599, :214-231)
// The ICP-failed branch (:161-176) is steps 0, 1 (every wave looks beams up, at the moved pose), 2 and a product.
// Same cells, same terms as the reference's brute force; only the ORDER of the products / sums differs (asserted <= 1e-9).
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
                                                                const double* __restrict__ mixlut) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (gate_prev && *gate_prev) return;  // the scan before this one resamples: see rbpf_raycast_box
  const int p = blockIdx.x;
  const int k = c.k;
  __shared__ double sh_mix[kMixLds];  // the head of the handle's mixture table (filled below, visible after the first barrier)
  const MixLut mixL{sh_mix, mixlut};
  double* smp = lds;               // [k][3]
  double* pscan = lds + 3 * k;     // [k]
  double* ppose = lds + 4 * k;     // [k]
  double* stf = lds + 5 * k;       // [k][4] sensor transform of sample j; later reused as wj[k]
  double* fac = lds + 12 * k;      // [k][kUnCap] per-(sample, unstable beam) terms of one chunk of unstable beams
  double2* lbeams = reinterpret_cast<double2*>(lds + (12 + kUnCap) * k);  // [Bv] the scan, staged: every later read is an LDS read
  double* cpz = lds + (12 + kUnCap) * k + 2 * c.Bv;  // [Bv] mixture term of beam b at the samples' centre
  unsigned int* ctag = reinterpret_cast<unsigned int*>(cpz + c.Bv);  // [Bv] the code it was computed for, or one of kTag*
  unsigned int* ccell = ctag + c.Bv;                                 // [Bv] the cell that code was looked up at (0xFFFFFFFF: none)
  float* marg = reinterpret_cast<float*>(ccell + c.Bv);              // [Bv] distance of the centre's end point from its cell's nearest border, rounded DOWN (-1: no code)
  int* ulist = reinterpret_cast<int*>(marg + c.Bv);                  // [<= Bv] the unstable beams, ascending
  constexpr unsigned int kTagNone = 0xFFFFFFFFu;    // a windowed lookup outside the window
  constexpr unsigned int kTagSearch = 0xFFFFFFFEu;  // query mode: the 7 x 7 look did not settle it — step 2
  constexpr unsigned int kTagOut = 0xFFFFFFFDu;     // the end point is outside the world
  constexpr unsigned int kBoxHi = 0x7FF8C0DEu;      // high word of a NaN that carries a cell index: a pair term waiting for step 4's search
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  const uint16_t* code = codes ? codes + (size_t)p * c.g.xsize * c.g.ysize : nullptr;  // NULL: no stored field (query mode)
  const double* z = normals + (size_t)p * c.stride_normals;
  const int nocc = n_occ[p];
  // a particle whose field is authoritative (injected / whole-field fresh) always reads it
  // (the tile pointers are set unconditionally — nW == 0 means "no tile" — so that the compiler can see they are LDS
  //  addresses and use ds_read instead of flat loads in the lookups)
  unsigned long long* const tile_bm = reinterpret_cast<unsigned long long*>(ulist + c.Bv);
  DistSrc ds{code, occ_of(P, M, trow_occ, p), win[p], skip[p] == skip_eq ? 0 : df_mode,
             tile_bm, reinterpret_cast<const int*>(tile_bm), 0, 0, 0, 0};
  int oob = 0;
  __shared__ int sh_def[2];  // [0] beams, [1] pairs whose lookup needs the full search (counts that only grow)
  if (tid == 0) { sh_def[0] = 0; sh_def[1] = 0; }

  // ---- 0. loads, the centre of the samples, the LDS slice of the occupancy bitmap
  WGP_IN();
  const double th0v = pose[p * 3 + 0], x0v = pose[p * 3 + 1], y0v = pose[p * 3 + 2];
  const double pv0 = prev_pose[p * 3 + 0], pv1 = prev_pose[p * 3 + 1], pv2 = prev_pose[p * 3 + 2];
  // (requested WITH the pose, used much later: the first 64 samples' normals by wave 0, the new pose's three normals and the
  //  particle's weight by the last step — each was a dependent round trip on the workgroup's critical path)
  double zj0 = 0.0, zj1 = 0.0, zj2 = 0.0;   // sample tid's normals (samples beyond the workgroup's size load theirs in step 1)
  if (c.icp_ok && tid < k) { zj0 = z[3 * tid + 0]; zj1 = z[3 * tid + 1]; zj2 = z[3 * tid + 2]; }
  double zz0 = 0.0, zz1 = 0.0, zz2 = 0.0, w_old = 0.0;
  if (c.icp_ok && wid == 0) { zz0 = z[3 * k + 0]; zz1 = z[3 * k + 1]; zz2 = z[3 * k + 2]; w_old = weight[p]; }
  // (a table of at most NT entries — maps up to 512 x 512 cells at 256 threads — is requested WHOLE here, with the pose: which
  //  entries the window needs depends on the pose, and waiting for it made the staging below three dependent round trips)
  const int tt_all = ds.occ.TW * ds.occ.TW;
  const bool whole_table = ds.mode == 2 && occ_half > 0 && nocc && tt_all <= NT && tt_all <= 256;
  unsigned int my_id = 0u;
  if (whole_table && tid < tt_all) my_id = ds.occ.tab[tid];
  TRACE_P(0);
  const double th0 = uniform_d(th0v), x0 = uniform_d(x0v), y0 = uniform_d(y0v);
  double mu0[3];
  if (!c.icp_ok) {
    // ICP failed: the pose moves by the odometry motion model (particle_filter.cpp:161-176, :295-322) — every thread works it
    // out (three draws, two sincos), and the LDS slice of the bitmap is staged round THAT pose's sensor
    const double w0 = c.Lm[0] * z[0], w1 = c.Lm[1] * z[1], w2 = c.Lm[2] * z[2];
    const double uw = c.u[0], uvx = c.u[1];
    if (almost_equal(uw, 0.0)) {
      mu0[0] = normalize_angle_PI(th0 + w0);
      mu0[1] = x0 + (uvx * cos(mu0[0]) + w1);
      mu0[2] = y0 + (uvx * sin(mu0[0]) + w2);
    } else {
      mu0[0] = normalize_angle_PI(th0 + uw + w0);
      mu0[1] = x0 + ((-uvx / uw) * sin(mu0[0]) + (uvx / uw) * sin(mu0[0] + uw) + w1);
      mu0[2] = y0 + ((uvx / uw) * cos(mu0[0]) - (uvx / uw) * cos(mu0[0] + uw) + w2);
    }
  } else {
    double s0, c0;
    sincos(th0, &s0, &c0);
    // the mode the samples are drawn round: T(pose) * T_icp, or the particle's own scan-matched pose (N1 option)
    mu0[0] = center ? center[p * 3 + 0] : th0 + c.Ticp[0];
    mu0[1] = center ? center[p * 3 + 1] : c0 * c.Ticp[1] - s0 * c.Ticp[2] + x0;
    mu0[2] = center ? center[p * 3 + 2] : s0 * c.Ticp[1] + c0 * c.Ticp[2] + y0;
  }
  mu0[0] = uniform_d(mu0[0]); mu0[1] = uniform_d(mu0[1]); mu0[2] = uniform_d(mu0[2]);
  const double pv[3] = {uniform_d(pv0), uniform_d(pv1), uniform_d(pv2)};
  for (int b = tid; b < c.Bv; b += NT) lbeams[b] = beams[b];  // visible after the next barrier
  for (int q = tid; q < kMixLds; q += NT) sh_mix[q] = mixlut[q];
  double Tc[4];  // sensor transform at the centre of the samples
  sensor_transform(c, mu0[0], mu0[1], mu0[2], Tc);
  Tc[0] = uniform_d(Tc[0]); Tc[1] = uniform_d(Tc[1]); Tc[2] = uniform_d(Tc[2]); Tc[3] = uniform_d(Tc[3]);
  // (the normals have arrived with the pose: parked in the samples' own LDS slots until wave 0 turns them into samples, so that
  //  they do not hold six registers through the staging)
  if (c.icp_ok && tid < k) { smp[3 * tid + 0] = zj0; smp[3 * tid + 1] = zj1; smp[3 * tid + 2] = zj2; }
  TRACE_P(1);
  bool staged = false;
  if (ds.mode == 2 && occ_half > 0 && nocc) {
    // query mode: stage the bitmap rows/columns within occ_half cells of the sensor in LDS — every lookup of this
    // block ends within range_max of it, and its nearest obstacle is usually a few cells further at most
    int sci, scj;
    if (world2cell(c.g, Tc[0], Tc[1], sci, scj)) {
      const int R0 = max(0, sci - occ_half), R1 = min(c.g.xsize - 1, sci + occ_half);
      const int W0 = max(0, scj - occ_half) >> 6, W1 = min(c.g.ysize - 1, scj + occ_half) >> 6, nW = W1 - W0 + 1;
      unsigned long long* tb = tile_bm;
      int* ta = reinterpret_cast<int*>(tile_bm + (size_t)(R1 - R0 + 1) * nW);
      // Two round trips instead of a chain of dependent ones per word: the ids of the tiles under the window go to LDS
      // first, then every row requests its (up to kStC) 32-bit pieces at once.
      constexpr int kStC = 12, kStIds = 256;
      __shared__ unsigned int st_ids[kStIds];
      const int tr0 = R0 >> kTSh, tc0 = 2 * W0, ntc = min(2 * nW, ds.occ.TW - tc0), n_ids = ((R1 >> kTSh) - tr0 + 1) * ntc;
      if (ntc <= kStC && (whole_table || n_ids <= kStIds)) {
        if (whole_table) { if (tid < tt_all) st_ids[tid] = my_id; }
        else
          for (int q = tid; q < n_ids; q += NT) {
            const int qi = floor_div_small(q, ntc);
            st_ids[q] = ds.occ.tab[(tr0 + qi) * ds.occ.TW + tc0 + (q - qi * ntc)];
          }
        __syncthreads();
        TRACE_P(2);
        for (int r = tid; r <= R1 - R0; r += NT) {
          const int row = R0 + r;
          const unsigned int* ids = whole_table ? st_ids + (row >> kTSh) * ds.occ.TW + tc0 : st_ids + ((row >> kTSh) - tr0) * ntc;
          unsigned int v32[kStC];
#pragma unroll
          for (int q = 0; q < kStC; ++q) v32[q] = q < ntc ? ds.occ.bm[(size_t)ids[q] * kTS + (row & (kTS - 1))] : 0u;
          unsigned long long acc = 0ull;
#pragma unroll
          for (int w = 0; w < kStC / 2; ++w) {
            if (w < nW) {
              const unsigned long long v = (unsigned long long)v32[2 * w] | ((unsigned long long)v32[2 * w + 1] << 32);
              tb[r * nW + w] = v;
              acc |= v;
            }
          }
          ta[r] = acc != 0ull;
        }
      } else {
        for (int r = tid; r <= R1 - R0; r += NT) {
          unsigned long long acc = 0ull;
          for (int w = 0; w < nW; ++w) {
            const unsigned long long v = ds.occ.word(R0 + r, W0 + w);
            tb[r * nW + w] = v;
            acc |= v;
          }
          ta[r] = acc != 0ull;
        }
      }
      // the 128-entry table of the 7 x 7 look (visible after the barrier below)
      __shared__ unsigned char sh_lut7[128];
      if (tid < 128) {
        int best = 100;
        for (int cbit = 0; cbit < 7; ++cbit) if ((tid >> cbit) & 1) { const int dc = cbit - 3; best = min(best, dc * dc); }
        sh_lut7[tid] = (unsigned char)best;
      }
      ds.tany = ta; ds.R0 = R0; ds.R1 = R1; ds.W0 = W0; ds.nW = nW; ds.lut7 = sh_lut7;
    }
    __syncthreads();
    staged = true;
  }
  if (!staged) __syncthreads();  // lbeams / sh_mix / sh_def
  TRACE_P(3);
  zz0 = uniform_d(zz0); zz1 = uniform_d(zz1); zz2 = uniform_d(zz2); w_old = uniform_d(w_old);  // (arrived long ago; wave-uniform: scalar registers from here on)

  // ---- 1. wave 0 (ICP ok): samples, their sensor transforms, their odometry likelihoods.  The other waves (ICP failed: every
  //      wave): one lookup per beam at the centre.
  constexpr int kPW = NT / kWave;
  __shared__ double sh_spread[2];
  double dxy = 0.0, dth = 0.0;  // wave 0: how far any sample's sensor is from the centre's
  if (c.icp_ok && wid == 0) {
    int var_err = 0;
    const double nrot1 = uniform_d(normalize_angle_PI(c.rot1)), nrot2 = uniform_d(normalize_angle_PI(c.rot2));
    for (int j = lane; j < k; j += kWave) {
      double sj[3];
      const bool parked = j < NT;
      const double n0 = parked ? smp[3 * j + 0] : z[3 * j + 0], n1 = parked ? smp[3 * j + 1] : z[3 * j + 1], n2 = parked ? smp[3 * j + 2] : z[3 * j + 2];
      sj[0] = mu0[0] + c.Ld[0] * n0; sj[1] = mu0[1] + c.Ld[1] * n1; sj[2] = mu0[2] + c.Ld[2] * n2;
      dth = fmax(dth, fabs(c.Ld[0] * n0));
      sj[0] = normalize_angle_PI(sj[0]);
      smp[3 * j + 0] = sj[0]; smp[3 * j + 1] = sj[1]; smp[3 * j + 2] = sj[2];
      {
        double T[4];
        sensor_transform(c, sj[0], sj[1], sj[2], T);
        stf[4 * j + 0] = T[0]; stf[4 * j + 1] = T[1]; stf[4 * j + 2] = T[2]; stf[4 * j + 3] = T[3];
        dxy = fmax(dxy, fmax(fabs(T[0] - Tc[0]), fabs(T[1] - Tc[1])));
      }
      // (the samples' spread first: the other waves' step 3 needs it, nothing needs the odometry likelihoods before step 4)
      ppose[j] = pose_likelihood_odom(c, &smp[3 * j], pv, &var_err, nrot1, nrot2);   // (:542: against prev_pose as it stands — updated only after this call)
    }
    dxy = wave_max_d(dxy); dth = wave_max_d(dth);
    if (lane == 0) { sh_spread[0] = dxy; sh_spread[1] = dth; }
    if (var_err) atomicOr(&err[2], 1);
  } else if (nocc) {
    const int b_first = c.icp_ok ? tid - kWave : tid, b_step = c.icp_ok ? NT - kWave : NT;
    bool deferred = false;
    for (int b = b_first; b < c.Bv; b += b_step) {
      const double2 pt = lbeams[b];
      const double ex = Tc[3] * pt.x - Tc[2] * pt.y + Tc[0], ey = Tc[2] * pt.x + Tc[3] * pt.y + Tc[1];
      int ci, cj;
      unsigned int tag = kTagOut, cell = 0xFFFFFFFFu;
      double pz = 0.0;
      float mg = -1.0f;
      if (world2cell(c.g, ex, ey, ci, cj)) {
        const int cd = lookup_code_fast(c.g, ds, radius, ci, cj);
        tag = kTagNone;
        if (cd != -1) {
          cell = (unsigned int)(ci * c.g.xsize + cj);
          if (cd == kNeedSearch) { tag = kTagSearch; deferred = true; }
          else { tag = (unsigned int)cd; pz = mix_term(c, mixL, cd); }
          const double x_lo = c.g.xmin + ci * c.g.res, x_hi = c.g.xmin + (ci + 1) * c.g.res;
          const double y_lo = c.g.ymin + cj * c.g.res, y_hi = c.g.ymin + (cj + 1) * c.g.res;
          // (kept as a float rounded DOWN: a beam can only become unstable by it, never wrongly stable)
          mg = __double2float_rd(fmin(fmin(ex - x_lo, x_hi - ex), fmin(ey - y_lo, y_hi - ey)));
        }
      }
      ctag[b] = tag; ccell[b] = cell; cpz[b] = pz; marg[b] = mg;
    }
    if (deferred) atomicAdd(&sh_def[0], 1);
  }
  TRACE_P(4);
  __syncthreads();
  // ---- 2. the lookups the 7 x 7 look did not settle (a beam that ends more than three cells from every obstacle the slice
  //      shows: the first scans of a map, a doorway): the full search, all threads, one inlined copy
  if (sh_def[0]) {  // workgroup-uniform
    for (int b = tid; b < c.Bv; b += NT) {
      if (ctag[b] != kTagSearch) continue;
      const int cell = (int)ccell[b], ci = cell / c.g.xsize, cj = cell - ci * c.g.xsize;
      const int cd = nearest_code_query_body(c.g, ds, radius, ci, cj);
      ctag[b] = (unsigned int)cd;
      cpz[b] = mix_term(c, mixL, cd);
    }
    __syncthreads();
  }
  TRACE_P(5);
  __shared__ double sh_pst[kPW];
  if (!c.icp_ok) {
    // weight *= likelihoodFieldModel(scan, T(new pose)) (:171-175): the product per lane, per wave, then over the waves in wave
    // order (a fixed order; the reference multiplies beam by beam: tolerance)
    double pr = 1.0;
    if (nocc)
      for (int b = tid; b < c.Bv; b += NT) {
        const unsigned int tg = ctag[b];
        if (tg == kTagOut) oob |= 1;          // the reference throws from world2RowMajor
        else if (tg == kTagNone) oob |= 2;    // (window mode: sized so that this cannot happen — reported, never read stale)
        else pr *= cpz[b];
      }
    pr = wave_prod(pr);
    if (lane == 0) sh_pst[wid] = pr;
    if (oob & 1) atomicOr(&err[0], 1);
    if (oob & 2) atomicOr(&err[3], 4);
    __syncthreads();
    if (tid == 0) {
      double sl = sh_pst[0];
      for (int w = 1; w < kPW; ++w) sl *= sh_pst[w];
      if (!nocc) sl = 1.0;  // grid_mapper.cpp:94-98
      prev_pose[p * 3 + 0] = th0; prev_pose[p * 3 + 1] = x0; prev_pose[p * 3 + 2] = y0;
      pose[p * 3 + 0] = mu0[0]; pose[p * 3 + 1] = mu0[1]; pose[p * 3 + 2] = mu0[2];
      tr.new_pose[p * 3 + 0] = mu0[0]; tr.new_pose[p * 3 + 1] = mu0[1]; tr.new_pose[p * 3 + 2] = mu0[2];
      const double w = weight[p] * sl;
      weight[p] = w;
      tr.p_scan[(size_t)p * k] = sl;
      tr.weight_raw[p] = w;
      sens[p * 4 + 0] = Tc[0]; sens[p * 4 + 1] = Tc[1]; sens[p * 4 + 2] = Tc[2]; sens[p * 4 + 3] = Tc[3];  // the sensor transform of the new pose, for the raycast kernel
    }
    WGP_OUT();
    return;
  }
  // ---- 3. which beams are stable, the product of their terms, the others listed: every wave over ITS contiguous range of
  //      beams [w C, (w + 1) C), its unstable ones compacted (in beam order) into its own segment of ulist — no wave waits for
  //      another's count; the pairs below walk the segments in wave order, i.e. the unstable beams in beam order
  __shared__ int sh_cnt[kPW];
  const int seg = (c.Bv + kPW - 1) / kPW;  // beams per wave's range
  {
    const double sdxy = sh_spread[0], sdth = sh_spread[1];
    int n = 0;
    double pst = 1.0;
    if (nocc) {
      const int b_lo = wid * seg, b_hi = min(c.Bv, b_lo + seg);
      for (int b0 = b_lo; b0 < b_hi; b0 += kWave) {
        const int b = b0 + lane;
        bool unstable = false;
        if (b < b_hi) {
          const double2 pt = lbeams[b];
          // (|beam| only has to be bounded from above: the fp32 root, rounded up by more than its error)
          const double delta = sdxy + (double)(sqrtf((float)(pt.x * pt.x + pt.y * pt.y)) * 1.000001f) * sdth + 1e-9;
          const bool stable = ctag[b] < 0x10000u && (double)marg[b] > delta;
          if (stable) pst *= cpz[b];
          unstable = !stable;
        }
        const unsigned long long m = __ballot(unstable);
        if (unstable) ulist[b_lo + n + __popcll(m & ((1ull << lane) - 1ull))] = b;
        n += __popcll(m);
      }
    }
    pst = wave_prod(pst);
    if (lane == 0) { sh_cnt[wid] = n; sh_pst[wid] = pst; }
  }
  __syncthreads();
  TRACE_P(6);
  // ---- 4. scan likelihood of every sample: (product over the stable beams) * (its own terms of the unstable ones)
  double* wj = stf;  // [k] likelihoods.at(j) (the sensor transforms are dead once the pairs are through)
  {
    int n_un = 0;
    double p_stable = 1.0;  // grid_mapper.cpp:94-98: 1.0 until the map has an occupied cell
    if (nocc)
      for (int w = 0; w < kPW; ++w) { n_un += sh_cnt[w]; p_stable *= sh_pst[w]; }
    // the i-th unstable beam of the scan: segment by segment
    auto unstable_beam = [&](int i) {
      int w = 0;
#pragma unroll
      for (int q = 0; q < kPW - 1; ++q) { const int cq = sh_cnt[q]; if (w == q && i >= cq) { i -= cq; ++w; } }
      return ulist[w * seg + i];
    };
    for (int j = tid; j < k; j += NT) pscan[j] = p_stable;
    int seen = 0;
    for (int u0 = 0; u0 < n_un; u0 += kUnCap) {
      const int nu = min(kUnCap, n_un - u0);
      // one THREAD per (sample, unstable beam) of this chunk (grid_mapper.cpp:100-121 for that sample's pose and that beam)
      bool deferred = false;
      for (int pair = tid; pair < k * nu; pair += NT) {
        const int j = floor_div_small(pair, nu), i = pair - j * nu;
        const int b = unstable_beam(u0 + i);
        const double2 pt = lbeams[b];
        const double X = stf[4 * j + 0], Y = stf[4 * j + 1], st = stf[4 * j + 2], ct = stf[4 * j + 3];
        const double ex = ct * pt.x - st * pt.y + X, ey = st * pt.x + ct * pt.y + Y;
        int ci, cj;
        double term = 1.0;
        if (!world2cell(c.g, ex, ey, ci, cj)) oob |= 1;  // (the reference throws from world2RowMajor)
        else {
          const unsigned int cell = (unsigned int)(ci * c.g.xsize + cj);
          if (cell == ccell[b]) term = cpz[b];  // the centre's cell -> its code -> its term
          else {
            const int cd = lookup_code_fast(c.g, ds, radius, ci, cj);
            if (cd == kNeedSearch) { term = __hiloint2double((int)kBoxHi, (int)cell); deferred = true; }
            else if (cd < 0) oob |= 2;
            else term = ((unsigned int)cd == ctag[b]) ? cpz[b] : mix_term(c, mixL, cd);
          }
        }
        fac[j * kUnCap + i] = term;
      }
      if (deferred) atomicAdd(&sh_def[1], 1);
      __syncthreads();
      const int def_now = sh_def[1];
      if (def_now != seen) {  // workgroup-uniform: some pair of this chunk waits for the full search
        seen = def_now;
        for (int pair = tid; pair < k * nu; pair += NT) {
          const int j = floor_div_small(pair, nu), i = pair - j * nu;
          const double v = fac[j * kUnCap + i];
          if ((unsigned int)__double2hiint(v) != kBoxHi) continue;
          const int cell = __double2loint(v), ci = cell / c.g.xsize, cj = cell - ci * c.g.xsize;
          const int cd = nearest_code_query_body(c.g, ds, radius, ci, cj);
          const int b = unstable_beam(u0 + i);
          fac[j * kUnCap + i] = ((unsigned int)cd == ctag[b]) ? cpz[b] : mix_term(c, mixL, cd);
        }
        __syncthreads();
      }
      for (int j = tid; j < k; j += NT) {
        double pr = pscan[j];
        for (int i = 0; i < nu; ++i) pr *= fac[j * kUnCap + i];
        pscan[j] = pr;
      }
      if (u0 + kUnCap < n_un) __syncthreads();  // (the next chunk rewrites fac)
    }
    if (oob & 1) atomicOr(&err[0], 1);
    if (oob & 2) atomicOr(&err[3], 4);
    if (n_un > 0) __syncthreads();  // (stf -> wj: every pair has read its sample's transform)
    // the sample's own thread: clamps, likelihoods.at(j), the trace (:541-556)
    for (int j = tid; j < k; j += NT) {
      const double psj = pscan[j], ppj = ppose[j];
      const double ps = fmin(fmax(psj, c.scan_min), c.scan_max);  // std::clamp
      const double pp = fmin(fmax(ppj, c.pose_min), c.pose_max);
      tr.p_scan[(size_t)p * k + j] = psj;
      tr.p_pose[(size_t)p * k + j] = ppj;
      tr.sampled[((size_t)p * k + j) * 3 + 0] = smp[3 * j + 0];
      tr.sampled[((size_t)p * k + j) * 3 + 1] = smp[3 * j + 1];
      tr.sampled[((size_t)p * k + j) * 3 + 2] = smp[3 * j + 2];
      wj[j] = ps * pp;
    }
  }
  __syncthreads();
  TRACE_P(7);
  // ---- 5. Gaussian proposal (:522-599), new pose (:214-231): wave 0 alone.  The weighted sums are lane-strided partial sums
  //      closed with a butterfly — a fixed order, not the reference's left-to-right one: the results agree to rounding
  //      (asserted at 1e-10 against the oracle) — and every lane of the wave holds them, so nothing goes through LDS again.
  if (wid != 0) return;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int j = lane; j < k; j += kWave) {
    const double pj = wj[j];
    for (int q = 0; q < 3; ++q) a[q] += smp[3 * j + q] * pj;
    a[3] += pj;
  }
  for (int q = 0; q < 4; ++q) a[q] = wave_sum_d(a[q]);
  const double eta = a[3];
  if (almost_equal(eta, 0.0)) {  // "eta is 0" (:563, reported): the pose stays, and so does its sensor transform
    if (lane == 0) {
      atomicOr(&err[1], 1);
      double Ts[4];
      sensor_transform(c, th0, x0, y0, Ts);
      for (int q = 0; q < 4; ++q) sens[p * 4 + q] = Ts[q];
    }
    return;
  }
  double mu[3] = {a[0] / eta, a[1] / eta, a[2] / eta};
  mu[0] = normalize_angle_PI(mu[0]);
  double su[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int j = lane; j < k; j += kWave) {
    const double d[3] = {smp[3 * j + 0] - mu[0], smp[3 * j + 1] - mu[1], smp[3 * j + 2] - mu[2]};
    const double w = wj[j];
    int o = 0;
    for (int r = 0; r < 3; ++r) for (int q = r; q < 3; ++q) su[o++] += (d[r] * d[q]) * w;
  }
  for (int o = 0; o < 6; ++o) su[o] = wave_sum_d(su[o]);
  TRACE_P(8);
  if (lane == 0) {
    double sigma[3][3];
    {
      int o = 0;
      for (int r = 0; r < 3; ++r) for (int q = r; q < 3; ++q) { sigma[r][q] = su[o] / eta; sigma[q][r] = sigma[r][q]; ++o; }
    }
    double L[3][3];
    llt3(sigma, L);
    double np[3];
    for (int r = 0; r < 3; ++r) np[r] = mu[r] + ((L[r][0] * zz0 + L[r][1] * zz1) + L[r][2] * zz2);
    prev_pose[p * 3 + 0] = th0; prev_pose[p * 3 + 1] = x0; prev_pose[p * 3 + 2] = y0;
    for (int q = 0; q < 3; ++q) { pose[p * 3 + q] = np[q]; tr.new_pose[p * 3 + q] = np[q]; tr.mu[p * 3 + q] = mu[q]; }
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) tr.sigma[p * 9 + r * 3 + q] = sigma[r][q];
    tr.eta[p] = eta;
    const double w = w_old * eta;
    weight[p] = w;
    tr.weight_raw[p] = w;
    double Ts[4];  // the sensor transform of the new pose, for the raycast kernel (saves it two sincos on its critical path)
    sensor_transform(c, np[0], np[1], np[2], Ts);
    for (int q = 0; q < 4; ++q) sens[p * 4 + q] = Ts[q];
  }
  TRACE_P(9);
  WGP_OUT();
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

__global__ __launch_bounds__(kWave) void rbpf_raycast(ScanC c, TilePool P, MapT M, const double2* __restrict__ beams,
                                                     const double* __restrict__ pose, int* __restrict__ trow_occ,
                                                     int* __restrict__ n_occ, int* __restrict__ err, OccLog log,
                                                     const int* __restrict__ gate_prev = nullptr) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];
  if (gate_prev && *gate_prev) return;  // the scan before this one resamples: see rbpf_raycast_box
  int* ex = lds_i;         // [Bv]
  int* ey = lds_i + c.Bv;  // [Bv]
  unsigned int* tbits = reinterpret_cast<unsigned int*>(lds_i + 2 * c.Bv);  // [(TT + 31) / 32] tiles this scan writes
  __shared__ int bad;
  const int p = c.p0 + blockIdx.x, lane = threadIdx.x;
  unsigned int* tab = M.table + (size_t)p * M.TT;
  unsigned int* shed = M.shed + (size_t)p * M.TT;
  int* rc = trow_occ + (size_t)p * M.TW;
  int* nocc = n_occ + p;
  const double th = pose[p * 3 + 0], x = pose[p * 3 + 1], y = pose[p * 3 + 2];
  if (lane == 0) bad = 0;
  const int tword = (M.TT + 31) / 32;
  for (int w = lane; w < tword; w += kWave) tbits[w] = 0u;
  __syncthreads();
  double s0, c0;
  sincos(th, &s0, &c0);
  const double X = c0 * c.Trs[1] - s0 * c.Trs[2] + x;
  const double Y = s0 * c.Trs[1] + c0 * c.Trs[2] + y;
  double st, ct;
  sincos(th + c.Trs[0], &st, &ct);
  for (int b = lane; b < c.Bv; b += kWave) {
    const double2 pt = beams[b];
    int ci = 0, cj = 0;
    if (!world2cell(c.g, ct * pt.x - st * pt.y + X, st * pt.x + ct * pt.y + Y, ci, cj)) bad = 1;
    ex[b] = ci; ey[b] = cj;
  }
  int rx = 0, ry = 0;
  if (!world2cell(c.g, x, y, rx, ry)) bad = 1;  // freeGridIndex: world2Grid of the ROBOT pose (:558)
  __syncthreads();
  if (bad) { if (lane == 0) atomicOr(&err[0], 1); return; }
  // which tiles does this scan write?  (one extra walk of the rays; this kernel is the fallback / reference-mode path)
  for (int b = 0; b < c.Bv; ++b) {
    const int x1 = ex[b], y1 = ey[b];
    const Ray r = make_ray(rx, ry, x1, y1);
    for (int n = lane; n < r.count; n += kWave) {
      int cx, cy;
      ray_cell(r, n, cx, cy);
      const int t = tile_of(M, cx, cy);
      atomicOr(&tbits[t >> 5], 1u << (t & 31));
    }
    if (lane == 0) { const int t = tile_of(M, x1, y1); atomicOr(&tbits[t >> 5], 1u << (t & 31)); }
  }
  __syncthreads();
  {
    int need = 0;  // tiles to clone: one pop of the ring for all of them
    for (int w = 0; w < tword; ++w) {
      unsigned int m = tbits[w];
      while (m) {
        const int t = w * 32 + __ffs((int)m) - 1;
        m &= m - 1;
        if (!tile_is_private(P, tab, t)) ++need;
      }
    }
    if (need) {
      unsigned long long base = 0ull;
      if (lane == 0) base = tile_pop_n(P, (unsigned int)need);
      base = ((unsigned long long)__shfl((int)(base >> 32), 0, kWave) << 32) | (unsigned int)__shfl((int)base, 0, kWave);
      if (base == ~0ull) { if (lane == 0) atomicOr(&err[3], 8); return; }  // tile pool exhausted: nothing has been written
      for (int w = 0; w < tword; ++w) {
        unsigned int m = tbits[w];
        while (m) {
          const int t = w * 32 + __ffs((int)m) - 1;
          m &= m - 1;
          if (!tile_is_private(P, tab, t)) { tile_clone_into(P, tab, shed, t, tile_at(P, base), lane); ++base; }
        }
      }
    }
  }
  __syncthreads();
  int n_log = 0;
  int* ev = log.ev ? log.ev + (size_t)p * log.cap : nullptr;
  for (int b = 0; b < c.Bv; ++b) {
    const int x1 = ex[b], y1 = ey[b];
    const Ray r = make_ray(rx, ry, x1, y1);
    for (int n0 = 0; n0 < r.count; n0 += kWave) {
      const int n = n0 + lane;
      bool flip = false;
      int cell = 0;
      if (n < r.count) {
        int cx, cy;
        ray_cell(r, n, cx, cy);
        cell = cx * c.g.xsize + cy;
        flip = add_log_odds(P, tab[tile_of(M, cx, cy)], c.d_free, c.cut_occ, cx, cy, rc, nocc);
      }
      if (ev) {  // a free add can only take a cell OUT of the occupied set
        const unsigned long long m = __ballot(flip);
        if (flip) { const int at = n_log + __popcll(m & ((1ull << lane) - 1ull)); if (at < log.cap) ev[at] = cell | (int)0x80000000; }
        n_log += __popcll(m);
      }
    }
    __syncthreads();  // free-cell adds of this beam land before the endpoint / next beam touch the cells
    int eflip = 0;
    if (lane == 0) {
      const unsigned int eid = tab[tile_of(M, x1, y1)];
      const double before = P.lo[(size_t)eid * kTileCells + in_tile(x1, y1)];
      const bool flip = add_log_odds(P, eid, c.d_occ, c.cut_occ, x1, y1, rc, nocc);
      if (ev && flip && n_log < log.cap) ev[n_log] = (x1 * c.g.xsize + y1) | (before >= c.cut_occ ? (int)0x80000000 : 0);
      eflip = flip ? 1 : 0;
    }
    if (ev) n_log += __shfl(eflip, 0, kWave);
    __syncthreads();
  }
  if (ev && lane == 0) log.count[p] = n_log;
}

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
__global__ void rbpf_add_repeated_test(const double* __restrict__ x, const double* __restrict__ d, const int* __restrict__ n, double* __restrict__ out, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = add_repeated(x[i], d[i], n[i]);
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
__global__ __launch_bounds__(256) void rbpf_normalize(int N, const double* __restrict__ zp, const double* weight, double* weight_out,
                                                      double* __restrict__ cs, int* __restrict__ parent, NormOut* __restrict__ out,
                                                      int* __restrict__ gate = nullptr, const int* __restrict__ gate_prev = nullptr,
                                                      unsigned int* seq = nullptr, unsigned int seq_val = 0,
                                                      int* __restrict__ children = nullptr) {
  __shared__ __attribute__((aligned(16))) double w[kNormChunk], cl[kNormChunk];
  if (gate_prev && *gate_prev) return;
  normalize_body<256, true>(N, zp, weight, weight_out, cs, parent, out, w, cl, gate, seq, seq_val, children);
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
// ev u16[Bv][kBoxEv] | val_e f64[Bv + 64] | exy own ecnt i32[Bv]
#ifdef TBNAV_PHASE_PROF
__device__ unsigned long long g_phase_w[16];
#endif
#if defined(TBNAV_PHASE_PROF) && !defined(TBNAV_TRACE_ONLY)
#define PHASE_STAMP_W(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_phase_w[i], now_ - t_prev_); t_prev_ = now_; } } while (0)
#else
#define PHASE_STAMP_W(i)
#endif
#ifdef TBNAV_PHASE_PROF
__device__ unsigned long long g_trace[2][16][16];  // [which][wave][stamp] of TWO workgroups (blockIdx.x == 100: first round of residents; 900: second): 10 ns ticks
#define TRACE_W(i) do { if ((blockIdx.x == 100 || blockIdx.x == 900) && (threadIdx.x & 63) == 0) g_trace[blockIdx.x == 900][threadIdx.x >> 6][i] = wall_clock64(); } while (0)
__device__ unsigned long long g_wg[4096][3];    // [workgroup] entry, exit (10 ns ticks), XCC_ID << 32 | HW_ID — of the LAST launch
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
__host__ __device__ constexpr size_t box_lds_bytes(size_t cap, size_t bv) { return 4 * cap + 8 * (bv + 64) + 4 * 2 * bv + 2 * kBoxEv * bv; }
// (512 threads: three workgroups = 24 waves per CU when the LDS array is sized by what the boxes need, see launch_raycast —
//  6 waves per SIMD leave 80 registers a lane: the kernel needs 77 and spills nothing; 1024 threads: two workgroups = 32 waves, 64)
#ifndef TBNAV_RC512_WAVES
#define TBNAV_RC512_WAVES 6
#endif
template <int NT>
__global__ __launch_bounds__(NT, NT == 512 ? TBNAV_RC512_WAVES : 8) void rbpf_raycast_box(ScanC c, TilePool P, MapT M, const double2* __restrict__ beams,
                                                          const double* __restrict__ pose, const double* __restrict__ sens,
                                                          int* __restrict__ trow_occ, int* __restrict__ n_occ, int* __restrict__ err,
                                                          int tile_cap, unsigned long long* __restrict__ touched, NormArgs nz,
                                                          int* __restrict__ box_need, int* __restrict__ box_need_host, int need_slot) {
  extern __shared__ __attribute__((aligned(16))) int lds_i[];
  // enqueued behind a scan whose resampling decision the host had not seen yet: if that scan resamples, this launch does
  // nothing (the host runs the copies and enqueues this scan again)
  if (nz.gate_prev && *nz.gate_prev) return;
  // nz.N > 0: workgroup 0 is not a particle's — it normalises the weights the proposal kernel left and selects the parents
  // (one workgroup of dependent adds, independent of the maps: it rides in this launch, beside the map updates, instead of
  // costing a second stream, an event and a dependent boundary); the particles' workgroups follow
  if (nz.N > 0 && blockIdx.x == 0) {
    double* w = reinterpret_cast<double*>(lds_i);
    normalize_body<NT, false>(nz.N, nz.zp, nz.weight, nz.weight_out, nz.cs, nz.parent, nz.out, w, w + kNormChunk, nz.gate, nz.seq, nz.seq_val,
                   nz.children);
    return;
  }
  const int Bv = c.Bv;
  unsigned int* tile = reinterpret_cast<unsigned int*>(lds_i);         // (tile_cap is a multiple of 8)
  unsigned short* ev = reinterpret_cast<unsigned short*>(lds_i + tile_cap);  // [Bv][kBoxEv]  beam << 1 | occupied  (16 bytes a slot, 16-byte aligned)
  double* val_e = reinterpret_cast<double*>(lds_i + tile_cap + 4 * Bv);      // [Bv + 64] the value replayed for an end-point cell / a hot cell
  int* exy = lds_i + tile_cap + 4 * Bv + 2 * (Bv + 64);  // [Bv] end-point cell, x | y << 16
  int* ecnt = exy + Bv;                                  // [Bv] events recorded in the slot of beam b — the FIRST beam that ended in its cell (0: b opened
                                                         //      no slot; may exceed kBoxEv: overflow)
  constexpr unsigned int kFlag = 0x80000000u;
  constexpr int kEv = kBoxEv;
  __shared__ int bad, bx0, bx1, by0, by1, srx, sry, nocc_delta, n_ovf;
  __shared__ unsigned long long need_base;
  __shared__ unsigned int mt_id[kMapTilesMax];  // map tiles under the box: the tile the particle's table names (once written: its private tile)
  __shared__ int mt_touch[kMapTilesMax], mt_slot[kMapTilesMax], mt_priv[kMapTilesMax];
  __shared__ int rc_delta[kBoxSideMax / kTS + 2];
  __shared__ int ovf[kWave];                    // slots whose event list overflowed (more than these: found by scanning)
  __shared__ double sh_pose[4];
  __shared__ double robot_v0, robot_v;  // the robot's own cell: its log-odds before the scan / after the adds applied so far
  __shared__ int robot_cnt;             // beams with a free cell (each adds l_free to the robot's cell once)
  constexpr int nthr = NT, nw = NT / kWave;
  const int p = c.p0 + blockIdx.x - (nz.N > 0 ? 1 : 0), tid_k = threadIdx.x, tid = tid_k, lane = tid & (kWave - 1), wid = tid / kWave;
#ifdef TBNAV_PHASE_PROF
  unsigned long long t_prev_ = wall_clock64();
#endif
  unsigned int* tab = M.table + (size_t)p * M.TT;
  unsigned int* shed = M.shed + (size_t)p * M.TT;
  TRACE_W(0);
  WG_IN();
  if (wid == 0) {
    const double x = pose[p * 3 + 1], y = pose[p * 3 + 2];
    int rx0 = 0, ry0 = 0;
    const bool robot_ok = world2cell(c.g, x, y, rx0, ry0);  // freeGridIndex: world2Grid of the ROBOT pose (:558)
    double X, Y, st0, ct0;
    if (sens) { X = sens[p * 4 + 0]; Y = sens[p * 4 + 1]; st0 = sens[p * 4 + 2]; ct0 = sens[p * 4 + 3]; }
    else {
      const double th = pose[p * 3 + 0];
      double s0, c0;
      sincos(th, &s0, &c0);
      if (c.Trs[0] == 0.0) { st0 = s0; ct0 = c0; } else sincos(th + c.Trs[0], &st0, &ct0);
      X = c0 * c.Trs[1] - s0 * c.Trs[2] + x;
      Y = s0 * c.Trs[1] + c0 * c.Trs[2] + y;
    }
    if (lane == 0) {
      sh_pose[0] = X; sh_pose[1] = Y; sh_pose[2] = st0; sh_pose[3] = ct0;
      bad = robot_ok ? 0 : 1; bx0 = bx1 = rx0; by0 = by1 = ry0; srx = rx0; sry = ry0;
      nocc_delta = 0; n_ovf = 0; robot_cnt = 0;
    }
  } else {
    uint4* t4 = reinterpret_cast<uint4*>(tile);
    for (int t = tid - kWave; t < tile_cap / 4; t += nthr - kWave) t4[t] = uint4{0u, 0u, 0u, 0u};
    for (int b = tid - kWave; b < Bv; b += nthr - kWave) ecnt[b] = 0;
    for (int t = tid - kWave; t < kMapTilesMax; t += nthr - kWave) mt_touch[t] = 0;
    for (int t = tid - kWave; t < kBoxSideMax / kTS + 2; t += nthr - kWave) rc_delta[t] = 0;
  }
  __syncthreads();
  TRACE_W(1);
  // (workgroup-uniform values read from LDS are moved to scalar registers: the kernel has 64 VGPRs to live in)
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  const int rx = uni(srx), ry = uni(sry);
  {
    const double X = sh_pose[0], Y = sh_pose[1], st = sh_pose[2], ct = sh_pose[3];
    for (int b0 = wid * kWave; b0 < Bv; b0 += nthr) {
      const int b = b0 + lane;
      int ci = rx, cj = ry;
      if (b < Bv) {
        const double2 pt = beams[b];
        if (!world2cell(c.g, ct * pt.x - st * pt.y + X, st * pt.x + ct * pt.y + Y, ci, cj)) { bad = 1; ci = rx; cj = ry; }
        exy[b] = ci | (cj << 16);
      }
      const int lo_x = wave_min_dpp(ci), hi_x = wave_max_dpp(ci), lo_y = wave_min_dpp(cj), hi_y = wave_max_dpp(cj);
      const unsigned long long has_free = __ballot(b < Bv && (ci != rx || cj != ry));  // the ray has a free cell: its first is the robot's
      if (lane == 0) {
        atomicMin(&bx0, lo_x); atomicMax(&bx1, hi_x); atomicMin(&by0, lo_y); atomicMax(&by1, hi_y);
        if (has_free) atomicAdd(&robot_cnt, __popcll(has_free));
      }
    }
    // (The robot's own cell takes one add per beam: lane 0 of the last wave, which walks no ray, fetches it and works the adds
    //  out beside the walk — add_repeated: no chain of dependent adds.  Fetched HERE and looked at in front of the flag barrier, its
    //  two dependent loads held the whole workgroup up: 1-3 us per particle once the chip is loaded.)
  }
  __syncthreads();
  TRACE_W(2);
  if (bad) { if (tid == 0) atomicOr(&err[0], 1); return; }
  const int minx = uni(bx0), maxx = uni(bx1), maxy = uni(by1);
  const int miny = uni(by0) & ~1;                                   // the box starts on an even column and is an even number of
  const int bw = ((maxy | 1) + 1) - miny;                           // columns wide: a PAIR of cells never straddles a row or a map tile
  const int bh = maxx - minx + 1;
  const int tx0 = minx >> kTSh, ty0 = miny >> kTSh, mty = (maxy >> kTSh) - ty0 + 1, mtn = ((maxx >> kTSh) - tx0 + 1) * mty;
  const int rows_fit = uni(floor_div_small(tile_cap, bw));          // rows of the box the LDS array holds at a time
  if (rows_fit < 1 || mtn > kMapTilesMax || bh > kBoxSideMax) { if (tid == 0) atomicOr(&err[3], 2); return; }  // cannot happen: see launch_raycast
  // What the LDS array would have to hold for this particle's box to be ONE band: the host sizes the array of the scans to
  // come from it (launch_raycast: less LDS per workgroup = three workgroups per CU instead of two).  Three slots take turns:
  // this launch accumulates into need_slot; one workgroup hands the PREVIOUS launch's maximum (complete: stream order) to the
  // host through mapped memory and clears the slot of the next launch.  Nothing waits for any of it.
  // (one particle in sixteen reports: the particles' boxes are a cell or two apart, and a thousand atomics on one word drain at
  //  ~12 ns each while every later load of the wave waits behind its own — 5 us on the first residents' critical path)
  if (box_need && tid == 0 && (blockIdx.x & 15u) == 1u) {
    atomicMax(&box_need[need_slot], bh * bw);
    if ((int)blockIdx.x == 1) {
      *box_need_host = box_need[(need_slot + 2) % 3];
      box_need[(need_slot + 1) % 3] = 0;
    }
  }
  // the particle's table entries under the box, and the reference counts of the tiles they name (needed in phase C)
  // (the table work sits on the last threads of the LAST BUT ONE wave, which walks no ray — the last wave, which walks none
  //  either, has the robot cell's chain to work on)
  const int tq = nthr - kWave - 1 - tid;
  //  — the entry goes to LDS as it arrives (these threads have nothing else to do in the flag phase); the reference count of the
  //  tile it names is fetched beside the walk: nothing needs it before phase C)
  if (tq >= 0 && tq < mtn) {
    const int qi = floor_div_small(tq, mty), qj = tq - qi * mty;
    mt_id[tq] = tab[(tx0 + qi) * M.TW + (ty0 + qj)];
  }
  auto map_tile = [&](int cx, int cy) { return __mul24((cx >> kTSh) - tx0, mty) + ((cy >> kTSh) - ty0); };
  auto cell_ptr = [&](int cx, int cy) -> double* { return P.lo + (size_t)mt_id[map_tile(cx, cy)] * kTileCells + in_tile(cx, cy); };
  auto toggled = [&](int cx, int cy, bool now) {  // the cell crossed the occupied cut-off: its bit in the (private) tile, tile-row count, total
    atomicXor(&P.bm[(size_t)mt_id[map_tile(cx, cy)] * kTS + (cx & (kTS - 1))], 1u << (cy & (kTS - 1)));
    atomicAdd(&rc_delta[(cx >> kTSh) - tx0], now ? 1 : -1);
    atomicAdd(&nocc_delta, now ? 1 : -1);
  };
  auto record = [&](unsigned int word, int what) {  // an event for the flagged cell whose tile word this is
    const int o = (int)((word >> 16) & 0x7FFFu);
    const int en = atomicAdd(&ecnt[o], 1);
    if (en < kEv) ev[o * kEv + en] = (unsigned short)what;
  };
  const int step_r = uni(floor_div_small(2 * nthr, bw)), step_c = 2 * nthr - step_r * bw;  // pair pi + nthr in (row, column) terms
  int n_distinct = 0, n_ends = 0;
  for (int x0 = minx; x0 <= maxx; x0 += rows_fit) {  // one band of rows at a time (one trip unless the box is larger than the LDS array)
    // (per-thread values are re-derived from an opaque copy of the thread index in every trip: hoisted out of this loop they
    //  would be spilled — the kernel has 64 VGPRs — and a spill reload between memory requests serialises them)
    int tid = tid_k;
    asm volatile("" : "+v"(tid));
    const int lane = tid & (kWave - 1), wid = tid / kWave, tq = nthr - kWave - 1 - tid;
    const int nr = (maxx - x0 + 1 < rows_fit) ? maxx - x0 + 1 : rows_fit;
    const int band_cells = __mul24(nr, bw);
    const bool clip = nr != bh;
    if (x0 != minx) {  // (a further band: the LDS state of the previous one is cleared)
      __syncthreads();
      uint4* t4 = reinterpret_cast<uint4*>(tile);
      for (int t = tid; t < tile_cap / 4; t += nthr) t4[t] = uint4{0u, 0u, 0u, 0u};
      for (int b = tid; b < Bv; b += nthr) ecnt[b] = 0;
      for (int t = tid; t < kMapTilesMax; t += nthr) mt_touch[t] = 0;
      if (tid == 0) n_ovf = 0;
      __syncthreads();
    }
    auto cell_t = [&](int e) { return __mul24((e & 0xFFFF) - x0, bw) + ((e >> 16) - miny); };
    auto in_band = [&](int e) { return (unsigned int)((e & 0xFFFF) - x0) < (unsigned int)nr; };
    // F. flag the end-point cells.  The first beam to reach a cell leaves its own index there as the cell's slot — one
    //    compare-and-swap against the cleared word: winner and losers alike know the slot at once — and every beam records its
    //    end-point event straight away (three dependent LDS operations; a flag, a slot counter, the slot number and then the
    //    event in a phase of its own were six).
    for (int b = tid; b < Bv; b += nthr) {
      const int e = exy[b];
      if (!in_band(e)) continue;
      const unsigned int mine = kFlag | ((unsigned int)b << 16);
      const unsigned int old = atomicCAS(&tile[cell_t(e)], 0u, mine);
      record(old ? old : mine, (b << 1) | 1);
    }
    TRACE_W(3);
    __syncthreads();
    PHASE_STAMP_W(0);
    TRACE_W(4);
    // 1. the walk
    TRACE_W(5);
    if (x0 == minx && tq >= 0 && tq < mtn) {
      const unsigned int id = mt_id[tq];
      const int rf = id ? P.ref[id] : 0;
      mt_priv[tq] = (id != 0u && rf == 1) ? 1 : 0;
    }
    if (x0 == minx && tid == nthr - kWave) {  // the robot's own cell (the last wave walks no ray)
      const unsigned int rt = tab[(rx >> kTSh) * M.TW + (ry >> kTSh)];
      const double old = P.lo[(size_t)rt * kTileCells + in_tile(rx, ry)];
      robot_v0 = old; robot_v = add_repeated(old, c.d_free, robot_cnt);
    }
    {
      int S = Bv > 0 ? nthr / Bv : 1;  // segments per ray: as many as give every thread at most one task
      S = S < 1 ? 1 : (S > 4 ? 4 : S);
      const int G = (Bv + kWave - 1) / kWave;
      const unsigned int band_bytes = 4u * (unsigned int)band_cells;
      int n_first = 0;
      for (int task = tid; task < kWave * G * S; task += nthr) {
        const int tb = floor_div_small(task, S), sgm = task - tb * S;
        const int b = __mul24(tb & (kWave - 1), G) + (tb >> 6);  // lanes of a wave take rays spread round the scan
        if (b >= Bv) continue;
        const int e = exy[b];
        const RayP pr = ray_packed(rx, ry, e & 0xFFFF, e >> 16);
        const int count = pr.dmaj, L = floor_div_small(count + S - 1, S);
        int n = __mul24(sgm, L);
        const int n1 = (n + L < count) ? n + L : count;
        if (n >= n1) continue;
        const int two_dmin = 2 * pr.dmin, two_dmaj = 2 * pr.dmaj;
        const int a0 = __mul24(two_dmin, n) - pr.dmaj;
        const int c0 = a0 > 0 ? floor_div_small(a0 + two_dmaj - 1, two_dmaj) : 0;  // operands < 2^24
        int rem = a0 - __mul24(two_dmaj, c0 - 1);
        const int sc = pr.neg ? -c0 : c0;
        // byte offset of the segment's first cell in the band's array, and the byte steps along / across the ray
        int at = 4 * (__mul24((pr.ymajor ? pr.xa + sc : pr.xa + n) - x0, bw) + ((pr.ymajor ? pr.ya + n : pr.ya + sc) - miny));
        const int d_major = 4 * (pr.ymajor ? 1 : bw);
        const int d_both = d_major + 4 * (pr.ymajor ? bw : 1) * (pr.neg ? -1 : 1);
        auto advance = [&]() {
          const int r2 = rem + two_dmin;
          const bool side = r2 > two_dmaj;
          rem = side ? r2 - two_dmaj : r2;
          at += side ? d_both : d_major;
        };
        if (n == 0) { ++n_first; advance(); ++n; }  // position 0 is the robot's cell (or, for a reversed ray, the end point): counted below
        char* const tile_b = reinterpret_cast<char*>(tile);
        // What an add returns is looked at TWO steps later, while the next two adds are in flight: three registers take turns
        // (no register is copied at the top of the loop, which would wait for the add just issued), so the walk never waits
        // for LDS unless it has an event to record.
        unsigned int r0 = 0u, r1 = 0u, r2 = 0u;
        auto look = [&](unsigned int& old) { if (old & kFlag) record(old, b << 1); old = 0u; };
        auto step = [&](auto clipped, unsigned int& fresh, unsigned int& old) {
          if (!decltype(clipped)::value || (unsigned int)at < band_bytes) fresh = atomicAdd(reinterpret_cast<unsigned int*>(tile_b + at), 1u);  // (clipped: the cells of the ray in this band of rows)
          advance();
          look(old);
        };
        auto walk = [&](auto clipped) {
          int m = n1 - n;
          for (; m >= 3; m -= 3) { step(clipped, r0, r1); step(clipped, r1, r2); step(clipped, r2, r0); }
          if (m >= 1) step(clipped, r0, r1);
          if (m >= 2) step(clipped, r1, r2);
        };
        if (clip) walk(std::true_type{}); else walk(std::false_type{});
        look(r0); look(r1); look(r2);
      }
      // the robot's own cell is the first free cell of every ray that has a free cell at all
      n_first = wave_sum_dpp(n_first);
      if (lane == 0 && n_first && (unsigned int)(rx - x0) < (unsigned int)nr) atomicAdd(&tile[__mul24(rx - x0, bw) + (ry - miny)], (unsigned int)n_first);
    }
    TRACE_W(6);
    __syncthreads();  // every event is recorded
    TRACE_W(7);
    PHASE_STAMP_W(1);
    // 2. requests and marks in one pass.  The band as PAIRS of cells (16 bytes of a map tile's row, two tile words): pair
    //    tid + i * nthr for i < 4 — consecutive lanes take consecutive pairs, so a wave's request is whole cache lines.  A
    //    counted or flagged pair marks its map tile as written and asks for its log-odds from whichever tile the particle's
    //    table names NOW (shared, private or the zero tile hold the same values: the loads fly while the tiles are made
    //    private).  Slots that overflowed are listed on the way.
    const int np = band_cells >> 1;
    const uint2* tile2 = reinterpret_cast<const uint2*>(tile);
    constexpr int kSl = NT == 512 ? 6 : 4;  // pairs a thread holds across the passes (512 threads: 6 fill the 80 registers exactly — 48.6 vs 50.1 us per 1000 particles; 8 spill)
    double2 v[kSl];
    auto pairs = [&](int first, auto&& fn) {  // fn(i, the pair's two tile words, cx, cy of its first cell), i < kSl
      const int pi0 = first + tid;
      int row = floor_div_small(2 * (pi0 < np ? pi0 : 0), bw), col = 2 * (pi0 < np ? pi0 : 0) - __mul24(row, bw);  // cell index < 2^16, bw < 2^8
#pragma unroll
      for (int i = 0; i < kSl; ++i) {
        const int pi = pi0 + i * nthr;
        uint2 w = uint2{0u, 0u};
        if (pi < np) w = tile2[pi];
        fn(i, w, x0 + row, miny + col);
        row += step_r; col += step_c;
        if (col >= bw) { col -= bw; ++row; }
      }
    };
    pairs(0, [&](int i, uint2 w, int cx, int cy) {
      v[i] = double2{0.0, 0.0};
      if (w.x | w.y) {
        const int mt = map_tile(cx, cy);
        mt_touch[mt] = 1;
        v[i] = *reinterpret_cast<const double2*>(P.lo + (size_t)mt_id[mt] * kTileCells + in_tile(cx, cy));
      }
    });
    for (int first = kSl * nthr; first < np; first += kSl * nthr)  // (bands of more than 8 * nthr cells: marks only, their loads follow)
      pairs(first, [&](int, uint2 w, int cx, int cy) { if (w.x | w.y) mt_touch[map_tile(cx, cy)] = 1; });
    for (int o = tid; o < Bv; o += nthr) {
      if (ecnt[o] == 0) continue;
      const int e = exy[o];
      const bool robot_cell = (e & 0xFFFF) == rx && (e >> 16) == ry;  // an end point too: no events from the walk, replayed against every beam
      if (robot_cell) ecnt[o] = kEv + 1;
      if (robot_cell || ecnt[o] > kEv) { const int i = atomicAdd(&n_ovf, 1); if (i < kWave) ovf[i] = o; }
    }
    TRACE_W(8);
    __syncthreads();
    PHASE_STAMP_W(2);
    TRACE_W(9);
    // C. make the written tiles private to the particle (first write after a resample, or first touch of the area): ONE pop
    //    of the free ring for all of them, then one wave per tile copies 8 KB.  Usually there is nothing to do — and every wave
    //    sees that for itself (one ballot over the at most 64 tiles under the box), without a barrier to agree on it.
    const unsigned long long need_m = __ballot(lane < mtn && mt_touch[lane < mtn ? lane : 0] && !mt_priv[lane < mtn ? lane : 0]);
    if (need_m) {  // workgroup-uniform
      if (wid == 0 && lane < mtn) mt_slot[lane] = ((need_m >> lane) & 1ull) ? __popcll(need_m & ((1ull << lane) - 1ull)) : -1;
      if (tid == 0) { need_base = tile_pop_n(P, (unsigned int)__popcll(need_m)); if (need_base == ~0ull) bad = 1; }
      __syncthreads();
      if (bad) { if (tid == 0) atomicOr(&err[3], 8); return; }  // tile pool exhausted (nothing has been written if this is the first band)
      for (int q = wid; q < mtn; q += nw) {
        if (!mt_touch[q] || mt_slot[q] < 0) continue;
        const int qi = floor_div_small(q, mty), qj = q - qi * mty;
        const unsigned int nid = tile_at(P, need_base + (unsigned long long)mt_slot[q]);
        tile_clone_into(P, tab, shed, (tx0 + qi) * M.TW + (ty0 + qj), nid, lane);
        if (lane == 0) { mt_id[q] = nid; mt_priv[q] = 1; }
      }
      __syncthreads();
    }
    PHASE_STAMP_W(3);
    TRACE_W(10);
    auto finish_end = [&](int slot, int cx, int cy, double v0o, double vv) {
      val_e[slot] = vv;
      const bool was = v0o >= c.cut_occ, now = vv >= c.cut_occ;
      if (was != now) toggled(cx, cy, now);
    };
    // 3a. end-point cells whose slot holds every event: one lane each, the events sorted by beam in registers (a
    //     19-comparator network on the 8 sixteen-bit entries; an empty entry sorts last) and applied in that order
    for (int o = tid; o < Bv; o += nthr) {
      const int ne = ecnt[o];
      if (ne == 0 || ne > kEv) continue;
      const int e = exy[o], cx = e & 0xFFFF, cy = e >> 16;
      const double v0o = *cell_ptr(cx, cy);
      // the events in beam order without sorting: every beam that reaches the cell lies within a few beams of the slot's
      // own beam b0, so bit (beam - b0 + 32) of a 64-bit mask per kind orders them (checked per event; a stray one sends
      // the slot to the exhaustive path).  Beam indices are circular: the bits are walked from the one that stands for the
      // lowest ABSOLUTE beam index.
      const uint4 raw = *reinterpret_cast<const uint4*>(ev + o * kEv);
      const unsigned int w4[4] = {raw.x, raw.y, raw.z, raw.w};
      const int base = o - 32;
      unsigned long long m_free = 0ull, m_occ = 0ull;
      bool stray = false;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned int k = (q & 1) ? (w4[q >> 1] >> 16) : (w4[q >> 1] & 0xFFFFu);
        int d = (int)(k >> 1) - base;
        d += d < 0 ? Bv : 0; d -= d >= Bv ? Bv : 0;  // circular distance from base, in [0, Bv)
        const bool valid = q < ne;
        stray |= valid && d > 63;
        const unsigned long long bit = valid ? 1ull << (d & 63) : 0ull;
        if (k & 1u) m_occ |= bit; else m_free |= bit;
      }
      // absolute beam of bit d is base + d (mod Bv): bits from d0 = (base < 0 ? -base : (base + 63 >= Bv ? Bv - base : 0)) up are
      // the low absolute indices when the window wraps
      int d0 = 0;
      if (base < 0) d0 = -base; else if (base + 63 >= Bv) d0 = Bv - base;
      d0 = d0 > 63 ? 0 : d0;
      double vv = v0o;
      if (!stray) {
#pragma unroll 1
        for (int part = 0; part < 2; ++part) {
          const unsigned long long keep = part == 0 ? ~0ull << d0 : ~(~0ull << d0);
          unsigned long long m = (m_free | m_occ) & keep;
          while (m) {
            const int bit = __ffsll((long long)m) - 1;
            vv += ((m_occ >> bit) & 1ull) ? c.d_occ : c.d_free;
            m &= m - 1;
          }
        }
      } else {  // (never seen: an event more than 31 beams from the slot's own) selection by ascending beam from LDS
        int last = -1;
        for (int i = 0; i < ne; ++i) {
          int best = 0x10000;
          for (int j = 0; j < ne; ++j) { const int k = ev[o * kEv + j]; if (k > last && k < best) best = k; }
          vv += (best & 1) ? c.d_occ : c.d_free;
          last = best;
        }
      }
      ++n_ends;
      finish_end(o, cx, cy, v0o, vv);
    }
    TRACE_W(11);
    // 3b. overflowed slots: one wave per cell.  Lanes test beams q = 64*i + lane against the cell (is it q's end point /
    //     one of q's free cells); the two ballots are the cell's update sequence for those 64 beams, replayed in bit (=
    //     beam) order.  Pre-filter: a Bresenham cell lies within one cell of the line robot -> end point.
    {
      const int n_over = uni(n_ovf);
      const int trips = (Bv + kWave - 1) / kWave;
      for (int i0 = wid; i0 < (n_over <= kWave ? n_over : Bv); i0 += nw) {
        const int o = n_over <= kWave ? ovf[i0] : i0;  // (more overflowed slots than the list holds: scan them all)
        if (ecnt[o] <= kEv) continue;
        const int eo = exy[o], cx = eo & 0xFFFF, cy = eo >> 16;
        const double v0o = *cell_ptr(cx, cy);
        double vv = v0o;
        const int ux = cx - rx, uy = cy - ry;
        for (int i = 0; i < trips; ++i) {
          const int q = i * kWave + lane;
          bool is_end = false, hit = false;
          if (q < Bv) {
            const int eq = exy[q], qx = eq & 0xFFFF, qy = eq >> 16;
            is_end = eq == eo;  // the end point is never one of its own ray's free cells
            const int dx = qx - rx, dy = qy - ry;
            const double cr = (double)(ux * dy - uy * dx), l2 = (double)(dx * dx + dy * dy);
            if (!is_end && cr * cr <= l2) hit = on_ray_packed(rx, ry, qx, qy, cx, cy);
          }
          const unsigned long long occm = __ballot(is_end), freem = __ballot(hit);
          unsigned long long m = occm | freem;
          while (m) {
            const int bit = __ffsll((long long)m) - 1;
            vv += ((occm >> bit) & 1ull) ? c.d_occ : c.d_free;
            m &= m - 1;
          }
        }
        if (lane == 0) { ++n_ends; finish_end(o, cx, cy, v0o, vv); }
      }
    }
    // 3h. the cells round the robot: every ray starts there, so they collect tens to hundreds of adds — one long dependent
    //     chain each.  They get a lane of their own in the last wave (which has no end-point cell to replay), are then
    //     flagged like end-point cells, and their group's owner takes the value from val_e.  The few that take kVeryHot adds or
    //     more (the robot's neighbours: up to half the beams each) go to the last wave but one instead, which works them out
    //     without the chain (add_repeated: a few hundred integer instructions per binade, worth it from about a hundred adds);
    //     the two waves run side by side, so the phase lasts as long as a chain of kVeryHot adds, not of the longest.
    if ((wid == nw - 1 || wid == nw - 2) && lane < kHotSide * kHotSide) {
      const int hi = floor_div_small(lane, kHotSide), hx = rx - kHotSide / 2 + hi, hy = ry - kHotSide / 2 + (lane - hi * kHotSide);
      if ((unsigned int)(hx - x0) < (unsigned int)nr && hy >= miny && hy < miny + bw) {
        const int t = __mul24(hx - x0, bw) + (hy - miny);
        const unsigned int f = tile[t];
        const int cnq = (int)(f & 0xFFFFu);
        const bool robot_cell = hx == rx && hy == ry;  // (worked out beside the walk)
        const bool very = !robot_cell && cnq >= kVeryHot;
        if (!(f & kFlag) && (robot_cell ? f != 0u : cnq >= kHotMin) && very == (wid == nw - 2)) {
          double v0o, vv;
          if (robot_cell) { v0o = robot_v0; vv = robot_v; }
          else if (very) { v0o = *cell_ptr(hx, hy); vv = add_repeated(v0o, c.d_free, cnq); }
          else {
            v0o = *cell_ptr(hx, hy); vv = v0o;
            int a = 0;
            for (; a + 4 <= cnq; a += 4) { vv += c.d_free; vv += c.d_free; vv += c.d_free; vv += c.d_free; }
            for (; a < cnq; ++a) vv += c.d_free;
          }
          tile[t] = kFlag | ((unsigned int)(Bv + lane) << 16);
          ++n_distinct;
          finish_end(Bv + lane, hx, hy, v0o, vv);
        }
      }
    }
    TRACE_W(12);
    __syncthreads();  // val_e is complete
    TRACE_W(13);
    PHASE_STAMP_W(4);
    // 3c. the pairs: a plain cell adds its count, an end-point or hot cell takes the value worked out for it, an untouched one
    //     keeps its own; the pair goes back as one 16-byte store (the tile is private to the particle and nobody else writes
    //     these cells)
    for (int first = 0; first < np; first += kSl * nthr) {
      if (first) pairs(first, [&](int i, uint2 w, int cx, int cy) { v[i] = (w.x | w.y) ? *reinterpret_cast<const double2*>(cell_ptr(cx, cy)) : double2{0.0, 0.0}; });
      pairs(first, [&](int i, uint2 w, int cx, int cy) {
        if (!(w.x | w.y)) return;
        // both cells of the pair in ONE loop (two independent chains of adds, predicated on each cell's count): a few
        // straight-line instructions instead of a nest of divergent branches and loops per cell
        const bool plain0 = w.x != 0u && !(w.x & kFlag), plain1 = w.y != 0u && !(w.y & kFlag);
        const int c0 = plain0 ? (int)(w.x & 0xFFFFu) : 0, c1 = plain1 ? (int)(w.y & 0xFFFFu) : 0;
        const double o0 = v[i].x, o1 = v[i].y;
        double n0 = o0, n1 = o1;
        const int cm = c0 > c1 ? c0 : c1;
        for (int a = 0; a < cm; ++a) {
          const double t0 = n0 + c.d_free, t1 = n1 + c.d_free;
          n0 = a < c0 ? t0 : n0;
          n1 = a < c1 ? t1 : n1;
        }
        if ((w.x | w.y) & kFlag) {  // an end-point or hot cell takes the value worked out for it
          if (w.x & kFlag) n0 = val_e[(w.x >> 16) & 0x7FFFu];
          if (w.y & kFlag) n1 = val_e[(w.y >> 16) & 0x7FFFu];
        }
        n_distinct += (plain0 ? 1 : 0) + (plain1 ? 1 : 0);
        *reinterpret_cast<double2*>(cell_ptr(cx, cy)) = double2{n0, n1};
        const bool tog0 = plain0 && ((o0 >= c.cut_occ) != (n0 >= c.cut_occ)), tog1 = plain1 && ((o1 >= c.cut_occ) != (n1 >= c.cut_occ));
        if (tog0 | tog1) {
          if (tog0) toggled(cx, cy, n0 >= c.cut_occ);
          if (tog1) toggled(cx, cy + 1, n1 >= c.cut_occ);
        }
      });
    }
    PHASE_STAMP_W(5);
  }
  TRACE_W(14);
  __syncthreads();
  // the tile-row counts / occupied count of the particle (this workgroup owns them; nothing waits for the adds)
  int* rc = trow_occ + (size_t)p * M.TW;
  for (int r = tid; r <= (maxx >> kTSh) - tx0; r += nthr) if (rc_delta[r]) atomicAdd(&rc[tx0 + r], rc_delta[r]);
  if (tid == 0 && nocc_delta) atomicAdd(&n_occ[p], nocc_delta);
  if (touched) {  // measurement hook (tbnav_rbpf_scan_counts): [0] += cell updates (free adds + end points), [1] += distinct cells written
    __shared__ int cnt_upd, cnt_dis;
    if (tid == 0) { cnt_upd = 0; cnt_dis = 0; }
    __syncthreads();
    int n_upd = 0;
    for (int b = tid; b < Bv; b += nthr) {
      const int e = exy[b], dx = (e & 0xFFFF) - rx, dy = (e >> 16) - ry;
      n_upd += max(dx < 0 ? -dx : dx, dy < 0 ? -dy : dy) + 1;  // free cells of the ray (its Chebyshev length) + the end point
    }
    n_upd = wave_sum_i(n_upd); n_distinct = wave_sum_i(n_distinct + n_ends);
    if (lane == 0) { atomicAdd(&cnt_upd, n_upd); atomicAdd(&cnt_dis, n_distinct); }
    __syncthreads();
    if (tid == 0) { atomicAdd(&touched[0], (unsigned long long)cnt_upd); atomicAdd(&touched[1], (unsigned long long)cnt_dis); }
  }
#if defined(TBNAV_PHASE_PROF) && !defined(TBNAV_TRACE_ONLY)  // (the sums are contended atomics on ONE address: they distort the very timeline TRACE_ONLY records)
  if (tid == 0) { atomicAdd(&g_phase_w[15], 1ull); }
#endif
  WG_OUT();
}

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
constexpr size_t edt_compact_lds(int smax) { return (size_t)smax * kWave * 4 + (size_t)smax * (8 + 6) + 16; }
constexpr int kEdtRowsA = 144;  // 38.9 KB -> 4 waves per CU
constexpr int kEdtRowsB = 288;  // 77.8 KB -> 2 waves per CU

// free ring = every tile but tile 0 (the shared zero tile, pinned)
__global__ void rbpf_pool_init(TilePool P) {
  for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i + 1 < P.cap; i += gridDim.x * blockDim.x) P.ring[i] = i + 1;
  if (blockIdx.x == 0 && threadIdx.x == 0) { P.ctr[0] = 0ull; P.ctr[1] = (unsigned long long)P.cap - 1ull; P.ref[0] = 1 << 30; }
}

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

// ---- particle migration between handles (sharded filter): a particle travels as its state, its
//      per-tile-row counts and ONLY the tiles (log-odds + occupancy bits) it does not share with the zero tile ------------------------------------
__global__ __launch_bounds__(256) void rbpf_pack_tiles(TilePool P, const unsigned int* __restrict__ ids, double* __restrict__ out,
                                                       unsigned int* __restrict__ out_bm) {
  const double2* src = reinterpret_cast<const double2*>(P.lo + (size_t)ids[blockIdx.x] * kTileCells);
  double2* dst = reinterpret_cast<double2*>(out + (size_t)blockIdx.x * kTileCells);
  for (int i = threadIdx.x; i < kTileCells / 2; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x < kTS) out_bm[(size_t)blockIdx.x * kTS + threadIdx.x] = P.bm[(size_t)ids[blockIdx.x] * kTS + threadIdx.x];
}
// one workgroup per received tile: take a free tile, name it in the (released) slot's table, fill it
__global__ __launch_bounds__(256) void rbpf_unpack_tiles(TilePool P, MapT M, int p, const unsigned int* __restrict__ tidx,
                                                         const double* __restrict__ in, const unsigned int* __restrict__ in_bm,
                                                         int* __restrict__ err) {
  __shared__ unsigned int sid;
  if (threadIdx.x == 0) {
    const unsigned int id = tile_pop(P);
    if (id) { P.ref[id] = 1; M.table[(size_t)p * M.TT + tidx[blockIdx.x]] = id; } else atomicOr(&err[3], 8);
    sid = id;
  }
  __syncthreads();
  if (sid == 0u) return;
  const double2* src = reinterpret_cast<const double2*>(in + (size_t)blockIdx.x * kTileCells);
  double2* dst = reinterpret_cast<double2*>(P.lo + (size_t)sid * kTileCells);
  for (int i = threadIdx.x; i < kTileCells / 2; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x < kTS) P.bm[(size_t)sid * kTS + threadIdx.x] = in_bm[(size_t)blockIdx.x * kTS + threadIdx.x];
}

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
// tiles named by each listed slot's table, and the slot's field state
__global__ __launch_bounds__(256) void rbpf_count_tiles(MapT M, const int* __restrict__ slots, const int* __restrict__ fstate, int2* __restrict__ out) {
  __shared__ int tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  const int slot = slots[blockIdx.x];
  const unsigned int* tab = M.table + (size_t)slot * M.TT;
  int c = 0;
  for (int t = threadIdx.x; t < M.TT; t += blockDim.x) c += tab[t] != 0u ? 1 : 0;
  if (c) atomicAdd(&tot, c);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = int2{tot, fstate[slot]};
}
// one workgroup per exported particle: header, state, tile indices (ascending), tile payloads, tile-row counts, stored field
__global__ __launch_bounds__(256) void rbpf_pack_batch(TilePool P, MapT M, const double* __restrict__ pose, const double* __restrict__ prev,
                                                       const double* __restrict__ weight, const int* __restrict__ trow, const int* __restrict__ nocc,
                                                       const int* __restrict__ fstate, const uint16_t* __restrict__ codes, size_t G, int xsize,
                                                       const BatchItem* __restrict__ items, char* __restrict__ buf) {
  const BatchItem it = items[blockIdx.x];
  const int slot = it.slot, tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const BlobLayout L = blob_layout_hd(M.TW, G, it.n_tiles, it.has_codes != 0);
  char* b = buf + it.off;
  const unsigned int* tab = M.table + (size_t)slot * M.TT;
  unsigned int* tidx = reinterpret_cast<unsigned int*>(b + L.tidx);
  if (tid == 0) {
    *reinterpret_cast<BlobHeader*>(b) = BlobHeader{kBlobMagic, it.n_tiles, it.has_codes ? 1u : 0u, nocc[slot], fstate[slot], (uint32_t)xsize, (uint32_t)M.TT};
    double* bs = reinterpret_cast<double*>(b + L.state);
    for (int q = 0; q < 3; ++q) { bs[q] = pose[(size_t)slot * 3 + q]; bs[3 + q] = prev[(size_t)slot * 3 + q]; }
    bs[6] = weight[slot];
  }
  __shared__ int base, wcnt[4];
  if (tid == 0) base = 0;
  __syncthreads();
  for (int t0 = 0; t0 < M.TT; t0 += 256) {
    const int t = t0 + tid;
    const bool f = t < M.TT && tab[t] != 0u;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wcnt[w];
    if (f) tidx[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)t;
    __syncthreads();
    if (tid == 0) base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  __threadfence_block();  // tidx was written by this workgroup: visible to all of it after the fence + the barrier above
  __syncthreads();
  double2* dst = reinterpret_cast<double2*>(b + L.tiles);
  for (size_t i = tid; i < (size_t)it.n_tiles * (kTileCells / 2); i += 256) {
    const unsigned int id = tab[tidx[i / (kTileCells / 2)]];
    dst[i] = reinterpret_cast<const double2*>(P.lo + (size_t)id * kTileCells)[i % (kTileCells / 2)];
  }
  unsigned int* dbm = reinterpret_cast<unsigned int*>(b + L.tile_bm);
  for (size_t i = tid; i < (size_t)it.n_tiles * kTS; i += 256) dbm[i] = P.bm[(size_t)tab[tidx[i / kTS]] * kTS + (i % kTS)];
  int* dtr = reinterpret_cast<int*>(b + L.trow);
  for (int r = tid; r < M.TW; r += 256) dtr[r] = trow[(size_t)slot * M.TW + r];
  if (it.has_codes) {
    uint16_t* dc = reinterpret_cast<uint16_t*>(b + L.codes);
    const uint16_t* sc = codes + (size_t)slot * G;
    for (size_t i = tid; i < G; i += 256) dc[i] = sc[i];
  }
}
__global__ __launch_bounds__(256) void rbpf_blob_headers(const BatchItem* __restrict__ items, const char* __restrict__ buf, BlobHeader* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = *reinterpret_cast<const BlobHeader*>(buf + items[i].off);
}
__global__ __launch_bounds__(256) void rbpf_release_slots(TilePool P, MapT M, const BatchItem* __restrict__ items) {
  const int p = items[blockIdx.x].slot;
  unsigned int* tab = M.table + (size_t)p * M.TT;
  unsigned int* shed = M.shed + (size_t)p * M.TT;
  for (int t = threadIdx.x; t < M.TT; t += blockDim.x) {
    const unsigned int id = tab[t], sh = shed[t];
    if (id && atomicSub(&P.ref[id], 1) == 1) tile_push(P, id);
    if (sh && atomicSub(&P.ref[sh], 1) == 1) tile_push(P, sh);
    tab[t] = 0u; shed[t] = 0u;
  }
}
// one workgroup per imported particle (its slot was released by the launch before): ONE pop for all its tiles
__global__ __launch_bounds__(256) void rbpf_unpack_batch(TilePool P, MapT M, double* __restrict__ pose, double* __restrict__ prev,
                                                         double* __restrict__ weight, int* __restrict__ trow, int* __restrict__ nocc,
                                                         int* __restrict__ fstate, uint16_t* __restrict__ codes, size_t G,
                                                         const BatchItem* __restrict__ items, const char* __restrict__ buf, int* __restrict__ err) {
  const BatchItem it = items[blockIdx.x];
  const int slot = it.slot, tid = threadIdx.x;
  const char* b = buf + it.off;
  const BlobHeader hd = *reinterpret_cast<const BlobHeader*>(b);
  const BlobLayout L = blob_layout_hd(M.TW, G, hd.n_tiles, hd.has_codes != 0);
  __shared__ unsigned long long sbase;
  if (tid == 0) sbase = hd.n_tiles ? tile_pop_n(P, hd.n_tiles) : 0ull;
  __syncthreads();
  const unsigned long long pos = sbase;
  if (pos == ~0ull) { if (tid == 0) atomicOr(&err[3], 8); return; }  // pool exhausted: the slot keeps the empty map
  const unsigned int* tidx = reinterpret_cast<const unsigned int*>(b + L.tidx);
  const double2* src = reinterpret_cast<const double2*>(b + L.tiles);
  for (size_t i = tid; i < (size_t)hd.n_tiles * (kTileCells / 2); i += 256)
    reinterpret_cast<double2*>(P.lo + (size_t)tile_at(P, pos + i / (kTileCells / 2)) * kTileCells)[i % (kTileCells / 2)] = src[i];
  const unsigned int* sbm = reinterpret_cast<const unsigned int*>(b + L.tile_bm);
  for (size_t i = tid; i < (size_t)hd.n_tiles * kTS; i += 256) P.bm[(size_t)tile_at(P, pos + i / kTS) * kTS + (i % kTS)] = sbm[i];
  for (unsigned int j = tid; j < hd.n_tiles; j += 256) {
    const unsigned int id = tile_at(P, pos + j);
    P.ref[id] = 1;
    M.table[(size_t)slot * M.TT + tidx[j]] = id;
  }
  const int* str = reinterpret_cast<const int*>(b + L.trow);
  for (int r = tid; r < M.TW; r += 256) trow[(size_t)slot * M.TW + r] = str[r];
  if (tid == 0) {
    const double* bs = reinterpret_cast<const double*>(b + L.state);
    for (int q = 0; q < 3; ++q) { pose[(size_t)slot * 3 + q] = bs[q]; prev[(size_t)slot * 3 + q] = bs[3 + q]; }
    weight[slot] = bs[6];
    nocc[slot] = hd.nocc;
    fstate[slot] = hd.has_codes ? 2 : 0;
  }
  if (hd.has_codes) {
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(b + L.codes);
    uint16_t* dc = codes + (size_t)slot * G;
    for (size_t i = tid; i < G; i += 256) dc[i] = sc[i];
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

}  // namespace

// =================================================================================================
// Handle + C-ABI
// =================================================================================================
struct tbnav_rbpf {
  tbnav_rbpf_params p;
  int device = 0, N = 0, k = 0, xsize = 0, ysize = 0, words = 0, radius = 0, edt_cols = 64;
  size_t G = 0;
  double l_prior = 0, l_occ = 0, l_free = 0, cut_occ = 0, max_occ_dist = 10.0;
  // particle state: [N][7] = pose(3), prev_pose(3), weight — double-buffered with the maps
  double* d_state[2] = {nullptr, nullptr};
  // log-odds: tiled, copy-on-write (see TilePool).  The tables are double-buffered with the rest of the particle state.
  TilePool pool{};
  unsigned int* d_table[2] = {nullptr, nullptr};  // [N][TT]
  unsigned int* d_shed = nullptr;                 // [N][TT]
  int TW = 0, TT = 0;
  double* d_dense = nullptr;   // [G] staging of one particle's dense log-odds (get/set_log_odds), allocated on first use
  double* d_cs = nullptr;      // [N] prefix scratch of the normalise kernel (N > kNormChunk)
  unsigned int* d_tile_scratch = nullptr;  // [TT] tile ids of a particle being exported
  // sharded filter: normalise / select over the all-gathered weights (tbnav_rbpf_resample_global_dev)
  double* d_gw = nullptr; double* d_gcs = nullptr; int* d_gparent = nullptr; double* d_gz = nullptr; size_t g_cap = 0;
  unsigned long long* d_touched = nullptr;  // [2] measurement hook: cell updates / distinct cells written (tbnav_rbpf_scan_counts)
  // rbpf_raycast_box's LDS array sized by what the particles' boxes needed in the last scans (device feedback, see the kernel)
  int* d_box_need = nullptr;      // [3] words of LDS array the largest box of a launch needed; the slots take turns
  int* h_box_need = nullptr;      // mapped pinned: the last complete launch's maximum
  int* d_box_need_host = nullptr; // device view of h_box_need
  unsigned int rc_launches = 0;   // box-counter launches so far (which slot accumulates)
  int raycast_adapt = 1;          // TBNAV_RBPF_OPT_RAYCAST_ADAPT: 0 = size the array for the worst case of the scan's longest beam
  bool count_touched = false;
  // stored distance field, u16 [N][G] x 2: allocated on first need (injection, materialisation, the stored-field
  // modes); the default query mode never touches it.  NULL until then.
  uint16_t* d_code[2] = {nullptr, nullptr};
  int* d_nocc[2] = {nullptr, nullptr};
  int cur = 0;
  int* d_trow[2] = {nullptr, nullptr};                   // [N][TW] occupied cells per tile row, kept current by the raycast kernel (the bits themselves live in the tiles)
  unsigned long long* d_bm_dense = nullptr;              // [N][xsize][words] dense rows for the exact-transform kernels, rebuilt from the tiles on demand
  int* d_rc_dense = nullptr;                             // [N][xsize]        (allocated with the stored field)
  double2* d_beams = nullptr;  // capacity max_beams
  int max_beams = 0;
  double* d_normals = nullptr;
  size_t normals_cap = 0;
  const double* last_normals = nullptr;  // the normals the last scan used (d_normals, or an entry of the batch ring)
  size_t last_z_index = 0;               // where in them its resampling offset sits (N * stride)
  // tbnav_rbpf_slam_batch draws the noise of a few scans ahead in one launch: normals and beam tables of ring_scans scans
  double* d_norm_ring = nullptr; size_t norm_ring_stride = 0;
  double2* d_beam_ring = nullptr; double2* h_beam_ring = nullptr; size_t beam_ring_stride = 0;
  int ring_scans = 0;
  int* d_parent = nullptr;     // [2][N]: the parent of every slot | how many slots chose each particle
  ExportCuts cuts{};           // host-derived (glibc) log-odds break points of the int8 map export
  int* d_best = nullptr;       // arg-max particle index
  double* d_best_pose = nullptr;
  int8_t* d_export = nullptr;  // [G]
  bool sm_on = false;          // N1 option: per-particle scan matching before sampling (tbnav_rbpf_set_scan_matching)
  ScanMatchC sm{0.05, 0.05, 5, 64};
  double* d_center = nullptr;  // [N][3] matched poses of the last call
  // scratch of the batched export / import (tbnav_rbpf_export_batch_dev ...): grown on demand
  int* d_bslots = nullptr; int2* d_bcount = nullptr; BatchItem* d_bitems = nullptr; BlobHeader* d_bhdr = nullptr; size_t batch_cap = 0;
  std::vector<int2> batch_counts;  // tiles / field state of the slots counted last
  double* d_mixlut = nullptr;  // [kMixLut] mixture term per distance code (constants of the handle: tabulated once at create)
  double* d_score = nullptr;   // [N]
  bool timing = false;         // record HIP events round the kernels (tbnav_rbpf_set_timing): each costs device time, so off by default
  int tile_cap = 0;            // cells of the raycast LDS tile (0 = use the beam-ordered kernel)
  int raycast_threads = 0;     // block size of the tile raycast: 0 = 1024 (TBNAV_RBPF_OPT_RAYCAST_THREADS)
  std::vector<double2> beam_cs;  // (cos, sin) of every beam's angle in the sensor frame, kept between scans
  std::vector<double2> beams_tmp;
  int raycast_band_rows = 0;   // > 0: cap the LDS array of rbpf_raycast_box at about this many box rows (TBNAV_RBPF_OPT_RAYCAST_BAND_ROWS, tests)
  int lk_raycast = -1, lk_raycast_grid = 0, lk_propose = 0;  // the instantiations the last launches were (tbnav_rbpf_last_kernel_names): raycast threads (0 = beam-ordered), its workgroups, propose threads
  int raycast_form = 0;        // 0 = box counters (rbpf_raycast_box), 1 = the beam-ordered kernel (rbpf_raycast) (TBNAV_RBPF_OPT_RAYCAST_FORM)
  double* d_sens = nullptr;    // [N][4] sensor transform (X, Y, sin, cos) of each particle's new pose, left by the proposal kernel
  uint64_t seed = 0x5EEDull, scan_index = 0;  // device noise source (normals == NULL)
  uint64_t rng_first = 0, rng_n_global = 0;   // sharded filters: this handle's particles are [rng_first, rng_first + N) of rng_n_global (0 = unsharded)
  // sharded filter inside the library (tbnav_rbpf_attach_comm / tbnav_rbpf_group_*): the weights' all-gather and the global
  // normalise / select run on a SECOND stream beside the local map update
  tbnav_comm* comm = nullptr;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_w = nullptr, ev_g = nullptr;   // "the proposal kernel has left the weights" / "the global normalise is through"
  double* d_gw_raw = nullptr;                  // [n_global] all-gathered raw weights
  char* d_sendbuf = nullptr; char* d_recvbuf = nullptr; size_t send_cap = 0, recv_cap = 0;   // particle blobs of a cross-rank resample
  unsigned long long* d_sizes = nullptr;       // [n_local + n_global] blob size of every particle this rank sends | of every particle
  int* d_status = nullptr;                     // [1 + nranks] this rank's status | everybody's
  bool full_edt = false;       // distance-field mode 0 (TBNAV_RBPF_DF=full): whole-map transform after every map update
  int df_mode = 2;             // 0 full, 1 windowed refresh before the update (TBNAV_RBPF_DF=window), 2 exact query at lookup (default)
  int* d_fstate = nullptr;     // [N] distance-field state: 0 stale, 1 window fresh, 2 whole field fresh / injected
  int* d_fstate_alt = nullptr; // [N] the other buffer of the resample gather
  // reference distance-field mode (tbnav_rbpf_set_option DF_MODE = REFERENCE): host-side brushfire state + the
  // device log of occupied-set changes it is fed from
  bool ref_field = false;
  tbnav::RefField* ref = nullptr;
  std::vector<tbnav::RefField::StatePtr> ref_on_dev;  // [N] the reference-field state whose field slot p of d_code holds (empty: unknown); holding the
                                                      //     pointer keeps the state alive, so an address is never re-used while it is compared
  int* d_log_pack = nullptr; unsigned long long* d_log_off = nullptr; size_t log_pack_cap = 0, log_off_cap = 0;  // the scan's logs, packed
  int* d_code_src = nullptr;              // [N] slot to copy the field from (rbpf_copy_codes)
  int host_threads = 1;        // host threads of the reference-field mode's per-particle work (TBNAV_RBPF_OPT_HOST_THREADS; set at create)
  int* d_log_ev = nullptr;     // [N][log_cap]
  int* d_log_cnt = nullptr;    // [N]
  int log_cap = 0;
  uint64_t scans_done = 0;
  int* d_skip = nullptr;       // [N] scratch: 1 = no refresh needed this call
  int4* d_win = nullptr;       // [N] refreshed window (i0, i1, j0, j1), inclusive
  int* d_tier = nullptr;       // [N] which distance-field kernel handles the particle this scan
  int* d_err = nullptr;
  NormOut* d_norm = nullptr;
  // pinned host staging: the scan going in, the error flags and the normalisation result coming out (pageable
  // buffers make every one of those small copies a blocking, staged transfer)
  double2* h_beams = nullptr;  // [kScanSlots] x capacity max_beams
  // error flags and normalisation result live in mapped pinned host memory: the kernels write them over the
  // fabric (a handful of bytes per scan) and the host reads them after the stream sync — no copy kernels, no memset
  // kScanSlots of each: a scan in flight owns slot (scan number % kScanSlots) — tbnav_rbpf_slam_batch keeps two scans in the
  // stream; every other entry point uses slot 0
  int* h_err = nullptr;        // [kScanSlots][4] host view; d_err is the device view of the same bytes
  NormOut* h_norm = nullptr;   // [kScanSlots] host view of d_norm
  int* d_gate = nullptr;       // [kScanSlots] device memory: 1 = that scan resamples (NormArgs::gate)
  unsigned int* h_seq = nullptr;  // [kScanSlots] mapped: the scan number whose normalisation result the slot holds (NormArgs::seq)
  unsigned int* d_seq = nullptr;  // device view of h_seq
  bool fstate_dirty = true;    // some d_fstate entry may be non-zero
  int batch_pipeline = 1;      // tbnav_rbpf_slam_batch keeps two scans in the stream (TBNAV_RBPF_OPT_BATCH_PIPELINE)
  double* d_trace = nullptr;   // sampled, p_scan, p_pose, mu, sigma, eta, new_pose, weight_raw
  Trace tr{};
  hipStream_t stream = nullptr;
  hipEvent_t ev[TBNAV_RBPF_NKERNELS + 2] = {};  // 0..5 bracket kernels 0..4; 6,7 bracket the gather
  float last_ms[TBNAV_RBPF_NKERNELS] = {0};
  std::vector<int> h_parent;
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true; }
  ~DeviceGuard() { if (ok && prev >= 0) (void)hipSetDevice(prev); }
};

// state layout helpers: the kernels take pose / prev_pose / weight pointers with [N][3] / [N] strides,
// so the 7-double record is split into three arrays inside one allocation.
struct StatePtrs { double *pose, *prev, *weight; };
StatePtrs state_ptrs(double* base, int N) { return {base, base + (size_t)3 * N, base + (size_t)6 * N}; }
MapT map_of(const tbnav_rbpf* h) { return MapT{h->d_table[h->cur], h->d_shed, h->TW, h->TT}; }

// The stored u16 distance field exists only once something needs it: 2 * N * G * 2 bytes (double-buffered like the
// rest of the particle state).  Refused beyond 32 GB — BASELINE configs[4]-sized handles run in query mode only.
int ensure_codes(tbnav_rbpf* h) {
  if (h->d_code[0]) return TBNAV_OK;
  const size_t bytes = sizeof(uint16_t) * h->G * (size_t)h->N;
  if (2 * bytes > ((size_t)32 << 30) || h->N > 65535) return TBNAV_ERR_UNSUPPORTED;
  for (int b = 0; b < 2; ++b) {
    TBNAV_HIP(hipMalloc((void**)&h->d_code[b], bytes));
    TBNAV_HIP(hipMemset(h->d_code[b], 0xFF, bytes));  // occ_dist = max_occ_dist_ (grid_mapper.cpp:49,58)
  }
  TBNAV_HIP(hipMalloc((void**)&h->d_bm_dense, sizeof(unsigned long long) * (size_t)h->N * h->xsize * h->words));
  TBNAV_HIP(hipMalloc((void**)&h->d_rc_dense, sizeof(int) * (size_t)h->N * h->xsize));
  return TBNAV_OK;
}

double logodds_to_prob(double l) { return 1 - (1 / (1 + std::exp(l))); }  // grid_mapper.hpp:27-30 (glibc on the host)

// smallest l with prob(l) >= p_occ, found by bisection on the host (prob is monotone in l)
double find_occ_cut(double l_occ_nominal, double p_occ) {
  double lo = l_occ_nominal - 1.0, hi = l_occ_nominal + 1.0;  // prob(lo) < p_occ <= prob(hi)
  for (int it = 0; it < 200; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (logodds_to_prob(mid) >= p_occ) hi = mid; else lo = mid;
  }
  return hi;
}

// Break points of the exported map value as a function of the log-odds, found with the HOST libm by bisection
// (the exported value is monotone in l apart from the prob == 0.5 plateau, which maps to -1).
int export_value_host(double l) {  // GridMapper::gridMap after updateCellState, evaluated as the reference does
  const double prob = logodds_to_prob(l);
  if (prob == 0.5) return -1;
  if (prob >= 0.90) return 100;
  if (prob <= 0.35) return 0;
  return (int)(int8_t)(prob * 100);
}
double bisect_first(double lo, double hi, bool (*pred)(double, int), int arg) {  // pred(lo) false, pred(hi) true, monotone
  for (int it = 0; it < 300; ++it) {
    const double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if (pred(mid, arg)) hi = mid; else lo = mid;
  }
  return hi;
}
ExportCuts derive_export_cuts(double cut_occ) {
  ExportCuts c{};
  c.occ_cut = cut_occ;
  // largest l with prob <= 0.35: the predecessor of the first l with prob > 0.35
  const double first_above = bisect_first(-3.0, 0.0, [](double l, int) { return logodds_to_prob(l) > 0.35; }, 0);
  c.free_cut = std::nextafter(first_above, -1.0e9);
  c.half_lo = bisect_first(-1.0, 1.0, [](double l, int) { return logodds_to_prob(l) >= 0.5; }, 0);
  const double first_gt = bisect_first(-1.0, 1.0, [](double l, int) { return logodds_to_prob(l) > 0.5; }, 0);
  c.half_hi = std::nextafter(first_gt, -1.0e9);
  c.n_steps = 0;
  for (int k = 36; k <= 89; ++k)  // smallest l (outside the 0.5 plateau) whose exported value is >= k
    c.step[c.n_steps++] = bisect_first(c.free_cut, cut_occ, [](double l, int kk) { const int v = export_value_host(l); return v == 100 || (v >= kk); }, k);
  return c;
}

size_t edt_lds_bytes(int xs, int words, int C) { return (size_t)xs * words * 8 + (size_t)xs * C * 5; }

// the part of ScanC the beam mixture term needs (also what the handle's table of it is built from at create)
bool mixture_consts(const tbnav_rbpf* h, ScanC& c) {
  const tbnav_rbpf_params& P = h->p;
  c.g = GridC{P.xmin, P.xmax, P.ymin, P.ymax, P.resolution, h->xsize, h->ysize, h->words, h->max_occ_dist, 1.0 / P.resolution};
  c.z_hit = P.z_hit;
  c.var_hit = P.sigma_hit * P.sigma_hit;                       // grid_mapper.cpp:77
  if (almost_equal(c.var_hit, 0.0)) return false;
  c.sqrt_inv_hit = 1.0 / std::sqrt(2.0 * kPI * c.var_hit);    // pdfNormal, grid_mapper.cpp:25
  c.rand_term = P.z_rand / P.z_max;                            // grid_mapper.cpp:121
  return true;
}

int build_scan_consts(tbnav_rbpf* h, ScanC& c, const float* scan, int n_beams, const double u[3],
                      const double cur_odom[3], const double prev_odom[3], int icp_ok, const double T_icp[3],
                      std::vector<double2>& beams) {
  const tbnav_rbpf_params& P = h->p;
  c.N = h->N; c.k = h->k; c.icp_ok = icp_ok ? 1 : 0;
  for (int q = 0; q < 3; ++q) { c.Trs[q] = P.Trs[q]; c.Ld[q] = std::sqrt(P.sample_range[q]); c.Lm[q] = std::sqrt(P.motion_noise[q]);
                                c.Ticp[q] = T_icp[q]; c.u[q] = u[q]; }
  if (!mixture_consts(h, c)) return TBNAV_ERR_PDF_VARIANCE;
  c.scan_min = P.scan_likelihood_min; c.scan_max = P.scan_likelihood_max;
  c.pose_min = P.pose_likelihood_min; c.pose_max = P.pose_likelihood_max;
  c.a1 = P.srr; c.a2 = P.srt; c.a3 = P.str_; c.a4 = P.stt;
  // odometry deltas (particle_filter.cpp:393-403), identical for every particle and sample
  c.rot1 = std::atan2(cur_odom[2] - prev_odom[2], cur_odom[1] - prev_odom[1]) - prev_odom[0];
  const double dxo = cur_odom[1] - prev_odom[1], dyo = cur_odom[2] - prev_odom[2];
  c.trans = std::sqrt(dxo * dxo + dyo * dyo);
  c.rot2 = normalize_angle_PI(normalize_angle_PI(cur_odom[0]) - normalize_angle_PI(prev_odom[0]) - c.rot1);
  c.d_free = h->l_free - h->l_prior;
  c.d_occ = h->l_occ - h->l_prior;
  c.cut_occ = h->cut_occ;
  c.stride_normals = icp_ok ? 3 * h->k + 3 : 3;
  c.p0 = 0;
  // valid beams in the sensor frame, sensor_model.cpp:73-108 (float limits, double angle accumulation)
  // (the angle of beam i does not depend on the scan: its cosine and sine — glibc's, in the reference's accumulation
  //  order — are kept from one call to the next; 2 x 360 libm calls were a tenth of the host's time per scan)
  if ((int)h->beam_cs.size() != n_beams) {
    h->beam_cs.resize(n_beams);
    double beam_angle = P.beam_min;
    for (int i = 0; i < n_beams; ++i) {
      h->beam_cs[i] = double2{std::cos(beam_angle), std::sin(beam_angle)};
      beam_angle += P.beam_delta;
      if (P.beam_max < 0.0 && beam_angle <= P.beam_max) beam_angle = P.beam_min;
      else if (P.beam_max >= 0.0 && beam_angle >= P.beam_max) beam_angle = P.beam_min;
    }
  }
  beams.clear();
  c.rmax = 0.0;
  for (int i = 0; i < n_beams; ++i) {
    const double range = scan[i];
    if (range >= P.range_min && range < P.range_max) {
      beams.push_back(double2{range * h->beam_cs[i].x, range * h->beam_cs[i].y});
      c.rmax = std::max(c.rmax, range);
    }
  }
  c.Bv = (int)beams.size();
  return TBNAV_OK;
}

int status_from_err(const int err[4]) {
  if (err[0]) return TBNAV_ERR_OUT_OF_WORLD;
  if (err[2]) return TBNAV_ERR_PDF_VARIANCE;
  if (err[1]) return TBNAV_ERR_ETA_ZERO;
  if (err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;  // no free log-odds tile left (the scan of that particle was not applied)
  if (err[3] & 4) return TBNAV_ERR_UNSUPPORTED;  // a likelihood lookup left the particle's refreshed window (cannot happen: see rbpf_window)
  if (err[3]) return TBNAV_ERR_BRESENHAM;
  return TBNAV_OK;
}

// The three tiers of the exact distance transform for particles [p0, p0 + count): windowed (tiles_x = the
// tiles a window can span) or whole-map (tiles_x = every tile; the caller has set win/skip accordingly).
int run_distance_field(tbnav_rbpf* h, const GridC& g, int p0, int count, int tiles64) {
  hipStream_t st = h->stream;
  // dense bitmap rows + row counts of these particles, from their tiles
  hipLaunchKernelGGL(rbpf_densify, dim3((h->xsize + 3) / 4, count), dim3(256), 0, st, g, p0, h->pool, map_of(h), h->d_trow[h->cur],
                     h->d_bm_dense, h->d_rc_dense);
  TBNAV_HIP(hipGetLastError());
  // tier 0: <= kEdtRowsA non-empty rows, tier 1: <= kEdtRowsB, tier 2: the general kernel (decided on the device)
  TBNAV_HIP(hipMemsetAsync(h->d_tier + p0, 0, sizeof(int) * count, st));
  const EdtJob job{h->d_win, h->d_skip, p0};
  const dim3 gridc(tiles64, count);
  hipLaunchKernelGGL(rbpf_edt_compact<kEdtRowsA>, gridc, dim3(kWave), edt_compact_lds(kEdtRowsA), st, g, h->radius,
                     h->d_bm_dense, h->d_rc_dense, h->d_code[h->cur], h->d_tier, 0, job);
  TBNAV_HIP(hipGetLastError());
  hipLaunchKernelGGL(rbpf_edt_compact<kEdtRowsB>, gridc, dim3(kWave), edt_compact_lds(kEdtRowsB), st, g, h->radius,
                     h->d_bm_dense, h->d_rc_dense, h->d_code[h->cur], h->d_tier, 1, job);
  TBNAV_HIP(hipGetLastError());
  const int C = h->edt_cols;
  const size_t lds = edt_lds_bytes(h->xsize, h->words, C);
  const dim3 grid(tiles64 * (kWave / C), count);
  if (C == 64) hipLaunchKernelGGL(rbpf_edt<64>, grid, dim3(64), lds, st, g, h->radius, h->d_bm_dense, h->d_code[h->cur], h->d_tier, 2, job);
  else hipLaunchKernelGGL(rbpf_edt<32>, grid, dim3(32), lds, st, g, h->radius, h->d_bm_dense, h->d_code[h->cur], h->d_tier, 2, job);
  TBNAV_HIP(hipGetLastError());
  return TBNAV_OK;
}

GridC grid_of(const tbnav_rbpf* h) {
  return GridC{h->p.xmin, h->p.xmax, h->p.ymin, h->p.ymax, h->p.resolution, h->xsize, h->ysize, h->words, h->max_occ_dist, 1.0 / h->p.resolution};
}

// Whole-field refresh of ONE particle, on demand (state 2 afterwards).
int ensure_full_field(tbnav_rbpf* h, int particle) {
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  hipStream_t st = h->stream;
  int stt = 0;
  TBNAV_HIP(hipStreamSynchronize(st));
  TBNAV_HIP(hipMemcpy(&stt, h->d_fstate + particle, sizeof(int), hipMemcpyDeviceToHost));
  if (stt == 2) return TBNAV_OK;
  const int zero = 0, two = 2;
  if (h->edt_cols == 0) {
    const GridC g = grid_of(h);
    hipLaunchKernelGGL(rbpf_field_by_query, dim3((unsigned)((h->G + 255) / 256)), dim3(256), 0, st, g, h->radius, particle,
                       h->pool, map_of(h), h->d_trow[h->cur], h->d_code[h->cur]);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipStreamSynchronize(st));
    TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
    h->fstate_dirty = true;
    return TBNAV_OK;
  }
  const int4 full = make_int4(0, h->xsize - 1, 0, h->ysize - 1);
  TBNAV_HIP(hipMemcpy(h->d_win + particle, &full, sizeof full, hipMemcpyHostToDevice));
  TBNAV_HIP(hipMemcpy(h->d_skip + particle, &zero, sizeof zero, hipMemcpyHostToDevice));
  int rc = run_distance_field(h, grid_of(h), particle, 1, (h->ysize + kWave - 1) / kWave);
  if (rc != TBNAV_OK) return rc;
  TBNAV_HIP(hipStreamSynchronize(st));
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
  h->fstate_dirty = true;
  return TBNAV_OK;
}

// lowVarianceResampling's copies on the device: d_parent holds the parent of every slot and, behind them, how many slots
// chose each particle.  One launch (rbpf_resample_apply): tables and reference counts, and state / counts / field state
// into the alternate buffers (the occupancy bits travel with the tiles: nothing of map size is copied).
int resample_on_device(tbnav_rbpf* h) {
  const int N = h->N, nxt = 1 - h->cur;
  hipStream_t st = h->stream;
  const size_t n = (size_t)N * h->TT;
  const int blocks = (int)std::min<size_t>((n + kResampleThreads - 1) / kResampleThreads, 8192);
  const size_t work = h->d_code[0] ? h->G / 4 : (size_t)0;
  const int chunks = (int)std::min<size_t>(std::max<size_t>(work / 2048, 1), 64);
  const GatherArgs ga{h->G, h->TW, h->d_state[h->cur], h->d_state[nxt], h->d_trow[h->cur], h->d_trow[nxt], h->d_nocc[h->cur], h->d_nocc[nxt],
                      h->d_fstate, h->d_fstate_alt, h->d_code[h->cur], h->d_code[nxt], h->df_mode != 2 ? 1 : 0};
  hipLaunchKernelGGL(rbpf_resample_apply, dim3(blocks + N * chunks), dim3(kResampleThreads), 0, st, N, h->TT, h->d_parent, h->d_parent + N,
                     h->d_table[h->cur], h->d_table[nxt], h->d_shed, h->pool, blocks, chunks, ga);
  TBNAV_HIP(hipGetLastError());
  std::swap(h->d_fstate, h->d_fstate_alt);
  h->cur = nxt;
  return TBNAV_OK;
}

// ---- reference distance-field mode (ref_field.hpp) ------------------------------------------------------------
// Before the raycast: a log big enough for every cell update of the scan (a cell can enter and leave the occupied set
// more than once in one scan).
int ref_field_prepare_log(tbnav_rbpf* h, int Bv, OccLog& log) {
  const double reach = (double)h->p.range_max + std::hypot(h->p.Trs[1], h->p.Trs[2]);
  const long per_ray = (long)std::ceil(reach / h->p.resolution) + 4;
  const long cap = (long)std::max(Bv, 1) * per_ray;
  if ((size_t)cap * h->N * sizeof(int) > ((size_t)1 << 30)) return TBNAV_ERR_UNSUPPORTED;
  if (cap > h->log_cap) {
    (void)hipFree(h->d_log_ev); h->d_log_ev = nullptr; h->log_cap = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_log_ev, sizeof(int) * (size_t)cap * h->N));
    h->log_cap = (int)cap;
  }
  if (!h->d_log_cnt) TBNAV_HIP(hipMalloc((void**)&h->d_log_cnt, sizeof(int) * h->N));
  TBNAV_HIP(hipMemsetAsync(h->d_log_cnt, 0, sizeof(int) * h->N, h->stream));
  log = OccLog{h->d_log_ev, h->d_log_cnt, h->log_cap};
  return TBNAV_OK;
}
// After the scan (and its resample, if one fired): replay the logged set changes, run the reference's brushfire for
// every particle as it was BEFORE the resample (the reference integrates the scan in the particle loop and resamples
// afterwards, particle_filter.cpp:158-249), copy like the resample did, and make the result the authoritative field.
// the particles' logged sequences packed back to back (one copy to the host instead of one per particle)
__global__ __launch_bounds__(256) void rbpf_pack_logs(const int* __restrict__ ev, int log_cap, int p_first, const unsigned long long* __restrict__ off,
                                                      int* __restrict__ out) {
  const int i = blockIdx.x;
  const unsigned long long o = off[i], n = off[i + 1] - o;
  const int* src = ev + (size_t)(p_first + i) * log_cap;
  for (unsigned long long q = threadIdx.x; q < n; q += blockDim.x) out[o + q] = src[q];
}
// slot p takes the field slot src[p] holds (src[p] == p: keep) — the particles that share a state with one already on the device
__global__ __launch_bounds__(256) void rbpf_copy_codes(uint16_t* __restrict__ code, size_t G, const int* __restrict__ src) {
  const int p = blockIdx.y, q = src[p];
  if (q == p) return;
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  if ((G & 7) == 0) {  // (every slot starts on a 16-byte boundary)
    const uint4* s = reinterpret_cast<const uint4*>(code + (size_t)q * G);
    uint4* d = reinterpret_cast<uint4*>(code + (size_t)p * G);
    for (size_t i = i0; i < G / 8; i += step) d[i] = s[i];
  } else {
    for (size_t i = i0; i < G; i += step) code[(size_t)p * G + i] = code[(size_t)q * G + i];
  }
}
int ref_field_after_scan(tbnav_rbpf* h, bool resampled, int p_first = 0, int p_count = -1) {
  const int N = h->N;
  if (p_count < 0) p_count = N;
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  hipStream_t st = h->stream;
  TBNAV_HIP(hipStreamSynchronize(st));
  std::vector<int> cnt(N);
  TBNAV_HIP(hipMemcpy(cnt.data(), h->d_log_cnt, sizeof(int) * N, hipMemcpyDeviceToHost));
  // the logs: packed on the device, ONE copy (a few thousand events per particle; one small copy each was 10-20 ms per 1000)
  std::vector<size_t> off((size_t)p_count + 1, 0);
  for (int i = 0; i < p_count; ++i) {
    if (cnt[p_first + i] > h->log_cap) return TBNAV_ERR_UNSUPPORTED;  // cannot happen: the log holds every cell update
    off[i + 1] = off[i] + (size_t)cnt[p_first + i];
  }
  const size_t total = off[p_count];
  std::vector<int> all(total ? total : 1);
  if (total) {
    if (total > h->log_pack_cap || (size_t)p_count + 1 > h->log_off_cap) {
      (void)hipFree(h->d_log_pack); (void)hipFree(h->d_log_off); h->d_log_pack = nullptr; h->d_log_off = nullptr; h->log_pack_cap = h->log_off_cap = 0;
      const size_t cap = total + total / 2, ocap = (size_t)N + 1;
      TBNAV_HIP(hipMalloc((void**)&h->d_log_pack, sizeof(int) * cap));
      TBNAV_HIP(hipMalloc((void**)&h->d_log_off, sizeof(unsigned long long) * ocap));
      h->log_pack_cap = cap; h->log_off_cap = ocap;
    }
    std::vector<unsigned long long> off64(off.begin(), off.end());
    TBNAV_HIP(hipMemcpy(h->d_log_off, off64.data(), sizeof(unsigned long long) * off64.size(), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rbpf_pack_logs, dim3(p_count), dim3(256), 0, st, h->d_log_ev, h->log_cap, p_first, h->d_log_off, h->d_log_pack);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipMemcpyAsync(all.data(), h->d_log_pack, sizeof(int) * total, hipMemcpyDeviceToHost, st));
    TBNAV_HIP(hipStreamSynchronize(st));
  }
  // one replay + brushfire per distinct (state, sequence) — ref_field.hpp — side by side on the host's cores: inside one state the
  // order of every set and heap operation is the reference's
  h->ref->step(p_first, p_count, h->host_threads, all.data(), off.data());
  if (resampled) {
    h->h_parent.resize(N);
    TBNAV_HIP(hipMemcpy(h->h_parent.data(), h->d_parent, sizeof(int) * N, hipMemcpyDeviceToHost));
    h->ref->resample(h->h_parent.data());
  }
  h->ref_on_dev.resize(N);
  if (resampled) { p_first = 0; p_count = N; for (auto& q : h->ref_on_dev) q.reset(); }  // (the device's own gather moved the slots)
  // to the device: a state no slot holds yet is uploaded once; the other particles that share it copy it on the device
  std::unordered_map<const void*, int> holder;  // state -> a slot whose device field is that state's
  for (int p = 0; p < N; ++p) if (h->ref_on_dev[p] && h->ref_on_dev[p].get() == h->ref->state(p)) holder.emplace(h->ref_on_dev[p].get(), p);
  std::vector<int> src(N);
  bool any_copy = false;
  for (int p = 0; p < N; ++p) {
    src[p] = p;
    if (p < p_first || p >= p_first + p_count) continue;
    const void* s = (const void*)h->ref->state(p);
    if (h->ref_on_dev[p].get() == s) continue;
    auto it = holder.find(s);
    if (it == holder.end()) {
      TBNAV_HIP(hipMemcpyAsync(h->d_code[h->cur] + (size_t)p * h->G, h->ref->codes(p), sizeof(uint16_t) * h->G, hipMemcpyHostToDevice, st));
      holder.emplace(s, p);
    } else { src[p] = it->second; any_copy = true; }
    h->ref_on_dev[p] = h->ref->state_ptr(p);
  }
  if (any_copy) {
    if (!h->d_code_src) TBNAV_HIP(hipMalloc((void**)&h->d_code_src, sizeof(int) * N));
    TBNAV_HIP(hipMemcpyAsync(h->d_code_src, src.data(), sizeof(int) * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rbpf_copy_codes, dim3(64, N), dim3(256), 0, st, h->d_code[h->cur], h->G, h->d_code_src);
    TBNAV_HIP(hipGetLastError());
  }
  TBNAV_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_fstate + p_first), 2, p_count, st));
  TBNAV_HIP(hipStreamSynchronize(st));  // (src and the states' host buffers are read by the copies)
  h->fstate_dirty = true;
  return TBNAV_OK;
}

// GridMapper::integrateScan's map update (grid_mapper.cpp:140-178) for particles [c.p0, c.p0 + count) at their poses.
// sens: the sensor transforms the proposal kernel left for exactly these poses (NULL: the raycast derives them).
// nz (optional): the weights' normalise / select step to run with this update — inside the box-counter kernel's launch as
// workgroup 0 (no second stream, no event), behind the other map-update kernels as a launch of its own.
int launch_raycast(tbnav_rbpf* h, const ScanC& c, int count, const double* sens, const NormArgs* nz = nullptr, int* err = nullptr,
                   const double2* beams_dev = nullptr) {
  if (!err) err = h->d_err;
  if (!beams_dev) beams_dev = h->d_beams;
  const int* gp = nz ? nz->gate_prev : nullptr;
  hipStream_t st = h->stream;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  const int bvn = c.Bv > 0 ? c.Bv : 1;
  const MapT M = map_of(h);
  int nt = h->raycast_threads;
  const bool nt_auto = nt == 0;
  if (nt == 0) nt = 1024;
  // rbpf_raycast_box: one u32 per cell of the box, the box padded to whole groups of 8 cells along y
  long cap_win = 0;
  if (h->tile_cap > 0) {
    // every end point lies within `reach` of the robot's position: at most floor(2 reach / res) + 2 rows or columns (+1 spare);
    // along y the box is padded to whole pairs of cells
    const double reach = c.rmax + std::hypot(h->p.Trs[1], h->p.Trs[2]);
    const long side = (long)std::floor(2.0 * reach / h->p.resolution) + 3;
    cap_win = (side * ((side + 2) & ~1L) + 7) & ~7L;
    // ... but no more than lets TWO workgroups share a CU's 160 KB (the kernel works a larger box through in bands of rows;
    // at least one padded row must fit)
    const long cap_fit = ((78L * 1024 - (long)box_lds_bytes(0, (size_t)bvn) - 1536) / 4) & ~7L;
    if (cap_win > cap_fit) cap_win = std::max(cap_fit, (side + 9) & ~7L);
    // (test hook: at most about this many rows of the box per band, to drive the band loop on small maps)
    if (h->raycast_band_rows > 0) cap_win = std::min(cap_win, (h->raycast_band_rows * ((side + 2) & ~1L) + 7) & ~7L);
    // ... and no more than the particles' boxes needed lately (+ 1/8 + 512 words for what a scan's motion changes): the bound
    // above is the scan's longest beam in every direction from every pose, a room's box is a fraction of that — the array is
    // what keeps a CU at two workgroups.  A box that outgrows the guess costs its particle a second band, not correctness.
    const int need = (h->raycast_adapt && h->h_box_need) ? *reinterpret_cast<volatile int*>(h->h_box_need) : 0;
    if (need > 0) {
      const long want = ((long)need + need / 8 + 512 + 7) & ~7L;
      cap_win = std::min(cap_win, std::max(want, (side + 9) & ~7L));
    }
  }
  const size_t lds_win = box_lds_bytes((size_t)cap_win, (size_t)bvn);
  // workgroup size: 512 threads when THREE workgroups fit a CU's LDS (24 waves, 80 registers a lane), else 1024 (two, 32 waves).
  // Measured at cfg3, N = 1000 / 4000 (56 KB of LDS saved by the adaptive array): 1024 x 2: 55.8 / 191 us; 512 x 2: 54.4 / 199;
  // 512 x 3: 49.1 / 163
  if (nt_auto && cap_win > 0 && 3 * (lds_win + 1536) <= (size_t)kMaxLds) nt = 512;
  if (cap_win > 0 && !h->ref_field && c.Bv < 32768 - kWave && h->raycast_form == 0 && nt >= 512 && lds_win <= (size_t)kMaxLds - 4096) {
    // default: box counters (rbpf_raycast_box)
    unsigned long long* touched = h->count_touched ? h->d_touched : nullptr;
    const NormArgs na = nz ? *nz : NormArgs{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr};
    const int blocks = count + (nz ? 1 : 0);
    const size_t lds_launch = nz ? std::max(lds_win, sizeof(double) * 2 * kNormChunk) : lds_win;  // (workgroup 0's two arrays)
    const int need_slot = (int)(h->rc_launches++ % 3u);
    h->lk_raycast = nt == 512 ? 512 : 1024; h->lk_raycast_grid = blocks;
    if (nt == 512)
      hipLaunchKernelGGL((rbpf_raycast_box<512>), dim3(blocks), dim3(512), lds_launch, st, c, h->pool, M, beams_dev, sp.pose, sens,
                         h->d_trow[h->cur], h->d_nocc[h->cur], err, (int)cap_win, touched, na, h->d_box_need, h->d_box_need_host, need_slot);
    else
      hipLaunchKernelGGL((rbpf_raycast_box<1024>), dim3(blocks), dim3(1024), lds_launch, st, c, h->pool, M, beams_dev, sp.pose, sens,
                         h->d_trow[h->cur], h->d_nocc[h->cur], err, (int)cap_win, touched, na, h->d_box_need, h->d_box_need_host, need_slot);
    TBNAV_HIP(hipGetLastError());
    return TBNAV_OK;
  }
  {
    // beam-ordered kernel: scans the LDS tile cannot hold, and the reference distance-field mode (it logs the
    // occupied-set changes in the reference's order)
    OccLog log{nullptr, nullptr, 0};
    if (h->ref_field) {
      const int rc2 = ref_field_prepare_log(h, c.Bv, log);
      if (rc2 != TBNAV_OK) return rc2;
    }
    h->lk_raycast = 0; h->lk_raycast_grid = count;
    hipLaunchKernelGGL(rbpf_raycast, dim3(count), dim3(kWave), sizeof(int) * (2 * bvn + (h->TT + 31) / 32), st, c, h->pool, M, beams_dev,
                       sp.pose, h->d_trow[h->cur], h->d_nocc[h->cur], err, log, gp);
  }
  TBNAV_HIP(hipGetLastError());
  if (nz) {
    hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, st, nz->N, nz->zp, nz->weight, nz->weight_out, nz->cs, nz->parent, nz->out,
                       nz->gate, nz->gate_prev, nz->seq, nz->seq_val, nz->children);
    TBNAV_HIP(hipGetLastError());
  }
  return TBNAV_OK;
}

// the scan's valid beams into d_beams (shared by slam_impl and the one-particle entry points)
int upload_beams(tbnav_rbpf* h, const std::vector<double2>& beams, int n_beams, int Bv, bool stage_only = false, int slot = 0) {
  if (n_beams > h->max_beams) {
    TBNAV_HIP(hipStreamSynchronize(h->stream));  // (a scan still in flight reads the buffers about to go)
    (void)hipFree(h->d_beams);
    (void)hipHostFree(h->h_beams);
    h->d_beams = nullptr; h->h_beams = nullptr; h->max_beams = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_beams, sizeof(double2) * n_beams));
    TBNAV_HIP(hipHostMalloc((void**)&h->h_beams, sizeof(double2) * n_beams * kScanSlots, hipHostMallocDefault));
    h->max_beams = n_beams;
  }
  if (Bv) {
    double2* hb = h->h_beams + (size_t)slot * h->max_beams;
    std::memcpy(hb, beams.data(), sizeof(double2) * Bv);
    if (!stage_only) TBNAV_HIP(hipMemcpyAsync(h->d_beams, hb, sizeof(double2) * Bv, hipMemcpyHostToDevice, h->stream));
  }
  return TBNAV_OK;
}

// One scan = scan_enqueue (everything up to and including the map update, on the handle's stream) + scan_finish (wait,
// read the stats, run the resampling copies if the scan decided to resample).  `slot`: which of the kScanSlots result slots
// the scan owns.  gate_prev (device pointer or NULL): the resampling decision of the scan enqueued before this one, when the
// host has not seen it yet — the kernels of this scan do nothing if it is set (tbnav_rbpf_slam_batch).
struct ScanTicket { int slot = 0; int n_valid = 0; bool local_only = false; bool poll = false; unsigned int seq = 0; };
// A scan whose constants, beam table and noise are on the device already (tbnav_rbpf_slam_batch prepares a few scans at a time).
struct Prefetched { ScanC c; int rc = TBNAV_OK; const double2* d_beams = nullptr; const double* d_normals = nullptr; };
int scan_enqueue(tbnav_rbpf* h, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* normals,
                 tbnav_rbpf_stats* out, bool local_only, int slot, const int* gate_prev, ScanTicket& tk,
                 const Prefetched* pre = nullptr, hipEvent_t weights_ready = nullptr) {
  hipStream_t st = h->stream;
  int* const d_err = h->d_err + 4 * slot;
  int* const h_err = h->h_err + 4 * slot;
  ++h->scans_done;
  ScanC c;
  std::vector<double2>& beams = h->beams_tmp;  // (kept between calls: no allocation per scan)
  int rc;
  if (pre) { c = pre->c; rc = pre->rc; }
  else rc = build_scan_consts(h, c, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, beams);
  std::memset(out, 0, sizeof *out);
  if (rc != TBNAV_OK) { out->status = rc; return rc; }
  out->n_valid_beams = c.Bv;
  tk.slot = slot; tk.n_valid = c.Bv; tk.local_only = local_only;
  if (!pre) {
    rc = upload_beams(h, beams, n_beams, c.Bv, /*stage_only=*/normals == nullptr, slot);  // device noise: the noise kernel carries the beams over
    if (rc != TBNAV_OK) return rc;
  }
  const size_t n_norm = (size_t)h->N * c.stride_normals + 1;
  if (!pre && n_norm > h->normals_cap) {
    TBNAV_HIP(hipStreamSynchronize(st));  // (a scan still in flight reads the old buffer)
    (void)hipFree(h->d_normals);
    h->d_normals = nullptr;
    TBNAV_HIP(hipMalloc((void**)&h->d_normals, sizeof(double) * n_norm));
    h->normals_cap = n_norm;
  }
  if (pre) {
    // (drawn with the rest of its chunk)
  } else if (normals) {
    TBNAV_HIP(hipMemcpyAsync(h->d_normals, normals, sizeof(double) * n_norm, hipMemcpyHostToDevice, st));
  } else {
    const int blocks = (int)std::min<size_t>((n_norm / 2 + 255) / 256, 4096);
    if (h->rng_n_global)  // this shard's slice of the ensemble's stream + the ensemble's resampling offset (same on every rank)
      hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, st, n_norm - 1, (unsigned long long)h->seed,
                         (unsigned long long)h->scan_index, h->d_normals, (const double2*)(h->h_beams + (size_t)slot * h->max_beams), h->d_beams, c.Bv,
                         (size_t)0, (size_t)0, (size_t)h->rng_first * c.stride_normals, (size_t)h->rng_n_global * c.stride_normals, n_norm - 1);
    else
    hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, st, n_norm, (unsigned long long)h->seed,
                       (unsigned long long)h->scan_index, h->d_normals, (const double2*)(h->h_beams + (size_t)slot * h->max_beams), h->d_beams, c.Bv);
    TBNAV_HIP(hipGetLastError());
  }
  ++h->scan_index;
  const double2* const beams_dev = pre ? pre->d_beams : h->d_beams;
  const double* const normals_dev = pre ? pre->d_normals : h->d_normals;
  h->last_normals = normals_dev;
  h->last_z_index = (size_t)h->N * c.stride_normals;
  for (int q = 0; q < 4; ++q) h_err[q] = 0;  // mapped: the scan that last owned the slot has been waited for
  h->h_norm[slot] = NormOut{};
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);

  // ---- distance-field refresh for this call's lookups (windowed), then the particle update
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[0], st));
  {
    // every lookup of this call lies within `half` metres of the particle's CURRENT position: the sampled poses
    // sit at T(pose)*T_icp (or the motion-model pose) +- the sampling noise, the laser at |Trs| from them, and
    // a valid beam ends less than range_max from the laser
    double sig = 0.0;
    for (int q = 1; q < 3; ++q) sig = std::max(sig, std::max(h->p.sample_range[q], h->p.motion_noise[q]));
    const double move = std::max(std::hypot(T_icp[1], T_icp[2]), std::fabs(u[1]));
    const double half = (double)h->p.range_max + std::hypot(h->p.Trs[1], h->p.Trs[2]) + move + 8.0 * std::sqrt(sig);
    int half_cells = (int)std::ceil(half / h->p.resolution) + 3;
    if (h->full_edt || half_cells > h->xsize) half_cells = h->xsize;  // whole map
    if (h->df_mode != 2) {  // query mode needs neither windows nor skip flags: the proposal kernel reads the field state itself
      hipLaunchKernelGGL(rbpf_window, dim3((h->N + 255) / 256), dim3(256), 0, st, c.g, h->N, half_cells, h->df_mode == 1 ? 1 : 0,
                         sp.pose, h->d_fstate, h->d_skip, h->d_win);
      TBNAV_HIP(hipGetLastError());
    }
    if (h->df_mode == 1) {
      const int tiles = std::min((2 * half_cells + 1 + kWave - 1) / kWave + 1, (h->ysize + kWave - 1) / kWave);
      rc = run_distance_field(h, c.g, 0, h->N, tiles);
      if (rc != TBNAV_OK) return rc;
    }
  }
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[1], st));
  size_t propose_lds = sizeof(double) * ((12 + kUnCap) * h->k + 3 * (c.Bv > 0 ? c.Bv : 1)) + sizeof(unsigned int) * 4 * (c.Bv > 0 ? c.Bv : 1);
  const size_t propose_lds_base = propose_lds;  // (what follows adds the LDS slice of the occupancy bitmap)
  int occ_half = 0;
  if (h->df_mode == 2) {
    // LDS copy of the occupancy bitmap round each particle's sensor: reach of a lookup (range_max + sampling
    // spread) plus a margin for the walk to the nearest obstacle; shrunk, then dropped, if it would not fit
    double sig = 0.0;
    for (int q = 1; q < 3; ++q) sig = std::max(sig, std::max(h->p.sample_range[q], h->p.motion_noise[q]));
    const int reach = (int)std::ceil(((double)h->p.range_max + 8.0 * std::sqrt(sig)) / h->p.resolution) + 2;
    for (int margin : {48, 16, 0}) {
      const int half = reach + margin;
      const int rows = std::min(h->xsize, 2 * half + 1), nw = std::min(h->words, (2 * half + 1 + 63) / 64 + 1);
      const size_t bytes = (size_t)rows * nw * 8 + (size_t)rows * 4;
      if (bytes <= 48 * 1024) { occ_half = half; propose_lds += bytes; break; }
    }
  }
  if (propose_lds > (size_t)kMaxLds - 3072) return TBNAV_ERR_UNSUPPORTED;  // scan x samples too large for one workgroup's LDS
  const int* skip_arr = h->df_mode == 2 ? h->d_fstate : h->d_skip;
  const int skip_eq = h->df_mode == 2 ? 2 : 1;
  const double* center = nullptr;
  if (h->sm_on && c.icp_ok) {
    // N1 option: every particle refines T(pose) * T_icp against its own map first; the samples are drawn round that
    const size_t sm_lds = sizeof(double2) * (c.Bv > 0 ? c.Bv : 1) + sizeof(double) * kMixLut + sizeof(unsigned long long) * 4 * (c.Bv > 0 ? c.Bv : 1) +
                          (propose_lds - propose_lds_base);
    hipLaunchKernelGGL(rbpf_scanmatch, dim3(h->N), dim3(kMatchThreads), sm_lds, st, c, h->sm, beams_dev, h->d_code[h->cur],
                       h->pool, map_of(h), h->d_trow[h->cur], skip_arr, skip_eq, h->df_mode, h->radius, occ_half,
                       h->d_nocc[h->cur], h->d_win, sp.pose, h->d_center, h->d_score, d_err, gate_prev, h->d_mixlut);
    TBNAV_HIP(hipGetLastError());
    center = h->d_center;
  }
  // workgroup size: four waves when four workgroups fit a CU's LDS (the 360-beam scans: 38 KB each), eight when the scan's
  // tables leave room for two or three only (1080 beams: 58 KB) — measured: 360 beams 32 us per 1000 particles with 256
  // threads against 43 with 512; the configs[4] shard 0.80 ms with 256 against 0.61 with 512
  h->lk_propose = (propose_lds + 3072 > (size_t)kMaxLds / 4) ? 2 * kProposeThreads : kProposeThreads;
  if (propose_lds + 3072 > (size_t)kMaxLds / 4)
    hipLaunchKernelGGL((rbpf_propose<2 * kProposeThreads>), dim3(h->N), dim3(2 * kProposeThreads), propose_lds, st, c, beams_dev,
                     h->d_code[h->cur], h->pool, map_of(h), h->d_trow[h->cur], skip_arr, skip_eq, h->df_mode, h->radius, occ_half,
                     h->d_nocc[h->cur], h->d_win, normals_dev, center, sp.pose, sp.prev, sp.weight, h->tr, h->d_sens, d_err, gate_prev, h->d_mixlut);
    else
    hipLaunchKernelGGL((rbpf_propose<kProposeThreads>), dim3(h->N), dim3(kProposeThreads), propose_lds, st, c, beams_dev,
                     h->d_code[h->cur], h->pool, map_of(h), h->d_trow[h->cur], skip_arr, skip_eq, h->df_mode, h->radius, occ_half,
                     h->d_nocc[h->cur], h->d_win, normals_dev, center, sp.pose, sp.prev, sp.weight, h->tr, h->d_sens, d_err, gate_prev, h->d_mixlut);
  TBNAV_HIP(hipGetLastError());
  if (weights_ready) TBNAV_HIP(hipEventRecord(weights_ready, st));  // (sharded filter: the exchange starts here, beside the map update)
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[2], st));
  // normalise / select needs only the weights the proposal kernel left: it rides in the map update's launch as one extra
  // workgroup (the chain of adds it is made of would otherwise sit on the critical path, and a second stream costs an
  // event and a dependent boundary).  With event timing on it is a launch of its own, so that the intervals mean what
  // they say.
  const double* z_norm = normals_dev + (size_t)h->N * c.stride_normals;
  tk.seq = (unsigned int)h->scans_done;
  const NormArgs nz{h->N, z_norm, sp.weight, sp.weight, h->d_cs, h->d_parent, h->d_norm + slot, h->d_gate + slot, gate_prev,
                    tk.poll ? h->d_seq + slot : nullptr, tk.seq, h->d_parent + h->N};
  auto launch_normalize = [&](hipStream_t s2) -> int {
    hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, s2, h->N, z_norm, sp.weight, sp.weight, h->d_cs, h->d_parent, h->d_norm + slot,
                       nullptr, nullptr, nullptr, 0u, h->d_parent + h->N);
    TBNAV_HIP(hipGetLastError());
    return TBNAV_OK;
  };
  const bool overlap = !local_only && !h->timing;
  if ((gate_prev || tk.poll) && !overlap) return TBNAV_ERR_INVALID_ARG;  // (a gated or polled scan is a batch scan: never local-only or timed)
  rc = launch_raycast(h, c, h->N, h->d_sens, overlap ? &nz : nullptr, d_err, beams_dev);
  if (rc != TBNAV_OK) return rc;
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[3], st));
  if (h->full_edt) {
    // legacy placement (TBNAV_RBPF_FULL_EDT=1): whole field of every particle right after the map update,
    // where the reference runs its brushfire (grid_mapper.cpp:181)
    TBNAV_HIP(hipMemsetAsync(h->d_skip, 0, sizeof(int) * h->N, st));
    rc = run_distance_field(h, c.g, 0, h->N, (h->ysize + kWave - 1) / kWave);
    if (rc != TBNAV_OK) return rc;
  }
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[4], st));
  if (!local_only && !overlap) {
    rc = launch_normalize(st);
    if (rc != TBNAV_OK) return rc;
  }
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[5], st));
  // the map changed: every field is stale until the next refresh (whole-map mode: fresh everywhere).  In query mode
  // the states are all zero already unless a field was injected or materialised since the last call.
  if (h->full_edt) {
    TBNAV_HIP(hipMemsetD32Async((hipDeviceptr_t)h->d_fstate, 2, h->N, st));
    h->fstate_dirty = true;
  } else if (h->fstate_dirty || h->df_mode != 2) {
    TBNAV_HIP(hipMemsetD32Async((hipDeviceptr_t)h->d_fstate, 0, h->N, st));
    h->fstate_dirty = h->df_mode != 2;
  }
  return TBNAV_OK;
}

int scan_finish(tbnav_rbpf* h, const ScanTicket& tk, tbnav_rbpf_stats* out) {
  hipStream_t st = h->stream;
  const bool local_only = tk.local_only;
  int rc = TBNAV_OK;
  if (tk.poll) {
    // the normalise / select workgroup raises the slot's flag as soon as its result is in host memory — the host need not
    // wait for the rest of the map update (error flags raised later in that launch: see tbnav_rbpf_slam_batch)
    volatile unsigned int* flag = h->h_seq + tk.slot;
    for (unsigned long spins = 1; *flag != tk.seq; ++spins) {
      __builtin_ia32_pause();
      if ((spins & 0xFFFF) == 0) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess && *flag != tk.seq) return TBNAV_ERR_HIP;  // the stream drained and the flag never came
        if (q != hipSuccess && q != hipErrorNotReady) TBNAV_HIP(q);
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    TBNAV_HIP(hipStreamSynchronize(st));
  }
  const int* err = h->h_err + 4 * tk.slot;
  const NormOut no = h->h_norm[tk.slot];
  out->status = status_from_err(err);
  out->sum_w = no.sum_w; out->sq_sum = no.sq_sum; out->neff = no.neff; out->resampled = no.resampled;
  bool gathered = false;
  if (!local_only && no.resampled && out->status == TBNAV_OK) {
    // lowVarianceResampling's deep copies (particle_filter.cpp:495): tables and state move on the device
    if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[6], st));
    rc = resample_on_device(h);
    if (rc != TBNAV_OK) return rc;
    if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[7], st));
    gathered = true;
  }
  if (gathered && !tk.poll) TBNAV_HIP(hipStreamSynchronize(st));  // (a batch goes straight on: the next scan is behind the copies in the stream)
  for (float& v : h->last_ms) v = 0.f;
  if (h->timing) {
    float e01, e12, e23, e34, e45;
    TBNAV_HIP(hipEventElapsedTime(&e01, h->ev[0], h->ev[1]));
    TBNAV_HIP(hipEventElapsedTime(&e12, h->ev[1], h->ev[2]));
    TBNAV_HIP(hipEventElapsedTime(&e23, h->ev[2], h->ev[3]));
    TBNAV_HIP(hipEventElapsedTime(&e34, h->ev[3], h->ev[4]));
    TBNAV_HIP(hipEventElapsedTime(&e45, h->ev[4], h->ev[5]));
    h->last_ms[0] = e12;        // propose
    h->last_ms[1] = e23;        // raycast
    h->last_ms[2] = 0.f;        // (occupancy pass: folded into the raycast)
    h->last_ms[3] = e01 + e34;  // distance field (windowed refresh before the update, or whole-map after it)
    h->last_ms[4] = e45;        // normalise / select
    if (gathered) TBNAV_HIP(hipEventElapsedTime(&h->last_ms[5], h->ev[6], h->ev[7]));
  }
  if (h->ref_field && out->status == TBNAV_OK) {
    rc = ref_field_after_scan(h, gathered);
    if (rc != TBNAV_OK) return rc;
  }
  return out->status;
}

int sharded_scan(int n, tbnav_rbpf* const* hs, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* const* normals, tbnav_rbpf_stats* out,
                 tbnav_rbpf_stats* local_out);

int slam_impl(tbnav_rbpf* h, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
              const double prev_odom[3], int icp_ok, const double T_icp[3], const double* normals,
              tbnav_rbpf_stats* out, bool local_only) {
  if (!h || !scan || n_beams <= 0 || !u || !cur_odom || !prev_odom || !T_icp || !out) return TBNAV_ERR_INVALID_ARG;
  if (h->ref_field && (local_only || h->comm)) return TBNAV_ERR_UNSUPPORTED;  // the reference-field mode is a single-handle mode
  if (h->comm && !local_only) {  // this process's rank of a sharded filter: the exchange is issued from here (sharded_scan)
    const double* nr[1] = {normals};
    return sharded_scan(1, &h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, nr, out, nullptr);
  }
  DeviceGuard guard(h->device);
  ScanTicket tk;
  const int rc = scan_enqueue(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals, out, local_only, 0, nullptr, tk);
  if (rc != TBNAV_OK) return rc;
  return scan_finish(h, tk, out);
}

}  // namespace

extern "C" {

namespace {
// host threads for the reference-field mode: the cores this process may run on (its affinity mask; a container's CPU quota is
// not visible here — TBNAV_RBPF_OPT_HOST_THREADS overrides), at most 32
int default_host_threads() {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  return n < 1 ? 1 : (n > 32 ? 32 : n);
}
int create_impl(const tbnav_rbpf_params* P, uint64_t max_pool_bytes, tbnav_rbpf** out) {
  if (!P || !out) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  if (P->num_particles <= 0 || P->num_samples_mode <= 0 || !(P->resolution > 0.0) || !(P->xmax > P->xmin) || !(P->ymax > P->ymin))
    return TBNAV_ERR_INVALID_ARG;
  const int xsize = (int)static_cast<unsigned int>(std::ceil((P->xmax - P->xmin) / P->resolution));  // mapSize, grid_mapper.cpp:31-34
  const int ysize = (int)static_cast<unsigned int>(std::ceil((P->ymax - P->ymin) / P->resolution));
  if (xsize != ysize) return TBNAV_ERR_UNSUPPORTED;  // the reference indexes both axes with xsize_ (grid_mapper.cpp:195-197,896)
  if (xsize < 4 || xsize > 32000 || (xsize & 1)) return TBNAV_ERR_UNSUPPORTED;  // vectorised code copies need G % 4 == 0
  if (P->num_particles > (1 << 20)) return TBNAV_ERR_UNSUPPORTED;
  const int radius = (int)static_cast<unsigned int>(std::ceil((10.0 - 0.0) / P->resolution));         // cell_radius_, grid_mapper.cpp:50
  if (radius > 254) return TBNAV_ERR_UNSUPPORTED;  // row-pass distances are stored as u8 (and radius^2 must fit the u16 code)
  const int words = (ysize + 63) / 64;
  int C = 64;
  if (edt_lds_bytes(xsize, words, C) > (size_t)kMaxLds) C = 32;
  // larger maps (xsize > ~640, e.g. BASELINE configs[4]'s 2000 x 2000): no LDS distance transform.  The SLAM path then
  // always answers lookups by query, and an on-demand field is produced cell by cell with the same query.
  if (edt_lds_bytes(xsize, words, C) > (size_t)kMaxLds) C = 0;
  int ndev = 0;
  {
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      return tbnav::hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount", __FILE__, __LINE__);
  }
  int dev = P->device;
  if (dev < 0) TBNAV_HIP(hipGetDevice(&dev));
  if (dev >= ndev) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(dev);
  if (!guard.ok) return TBNAV_ERR_NO_DEVICE;

  tbnav_rbpf* h = new (std::nothrow) tbnav_rbpf();
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->p = *P; h->device = dev; h->N = P->num_particles; h->k = P->num_samples_mode;
  h->host_threads = default_host_threads();
  h->xsize = xsize; h->ysize = ysize; h->words = words; h->radius = radius; h->edt_cols = C;
  h->G = (size_t)xsize * ysize;
  h->TW = (xsize + kTS - 1) / kTS; h->TT = h->TW * h->TW;
  {
    // every beam shorter than range_max ends within this many cells of the robot cell (+2 for the laser offset / rounding)
    const double reach = (double)P->range_max + std::sqrt(P->Trs[1] * P->Trs[1] + P->Trs[2] * P->Trs[2]);
    const long side = 2 * ((long)std::ceil(reach / P->resolution) + 2) + 1;
    h->tile_cap = (side * side <= 30000) ? (int)(side * side) : 0;
    h->df_mode = 2;  // exact query at lookup; the other modes are selected with tbnav_rbpf_set_option
    h->full_edt = false;
  }
  // log-odds constants with the host libm, exactly as the reference's ctor (grid_mapper.cpp:42-47)
  h->l_prior = std::log(0.5 / (1 - 0.5));
  h->l_occ = std::log(0.90 / (1 - 0.90));
  h->l_free = std::log(0.35 / (1 - 0.35));
  h->cut_occ = find_occ_cut(h->l_occ, 0.90);
  h->cuts = derive_export_cuts(h->cut_occ);
  const int N = h->N;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
  const size_t table_entries = (size_t)N * h->TT;
  for (int b = 0; b < 2; ++b) {
    A((void**)&h->d_state[b], sizeof(double) * 7 * N);
    A((void**)&h->d_table[b], sizeof(unsigned int) * table_entries);
    A((void**)&h->d_nocc[b], sizeof(int) * N);
    A((void**)&h->d_trow[b], sizeof(int) * (size_t)N * h->TW);
  }
  A((void**)&h->d_shed, sizeof(unsigned int) * table_entries);
  A((void**)&h->d_cs, sizeof(double) * N);
  A((void**)&h->d_sens, sizeof(double) * 4 * N);
  A((void**)&h->d_tile_scratch, sizeof(unsigned int) * h->TT);
  A((void**)&h->d_touched, sizeof(unsigned long long) * 2);
  A((void**)&h->d_box_need, sizeof(int) * 3);
  A((void**)&h->d_parent, sizeof(int) * 2 * N);
  A((void**)&h->d_best, sizeof(int));
  A((void**)&h->d_best_pose, sizeof(double) * 3);
  A((void**)&h->d_export, h->G);
  A((void**)&h->d_fstate, sizeof(int) * N);
  A((void**)&h->d_fstate_alt, sizeof(int) * N);
  A((void**)&h->d_center, sizeof(double) * 3 * N);
  A((void**)&h->d_score, sizeof(double) * N);
  A((void**)&h->d_skip, sizeof(int) * N);
  A((void**)&h->d_win, sizeof(int4) * N);
  A((void**)&h->d_tier, sizeof(int) * N);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_err, sizeof(int) * 4 * kScanSlots, hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_norm, sizeof(NormOut) * kScanSlots, hipHostMallocMapped);
  if (e == hipSuccess) { std::memset(h->h_err, 0, sizeof(int) * 4 * kScanSlots); std::memset((void*)h->h_norm, 0, sizeof(NormOut) * kScanSlots); }
  A((void**)&h->d_gate, sizeof(int) * kScanSlots);
  if (e == hipSuccess) e = hipMemset(h->d_gate, 0, sizeof(int) * kScanSlots);
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_seq, sizeof(unsigned int) * kScanSlots, hipHostMallocMapped);
  if (e == hipSuccess) { std::memset(h->h_seq, 0, sizeof(unsigned int) * kScanSlots); e = hipHostGetDevicePointer((void**)&h->d_seq, h->h_seq, 0); }
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->d_err, h->h_err, 0);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->d_norm, h->h_norm, 0);
  const size_t kk = (size_t)h->k;
  const size_t trace_doubles = (size_t)N * (kk * 3 + kk + kk + 3 + 9 + 1 + 3 + 1);
  A((void**)&h->d_trace, sizeof(double) * trace_doubles);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  for (auto& ev : h->ev) if (e == hipSuccess) e = hipEventCreate(&ev);
  // ---- the tile pool: everything the particles' maps can ever need if that fits the budget, else the budget.
  //      Default budget: half of the memory that is free now (several handles can live side by side).
  if (e == hipSuccess) {
    size_t free_b = 0, total_b = 0;
    e = hipMemGetInfo(&free_b, &total_b);
    const size_t per_tile = sizeof(double) * kTileCells + sizeof(unsigned int) * kTS + sizeof(int) + sizeof(unsigned int);
    const size_t budget = max_pool_bytes ? (size_t)max_pool_bytes : free_b / 2;
    // worst case: every table entry its own tile, plus the tiles the entries left since the last resample (still named by
    // the shed notes until the next resample settles them), plus the zero tile
    size_t cap = 2 * table_entries + 1;
    if (cap * per_tile > budget) cap = budget / per_tile;
    if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
    if (cap < (size_t)N + 2 && e == hipSuccess) e = hipErrorOutOfMemory;  // not even one tile per particle
    h->pool.cap = (unsigned int)cap;
    A((void**)&h->pool.lo, sizeof(double) * kTileCells * cap);
    A((void**)&h->pool.bm, sizeof(unsigned int) * kTS * cap);
    A((void**)&h->pool.ref, sizeof(int) * cap);
    A((void**)&h->pool.ring, sizeof(unsigned int) * cap);
    A((void**)&h->pool.ctr, sizeof(unsigned long long) * 2);
  }
  if (e == hipSuccess) {
    double* t = h->d_trace;
    h->tr.sampled = t; t += (size_t)N * kk * 3;
    h->tr.p_scan = t; t += (size_t)N * kk;
    h->tr.p_pose = t; t += (size_t)N * kk;
    h->tr.mu = t; t += (size_t)N * 3;
    h->tr.sigma = t; t += (size_t)N * 9;
    h->tr.eta = t; t += (size_t)N;
    h->tr.new_pose = t; t += (size_t)N * 3;
    h->tr.weight_raw = t;
    e = hipMemset(h->d_trace, 0, sizeof(double) * trace_doubles);
  }
  if (e == hipSuccess) {  // initParticleSet, particle_filter.cpp:125-138
    std::vector<double> s((size_t)7 * N);
    for (int i = 0; i < N; ++i) {
      for (int q = 0; q < 3; ++q) { s[(size_t)i * 3 + q] = P->pose0[q]; s[(size_t)3 * N + i * 3 + q] = P->pose0[q]; }
      s[(size_t)6 * N + i] = 1.0 / N;
    }
    e = hipMemcpy(h->d_state[0], s.data(), sizeof(double) * 7 * N, hipMemcpyHostToDevice);
    // empty maps: every table entry names the shared zero tile (log_odds_prior_ = log(1) = 0)
    if (e == hipSuccess) e = hipMemset(h->d_table[0], 0, sizeof(unsigned int) * table_entries);
    if (e == hipSuccess) e = hipMemset(h->d_shed, 0, sizeof(unsigned int) * table_entries);
    if (e == hipSuccess) e = hipMemset(h->pool.lo, 0, sizeof(double) * kTileCells);  // tile 0
    if (e == hipSuccess) e = hipMemset(h->pool.ref, 0, sizeof(int) * h->pool.cap);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(rbpf_pool_init, dim3(1024), dim3(256), 0, h->stream, h->pool);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemset(h->d_touched, 0, sizeof(unsigned long long) * 2);
    if (e == hipSuccess) e = hipMemset(h->d_box_need, 0, sizeof(int) * 3);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_box_need, sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) { *h->h_box_need = 0; e = hipHostGetDevicePointer((void**)&h->d_box_need_host, h->h_box_need, 0); }
    if (e == hipSuccess) e = hipMemset(h->d_nocc[0], 0, sizeof(int) * N);
    if (e == hipSuccess) e = hipMemset(h->d_skip, 0, sizeof(int) * N);
    // empty maps: the field "everything unreached" is what any lookup computes, no stored field needed (state 0)
    if (e == hipSuccess) e = hipMemset(h->d_fstate, 0, sizeof(int) * N);
    if (e == hipSuccess) e = hipMemset(h->d_fstate_alt, 0, sizeof(int) * N);
    h->fstate_dirty = false;
    if (e == hipSuccess) {
      std::vector<int4> w(N, make_int4(0, xsize - 1, 0, ysize - 1));
      e = hipMemcpy(h->d_win, w.data(), sizeof(int4) * N, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemset(h->d_trow[0], 0, sizeof(int) * (size_t)N * h->TW);
    if (e == hipSuccess) e = hipMemset(h->pool.bm, 0, sizeof(unsigned int) * kTS);  // tile 0
  }
  if (e == hipSuccess && C > 0) {
    const int lds = (int)edt_lds_bytes(xsize, words, C);
    e = (C == 64) ? hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt<64>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
                  : hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt<32>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  }
  // (2.3 KB of static LDS: the embedded normalise's scan scratch)
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 1024);
  // the proposal / scan-match kernels carry the scan, the per-sample data and the bitmap slice: more than the 64 KB
  // default for long scans or many samples
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<kProposeThreads>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);  // (2.3 KB static)
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<2 * kProposeThreads>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_scanmatch), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt_compact<kEdtRowsA>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)edt_compact_lds(kEdtRowsA));
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_edt_compact<kEdtRowsB>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)edt_compact_lds(kEdtRowsB));
  A((void**)&h->d_mixlut, sizeof(double) * kMixLut);
  if (e == hipSuccess) {
    ScanC cm{};
    if (mixture_consts(h, cm)) {  // (a zero variance is reported by the first scan, as the reference throws there)
      hipLaunchKernelGGL(rbpf_mix_lut, dim3(kMixLut / 256), dim3(256), 0, h->stream, cm, h->d_mixlut);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    const int rc = tbnav::hip_fail(e, "tbnav_rbpf_create allocation", __FILE__, __LINE__);
    tbnav_rbpf_destroy(h);
    return rc;
  }
  *out = h;
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_create(const tbnav_rbpf_params* P, tbnav_rbpf** out) { return create_impl(P, 0, out); }
int tbnav_rbpf_create_pool(const tbnav_rbpf_params* P, uint64_t max_pool_bytes, tbnav_rbpf** out) { return create_impl(P, max_pool_bytes, out); }

int tbnav_rbpf_pool_stats(tbnav_rbpf* h, uint64_t* capacity_tiles, uint64_t* free_tiles, uint64_t* tile_bytes) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  unsigned long long ctr[2] = {0, 0};
  TBNAV_HIP(hipMemcpy(ctr, h->pool.ctr, sizeof ctr, hipMemcpyDeviceToHost));
  if (capacity_tiles) *capacity_tiles = h->pool.cap - 1;  // tile 0 is the shared zero tile
  if (free_tiles) *free_tiles = ctr[1] - ctr[0];
  if (tile_bytes) *tile_bytes = sizeof(double) * kTileCells;
  return TBNAV_OK;
}

void tbnav_rbpf_destroy(tbnav_rbpf* h) {
#ifdef TBNAV_PHASE_PROF
  {
    unsigned long long ph[8];
    if (hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase_p), sizeof(ph)) == hipSuccess && ph[7])
      std::fprintf(stderr, "[propose phases, 10 ns ticks per workgroup] sampling %.1f | up to the per-beam lookups %.1f | per-sample products %.1f | "
                           "Gaussian fit %.1f | unstable beams %.1f\n",
                   (double)ph[5] / ph[7], (double)ph[0] / ph[7], (double)ph[1] / ph[7], (double)ph[2] / ph[7], (double)ph[6] / ph[7]);
    if (hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph)) == hipSuccess && ph[7])
      std::fprintf(stderr, "[raycast phases, 10 ns ticks per workgroup] set-up %.1f | flags/slots %.1f | ray walk %.1f | end-point replay %.1f | "
                           "other cells %.1f | overflowed slots %.2f | end-point cells %.1f\n",
                   (double)ph[0] / ph[7], (double)ph[1] / ph[7], (double)ph[2] / ph[7], (double)ph[3] / ph[7], (double)ph[4] / ph[7],
                   (double)ph[5] / ph[7], (double)ph[6] / ph[7]);
    unsigned long long tp[2][4][16];
    if (hipMemcpyFromSymbol(tp, HIP_SYMBOL(g_trace_p), sizeof(tp)) == hipSuccess && tp[0][0][0]) {
      for (int g = 0; g < 2; ++g) {
        std::fprintf(stderr, "[rbpf_propose trace of workgroup %d, us; columns: loads requested, centre's sensor transform known, table ids in LDS (barrier), "
                             "slice staged (barrier), step 1 done (wave 0: samples + odometry likelihoods; others: centre lookups), barrier (+ deferred searches), "
                             "stable product + unstable list (barrier), pairs / products / weights (barrier), sums, end]\n", g ? 100 : 96);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 4; ++w) if (tp[g][w][0] && tp[g][w][0] < t0) t0 = tp[g][w][0];
        for (int w = 0; w < 4; ++w) {
          std::fprintf(stderr, "  wave %d:", w);
          for (int i = 0; i < 10; ++i) std::fprintf(stderr, " %6.2f", tp[g][w][i] ? (double)(tp[g][w][i] - t0) * 0.01 : -1.0);
          std::fprintf(stderr, "\n");
        }
      }
    }
    unsigned long long tr[2][16][16];
    if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr)) == hipSuccess && tr[0][0][0]) {
      for (int g = 0; g < 2; ++g) {
        std::fprintf(stderr, "[raycast_box trace of workgroup %d, us since its first stamp; columns: entry, pose barrier, end-point barrier, flags, flag barrier, own events, walk, walk barrier, requests, barrier, tiles private, replay, overflow+hot, barrier, stores]\n", g ? 900 : 100);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 16; ++w) if (tr[g][w][0] && tr[g][w][0] < t0) t0 = tr[g][w][0];
        for (int w = 0; w < 16; ++w) {
          std::fprintf(stderr, "  wave %2d:", w);
          for (int i = 0; i < 15; ++i) std::fprintf(stderr, " %5.2f", tr[g][w][i] ? (double)(tr[g][w][i] - t0) * 0.01 : -1.0);
          std::fprintf(stderr, "\n");
        }
      }
    }
    {
      static unsigned long long wgp[4096][3];
      if (hipMemcpyFromSymbol(wgp, HIP_SYMBOL(g_wgp), sizeof(wgp)) == hipSuccess && wgp[1][0]) {
        int n = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < 4096; ++i) if (wgp[i][0] && wgp[i][1]) { ++n; t0 = std::min(t0, wgp[i][0]); t1 = std::max(t1, wgp[i][1]); }
        const int nb = 16;
        const double span = (double)(t1 - t0);
        int active[nb] = {0}, starts[nb] = {0};
        double dur[nb] = {0};
        for (int i = 0; i < 4096; ++i) if (wgp[i][0] && wgp[i][1]) {
          const int bs = std::min(nb - 1, (int)((double)(wgp[i][0] - t0) / span * nb));
          ++starts[bs]; dur[bs] += (double)(wgp[i][1] - wgp[i][0]) * 0.01;
          for (int b = 0; b < nb; ++b) { const double tm = t0 + (b + 0.5) * span / nb; if ((double)wgp[i][0] <= tm && tm < (double)wgp[i][1]) ++active[b]; }
        }
        std::fprintf(stderr, "[rbpf_propose workgroups of the last launch] %d recorded, first entry to last exit %.2f us; bins of %.2f us\n  resident at mid-bin:", n, span * 0.01, span * 0.01 / nb);
        for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %5d", active[b]);
        std::fprintf(stderr, "\n  entered in bin:     ");
        for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %5d", starts[b]);
        std::fprintf(stderr, "\n  mean residence (us):");
        for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %5.1f", starts[b] ? dur[b] / starts[b] : 0.0);
        // by XCC and by CU: is a slow workgroup's CU slow as a whole?
        std::map<unsigned long long, std::vector<double>> by_cu;
        double xs[16] = {0}; int xn[16] = {0};
        for (int i = 0; i < 4096; ++i) if (wgp[i][0] && wgp[i][1]) {
          const unsigned int hw = (unsigned int)wgp[i][2], xcc = (unsigned int)(wgp[i][2] >> 32) & 0xF;
          const double d = (double)(wgp[i][1] - wgp[i][0]) * 0.01;
          by_cu[((unsigned long long)xcc << 16) | (hw & 0xFF00u)].push_back(d);
          xs[xcc] += d; ++xn[xcc];
        }
        std::fprintf(stderr, "\n  mean residence by XCC:");
        for (int x = 0; x < 16; ++x) if (xn[x]) std::fprintf(stderr, " %.1f", xs[x] / xn[x]);
        double spread_in = 0.0; int ncu = 0; double cu_min = 1e9, cu_max = 0; int n3 = 0, n4 = 0; double d3 = 0, d4 = 0;
        for (auto& kv : by_cu) {
          double lo = 1e9, hi = 0, sum = 0;
          for (double d : kv.second) { lo = std::min(lo, d); hi = std::max(hi, d); sum += d; }
          spread_in += hi - lo; ++ncu;
          const double mean = sum / kv.second.size();
          cu_min = std::min(cu_min, mean); cu_max = std::max(cu_max, mean);
          if (kv.second.size() <= 3) { ++n3; d3 += mean; } else { ++n4; d4 += mean; }
        }
        std::fprintf(stderr, "\n  %d CUs; mean (max - min) inside a CU %.1f us; CU means from %.1f to %.1f us; CUs with <= 3 workgroups: %d, mean %.1f us; with 4+: %d, mean %.1f us\n",
                     ncu, spread_in / std::max(1, ncu), cu_min, cu_max, n3, n3 ? d3 / n3 : 0.0, n4, n4 ? d4 / n4 : 0.0);
      }
    }
    {
      static unsigned long long wg[4096][3];
      if (hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_wg), sizeof(wg)) == hipSuccess && wg[1][0]) {
        int n = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < 4096; ++i) if (wg[i][0] && wg[i][1]) { ++n; t0 = std::min(t0, wg[i][0]); t1 = std::max(t1, wg[i][1]); }
        std::fprintf(stderr, "[raycast_box workgroups of the last launch] %d recorded, first entry to last exit %.2f us\n", n, (double)(t1 - t0) * 0.01);
        const int nb = 16;
        const double span = (double)(t1 - t0);
        int active[nb] = {0}, starts[nb] = {0};
        double dur_by_start[nb] = {0};
        for (int i = 0; i < 4096; ++i) if (wg[i][0] && wg[i][1]) {
          const int bs = std::min(nb - 1, (int)((double)(wg[i][0] - t0) / span * nb));
          ++starts[bs]; dur_by_start[bs] += (double)(wg[i][1] - wg[i][0]) * 0.01;
          for (int b = 0; b < nb; ++b) { const double tm = t0 + (b + 0.5) * span / nb; if ((double)wg[i][0] <= tm && tm < (double)wg[i][1]) ++active[b]; }
        }
        std::fprintf(stderr, "  time bin (%.2f us each):", span * 0.01 / nb);
        for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %5d", b);
        std::fprintf(stderr, "\n  resident at mid-bin:    ");
        for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %5d", active[b]);
        std::fprintf(stderr, "\n  entered in bin:         ");
        for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %5d", starts[b]);
        std::fprintf(stderr, "\n  mean residence (us):    ");
        for (int b = 0; b < nb; ++b) std::fprintf(stderr, " %5.1f", starts[b] ? dur_by_start[b] / starts[b] : 0.0);
        std::map<unsigned long long, int> per_cu, per_xcc;
        for (int i = 0; i < 4096; ++i) if (wg[i][0] && wg[i][1]) {
          const unsigned int hw = (unsigned int)wg[i][2], xcc = (unsigned int)(wg[i][2] >> 32) & 0xF;
          ++per_cu[((unsigned long long)xcc << 16) | (hw & 0xFF00u)];   // cu_id [11:8], sh_id [12], se_id [15:13]
          ++per_xcc[xcc];
        }
        int hist[16] = {0};
        for (auto& kv : per_cu) ++hist[std::min(15, kv.second)];
        std::fprintf(stderr, "\n  CUs that ran workgroups: %zu; CUs by number of workgroups run:", per_cu.size());
        for (int k = 1; k < 16; ++k) if (hist[k]) std::fprintf(stderr, " %d:%d", k, hist[k]);
        std::fprintf(stderr, "\n  workgroups per XCC:");
        for (auto& kv : per_xcc) std::fprintf(stderr, " %d", kv.second);
        double xs[16] = {0}; int xn[16] = {0};
        for (int i = 0; i < 4096; ++i) if (wg[i][0] && wg[i][1]) { const unsigned int xcc = (unsigned int)(wg[i][2] >> 32) & 0xF; xs[xcc] += (double)(wg[i][1] - wg[i][0]) * 0.01; ++xn[xcc]; }
        std::fprintf(stderr, "\n  mean residence by XCC (us):");
        for (int x = 0; x < 16; ++x) if (xn[x]) std::fprintf(stderr, " %.1f", xs[x] / xn[x]);
        std::fprintf(stderr, "\n");
      }
    }
    unsigned long long pw[16];
    if (hipMemcpyFromSymbol(pw, HIP_SYMBOL(g_phase_w), sizeof(pw)) == hipSuccess && pw[15])
      std::fprintf(stderr, "[raycast_box phases, 10 ns ticks per workgroup] set-up + flags %.1f | events + walk %.1f | requests + marks %.1f | "
                           "private tiles %.1f | end-point replay + hot cells %.1f | pairs %.1f | (unused) %.1f\n",
                   (double)pw[0] / pw[15], (double)pw[1] / pw[15], (double)pw[2] / pw[15], (double)pw[3] / pw[15], (double)pw[4] / pw[15],
                   (double)pw[5] / pw[15], (double)pw[14] / pw[15]);
  }
#endif
  if (!h) return;
  DeviceGuard guard(h->device);
  for (int b = 0; b < 2; ++b) { (void)hipFree(h->d_state[b]); (void)hipFree(h->d_table[b]); (void)hipFree(h->d_code[b]); (void)hipFree(h->d_nocc[b]); (void)hipFree(h->d_trow[b]); }
  (void)hipFree(h->d_bm_dense); (void)hipFree(h->d_rc_dense);
  (void)hipFree(h->pool.lo); (void)hipFree(h->pool.bm); (void)hipFree(h->pool.ref); (void)hipFree(h->pool.ring); (void)hipFree(h->pool.ctr);
  (void)hipFree(h->d_sens); (void)hipFree(h->d_shed); (void)hipFree(h->d_dense); (void)hipFree(h->d_cs); (void)hipFree(h->d_touched); (void)hipFree(h->d_box_need); if (h->h_box_need) (void)hipHostFree(h->h_box_need); (void)hipFree(h->d_fstate_alt);
  (void)hipFree(h->d_log_ev); (void)hipFree(h->d_log_cnt); (void)hipFree(h->d_tile_scratch); (void)hipFree(h->d_log_pack); (void)hipFree(h->d_log_off); (void)hipFree(h->d_code_src);
  (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); (void)hipFree(h->d_gz);
  (void)hipFree(h->d_gw_raw); (void)hipFree(h->d_sendbuf); (void)hipFree(h->d_recvbuf); (void)hipFree(h->d_sizes); (void)hipFree(h->d_status);
  if (h->ev_w) (void)hipEventDestroy(h->ev_w);
  if (h->ev_g) (void)hipEventDestroy(h->ev_g);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  (void)hipFree(h->d_beams); (void)hipFree(h->d_normals); (void)hipFree(h->d_parent); (void)hipFree(h->d_best); (void)hipFree(h->d_best_pose); (void)hipFree(h->d_export); (void)hipFree(h->d_tier); (void)hipFree(h->d_fstate); (void)hipFree(h->d_skip); (void)hipFree(h->d_win); (void)hipFree(h->d_center); (void)hipFree(h->d_mixlut); (void)hipFree(h->d_bslots); (void)hipFree(h->d_bcount); (void)hipFree(h->d_bitems); (void)hipFree(h->d_bhdr); (void)hipFree(h->d_score);
  (void)hipFree(h->d_trace);
  (void)hipHostFree(h->h_beams); (void)hipHostFree(h->h_err); (void)hipHostFree(h->h_norm); (void)hipFree(h->d_gate);
  for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
  (void)hipHostFree(h->h_seq); (void)hipHostFree(h->h_beam_ring); (void)hipFree(h->d_beam_ring); (void)hipFree(h->d_norm_ring);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h->ref;
  delete h;
}

int tbnav_rbpf_grid_size(const tbnav_rbpf* h, int32_t* xsize, int32_t* ysize) {
  if (!h || !xsize || !ysize) return TBNAV_ERR_INVALID_ARG;
  *xsize = h->xsize; *ysize = h->ysize;
  return TBNAV_OK;
}

int tbnav_rbpf_set_seed(tbnav_rbpf* h, uint64_t seed) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->seed = seed;
  h->scan_index = 0;
  return TBNAV_OK;
}

int tbnav_rbpf_set_rng_shard(tbnav_rbpf* h, uint64_t first_particle, uint64_t particles_global) {
  if (!h || (particles_global && particles_global < first_particle + (uint64_t)h->N)) return TBNAV_ERR_INVALID_ARG;
  h->rng_first = first_particle;
  h->rng_n_global = particles_global;
  return TBNAV_OK;
}

int tbnav_rbpf_get_normals(tbnav_rbpf* h, double* out, int64_t n) {
  if (!h || !out || n <= 0 || (size_t)n > std::max(h->normals_cap, h->norm_ring_stride)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(out, h->last_normals ? h->last_normals : h->d_normals, sizeof(double) * n, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int64_t tbnav_rbpf_num_normals(const tbnav_rbpf* h, int32_t icp_ok) {
  if (!h) return -1;
  return (int64_t)h->N * (icp_ok ? 3 * h->k + 3 : 3) + 1;
}

int tbnav_rbpf_slam(tbnav_rbpf* h, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                    const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals,
                    tbnav_rbpf_stats* out) {
  return slam_impl(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals, out, false);
}

int tbnav_rbpf_slam_batch(tbnav_rbpf* h, const float* scans, int32_t n_beams, int32_t n_scans, const double* u, const double* odom,
                          const int32_t* icp_ok, const double* T_icp, tbnav_rbpf_stats* out) {
  if (!h || !scans || n_scans <= 0 || !u || !odom || !T_icp || !out) return TBNAV_ERR_INVALID_ARG;
  // Two scans in the stream at a time: scan s + 1 is enqueued BEFORE the host waits for scan s, on the assumption that scan
  // s does not resample — its kernels check scan s's decision on the device (NormArgs::gate) and do nothing if it does; the
  // host then runs the copies and enqueues scan s + 1 again.  Between scans the device waits for nothing, and the results
  // are those of n_scans synchronous calls, bit for bit.  Only in the default configuration (distance look-ups by query: no
  // per-scan field refresh on the stream; no event timing; not the reference-field mode).
  const bool pipelined = n_scans > 1 && h->batch_pipeline && h->df_mode == 2 && !h->full_edt && !h->ref_field && !h->timing && !h->rng_n_global;
  if (!pipelined) {
    for (int s = 0; s < n_scans; ++s) {
      const int rc = slam_impl(h, scans + (size_t)s * n_beams, n_beams, u + 3 * s, odom + 3 * (s + 1), odom + 3 * s, icp_ok ? icp_ok[s] : 1,
                               T_icp + 3 * s, nullptr, out + s, false);
      if (rc != TBNAV_OK) return rc;
    }
    return TBNAV_OK;
  }
  if (n_beams <= 0) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  // The noise and the beam tables of the next few scans are put on the device by ONE launch per chunk (same Philox counters
  // as one launch per scan: same values), so that a scan is two launches — proposal, map update — back to back.
  const size_t norm_stride = (((size_t)h->N * (3 * (size_t)h->k + 3) + 1) + 1) & ~(size_t)1;
  int chunk = 8;
  while (chunk > 2 && (size_t)chunk * norm_stride * sizeof(double) > ((size_t)512 << 20)) --chunk;
  const bool ahead = chunk >= 3;  // (the host rewrites the pinned staging of chunk c + 1 once scan 0 of chunk c is through)
  if (ahead && (h->ring_scans != chunk || h->norm_ring_stride != norm_stride || h->beam_ring_stride != (size_t)n_beams)) {
    TBNAV_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_norm_ring); (void)hipFree(h->d_beam_ring); (void)hipHostFree(h->h_beam_ring);
    h->d_norm_ring = nullptr; h->d_beam_ring = nullptr; h->h_beam_ring = nullptr; h->ring_scans = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_norm_ring, sizeof(double) * norm_stride * chunk));
    TBNAV_HIP(hipMalloc((void**)&h->d_beam_ring, sizeof(double2) * (size_t)n_beams * chunk));
    TBNAV_HIP(hipHostMalloc((void**)&h->h_beam_ring, sizeof(double2) * (size_t)n_beams * chunk, hipHostMallocDefault));
    std::memset(h->h_beam_ring, 0, sizeof(double2) * (size_t)n_beams * chunk);
    h->ring_scans = chunk; h->norm_ring_stride = norm_stride; h->beam_ring_stride = (size_t)n_beams;
  }
  const unsigned long long scan0 = h->scan_index;  // noise counter of the batch's first scan
  std::vector<Prefetched> pre(ahead ? chunk : 0);
  int prepared_to = 0;  // scans [0, prepared_to) have had their chunk prepared
  auto prepare = [&](int first) -> int {
    const int m = std::min(chunk, n_scans - first);
    for (int j = 0; j < m; ++j) {
      const int s = first + j;
      Prefetched& q = pre[j];
      q.rc = build_scan_consts(h, q.c, scans + (size_t)s * n_beams, n_beams, u + 3 * s, odom + 3 * (s + 1), odom + 3 * s,
                               icp_ok ? icp_ok[s] : 1, T_icp + 3 * s, h->beams_tmp);
      q.d_beams = h->d_beam_ring + (size_t)j * n_beams;
      q.d_normals = h->d_norm_ring + (size_t)j * norm_stride;
      if (q.rc == TBNAV_OK && q.c.Bv) std::memcpy(h->h_beam_ring + (size_t)j * n_beams, h->beams_tmp.data(), sizeof(double2) * q.c.Bv);
    }
    const int blocks = (int)std::min<size_t>((norm_stride / 2 + 255) / 256, 4096);
    hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks, m), dim3(256), 0, h->stream, norm_stride, (unsigned long long)h->seed,
                       scan0 + (unsigned long long)first, h->d_norm_ring, (const double2*)h->h_beam_ring, h->d_beam_ring, n_beams,
                       norm_stride, (size_t)n_beams);
    TBNAV_HIP(hipGetLastError());
    prepared_to = first + m;
    return TBNAV_OK;
  };
  ScanTicket tk[2];
  auto enqueue = [&](int s, const int* gate_prev) -> int {
    if (ahead && s >= prepared_to) { const int rc = prepare(s); if (rc != TBNAV_OK) return rc; }
    ScanTicket& t = tk[s & 1];
    t = ScanTicket{};
    t.poll = true;
    h->scan_index = scan0 + (unsigned long long)s;  // (scan_enqueue counts it)
    return scan_enqueue(h, scans + (size_t)s * n_beams, n_beams, u + 3 * s, odom + 3 * (s + 1), odom + 3 * s, icp_ok ? icp_ok[s] : 1,
                        T_icp + 3 * s, nullptr, out + s, false, s % kScanSlots, gate_prev, t, ahead ? &pre[s % chunk] : nullptr);
  };
  int rc = enqueue(0, nullptr);
  if (rc != TBNAV_OK) return rc;
  for (int s = 0; s < n_scans; ++s) {
    const int rc_next = s + 1 < n_scans ? enqueue(s + 1, h->d_gate + s % kScanSlots) : TBNAV_OK;
    rc = scan_finish(h, tk[s & 1], out + s);
    if (rc == TBNAV_OK && s > 0) {
      // scan s - 1 was finished when its weights were normalised, while its map update was still running; that launch is
      // complete now (scan s ran behind it): anything it flagged after that?
      const int late = status_from_err(h->h_err + 4 * ((s - 1) % kScanSlots));
      if (late != TBNAV_OK) {
        (void)hipStreamSynchronize(h->stream);
        out[s - 1].status = late;
        std::memset(out + s, 0, sizeof(tbnav_rbpf_stats) * (size_t)(n_scans - s));
        return late;
      }
    }
    if (rc != TBNAV_OK || rc_next != TBNAV_OK) {
      (void)hipStreamSynchronize(h->stream);  // (whatever of scan s + 1 is in the stream: the filter's state after an error is unspecified)
      return rc != TBNAV_OK ? rc : rc_next;
    }
    if (out[s].resampled && s + 1 < n_scans) {
      // scan s + 1's launches did nothing: same scan number, same noise, again — on the resampled particles
      --h->scans_done;
      rc = enqueue(s + 1, nullptr);
      if (rc != TBNAV_OK) { (void)hipStreamSynchronize(h->stream); return rc; }
    }
  }
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // the last scan's map update
  out[n_scans - 1].status = status_from_err(h->h_err + 4 * ((n_scans - 1) % kScanSlots));
  return out[n_scans - 1].status;
}

int tbnav_rbpf_slam_local(tbnav_rbpf* h, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                          const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals,
                          tbnav_rbpf_stats* out) {
  return slam_impl(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals, out, true);
}

// Host-side, sequential, bit-faithful: O(n_global) double adds — the exchange step of the sharded
// filter; every rank runs it on the same all-gathered weights.
int tbnav_rbpf_resample_global(const double* w, int64_t n, double z, int32_t* parents, double* wn, tbnav_rbpf_stats* out) {
  if (!w || n <= 0 || !parents || !wn || !out) return TBNAV_ERR_INVALID_ARG;
  std::memset(out, 0, sizeof *out);
  double sum = 0.0;
  for (int64_t i = 0; i < n; ++i) sum += w[i];
  double sq = 0.0;
  for (int64_t i = 0; i < n; ++i) { wn[i] = w[i] / sum; sq += wn[i] * wn[i]; }
  out->sum_w = sum; out->sq_sum = sq;
  out->neff = static_cast<int>(1.0 / sq);
  const int N = (int)n;
  out->resampled = (out->neff < (N / 2)) ? 1 : 0;
  if (!out->resampled) { for (int m = 0; m < N; ++m) parents[m] = m; return TBNAV_OK; }
  const double r = z / static_cast<double>(N);
  double c = wn[0];
  int i = 0;
  for (int m = 0; m < N; ++m) {
    const double U = r + static_cast<double>(m * (1.0 / (N - 1)));
    while (U > c) {
      i++;
      if (i > N - 1) { i = N - 1; break; }
      c += wn[i];
    }
    parents[m] = i;
  }
  return TBNAV_OK;
}

int tbnav_rbpf_add_repeated(const double* x, const double* d, const int32_t* n, double* out, int64_t count) {
  if (!x || !d || !n || !out || count <= 0 || count > (1 << 26)) return TBNAV_ERR_INVALID_ARG;
  double *dx = nullptr, *dd = nullptr, *dout = nullptr;
  int* dn = nullptr;
  int rc = TBNAV_OK;
  auto body = [&]() -> int {
    TBNAV_HIP(hipMalloc((void**)&dx, sizeof(double) * count)); TBNAV_HIP(hipMalloc((void**)&dd, sizeof(double) * count));
    TBNAV_HIP(hipMalloc((void**)&dout, sizeof(double) * count)); TBNAV_HIP(hipMalloc((void**)&dn, sizeof(int) * count));
    TBNAV_HIP(hipMemcpy(dx, x, sizeof(double) * count, hipMemcpyHostToDevice)); TBNAV_HIP(hipMemcpy(dd, d, sizeof(double) * count, hipMemcpyHostToDevice));
    TBNAV_HIP(hipMemcpy(dn, n, sizeof(int) * count, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rbpf_add_repeated_test, dim3((unsigned int)((count + 255) / 256)), dim3(256), 0, 0, dx, dd, dn, dout, (int)count);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipMemcpy(out, dout, sizeof(double) * count, hipMemcpyDeviceToHost));
    return TBNAV_OK;
  };
  rc = body();
  (void)hipFree(dx); (void)hipFree(dd); (void)hipFree(dout); (void)hipFree(dn);
  return rc;
}

int tbnav_rbpf_gather_local(tbnav_rbpf* h, const int32_t* local_parent) {
  if (!h || !local_parent) return TBNAV_ERR_INVALID_ARG;
  if (h->ref_field) return TBNAV_ERR_UNSUPPORTED;  // the reference-field mode is a single-handle mode
  DeviceGuard guard(h->device);
  const int N = h->N;
  // slots with parent -1 keep their own content: copy self
  std::vector<int> par(local_parent, local_parent + N);
  par.resize(2 * (size_t)N, 0);  // [N, 2N): how many slots chose each particle
  for (int m = 0; m < N; ++m) { if (par[m] < 0) par[m] = m; if (par[m] >= N) return TBNAV_ERR_INVALID_ARG; ++par[N + par[m]]; }
  TBNAV_HIP(hipMemcpyAsync(h->d_parent, par.data(), sizeof(int) * 2 * N, hipMemcpyHostToDevice, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // par is a local
  const int rc = resample_on_device(h);
  if (rc != TBNAV_OK) return rc;
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}

// ---- device-side exchange for the sharded filter -------------------------------------------------------------
int tbnav_rbpf_copy_weights_dev(tbnav_rbpf* h, double* d_dst) {
  if (!h || !d_dst) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipMemcpyAsync(d_dst, sp.weight, sizeof(double) * h->N, hipMemcpyDeviceToDevice, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}

int tbnav_rbpf_resample_global_dev(tbnav_rbpf* h, const double* d_weights_all, int64_t n_global, int64_t offset, double z,
                                   int32_t* parents_out, tbnav_rbpf_stats* out) {
  if (!h || !d_weights_all || n_global <= 0 || offset < 0 || offset + h->N > n_global || !out || n_global > (1 << 24)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  if ((size_t)n_global > h->g_cap) {
    (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); h->d_gw = h->d_gcs = nullptr; h->d_gparent = nullptr; h->g_cap = 0;
    (void)hipFree(h->d_gw_raw); h->d_gw_raw = nullptr;  // (the in-library sharded scan sizes its buffers with the same capacity: it re-creates them)
    TBNAV_HIP(hipMalloc((void**)&h->d_gw, sizeof(double) * n_global));
    TBNAV_HIP(hipMalloc((void**)&h->d_gcs, sizeof(double) * n_global));
    TBNAV_HIP(hipMalloc((void**)&h->d_gparent, sizeof(int) * n_global));
    h->g_cap = (size_t)n_global;
  }
  if (!h->d_gz) TBNAV_HIP(hipMalloc((void**)&h->d_gz, sizeof(double)));
  if (z != z) {  // NaN: the offset the last scan's device noise carries (with tbnav_rbpf_set_rng_shard: the ENSEMBLE's, same on every rank)
    if (!h->last_normals || !h->last_z_index) return TBNAV_ERR_INVALID_ARG;
    TBNAV_HIP(hipMemcpyAsync(h->d_gz, h->last_normals + h->last_z_index, sizeof z, hipMemcpyDeviceToDevice, st));
  } else
  TBNAV_HIP(hipMemcpyAsync(h->d_gz, &z, sizeof z, hipMemcpyHostToDevice, st));
  *h->h_norm = NormOut{};
  // the reference's sequential normalise / Neff / selection (particle_filter.cpp:442-500) over the GLOBAL vector:
  // every rank runs the same kernel on the same values, so all ranks agree bit for bit
  hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, st, (int)n_global, h->d_gz, d_weights_all, h->d_gw, h->d_gcs, h->d_gparent, h->d_norm);
  TBNAV_HIP(hipGetLastError());
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipMemcpyAsync(sp.weight, h->d_gw + offset, sizeof(double) * h->N, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  const NormOut no = *h->h_norm;
  std::memset(out, 0, sizeof *out);
  out->sum_w = no.sum_w; out->sq_sum = no.sq_sum; out->neff = no.neff; out->resampled = no.resampled;
  if (no.resampled && parents_out) TBNAV_HIP(hipMemcpy(parents_out, h->d_gparent, sizeof(int) * n_global, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

namespace {
int batch_scratch(tbnav_rbpf* h, size_t n) {
  if (n <= h->batch_cap) return TBNAV_OK;
  (void)hipFree(h->d_bslots); (void)hipFree(h->d_bcount); (void)hipFree(h->d_bitems); (void)hipFree(h->d_bhdr);
  h->d_bslots = nullptr; h->d_bcount = nullptr; h->d_bitems = nullptr; h->d_bhdr = nullptr; h->batch_cap = 0;
  const size_t cap = n + n / 2 + 64;
  TBNAV_HIP(hipMalloc((void**)&h->d_bslots, sizeof(int) * cap));
  TBNAV_HIP(hipMalloc((void**)&h->d_bcount, sizeof(int2) * cap));
  TBNAV_HIP(hipMalloc((void**)&h->d_bitems, sizeof(BatchItem) * cap));
  TBNAV_HIP(hipMalloc((void**)&h->d_bhdr, sizeof(BlobHeader) * cap));
  h->batch_cap = cap;
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_set_weights_from_global_dev(tbnav_rbpf* h, const int32_t* global_parent_of_slot /*[N]*/) {
  // after a resample every slot carries its parent's normalised weight (weights are NOT reset, particle_filter.cpp:495)
  if (!h || !global_parent_of_slot || !h->d_gw) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  for (int m = 0; m < h->N; ++m)
    if (global_parent_of_slot[m] < 0 || (size_t)global_parent_of_slot[m] >= h->g_cap) return TBNAV_ERR_INVALID_ARG;
  { const int rc = batch_scratch(h, (size_t)h->N); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipMemcpyAsync(h->d_bslots, global_parent_of_slot, sizeof(int) * h->N, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(rbpf_gather_weights, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->N, h->d_gw, h->d_bslots, sp.weight);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // (the parent list is the caller's)
  return TBNAV_OK;
}

namespace {
BlobLayout blob_layout(const tbnav_rbpf* h, uint32_t n_tiles, bool has_codes) { return blob_layout_hd(h->TW, h->G, n_tiles, has_codes); }
int slot_tiles(tbnav_rbpf* h, int slot, std::vector<uint32_t>& tidx, std::vector<uint32_t>& ids, int& fstate) {
  std::vector<uint32_t> row(h->TT);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(row.data(), h->d_table[h->cur] + (size_t)slot * h->TT, sizeof(uint32_t) * h->TT, hipMemcpyDeviceToHost));
  TBNAV_HIP(hipMemcpy(&fstate, h->d_fstate + slot, sizeof(int), hipMemcpyDeviceToHost));
  tidx.clear(); ids.clear();
  for (int t = 0; t < h->TT; ++t) if (row[t]) { tidx.push_back((uint32_t)t); ids.push_back(row[t]); }
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_export_size(tbnav_rbpf* h, int32_t slot, uint64_t* bytes) {
  if (!h || !bytes || slot < 0 || slot >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  std::vector<uint32_t> tidx, ids; int fs = 0;
  { const int rc = slot_tiles(h, slot, tidx, ids, fs); if (rc != TBNAV_OK) return rc; }
  *bytes = blob_layout(h, (uint32_t)tidx.size(), fs == 2 && h->d_code[0]).total;
  return TBNAV_OK;
}

int tbnav_rbpf_export_particle_dev(tbnav_rbpf* h, int32_t slot, void* d_buf, uint64_t capacity, uint64_t* bytes) {
  if (!h || !d_buf || slot < 0 || slot >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  std::vector<uint32_t> tidx, ids; int fs = 0;
  { const int rc = slot_tiles(h, slot, tidx, ids, fs); if (rc != TBNAV_OK) return rc; }
  const bool has_codes = fs == 2 && h->d_code[0];
  const uint32_t n = (uint32_t)tidx.size();
  const BlobLayout L = blob_layout(h, n, has_codes);
  if (bytes) *bytes = L.total;
  if (L.total > capacity) return TBNAV_ERR_INVALID_ARG;
  char* b = static_cast<char*>(d_buf);
  BlobHeader hd{kBlobMagic, n, has_codes ? 1u : 0u, 0, fs, (uint32_t)h->xsize, (uint32_t)h->TT};
  TBNAV_HIP(hipMemcpy(&hd.nocc, h->d_nocc[h->cur] + slot, sizeof(int), hipMemcpyDeviceToHost));
  TBNAV_HIP(hipMemcpy(b, &hd, sizeof hd, hipMemcpyHostToDevice));
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  double* bs = reinterpret_cast<double*>(b + L.state);
  TBNAV_HIP(hipMemcpyAsync(bs, sp.pose + (size_t)slot * 3, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(bs + 3, sp.prev + (size_t)slot * 3, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(bs + 6, sp.weight + slot, sizeof(double), hipMemcpyDeviceToDevice, st));
  if (n) {
    TBNAV_HIP(hipMemcpy(b + L.tidx, tidx.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    TBNAV_HIP(hipMemcpy(h->d_tile_scratch, ids.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rbpf_pack_tiles, dim3(n), dim3(256), 0, st, h->pool, h->d_tile_scratch, reinterpret_cast<double*>(b + L.tiles),
                       reinterpret_cast<unsigned int*>(b + L.tile_bm));
    TBNAV_HIP(hipGetLastError());
  }
  TBNAV_HIP(hipMemcpyAsync(b + L.trow, h->d_trow[h->cur] + (size_t)slot * h->TW, sizeof(int) * h->TW, hipMemcpyDeviceToDevice, st));
  if (has_codes) TBNAV_HIP(hipMemcpyAsync(b + L.codes, h->d_code[h->cur] + (size_t)slot * h->G, sizeof(uint16_t) * h->G, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  return TBNAV_OK;
}

int tbnav_rbpf_import_particle_dev(tbnav_rbpf* h, int32_t slot, const void* d_buf, uint64_t bytes) {
  if (!h || !d_buf || slot < 0 || slot >= h->N || bytes < sizeof(BlobHeader)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  const char* b = static_cast<const char*>(d_buf);
  BlobHeader hd{};
  TBNAV_HIP(hipStreamSynchronize(st));
  TBNAV_HIP(hipMemcpy(&hd, b, sizeof hd, hipMemcpyDeviceToHost));
  if (hd.magic != kBlobMagic || hd.xsize != (uint32_t)h->xsize || hd.TT != (uint32_t)h->TT || hd.n_tiles > (uint32_t)h->TT) return TBNAV_ERR_INVALID_ARG;
  const BlobLayout L = blob_layout(h, hd.n_tiles, hd.has_codes != 0);
  if (L.total > bytes) return TBNAV_ERR_INVALID_ARG;
  if (hd.has_codes) { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  const MapT M = map_of(h);
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  hipLaunchKernelGGL(rbpf_release_slot, dim3((h->TT + 255) / 256), dim3(256), 0, st, h->pool, M, slot);
  TBNAV_HIP(hipGetLastError());
  if (hd.n_tiles) {
    hipLaunchKernelGGL(rbpf_unpack_tiles, dim3(hd.n_tiles), dim3(256), 0, st, h->pool, M, slot, reinterpret_cast<const unsigned int*>(b + L.tidx),
                       reinterpret_cast<const double*>(b + L.tiles), reinterpret_cast<const unsigned int*>(b + L.tile_bm), h->d_err);
    TBNAV_HIP(hipGetLastError());
  }
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  const double* bs = reinterpret_cast<const double*>(b + L.state);
  TBNAV_HIP(hipMemcpyAsync(sp.pose + (size_t)slot * 3, bs, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(sp.prev + (size_t)slot * 3, bs + 3, sizeof(double) * 3, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(sp.weight + slot, bs + 6, sizeof(double), hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(h->d_trow[h->cur] + (size_t)slot * h->TW, b + L.trow, sizeof(int) * h->TW, hipMemcpyDeviceToDevice, st));
  TBNAV_HIP(hipMemcpyAsync(h->d_nocc[h->cur] + slot, &hd.nocc, sizeof(int), hipMemcpyHostToDevice, st));
  const int fs = hd.has_codes ? 2 : 0;
  if (hd.has_codes) {
    TBNAV_HIP(hipMemcpyAsync(h->d_code[h->cur] + (size_t)slot * h->G, b + L.codes, sizeof(uint16_t) * h->G, hipMemcpyDeviceToDevice, st));
    h->fstate_dirty = true;
  }
  TBNAV_HIP(hipMemcpyAsync(h->d_fstate + slot, &fs, sizeof(int), hipMemcpyHostToDevice, st));
  TBNAV_HIP(hipStreamSynchronize(st));  // hd / fs are locals
  if (h->h_err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;
  return TBNAV_OK;
}

// ---- the same for many particles at once (what a cross-rank resample needs: hundreds of particles per rank) ------------
namespace {
int count_batch(tbnav_rbpf* h, int32_t n, const int32_t* slots) {
  for (int i = 0; i < n; ++i) if (slots[i] < 0 || slots[i] >= h->N) return TBNAV_ERR_INVALID_ARG;
  { const int rc = batch_scratch(h, (size_t)n); if (rc != TBNAV_OK) return rc; }
  h->batch_counts.resize(n);
  TBNAV_HIP(hipMemcpyAsync(h->d_bslots, slots, sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(rbpf_count_tiles, dim3(n), dim3(256), 0, h->stream, map_of(h), h->d_bslots, h->d_fstate, h->d_bcount);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(h->batch_counts.data(), h->d_bcount, sizeof(int2) * n, hipMemcpyDeviceToHost, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}
}  // namespace

int tbnav_rbpf_export_batch_sizes(tbnav_rbpf* h, int32_t n, const int32_t* slots, uint64_t* sizes_out) {
  if (!h || n < 0 || (n && (!slots || !sizes_out))) return TBNAV_ERR_INVALID_ARG;
  if (n == 0) return TBNAV_OK;
  DeviceGuard guard(h->device);
  { const int rc = count_batch(h, n, slots); if (rc != TBNAV_OK) return rc; }
  for (int i = 0; i < n; ++i)
    sizes_out[i] = blob_layout(h, (uint32_t)h->batch_counts[i].x, h->batch_counts[i].y == 2 && h->d_code[0]).total;
  return TBNAV_OK;
}

int tbnav_rbpf_export_batch_dev(tbnav_rbpf* h, int32_t n, const int32_t* slots, void* d_buf, uint64_t capacity, uint64_t* offsets_out) {
  if (!h || n < 0 || (n && (!slots || !d_buf || !offsets_out))) return TBNAV_ERR_INVALID_ARG;
  if (n == 0) { if (offsets_out) offsets_out[0] = 0; return TBNAV_OK; }
  DeviceGuard guard(h->device);
  // (counted again rather than trusting what tbnav_rbpf_export_batch_sizes saw: a scan in between would change the tables;
  //  a tiny launch and one 8-byte-per-particle copy)
  { const int rc = count_batch(h, n, slots); if (rc != TBNAV_OK) return rc; }
  std::vector<BatchItem> items(n);
  uint64_t off = 0;
  for (int i = 0; i < n; ++i) {
    const bool has_codes = h->batch_counts[i].y == 2 && h->d_code[0];
    items[i] = BatchItem{slots[i], (unsigned int)h->batch_counts[i].x, has_codes ? 1 : 0, 0, off};
    offsets_out[i] = off;
    off += blob_layout(h, items[i].n_tiles, has_codes).total;
  }
  offsets_out[n] = off;
  if (off > capacity) return TBNAV_ERR_INVALID_ARG;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipMemcpyAsync(h->d_bitems, items.data(), sizeof(BatchItem) * n, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(rbpf_pack_batch, dim3(n), dim3(256), 0, h->stream, h->pool, map_of(h), sp.pose, sp.prev, sp.weight, h->d_trow[h->cur],
                     h->d_nocc[h->cur], h->d_fstate, h->d_code[h->cur], h->G, h->xsize, h->d_bitems, static_cast<char*>(d_buf));
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(h->stream));  // (items is a local; the caller sends the buffer next)
  return TBNAV_OK;
}

int tbnav_rbpf_import_batch_dev(tbnav_rbpf* h, int32_t n, const int32_t* slots, const void* d_buf, uint64_t bytes, const uint64_t* offsets) {
  if (!h || n < 0 || (n && (!slots || !d_buf || !offsets))) return TBNAV_ERR_INVALID_ARG;
  if (n == 0) return TBNAV_OK;
  DeviceGuard guard(h->device);
  std::vector<char> seen(h->N, 0);
  std::vector<BatchItem> items(n);
  for (int i = 0; i < n; ++i) {
    if (slots[i] < 0 || slots[i] >= h->N || seen[slots[i]] || offsets[i] + sizeof(BlobHeader) > bytes || (offsets[i] & 7)) return TBNAV_ERR_INVALID_ARG;
    seen[slots[i]] = 1;  // (a slot receives one particle; one particle may fill several slots)
    items[i] = BatchItem{slots[i], 0u, 0, 0, offsets[i]};
  }
  { const int rc = batch_scratch(h, (size_t)n); if (rc != TBNAV_OK) return rc; }
  hipStream_t st = h->stream;
  const char* b = static_cast<const char*>(d_buf);
  TBNAV_HIP(hipMemcpyAsync(h->d_bitems, items.data(), sizeof(BatchItem) * n, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(rbpf_blob_headers, dim3((n + 255) / 256), dim3(256), 0, st, h->d_bitems, b, h->d_bhdr, n);
  TBNAV_HIP(hipGetLastError());
  std::vector<BlobHeader> hd(n);
  TBNAV_HIP(hipMemcpyAsync(hd.data(), h->d_bhdr, sizeof(BlobHeader) * n, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  bool any_codes = false;
  for (int i = 0; i < n; ++i) {
    if (hd[i].magic != kBlobMagic || hd[i].xsize != (uint32_t)h->xsize || hd[i].TT != (uint32_t)h->TT || hd[i].n_tiles > (uint32_t)h->TT) return TBNAV_ERR_INVALID_ARG;
    if (offsets[i] + blob_layout(h, hd[i].n_tiles, hd[i].has_codes != 0).total > bytes) return TBNAV_ERR_INVALID_ARG;
    any_codes |= hd[i].has_codes != 0;
  }
  if (any_codes) { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; h->fstate_dirty = true; }
  {
    // Does the pool hold what is coming?  Checked BEFORE the destination slots give their tiles up: the incoming tiles against
    // the free ones plus every tile the slots name now (an upper bound of what releasing them returns).  Beyond that the import
    // cannot succeed and nothing is touched; inside the bound it goes ahead (tiles the slots share with particles that stay do
    // not come back: the unpack kernel then reports the exhaustion, with the slots' maps already released — see the header).
    uint64_t incoming = 0;
    for (int i = 0; i < n; ++i) incoming += hd[i].n_tiles;
    unsigned long long ctr[2] = {0, 0};
    TBNAV_HIP(hipMemcpy(ctr, h->pool.ctr, sizeof ctr, hipMemcpyDeviceToHost));
    const uint64_t free_now = ctr[1] - ctr[0];
    if (incoming > free_now) {
      std::vector<int> sl(n);
      for (int i = 0; i < n; ++i) sl[i] = slots[i];
      { const int rc = count_batch(h, n, sl.data()); if (rc != TBNAV_OK) return rc; }
      uint64_t named = 0;
      for (int i = 0; i < n; ++i) named += (uint64_t)h->batch_counts[i].x;
      if (incoming > free_now + named) return TBNAV_ERR_POOL_EXHAUSTED;
      TBNAV_HIP(hipMemcpyAsync(h->d_bitems, items.data(), sizeof(BatchItem) * n, hipMemcpyHostToDevice, st));  // (count_batch reused the scratch's slot list only)
    }
  }
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  const MapT M = map_of(h);
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  hipLaunchKernelGGL(rbpf_release_slots, dim3(n), dim3(256), 0, st, h->pool, M, h->d_bitems);  // pushes: all before the first pop
  TBNAV_HIP(hipGetLastError());
  hipLaunchKernelGGL(rbpf_unpack_batch, dim3(n), dim3(256), 0, st, h->pool, M, sp.pose, sp.prev, sp.weight, h->d_trow[h->cur], h->d_nocc[h->cur],
                     h->d_fstate, h->d_code[h->cur], h->G, h->d_bitems, b, h->d_err);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(st));
  if (h->h_err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;
  return TBNAV_OK;
}

}  // extern "C"

// =================================================================================================
// The sharded filter inside the library (include/tbnav_comm.h)
// =================================================================================================
namespace {

int ensure_shard_state(tbnav_rbpf* h) {
  const int P = tbnav::comm_size(h->comm);
  const size_t ng = (size_t)P * h->N;
  if (!h->stream2) TBNAV_HIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  if (!h->ev_w) TBNAV_HIP(hipEventCreateWithFlags(&h->ev_w, hipEventDisableTiming));
  if (!h->ev_g) TBNAV_HIP(hipEventCreateWithFlags(&h->ev_g, hipEventDisableTiming));
  if (ng > h->g_cap || !h->d_gw_raw) {
    TBNAV_HIP(hipStreamSynchronize(h->stream));
    TBNAV_HIP(hipStreamSynchronize(h->stream2));
    (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); (void)hipFree(h->d_gw_raw); (void)hipFree(h->d_sizes); (void)hipFree(h->d_status);
    h->d_gw = h->d_gcs = h->d_gw_raw = nullptr; h->d_gparent = nullptr; h->d_sizes = nullptr; h->d_status = nullptr; h->g_cap = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_gw, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gcs, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gw_raw, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gparent, sizeof(int) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_sizes, sizeof(unsigned long long) * ((size_t)h->N + ng)));
    TBNAV_HIP(hipMalloc((void**)&h->d_status, sizeof(int) * (1 + (size_t)P)));
    h->g_cap = ng;
  }
  return TBNAV_OK;
}

int grow(char*& buf, size_t& cap, size_t need) {
  if (need <= cap) return TBNAV_OK;
  (void)hipFree(buf); buf = nullptr; cap = 0;
  const size_t want = need + need / 4 + 4096;
  TBNAV_HIP(hipMalloc((void**)&buf, want));
  cap = want;
  return TBNAV_OK;
}

// ParticleFilter::SLAM over the members' shards.  n == 1: this process's rank of a multi-process filter; n > 1: every member of a
// one-process group, in rank order.  Per scan and member, on the device:
//   main stream   noise -> propose -> [event: weights final] -> map update ........................ -> (resample: migration)
//   second stream                      wait -> ONE all-gather of the raw weights -> the reference's sequential normalise /
//                                      Neff / selection over the GLOBAL vector (identical on every rank) -> own slice back
// so the chain of adds of the global normalise (which grows with the ensemble, not with the shard) runs BESIDE the local map
// update, and the host waits once, for both streams.  Only when resampling fires do particles move: one all-gather of blob
// sizes, one batched export per rank, one message per (source, destination) pair, one batched import (tbnav_rbpf_export_batch_*
// / _import_batch_dev), and an all-gather of the ranks' statuses so that a rank whose pool is exhausted stops everybody.
int sharded_scan(int n, tbnav_rbpf* const* hs, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* const* normals, tbnav_rbpf_stats* out,
                 tbnav_rbpf_stats* local_out) {
  if (n <= 0 || !hs || !scan || n_beams <= 0 || !u || !cur_odom || !prev_odom || !T_icp || !out) return TBNAV_ERR_INVALID_ARG;
  for (int r = 0; r < n; ++r) if (!hs[r] || !hs[r]->comm || hs[r]->N != hs[0]->N || hs[r]->ref_field) return TBNAV_ERR_INVALID_ARG;
  const int P = tbnav::comm_size(hs[0]->comm), nl = hs[0]->N;
  const size_t ng = (size_t)P * nl;
  if (ng > ((size_t)1 << 24)) return TBNAV_ERR_UNSUPPORTED;
  // (everything above is a function of arguments every rank shares: all ranks return together.  From here on a failure that
  //  only THIS rank sees — a launch that fails, an allocation, a pool that runs dry — must not make it leave while its peers
  //  wait in a collective that has no timeout: the rank notes the code in lerr[], skips its own work, KEEPS JOINING the
  //  collectives, and the ranks agree on a status before anyone acts on data that may be missing.  Round-3 advisor finding.)
  std::vector<tbnav_comm*> comms(n);
  std::vector<hipStream_t> s1(n), s2(n);
  std::vector<ScanTicket> tk(n);
  std::vector<tbnav_rbpf_stats> lst(n);
  std::vector<int> lerr(n, TBNAV_OK);
  auto note = [&](int r, int rc) { if (rc != TBNAV_OK && lerr[r] == TBNAV_OK) lerr[r] = rc; };
  auto hipok = [&](int r, hipError_t e, const char* what, int line) { if (e != hipSuccess) note(r, tbnav::hip_fail(e, what, __FILE__, line)); return e == hipSuccess; };
#define TBNAV_L(r, call) hipok(r, (call), #call, __LINE__)
  // the ranks' codes -> one status, the same on every rank (the lowest rank's failure); collective when ranks live elsewhere
  auto agree = [&](const std::vector<int>& codes, int& status) -> int {
    status = TBNAV_OK;
    if (n == P) { for (int r = 0; r < n; ++r) if (codes[r] != TBNAV_OK && status == TBNAV_OK) status = codes[r]; return TBNAV_OK; }
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    for (int r = 0; r < n; ++r) {
      DeviceGuard guard(hs[r]->device);
      TBNAV_HIP(hipMemcpyAsync(hs[r]->d_status, &codes[r], sizeof(int), hipMemcpyHostToDevice, hs[r]->stream));
      send[r] = hs[r]->d_status; recv[r] = hs[r]->d_status + 1;
    }
    { const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(int), s1.data()); if (rc != TBNAV_OK) return rc; }
    std::vector<int> all(P);
    { DeviceGuard guard(hs[0]->device);
      TBNAV_HIP(hipMemcpyAsync(all.data(), hs[0]->d_status + 1, sizeof(int) * P, hipMemcpyDeviceToHost, hs[0]->stream));
      TBNAV_HIP(hipStreamSynchronize(hs[0]->stream)); }
    for (int q = 0; q < P; ++q) if (all[q] != TBNAV_OK) { status = all[q]; break; }
    return TBNAV_OK;
  };
  std::memset(out, 0, sizeof *out);
  // ---- A: every member's local scan (no normalise tail), its "weights are final" event recorded behind the proposal kernel
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    comms[r] = h->comm; s1[r] = h->stream; s2[r] = h->stream2;
    note(r, ensure_shard_state(h));   // (sized at attach: a no-op here unless the handle was resized since)
    if (!h->d_gw_raw || !h->d_status) { out->status = lerr[r]; return lerr[r]; }   // nothing to join a collective WITH: only before the first scan, at attach
    comms[r] = h->comm; s1[r] = h->stream; s2[r] = h->stream2;
    if (lerr[r] == TBNAV_OK)
      note(r, scan_enqueue(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals ? normals[r] : nullptr, &lst[r], true, 0, nullptr, tk[r], nullptr, h->ev_w));
    if (lerr[r] == TBNAV_OK) TBNAV_L(r, hipStreamWaitEvent(h->stream2, h->ev_w, 0));
  }
  // ---- B: the ONE collective of the update + the global normalise / selection, on the second streams
  //         (a member that failed above still takes part — with whatever its weight buffer holds: nobody will use the result)
  {
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    for (int r = 0; r < n; ++r) { send[r] = state_ptrs(hs[r]->d_state[hs[r]->cur], nl).weight; recv[r] = hs[r]->d_gw_raw; }
    const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(double) * nl, s2.data());
    if (rc != TBNAV_OK) return rc;   // (the communicator itself failed: it reports on every rank)
  }
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    if (lerr[r] != TBNAV_OK) continue;
    DeviceGuard guard(h->device);
    h->h_norm[1] = NormOut{};
    // the resampling offset: the scan's last normal — with tbnav_rbpf_set_rng_shard (device noise) the ENSEMBLE's, identical on every rank
    const double* zp = h->last_normals + h->last_z_index;
    hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, h->stream2, (int)ng, zp, h->d_gw_raw, h->d_gw, h->d_gcs, h->d_gparent, h->d_norm + 1,
                       nullptr, nullptr, nullptr, 0u);
    const size_t off = (size_t)tbnav::comm_rank(h->comm) * nl;
    (void)(TBNAV_L(r, hipGetLastError()) &&
           TBNAV_L(r, hipMemcpyAsync(state_ptrs(h->d_state[h->cur], nl).weight, h->d_gw + off, sizeof(double) * nl, hipMemcpyDeviceToDevice, h->stream2)) &&
           TBNAV_L(r, hipEventRecord(h->ev_g, h->stream2)) &&
           TBNAV_L(r, hipStreamWaitEvent(h->stream, h->ev_g, 0)));  // whatever the main stream does next sees the normalised weights
  }
  // ---- C: the host waits once per member (the reference's SLAM() is synchronous)
  std::vector<int> lstat(n, TBNAV_OK);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    TBNAV_L(r, hipStreamSynchronize(h->stream2));
    // what the reference reports by throwing (a particle left the world, eta is 0 ...) happens to the rank that holds the particle
    lstat[r] = lerr[r] != TBNAV_OK ? lerr[r] : scan_finish(h, tk[r], &lst[r]);
    if (local_out) local_out[r] = lst[r];
  }
  // Every rank must stop at the SAME scan with the same status — a rank that went on alone would sit in the next scan's
  // all-gather for ever: one all-gather of the ranks' statuses per scan (4 bytes each; ~1 % of a scan) when ranks live elsewhere.
  int status = TBNAV_OK;
  { const int rc = agree(lstat, status); if (rc != TBNAV_OK) return rc; }
  const NormOut no = hs[0]->h_norm[1];
  out->sum_w = no.sum_w; out->sq_sum = no.sq_sum; out->neff = no.neff; out->resampled = no.resampled;
  out->n_valid_beams = lst[0].n_valid_beams;
  out->status = status;
  if (status != TBNAV_OK) return status;
  if (!no.resampled) return TBNAV_OK;
  // ---- D: lowVarianceResampling's copies across shards.  Slot m (global) takes particle parents[m].
  std::vector<int> parents(ng);
  { DeviceGuard guard(hs[0]->device); if (!TBNAV_L(0, hipMemcpy(parents.data(), hs[0]->d_gparent, sizeof(int) * ng, hipMemcpyDeviceToHost))) std::fill(parents.begin(), parents.end(), 0); }
  struct Plan { std::vector<std::pair<int, int>> sends, recvs; std::vector<int32_t> send_slots; std::vector<uint64_t> send_sizes, send_offs; std::vector<unsigned long long> sizes_local; };
  std::vector<Plan> plan(n);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const int me = tbnav::comm_rank(h->comm), lo = me * nl;
    Plan& pl = plan[r];
    for (size_t m = 0; m < ng; ++m) {  // (dst, q): every particle of mine some other rank's slot chose — once per destination
      const int q = parents[m], dst = (int)(m / nl);
      if (q / nl == me && dst != me) pl.sends.emplace_back(dst, q);
    }
    std::sort(pl.sends.begin(), pl.sends.end());
    pl.sends.erase(std::unique(pl.sends.begin(), pl.sends.end()), pl.sends.end());
    for (int m = lo; m < lo + nl; ++m) { const int q = parents[m]; if (q / nl != me) pl.recvs.emplace_back(q / nl, q); }
    std::sort(pl.recvs.begin(), pl.recvs.end());
    pl.recvs.erase(std::unique(pl.recvs.begin(), pl.recvs.end()), pl.recvs.end());
    pl.send_slots.resize(pl.sends.size());
    for (size_t i = 0; i < pl.sends.size(); ++i) pl.send_slots[i] = pl.sends[i].second - lo;
    pl.send_sizes.assign(pl.sends.size(), 0);
    if (lerr[r] == TBNAV_OK) note(r, tbnav_rbpf_export_batch_sizes(h, (int32_t)pl.sends.size(), pl.send_slots.data(), pl.send_sizes.data()));
    if (lerr[r] != TBNAV_OK) std::fill(pl.send_sizes.begin(), pl.send_sizes.end(), 0);
    // what a particle of mine weighs, for whoever receives it (a particle sent to several ranks weighs the same for each)
    pl.sizes_local.assign(nl, 0ull);
    for (size_t i = 0; i < pl.sends.size(); ++i) pl.sizes_local[pl.sends[i].second - lo] = pl.send_sizes[i];
    TBNAV_L(r, hipMemcpyAsync(h->d_sizes, pl.sizes_local.data(), sizeof(unsigned long long) * nl, hipMemcpyHostToDevice, h->stream));
  }
  {
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    for (int r = 0; r < n; ++r) { send[r] = hs[r]->d_sizes; recv[r] = hs[r]->d_sizes + nl; }
    const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(unsigned long long) * nl, s1.data());
    if (rc != TBNAV_OK) return rc;
  }
  std::vector<std::vector<tbnav::P2P>> p_send(n), p_recv(n);
  std::vector<std::vector<uint64_t>> recv_offs(n);
  std::vector<unsigned long long> sizes_all(ng);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    Plan& pl = plan[r];
    if (!(TBNAV_L(r, hipMemcpyAsync(sizes_all.data(), h->d_sizes + nl, sizeof(unsigned long long) * ng, hipMemcpyDeviceToHost, h->stream)) &&
          TBNAV_L(r, hipStreamSynchronize(h->stream)))) std::fill(sizes_all.begin(), sizes_all.end(), 0ull);
    // everything this rank sends: ONE export, the blobs back to back in (destination, particle) order
    uint64_t total = 0;
    for (uint64_t b : pl.send_sizes) total += b;
    pl.send_offs.assign(pl.sends.size() + 1, 0);
    if (lerr[r] == TBNAV_OK) note(r, grow(h->d_sendbuf, h->send_cap, (size_t)total));
    if (lerr[r] == TBNAV_OK) note(r, tbnav_rbpf_export_batch_dev(h, (int32_t)pl.sends.size(), pl.send_slots.data(), h->d_sendbuf, total, pl.send_offs.data()));
    for (size_t i = 0; i < pl.sends.size();) {  // one message per destination
      size_t j = i;
      while (j < pl.sends.size() && pl.sends[j].first == pl.sends[i].first) ++j;
      p_send[r].push_back(tbnav::P2P{pl.sends[i].first, h->d_sendbuf + pl.send_offs[i], (size_t)(pl.send_offs[j] - pl.send_offs[i])});
      i = j;
    }
    // everything it receives: one buffer, the blobs in (source, particle) order
    recv_offs[r].assign(pl.recvs.size() + 1, 0);
    for (size_t i = 0; i < pl.recvs.size(); ++i) recv_offs[r][i + 1] = recv_offs[r][i] + sizes_all[pl.recvs[i].second];
    if (lerr[r] == TBNAV_OK) note(r, grow(h->d_recvbuf, h->recv_cap, (size_t)recv_offs[r].back()));
    for (size_t i = 0; i < pl.recvs.size();) {
      size_t j = i;
      while (j < pl.recvs.size() && pl.recvs[j].first == pl.recvs[i].first) ++j;
      p_recv[r].push_back(tbnav::P2P{pl.recvs[i].first, h->d_recvbuf + recv_offs[r][i], (size_t)(recv_offs[r][j] - recv_offs[r][i])});
      i = j;
    }
  }
  // is every rank ready to send what the sizes promised and to receive it?  A rank whose export or allocation failed cannot
  // honour its messages (its peers would wait for bytes that never come): agree BEFORE the exchange; nobody has touched a slot yet
  { const int rc = agree(lerr, status); if (rc != TBNAV_OK) return rc; }
  if (status != TBNAV_OK) { out->status = status; return status; }
  { const int rc = tbnav::comm_exchange(n, comms.data(), p_send.data(), p_recv.data(), s1.data()); if (rc != TBNAV_OK) return rc; }
  // local parents inside the handle (tile tables + reference counts), then the imported ones; weights are NOT reset by the
  // reference: every slot carries its parent's normalised weight
  std::vector<int> mstat(n, TBNAV_OK);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const int me = tbnav::comm_rank(h->comm), lo = me * nl;
    Plan& pl = plan[r];
    std::vector<int32_t> local_parent(nl), imp_slots;
    std::vector<uint64_t> imp_offs;
    for (int m = 0; m < nl; ++m) {
      const int q = parents[lo + m];
      if (q / nl == me) local_parent[m] = q - lo;
      else {
        local_parent[m] = -1;
        const auto it = std::lower_bound(pl.recvs.begin(), pl.recvs.end(), std::make_pair(q / nl, q));
        imp_slots.push_back(m);
        imp_offs.push_back(recv_offs[r][(size_t)(it - pl.recvs.begin())]);
      }
    }
    int rc = tbnav_rbpf_gather_local(h, local_parent.data());
    if (rc == TBNAV_OK && !imp_slots.empty())
      rc = tbnav_rbpf_import_batch_dev(h, (int32_t)imp_slots.size(), imp_slots.data(), h->d_recvbuf, recv_offs[r].back(), imp_offs.data());
    if (rc == TBNAV_OK) rc = tbnav_rbpf_set_weights_from_global_dev(h, parents.data() + lo);
    mstat[r] = rc;
  }
  // a rank that failed (tile pool exhausted) must not leave the others waiting in the next scan's collective: agree on it
  { const int rc = agree(mstat, status); if (rc != TBNAV_OK) return rc; }
#undef TBNAV_L
  out->status = status;
  return status;
}

}  // namespace

// One process driving several GPUs: the whole filter behind one object (what bmapping::ParticleFilter built with n_gpus > 1 holds).
struct tbnav_rbpf_group {
  int n = 0, n_global = 0;
  std::vector<tbnav_rbpf*> m;
  std::vector<tbnav_comm*> c;
  std::vector<std::vector<double>> normals;  // parity mode: each member's slice of the ensemble's draw stream + the offset
};

extern "C" {

int tbnav_rbpf_attach_comm(tbnav_rbpf* h, tbnav_comm* comm) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (comm && (h->ref_field || tbnav_comm_device(comm) != h->device)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  h->comm = comm;
  if (!comm) { h->rng_first = 0; h->rng_n_global = 0; return TBNAV_OK; }
  // equal shards: this rank's particles are [rank * N, (rank + 1) * N) of nranks * N — also for the device noise source
  h->rng_first = (uint64_t)tbnav::comm_rank(comm) * (uint64_t)h->N;
  h->rng_n_global = (uint64_t)tbnav::comm_size(comm) * (uint64_t)h->N;
  return ensure_shard_state(h);
}

void tbnav_rbpf_group_destroy(tbnav_rbpf_group* g) {
  if (!g) return;
  for (int r = 0; r < g->n; ++r) {
    if (r < (int)g->m.size()) tbnav_rbpf_destroy(g->m[r]);
    if (r < (int)g->c.size()) tbnav_comm_destroy(g->c[r]);
  }
  delete g;
}

int tbnav_rbpf_group_create(const tbnav_rbpf_params* params, int32_t n_gpus, const int32_t* devices, uint64_t max_pool_bytes_per_member,
                            tbnav_rbpf_group** out) {
  if (!params || !out || n_gpus <= 0 || params->num_particles <= 0 || params->num_particles % n_gpus != 0) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  tbnav_rbpf_group* g = new (std::nothrow) tbnav_rbpf_group();
  if (!g) return TBNAV_ERR_INVALID_ARG;
  g->n = n_gpus; g->n_global = params->num_particles;
  g->m.assign(n_gpus, nullptr); g->c.assign(n_gpus, nullptr); g->normals.resize(n_gpus);
  int rc = tbnav_comm_create_local(n_gpus, devices, g->c.data());
  for (int r = 0; r < n_gpus && rc == TBNAV_OK; ++r) {
    tbnav_rbpf_params p = *params;
    p.num_particles = params->num_particles / n_gpus;
    p.device = tbnav_comm_device(g->c[r]);
    rc = create_impl(&p, max_pool_bytes_per_member, &g->m[r]);
    if (rc == TBNAV_OK) {
      // initParticleSet gives every particle weight 1 / N of the WHOLE filter (particle_filter.cpp:134)
      std::vector<double> w((size_t)p.num_particles, 1.0 / params->num_particles);
      rc = tbnav_rbpf_set_particles(g->m[r], nullptr, nullptr, w.data());
    }
    if (rc == TBNAV_OK) rc = tbnav_rbpf_attach_comm(g->m[r], g->c[r]);
  }
  if (rc != TBNAV_OK) { tbnav_rbpf_group_destroy(g); return rc; }
  *out = g;
  return TBNAV_OK;
}

int tbnav_rbpf_group_size(const tbnav_rbpf_group* g) { return g ? g->n : -1; }
int tbnav_rbpf_group_member(tbnav_rbpf_group* g, int32_t rank, tbnav_rbpf** out) {
  if (!g || !out || rank < 0 || rank >= g->n) return TBNAV_ERR_INVALID_ARG;
  *out = g->m[rank];
  return TBNAV_OK;
}
int tbnav_rbpf_group_set_seed(tbnav_rbpf_group* g, uint64_t seed) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_rbpf* h : g->m) { const int rc = tbnav_rbpf_set_seed(h, seed); if (rc != TBNAV_OK) return rc; }  // one seed: the members draw disjoint slices of its stream
  return TBNAV_OK;
}
int tbnav_rbpf_group_set_option(tbnav_rbpf_group* g, int32_t option, int32_t value) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_rbpf* h : g->m) { const int rc = tbnav_rbpf_set_option(h, option, value); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int64_t tbnav_rbpf_group_num_normals(const tbnav_rbpf_group* g, int32_t icp_ok) {
  if (!g) return -1;
  return (int64_t)g->n_global * (icp_ok ? 3 * g->m[0]->k + 3 : 3) + 1;
}

// normals: the ENSEMBLE's draw stream in the reference's order (tbnav_rbpf_group_num_normals values: particle-major, the
// resampling offset last) or NULL (device noise: every member draws its slice of one stream).
int tbnav_rbpf_group_slam(tbnav_rbpf_group* g, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                          const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals, tbnav_rbpf_stats* out) {
  if (!g || !out) return TBNAV_ERR_INVALID_ARG;
  std::vector<const double*> nr(g->n, nullptr);
  if (normals) {
    const size_t stride = icp_ok ? 3 * (size_t)g->m[0]->k + 3 : 3, nl = (size_t)g->m[0]->N;
    for (int r = 0; r < g->n; ++r) {
      std::vector<double>& v = g->normals[r];
      v.resize(nl * stride + 1);
      std::memcpy(v.data(), normals + (size_t)r * nl * stride, sizeof(double) * nl * stride);
      v[nl * stride] = normals[(size_t)g->n_global * stride];
      nr[r] = v.data();
    }
  }
  return sharded_scan(g->n, g->m.data(), scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals ? nr.data() : nullptr, out, nullptr);
}

// ParticleFilter::getRobotState over the ensemble: strict >, first wins (particle_filter.cpp:255-274) — members in rank order
int tbnav_rbpf_group_best_state(tbnav_rbpf_group* g, double pose[3], int32_t* best_index) {
  if (!g || !pose) return TBNAV_ERR_INVALID_ARG;
  double best_w = 0.0; int best_r = 0, best_i = 0; double best_pose[3] = {0, 0, 0};
  bool have = false;
  for (int r = 0; r < g->n; ++r) {
    double p[3]; int32_t idx = 0;
    int rc = tbnav_rbpf_best_state(g->m[r], p, &idx);
    if (rc != TBNAV_OK) return rc;
    double w = 0.0;
    { DeviceGuard guard(g->m[r]->device); TBNAV_HIP(hipMemcpy(&w, state_ptrs(g->m[r]->d_state[g->m[r]->cur], g->m[r]->N).weight + idx, sizeof w, hipMemcpyDeviceToHost)); }
    // (a member whose weights are all <= 0.0 reports its slot 0, as the reference's loop would keep index 0)
    if (!have || w > best_w) { best_w = w; best_r = r; best_i = idx; std::memcpy(best_pose, p, sizeof p); have = true; }
  }
  std::memcpy(pose, best_pose, sizeof best_pose);
  if (best_index) *best_index = best_r * g->m[0]->N + best_i;
  return TBNAV_OK;
}
int tbnav_rbpf_group_best_map(tbnav_rbpf_group* g, int8_t* map) {
  if (!g || !map) return TBNAV_ERR_INVALID_ARG;
  double pose[3]; int32_t idx = 0;
  const int rc = tbnav_rbpf_group_best_state(g, pose, &idx);
  if (rc != TBNAV_OK) return rc;
  const int nl = g->m[0]->N;
  return tbnav_rbpf_particle_map(g->m[idx / nl], idx % nl, map);
}

int tbnav_rbpf_copy_particle(tbnav_rbpf* dst, int32_t dst_slot, tbnav_rbpf* src, int32_t src_slot) {
  if (!dst || !src || dst_slot < 0 || dst_slot >= dst->N || src_slot < 0 || src_slot >= src->N) return TBNAV_ERR_INVALID_ARG;
  if (dst->xsize != src->xsize || dst->ref_field != src->ref_field || dst->device != src->device) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(src->device);
  uint64_t bytes = 0;
  int rc = tbnav_rbpf_export_size(src, src_slot, &bytes);
  if (rc != TBNAV_OK) return rc;
  void* buf = nullptr;
  TBNAV_HIP(hipMalloc(&buf, bytes));
  rc = tbnav_rbpf_export_particle_dev(src, src_slot, buf, bytes, nullptr);
  if (rc == TBNAV_OK) rc = tbnav_rbpf_import_particle_dev(dst, dst_slot, buf, bytes);
  (void)hipFree(buf);
  if (rc == TBNAV_OK && src->ref_field) {  // the set with its history, the field with its stale cells
    dst->ref->copy_slot(dst_slot, *src->ref, src_slot);
    if ((size_t)dst_slot < dst->ref_on_dev.size()) dst->ref_on_dev[dst_slot].reset();
  }
  return rc;
}

int tbnav_rbpf_get_particles(tbnav_rbpf* h, double* pose, double* prev_pose, double* weight) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const int N = h->N;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], N);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (pose) TBNAV_HIP(hipMemcpy(pose, sp.pose, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
  if (prev_pose) TBNAV_HIP(hipMemcpy(prev_pose, sp.prev, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
  if (weight) TBNAV_HIP(hipMemcpy(weight, sp.weight, sizeof(double) * N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_set_particles(tbnav_rbpf* h, const double* pose, const double* prev_pose, const double* weight) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const int N = h->N;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], N);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (pose) TBNAV_HIP(hipMemcpy(sp.pose, pose, sizeof(double) * 3 * N, hipMemcpyHostToDevice));
  if (prev_pose) TBNAV_HIP(hipMemcpy(sp.prev, prev_pose, sizeof(double) * 3 * N, hipMemcpyHostToDevice));
  if (weight) TBNAV_HIP(hipMemcpy(sp.weight, weight, sizeof(double) * N, hipMemcpyHostToDevice));
  return TBNAV_OK;
}

int tbnav_rbpf_get_log_odds(tbnav_rbpf* h, int32_t particle, double* out) {
  if (!h || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  if (!h->d_dense) TBNAV_HIP(hipMalloc((void**)&h->d_dense, sizeof(double) * h->G));
  const int blocks = (int)std::min<size_t>((h->G + 255) / 256, 4096);
  hipLaunchKernelGGL(rbpf_tiles_to_dense, dim3(blocks), dim3(256), 0, h->stream, h->xsize, h->G, h->pool, map_of(h), particle, h->d_dense);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(out, h->d_dense, sizeof(double) * h->G, hipMemcpyDeviceToHost, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return TBNAV_OK;
}

int tbnav_rbpf_set_log_odds(tbnav_rbpf* h, int32_t particle, const double* in) {
  if (!h || !in || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  if (!h->d_dense) TBNAV_HIP(hipMalloc((void**)&h->d_dense, sizeof(double) * h->G));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(h->d_dense, in, sizeof(double) * h->G, hipMemcpyHostToDevice));
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  // the tiles take the new log-odds and the occupancy bits they imply; the particle's occupied counts are rebuilt
  TBNAV_HIP(hipMemsetAsync(h->d_nocc[h->cur] + particle, 0, sizeof(int), h->stream));
  TBNAV_HIP(hipMemsetAsync(h->d_trow[h->cur] + (size_t)particle * h->TW, 0, sizeof(int) * h->TW, h->stream));
  hipLaunchKernelGGL(rbpf_dense_to_tiles, dim3(h->TT), dim3(kWave), 0, h->stream, h->xsize, h->cut_occ, h->pool, map_of(h), particle, h->d_dense,
                     h->d_trow[h->cur], h->d_nocc[h->cur], h->d_err);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (h->h_err[3] & 8) return TBNAV_ERR_POOL_EXHAUSTED;
  const int zero = 0;  // the distance field no longer matches the map
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &zero, sizeof zero, hipMemcpyHostToDevice));
  if (h->ref_field) {  // the occupied set's history is unknown from here on: ascending order (documented in tbnav_rbpf.h)
    std::vector<int> cells;
    for (size_t c = 0; c < h->G; ++c) if (in[c] >= h->cut_occ) cells.push_back((int)c);
    h->ref->reset(particle, cells);
    if ((size_t)particle < h->ref_on_dev.size()) h->ref_on_dev[particle].reset();
  }
  return TBNAV_OK;
}

int tbnav_rbpf_get_dist_code(tbnav_rbpf* h, int32_t particle, uint16_t* out) {
  if (!h || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  { const int rc = ensure_full_field(h, particle); if (rc != TBNAV_OK) return rc; }  // whole field on demand
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(out, h->d_code[h->cur] + (size_t)particle * h->G, sizeof(uint16_t) * h->G, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_get_occ_dist(tbnav_rbpf* h, int32_t particle, double* out) {
  if (!h || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  std::vector<uint16_t> code(h->G);
  int rc = tbnav_rbpf_get_dist_code(h, particle, code.data());
  if (rc != TBNAV_OK) return rc;
  for (size_t c = 0; c < h->G; ++c)
    out[c] = code[c] == kCodeUnreached ? h->max_occ_dist : std::sqrt((double)code[c]) * h->p.resolution;
  return TBNAV_OK;
}

int tbnav_rbpf_set_occ_dist(tbnav_rbpf* h, int32_t particle, const double* in) {
  if (!h || !in || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  std::vector<uint16_t> code(h->G);
  const double res = h->p.resolution;
  for (size_t c = 0; c < h->G; ++c) {
    const double v = in[c];
    const double cells = v / res;
    const long d2 = std::lround(cells * cells);
    if (d2 >= 0 && d2 < 65535 && std::sqrt((double)d2) * res == v) { code[c] = (uint16_t)d2; continue; }
    if (v == h->max_occ_dist) { code[c] = kCodeUnreached; continue; }
    return TBNAV_ERR_INVALID_ARG;
  }
  DeviceGuard guard(h->device);
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (h->ref_field) { h->ref->set_codes(particle, code.data()); if ((size_t)particle < h->ref_on_dev.size()) h->ref_on_dev[particle].reset(); }
  TBNAV_HIP(hipMemcpy(h->d_code[h->cur] + (size_t)particle * h->G, code.data(), sizeof(uint16_t) * h->G, hipMemcpyHostToDevice));
  const int two = 2;  // an injected field is authoritative: the next call does not refresh it
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
  h->fstate_dirty = true;
  return TBNAV_OK;
}

int tbnav_rbpf_get_occupied_count(tbnav_rbpf* h, int32_t* counts) {
  if (!h || !counts) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(counts, h->d_nocc[h->cur], sizeof(int) * h->N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_get_trace(tbnav_rbpf* h, double* sampled, double* p_scan, double* p_pose, double* mu, double* sigma,
                         double* eta, double* new_pose, double* weight_raw, int32_t* resample_parent) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  const size_t N = h->N, k = h->k;
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  auto get = [&](double* dst, const double* src, size_t n) -> hipError_t {
    return dst ? hipMemcpy(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost) : hipSuccess;
  };
  TBNAV_HIP(get(sampled, h->tr.sampled, N * k * 3));
  TBNAV_HIP(get(p_scan, h->tr.p_scan, N * k));
  TBNAV_HIP(get(p_pose, h->tr.p_pose, N * k));
  TBNAV_HIP(get(mu, h->tr.mu, N * 3));
  TBNAV_HIP(get(sigma, h->tr.sigma, N * 9));
  TBNAV_HIP(get(eta, h->tr.eta, N));
  TBNAV_HIP(get(new_pose, h->tr.new_pose, N * 3));
  TBNAV_HIP(get(weight_raw, h->tr.weight_raw, N));
  if (resample_parent) TBNAV_HIP(hipMemcpy(resample_parent, h->d_parent, sizeof(int) * N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

int tbnav_rbpf_best_state(tbnav_rbpf* h, double pose[3], int32_t* best_index) {
  if (!h || !pose) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  hipLaunchKernelGGL(rbpf_argmax, dim3(1), dim3(256), 0, st, h->N, sp.weight, sp.pose, h->d_best, h->d_best_pose);
  TBNAV_HIP(hipGetLastError());
  int idx = 0;
  TBNAV_HIP(hipMemcpyAsync(pose, h->d_best_pose, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipMemcpyAsync(&idx, h->d_best, sizeof(int), hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  if (best_index) *best_index = idx;
  return TBNAV_OK;
}

int tbnav_rbpf_best_map(tbnav_rbpf* h, int8_t* map) {
  if (!h || !map) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  hipLaunchKernelGGL(rbpf_argmax, dim3(1), dim3(256), 0, st, h->N, sp.weight, sp.pose, h->d_best, h->d_best_pose);
  TBNAV_HIP(hipGetLastError());
  const int blocks = (int)std::min<size_t>((h->G + 255) / 256, 2048);
  hipLaunchKernelGGL(rbpf_export_map, dim3(blocks), dim3(256), 0, st, h->xsize, h->G, h->cuts, h->d_best, h->pool, map_of(h),
                     h->d_export);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(map, h->d_export, h->G, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  return TBNAV_OK;
}

int tbnav_rbpf_set_scan_matching(tbnav_rbpf* h, int32_t enable, double lstep, double astep, int32_t iterations) {
  if (!h || (enable && (!(lstep > 0.0) || !(astep > 0.0) || iterations < 1 || iterations > 32))) return TBNAV_ERR_INVALID_ARG;
  h->sm_on = enable != 0;
  if (enable) { h->sm.lstep = lstep; h->sm.astep = astep; h->sm.iters = iterations; h->sm.max_moves = 64; }
  return TBNAV_OK;
}

int tbnav_rbpf_get_scan_match(tbnav_rbpf* h, double* centers, double* scores) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (centers) TBNAV_HIP(hipMemcpy(centers, h->d_center, sizeof(double) * 3 * h->N, hipMemcpyDeviceToHost));
  if (scores) TBNAV_HIP(hipMemcpy(scores, h->d_score, sizeof(double) * h->N, hipMemcpyDeviceToHost));
  return TBNAV_OK;
}

// ---- one particle's GridMapper, for the host class bmapping::GridMapper (grid_mapper.hpp:128-140) ---------------
namespace {
int one_particle_consts(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, ScanC& c) {
  const double zero[3] = {0.0, 0.0, 0.0};
  std::vector<double2> beams;
  int rc = build_scan_consts(h, c, scan, n_beams, zero, zero, zero, 1, zero, beams);
  if (rc != TBNAV_OK) return rc;
  c.p0 = particle;
  return upload_beams(h, beams, n_beams, c.Bv);
}
}  // namespace

int tbnav_rbpf_integrate_scan(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, const double pose[3]) {
  if (!h || !scan || n_beams <= 0 || !pose || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  ++h->scans_done;
  ScanC c;
  int rc = one_particle_consts(h, particle, scan, n_beams, c);
  if (rc != TBNAV_OK) return rc;
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  TBNAV_HIP(hipMemcpy(sp.pose + (size_t)particle * 3, pose, sizeof(double) * 3, hipMemcpyHostToDevice));
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  rc = launch_raycast(h, c, 1, nullptr);
  if (rc != TBNAV_OK) return rc;
  const int zero = 0;  // the map changed: a stored field of this particle is stale
  TBNAV_HIP(hipMemcpyAsync(h->d_fstate + particle, &zero, sizeof zero, hipMemcpyHostToDevice, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  rc = status_from_err(h->h_err);
  if (rc != TBNAV_OK) return rc;
  if (h->ref_field) return ref_field_after_scan(h, false, particle, 1);
  if (h->df_mode != 2) return ensure_full_field(h, particle);  // stored-field modes: the whole field after the update
  return TBNAV_OK;
}

int tbnav_rbpf_likelihood(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, const double pose[3], double* out) {
  if (!h || !scan || n_beams <= 0 || !pose || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  ScanC c;
  int rc = one_particle_consts(h, particle, scan, n_beams, c);
  if (rc != TBNAV_OK) return rc;
  for (int q = 0; q < 4; ++q) h->h_err[q] = 0;
  hipLaunchKernelGGL(rbpf_likelihood_one, dim3(1), dim3(kWave), 0, h->stream, c, h->d_beams, h->d_code[h->cur], h->pool, map_of(h),
                     h->d_trow[h->cur], h->d_fstate, h->radius, h->d_nocc[h->cur], pose[0], pose[1], pose[2], h->d_score, h->d_err, h->d_mixlut);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(out, h->d_score, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  return status_from_err(h->h_err);
}

int tbnav_rbpf_particle_map(tbnav_rbpf* h, int32_t particle, int8_t* map) {
  if (!h || !map || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  hipStream_t st = h->stream;
  TBNAV_HIP(hipMemcpyAsync(h->d_best, &particle, sizeof(int), hipMemcpyHostToDevice, st));
  const int blocks = (int)std::min<size_t>((h->G + 255) / 256, 2048);
  hipLaunchKernelGGL(rbpf_export_map, dim3(blocks), dim3(256), 0, st, h->xsize, h->G, h->cuts, h->d_best, h->pool, map_of(h), h->d_export);
  TBNAV_HIP(hipGetLastError());
  TBNAV_HIP(hipMemcpyAsync(map, h->d_export, h->G, hipMemcpyDeviceToHost, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  return TBNAV_OK;
}

int tbnav_rbpf_set_option(tbnav_rbpf* h, int32_t option, int32_t value) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  switch (option) {
    case TBNAV_RBPF_OPT_DF_MODE: {
      if (value < TBNAV_RBPF_DF_FULL || value > TBNAV_RBPF_DF_REFERENCE) return TBNAV_ERR_INVALID_ARG;
      if (h->scans_done) return TBNAV_ERR_INVALID_ARG;  // the mode belongs to the filter's whole life
      if (value == TBNAV_RBPF_DF_REFERENCE) {
        if (h->N > 4096) return TBNAV_ERR_UNSUPPORTED;  // serial host brushfire per particle: small ensembles only
        { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
        delete h->ref;
        h->ref_on_dev.clear();
        h->ref = new (std::nothrow) tbnav::RefField(h->N, h->xsize, h->radius);
        if (!h->ref) return TBNAV_ERR_INVALID_ARG;
        h->ref_field = true; h->df_mode = 2; h->full_edt = false;
        return TBNAV_OK;
      }
      if (value != TBNAV_RBPF_DF_QUERY) {
        if (h->edt_cols == 0) return TBNAV_ERR_UNSUPPORTED;  // no LDS transform for this map size
        { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
        // empty maps: the stored field "everything unreached" IS the whole, fresh field
        TBNAV_HIP(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(h->d_fstate), 2, h->N));
        h->fstate_dirty = true;
      }
      h->ref_field = false; h->df_mode = value; h->full_edt = value == TBNAV_RBPF_DF_FULL;
      return TBNAV_OK;
    }
    case TBNAV_RBPF_OPT_RAYCAST_ORDERED:
      if (value) h->tile_cap = 0;
      else {
        const double reach = (double)h->p.range_max + std::hypot(h->p.Trs[1], h->p.Trs[2]);
        const long side = 2 * ((long)std::ceil(reach / h->p.resolution) + 2) + 1;
        h->tile_cap = (side * side <= 30000) ? (int)(side * side) : 0;
      }
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_THREADS:
      if (value != 0 && value != 256 && value != 512 && value != 1024) return TBNAV_ERR_INVALID_ARG;
      h->raycast_threads = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_BAND_ROWS:
      if (value < 0) return TBNAV_ERR_INVALID_ARG;
      h->raycast_band_rows = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_ADAPT:
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->raycast_adapt = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_BATCH_PIPELINE:
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->batch_pipeline = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_HOST_THREADS:
      if (value < 0 || value > 256) return TBNAV_ERR_INVALID_ARG;
      h->host_threads = value ? value : default_host_threads();
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_FORM:
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->raycast_form = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_COUNT_CELLS:
      h->count_touched = value != 0;
      return TBNAV_OK;
    default: return TBNAV_ERR_INVALID_ARG;
  }
}

int tbnav_rbpf_scan_counts(tbnav_rbpf* h, uint64_t* cell_updates, uint64_t* distinct_cells, int32_t reset) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  unsigned long long v[2] = {0, 0};
  TBNAV_HIP(hipMemcpy(v, h->d_touched, sizeof v, hipMemcpyDeviceToHost));
  if (cell_updates) *cell_updates = v[0];
  if (distinct_cells) *distinct_cells = v[1];
  if (reset) TBNAV_HIP(hipMemset(h->d_touched, 0, sizeof v));
  return TBNAV_OK;
}

int tbnav_rbpf_reference_field_counts(tbnav_rbpf* h, int32_t* distinct_states, int32_t* last_brushfires, int64_t* total_brushfires) {
  if (!h || !h->ref_field || !h->ref) return TBNAV_ERR_INVALID_ARG;
  if (distinct_states) *distinct_states = h->ref->distinct_states();
  if (last_brushfires) *last_brushfires = h->ref->last_step_brushfires();
  if (total_brushfires) *total_brushfires = h->ref->total_brushfires();
  return TBNAV_OK;
}

int tbnav_rbpf_set_timing(tbnav_rbpf* h, int32_t enable) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->timing = enable != 0;
  return TBNAV_OK;
}

int tbnav_rbpf_last_kernel_names(const tbnav_rbpf* h, char* propose, int32_t propose_cap, char* raycast, int32_t raycast_cap, int32_t* raycast_workgroups) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (propose && propose_cap > 0) { if (h->lk_propose) snprintf(propose, (size_t)propose_cap, "rbpf_propose<%d>", h->lk_propose); else propose[0] = 0; }
  if (raycast && raycast_cap > 0) {
    if (h->lk_raycast > 0) snprintf(raycast, (size_t)raycast_cap, "rbpf_raycast_box<%d>", h->lk_raycast);
    else if (h->lk_raycast == 0) snprintf(raycast, (size_t)raycast_cap, "rbpf_raycast");
    else raycast[0] = 0;
  }
  if (raycast_workgroups) *raycast_workgroups = h->lk_raycast_grid;
  return TBNAV_OK;
}

int tbnav_rbpf_last_kernel_ms(tbnav_rbpf* h, float ms[TBNAV_RBPF_NKERNELS]) {
  if (!h || !ms) return TBNAV_ERR_INVALID_ARG;
  for (int i = 0; i < TBNAV_RBPF_NKERNELS; ++i) ms[i] = h->last_ms[i];
  return TBNAV_OK;
}

}  // extern "C"
