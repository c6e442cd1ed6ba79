// rbpf_device.hpp — what the RBPF kernel files and the host side (rbpf_host.hpp + rbpf*.hip: handle, launches, C-ABI) share: the launch-argument
// structs, the tiled copy-on-write map's accessors, world -> cell, wave reductions, the exact-transform / resampling / migration
// helpers and the declarations of every kernel.  What only one family needs lives in that family's file: the distance lookups
// and likelihoods in rbpf_propose.hip, rays and add_repeated in rbpf_raycast.hip; the normalise / selection body, which runs in
// a kernel of its own AND as workgroup 0 of the map update, in rbpf_normalize.hpp.  Kernels are defined in their family's file
// and launched from the host files; the template kernels are explicitly instantiated where they are defined.  Everything is compiled
// -ffp-contract=off (csrc/Makefile): grid indices, log-odds, Neff and parent lists are bit-exact targets.
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
// reference-field mode only (ref_field.hpp): the particle's brushfire has not written this cell yet — a lookup that reads it is
// reported to the host (DistSrc::pend), which resumes that brushfire and has the proposal run again.  The largest real code is
// cell_radius^2 = 40 000.
constexpr uint16_t kCodePending = 0xFFFE;
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
// The free tiles are kept in kPoolShards lists (rings), list s holding the ids = s (mod kPoolShards), each with its head and tail on
// a 128-byte line of its own: the map update that follows a resampling pops once per particle, all within a few microseconds, and
// one word retires ~80 returning atomics per microsecond — a thousand co-resident workgroups waited up to 17 us for their ring
// position, sixteen words 1.3 us (tools/atomic_queue_probe.hip).  A caller pops from the list its hint names and moves on to the
// next when that one cannot supply the request; a tile goes back to the list of its id, so no list ever holds more than its share.
// A request is ONE fetch-add on a list's head: it is granted what lies between the position it got and the list's tail — all it
// asked for, usually; a part, or nothing, when the list is short — and takes the rest from the next lists the same way, keeping
// the ids where it has room for them (the map update in its LDS table of the box's tiles, an import in the slot's tile table).  A
// head that was pushed past its tail is brought back TO the tail by the caller that did it (atomicMin: never below a position
// that was granted — the add-then-subtract this replaces could hand a position out twice once failing on one list had become a
// normal event).  So a launch whose requests fit the pool's free tiles never sees "exhausted", whatever the lists' lengths; one
// that asks for more fails, possibly for several of its callers racing for the last tiles: the scan is an error as a whole.
// Pools below kPoolShardMin tiles keep ONE list.
constexpr int kPoolShards = 16, kPoolShardsLog2 = 4;
constexpr unsigned int kPoolShardMin = 16384u;   // tiles (128 MB of log-odds)
constexpr int kPoolCtrStride = 16;               // 64-bit words between two lists' counters
constexpr int kPoolCtrWords = kPoolCtrStride * kPoolShards;
constexpr int kPoolPosShift = 40;                // a ring position as the callers hold it: list << 40 | tiles popped from it so far
struct TilePool {
  double* lo;               // [cap][kTileCells], in-tile index = (i & 31) * 32 + (j & 31)
  unsigned int* bm;         // [cap][kTS] occupancy bits of the tile's cells (prob >= 0.90): row i & 31, bit j & 31
  int* ref;                 // [cap]
  unsigned int* ring;       // [shards][shard_cap] free tile ids
  unsigned long long* ctr;  // [list][kPoolCtrStride]: [0] head: tiles popped, [1] tail: tiles pushed (free = tail - head)
  unsigned int cap;
  unsigned int shards;      // 1 or kPoolShards
  unsigned int shard_cap;   // ceil(cap / shards)
};
struct MapT {
  unsigned int* table;  // [N][TT] of the current buffer
  unsigned int* shed;   // [N][TT] tile this slot stopped using since the last resample (0 = none)
  int TW, TT;           // tiles per side, tiles per map
};
__device__ __forceinline__ int tile_of(const MapT& M, int ci, int cj) { return (ci >> kTSh) * M.TW + (cj >> kTSh); }
__device__ __forceinline__ int in_tile(int ci, int cj) { return ((ci & (kTS - 1)) << kTSh) | (cj & (kTS - 1)); }
__device__ __forceinline__ unsigned int* ring_slot(const TilePool& P, unsigned int s, unsigned long long pos) {
  return P.ring + (size_t)s * P.shard_cap + (size_t)(pos % P.shard_cap);
}
__device__ __forceinline__ void tile_push(const TilePool& P, unsigned int id) {
  const unsigned int s = id & (P.shards - 1u);
  const unsigned long long pos = atomicAdd(P.ctr + (size_t)s * kPoolCtrStride + 1, 1ull);
  *ring_slot(P, s, pos) = id;
}
__device__ __forceinline__ unsigned int tile_at(const TilePool& P, unsigned long long pos) {
  return *ring_slot(P, (unsigned int)(pos >> kPoolPosShift), pos & ((1ull << kPoolPosShift) - 1ull));
}
// `want` tiles from list `list`: ONE atomic on its head (a workgroup that clones 15 tiles after a resample would otherwise queue 15
// times).  Granted: tile_at(P, pos + i), i < n — n == want unless the list is short (then n < want, possibly 0).
struct TileGrant { unsigned long long pos; unsigned int n; };
__device__ __forceinline__ TileGrant tile_grab(const TilePool& P, unsigned int want, unsigned int list) {
  unsigned long long* const c = P.ctr + (size_t)list * kPoolCtrStride;
  const unsigned long long pos = atomicAdd(c, (unsigned long long)want), tail = c[1];   // (no push can be in flight: see above)
  const unsigned long long at = ((unsigned long long)list << kPoolPosShift) | pos;
  if (pos + want <= tail) return TileGrant{at, want};
  atomicMin(c, tail);   // the head went past the tail: back to it (nobody waits for this one)
  return TileGrant{at, pos < tail ? (unsigned int)(tail - pos) : 0u};
}
// Tile by tile for a caller whose first grant was short: the grant in hand, then the next lists'.  want = how many the caller
// still needs, this one included.  0 = the pool is exhausted.
struct TileTaker { unsigned long long pos; unsigned int left, list, tried; };
__device__ __forceinline__ TileTaker tile_taker(unsigned int hint) { return TileTaker{0ull, 0u, hint, 0u}; }
__device__ __forceinline__ unsigned int tile_take(const TilePool& P, TileTaker& t, unsigned int want) {
  while (t.left == 0u) {
    if (t.tried >= P.shards) return 0u;
    const TileGrant g = tile_grab(P, want, (t.list + t.tried) & (P.shards - 1u));
    ++t.tried;
    t.pos = g.pos; t.left = g.n;
  }
  --t.left;
  return tile_at(P, t.pos++);
}
__device__ __forceinline__ unsigned int tile_pop(const TilePool& P, unsigned int hint) {  // one tile, from whichever list has one; 0 = pool exhausted
  TileTaker t = tile_taker(hint);
  return tile_take(P, t, 1u);
}
constexpr unsigned long long kPoolScattered = ~1ull;   // what a caller keeps in place of a position when its tiles came from several grants
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
  if (lane == 0) nid = tile_pop(P, blockIdx.x);
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

// ---- proposal side: what the host needs of it (lookups, likelihoods and the kernels' helpers are in rbpf_propose.hip) ----
constexpr int kMixLut = 1024, kMixLds = 128;
struct Trace {
  double *sampled, *p_scan, *p_pose, *mu, *sigma, *eta, *new_pose, *weight_raw;
};
struct ScanMatchC { double lstep, astep; int iters, max_moves; };
constexpr int kMatchThreads = 6 * kWave;
// ---- map update: what the host needs of it (the kernels and their device helpers are in rbpf_raycast.hip) ----
struct OccLog { int* ev; int* count; int cap; };
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
constexpr int kBoxSideMax = 176;  // rows a scan's bounding box can have (tile_cap <= 30000 -> side <= 173)
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
struct NormArgs { int N; const double* zp; const double* weight; double* weight_out; double* cs; int* parent; NormOut* out;
                  int* gate; const int* gate_prev; unsigned int* seq; unsigned int seq_val; int* children; };
constexpr int kBoxEv = 8;     // events a slot holds before it is replayed exhaustively
constexpr int kHotSide = 7;   // the kHotSide x kHotSide cells round the robot are candidates for a lane of their own ...
constexpr int kHotMin = 16;   // ... when they collect at least this many free adds
constexpr int kVeryHot = 80;  // ... and from this many on they are worked out without the chain of adds (add_repeated)
// tile u32[cap] | ev u16[bv][kBoxEv] (a slot's first 8 bytes become its replayed value once its events are read) | hot values f64[64] | exy i32[bv] | ecnt u16[bv]
constexpr int kBoxEvFour = 4; // ... in the four-workgroups-per-CU form (8 bytes a slot: exactly the replayed value; the bench room's box fits with them)
__host__ __device__ constexpr size_t box_lds_bytes(size_t cap, size_t bv, size_t ev = kBoxEv) { return 4 * cap + 2 * ev * bv + 8 * 64 + 4 * bv + 2 * ((bv + 1) & ~(size_t)1); }
// the 16-bit cell form: half the cell array + the slot table (hash_words u32, a power of two >= 2 x (bv + 64) slots)
__host__ __device__ constexpr size_t box16_hash_words(size_t bv) { size_t h = 256; while (h < 2 * (bv + 64)) h *= 2; return h; }
__host__ __device__ constexpr size_t box16_lds_bytes(size_t cap, size_t bv, size_t ev = kBoxEv) { return box_lds_bytes(cap, bv, ev) - 2 * cap + 4 * box16_hash_words(bv); }
constexpr size_t kBoxStaticLds = 896;  // the kernel's __shared__ variables (tools/kernel_resources.py rbpf_raycast: 856 B since mt_src, round 5), rounded up
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
// Exact n / d for 0 <= n < 2^16 by a divisor 1 <= d <= 2^8 that is the SAME for the whole workgroup (pairs in a row of the box, map tiles
// under it, segments per ray): with m = ceil(2^24 / d), worked out once, the quotient is the high word of (n << 8) * m — two
// instructions where floor_div_small is about twenty.  (m d - 2^24 < d, so n m / 2^24 exceeds n / d by less than n / 2^24 < 2^-8 <= 1 / d:
// the floor is the same.  tests/test_raycast_step_arithmetic.py holds the formula against // for every such n and d.)
__device__ __forceinline__ unsigned int udiv16_magic(int d) { return ((1u << 24) + (unsigned int)d - 1u) / (unsigned int)d; }
__device__ __forceinline__ int udiv16(int n, unsigned int m) { return (int)__umulhi((unsigned int)n << 8, m); }
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
// Freed tiles go back with one atomic per free list, WORKGROUP and round (lane-private pushes queue on the lists' tails:
// ~90 atomics per microsecond on one address, and a resample that kills 900 of 1000 particles frees 13 000 tiles).  Instead of one atomic per child and tile plus one per old entry, in two launches.
__device__ __forceinline__ void resample_tables_body(int N, int TT, const int* __restrict__ parent, const int* __restrict__ children,
                                                     const unsigned int* __restrict__ tab_old, unsigned int* __restrict__ tab_new,
                                                     unsigned int* __restrict__ shed, const TilePool& P, int block, int nblocks) {
  const size_t n = (size_t)N * TT;
  __shared__ int s_cnt[kPoolShards];
  __shared__ unsigned long long s_base[kPoolShards];
  const unsigned int smask = P.shards - 1u;
  for (size_t e0 = (size_t)block * blockDim.x; e0 < n; e0 += (size_t)nblocks * blockDim.x) {
    const size_t e = e0 + threadIdx.x;
    unsigned int freed[2] = {0u, 0u};
    if (threadIdx.x < kPoolShards) s_cnt[threadIdx.x] = 0;
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
    // a freed tile goes to the list of its id: counted per list in LDS, then ONE atomic per list and round for the workgroup
    __syncthreads();
    int off0 = 0, off1 = 0;
    if (freed[0]) off0 = atomicAdd(&s_cnt[freed[0] & smask], 1);
    if (freed[1]) off1 = atomicAdd(&s_cnt[freed[1] & smask], 1);
    __syncthreads();
    if (threadIdx.x < P.shards && s_cnt[threadIdx.x])
      s_base[threadIdx.x] = atomicAdd(P.ctr + (size_t)threadIdx.x * kPoolCtrStride + 1, (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    if (freed[0]) *ring_slot(P, freed[0] & smask, s_base[freed[0] & smask] + (unsigned long long)off0) = freed[0];
    if (freed[1]) *ring_slot(P, freed[1] & smask, s_base[freed[1] & smask] + (unsigned long long)off1) = freed[1];
    __syncthreads();  // (s_cnt is cleared by the next round)
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
    if ((G & 3) == 0) {   // (every slot starts on an 8-byte boundary)
      const uint2* ca = reinterpret_cast<const uint2*>(cd_src + (size_t)src * G);
      uint2* cb = reinterpret_cast<uint2*>(cd_dst + (size_t)m * G);
      for (size_t t = t0; t < G / 4; t += stride) cb[t] = ca[t];
    } else {
      for (size_t t = t0; t < G; t += stride) cd_dst[(size_t)m * G + t] = cd_src[(size_t)src * G + t];
    }
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
// Device noise drawn INSIDE rbpf_propose (round 5; the kernel's DN = true instantiation): the values rbpf_sample_normals would have stored —
// same Philox counters, same Box-Muller — for the particle's own 3k + 3 (or 3) normals, and the scan's beam table carried over by
// an extra leading workgroup instead of a launch of its own: workgroup 0 copies the table from pinned host memory to dev_beams,
// leaves the resampling offset's normal in *z_out (read by the normalise, a later launch) and publishes `seq` in *ready; the
// particles' workgroups (blockIdx.x - 1) look for it once without waiting when they start — those dispatched later find it —
// and otherwise wait for it, bounded, just before they first need the table (by then it has long arrived: the copy takes one
// PCIe round trip, their own first loads two dependent trips to HBM).
constexpr int kReadyCopies = 64, kReadyStride = 32;   // copies of the "beam table is there" word (NoiseSrc::ready), 128 bytes apart
struct NoiseSrc {
  unsigned long long seed, scan;   // key and counter prefix (scan << 40) of the scan's stream
  size_t base;                     // index of this handle's first normal in the ENSEMBLE's stream (sharded filters; 0 otherwise)
  size_t z_index;                  // index of the resampling offset in it
  double* z_out;
  const double2* host_beams;       // pinned host memory, device-visible
  double2* dev_beams;              // for the launches behind this one
  double2* fg_beams;               // FINE-GRAINED device memory (uncached): for the other workgroups of this one
  unsigned int* ready;             // fine-grained too: kReadyCopies copies, kReadyStride words apart
  unsigned int seq;
};
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
template <int NT, bool DN>
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
                                                                const double* __restrict__ mixlut, NoiseSrc ns, int* __restrict__ pend = nullptr);
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
template <int NT, int WPS, bool C16, int EV>
__global__ __launch_bounds__(NT, WPS) void rbpf_raycast_box(ScanC c, TilePool P, MapT M, const double2* __restrict__ beams,
                                                          const double* __restrict__ pose, const double* __restrict__ sens,
                                                          int* __restrict__ trow_occ, int* __restrict__ n_occ, int* __restrict__ err,
                                                          int tile_cap, unsigned long long* __restrict__ touched, NormArgs nz,
                                                          int* __restrict__ box_need, int* __restrict__ box_need_host, int need_slot, int hash_words);
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
__global__ void rbpf_warm_scratch(int* sink, int n);  // (rbpf_raycast.hip)
constexpr int kWarmScratchInts = 40;  // 160 bytes a lane: more than any map-update instantiation spills (tools/kernel_resources.py: <= 136)
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
