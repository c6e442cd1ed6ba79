// rbpf_raycast.hip — GridMapper::integrateScan's map update (grid_mapper.cpp:140-182, :549-807, :438-546): the box-counter
// kernel rbpf_raycast_box (default: LDS counters over the scan's bounding box, end-point cells replayed in beam order, the
// weights' normalise / selection riding along as workgroup 0) and the beam-ordered rbpf_raycast (scans the box kernel cannot
// hold; the reference-field mode, whose occupied-set log it writes).  Bit-identical maps.
#include "rbpf_device.hpp"
#include "rbpf_normalize.hpp"

namespace tbnav_rk {

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
    // j steps at once: all n if they stay in the binade (m + j s <= 2^53 - 1), else floor(room / s) of them and a plain add across the
    // binade's end.  The quotient is below n there, so a single-precision estimate (both operands to 24 bits, the hardware reciprocal:
    // ~2^-21 relative) is within one or two of it for the counts a cell can take, and the two loops on the remainder make it exact for any count.  (An fp64
    // division stood here: its u64 -> f64 conversions were the last spill of the 16-bit-cell instantiations.  A cell that starts at zero
    // crosses nine binades in its first scans, so this branch is not rare enough for a search over j either — tried: the map update 40.4 -> 42.3 us at N = 1000 / 360 beams,
    // 0.85 -> 1.14 ms at N = 12 500 / 1080 beams.)
    unsigned long long j = (unsigned long long)n;
    if (__umul64hi(j, s) != 0ull || ((j * s) >> 53) != 0ull || ((m + j * s) >> 53) != 0ull) {
      const unsigned long long room = (1ull << 53) - 1ull - m;
      const float rf = (float)(unsigned int)(room >> 32) * 4294967296.0f + (float)(unsigned int)room;
      const float sf = (float)(unsigned int)(s >> 32) * 4294967296.0f + (float)(unsigned int)s;
      j = (unsigned long long)(unsigned int)(rf * __builtin_amdgcn_rcpf(sf));
      long long r = (long long)(room - j * s);   // (j s <= room (1 + 2^-20) < 2^54: no wrap) the remainder, made to lie in [0, s)
      while (r < 0ll) { --j; r += (long long)s; }
      while (r >= (long long)s) { ++j; r -= (long long)s; }
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
// What bounds it: VALU issue — ~18 k VALU wave-instructions per particle; with four workgroups per CU the
// vector units are busy 29 us of a 42 us launch (SQ_ACTIVE_INST_VALU); memory traffic is the distinct cells once each way.
// LDS: tile u32[tile_cap] (rows padded to an even number of columns: pair i = words 2i, 2i+1) |
// ev u16[Bv][8, or 4 in the four-per-CU form] (slot o's first 8 bytes double as its replayed value) | hot-cell values f64[64] | exy i32[Bv] | ecnt u16[Bv]
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
__global__ __launch_bounds__(kWave) void rbpf_raycast(ScanC c, TilePool P, MapT M, const double2* __restrict__ beams,
                                                     const double* __restrict__ pose, int* __restrict__ trow_occ,
                                                     int* __restrict__ n_occ, int* __restrict__ err, OccLog log,
                                                     const int* __restrict__ gate_prev) {
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
      // one request for all of them; a list that is short grants what it has and the next lists supply the rest (lane 0 holds the
      // grants).  A tile cloned before the pool runs out stays the particle's (same values as the tile it left): no log-odds has
      // been written either way
      TileTaker tk = tile_taker(blockIdx.x);
      int left = need;
      for (int w = 0; w < tword; ++w) {
        unsigned int m = tbits[w];
        while (m) {
          const int t = w * 32 + __ffs((int)m) - 1;
          m &= m - 1;
          if (tile_is_private(P, tab, t)) continue;
          unsigned int nid = 0u;
          if (lane == 0) nid = tile_take(P, tk, (unsigned int)left);
          nid = (unsigned int)__shfl((int)nid, 0, kWave);
          if (nid == 0u) { if (lane == 0) atomicOr(&err[3], 8); return; }  // tile pool exhausted
          tile_clone_into(P, tab, shed, t, nid, lane);
          --left;
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
__global__ void rbpf_add_repeated_test(const double* __restrict__ x, const double* __restrict__ d, const int* __restrict__ n, double* __restrict__ out, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = add_repeated(x[i], d[i], n[i]);
}
// Launched once when a handle is created: a kernel with a private array per lane, over a grid that fills the device, so that the
// stream's queue has its scratch memory BEFORE the first map update whose instantiation spills (the four-per-CU form: ~100 bytes
// a lane).  Without it that launch pays the allocation: 171 us instead of 44 — a whole scan's worth, inside somebody's timed call.
__global__ __launch_bounds__(512) void rbpf_warm_scratch(int* sink, int n) {
  volatile int a[kWarmScratchInts];
  for (int i = 0; i < kWarmScratchInts; ++i) a[i] = i + n;
  if (n == 0x7ead) sink[0] = a[(threadIdx.x + n) % kWarmScratchInts];  // (never: keeps the array alive and dynamically indexed)
}
template <int NT, int WPS, bool C16, int EV>
__global__ __launch_bounds__(NT, WPS) void rbpf_raycast_box(ScanC c, TilePool P, MapT M, const double2* __restrict__ beams,
                                                          const double* __restrict__ pose, const double* __restrict__ sens,
                                                          int* __restrict__ trow_occ, int* __restrict__ n_occ, int* __restrict__ err,
                                                          int tile_cap, unsigned long long* __restrict__ touched, NormArgs nz,
                                                          int* __restrict__ box_need, int* __restrict__ box_need_host, int need_slot, int hash_words) {
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
  // C16 (round 4): a cell is a 16-BIT word — bit 15: end-point (or hot) cell, low 15 bits: free adds — and the slot of a flagged
  // cell is found by look-up in a small open-addressing table keyed by the cell (hash_words entries, a power of two >= 2 x the
  // slots; entry = (cell + 1) << 16 | slot).  Half the LDS per cell of the box: what makes room for a fourth workgroup per CU
  // (or a third / fourth where the 32-bit form fits two) at the price of a few instructions per walk step (LDS atomics are
  // 32-bit: the add goes to the containing dword, shifted) and a probe per event.  The 32-bit form (!C16) keeps the slot in
  // bits 16-30 of the cell's own word.
  const int tile_words = C16 ? tile_cap / 2 + hash_words : tile_cap;   // ints the cell array (+ the table) take
  unsigned int* tile = reinterpret_cast<unsigned int*>(lds_i);         // (tile_cap is a multiple of 8)
  unsigned short* tile16 = reinterpret_cast<unsigned short*>(lds_i);
  unsigned int* htab = reinterpret_cast<unsigned int*>(lds_i + tile_cap / 2);
  const unsigned int hmask = (unsigned int)hash_words - 1u;
  // events a slot holds before its cell is replayed exhaustively: 8 — or 4 where that is what lets a FOURTH workgroup share
  // the CU (2 x 360 x 4 bytes less: what the bench room's 94 x 88 box lacked; its cells take 1-3 events each.  Not where 8 fit as
  // well: a corridor's cells take 5-10 events, and replaying most slots exhaustively costs more than the residency buys)
  static_assert(EV == kBoxEv || EV == kBoxEvFour, "8 or 4 events a slot");
  constexpr int kEv = EV;
  unsigned short* ev = reinterpret_cast<unsigned short*>(lds_i + tile_words);  // [Bv][kEv]  beam << 1 | occupied  (16 or 8 bytes a slot, aligned)
  double* val_hot = reinterpret_cast<double*>(lds_i + tile_words + (kEv / 2) * Bv);    // [64] the value worked out for a hot cell (slot Bv + lane)
  int* exy = lds_i + tile_words + (kEv / 2) * Bv + 2 * 64;         // [Bv] end-point cell, x | y << 16
  unsigned short* ecnt = reinterpret_cast<unsigned short*>(exy + Bv);  // [Bv] events recorded in the slot of beam b — the FIRST beam that ended in its
                                                                       //      cell (0: b opened no slot; may exceed kBoxEv: overflow).  Two counts a dword:
                                                                       //      LDS atomics are 32-bit, a count never reaches 2^16 (one event per beam at most)
  constexpr unsigned int kFlag = 0x80000000u;   // (the flag as the passes below see a cell: C16 cells are widened to this form when read)
  // C16: the slot of flagged cell t.  claim: the first caller's slot wins and is returned to everybody (one compare-and-swap per
  // probe); find: the cell IS in the table (its flag is set only after its claim).
  auto h_of = [&](int t) { return ((unsigned int)t * 0x9E3779B1u >> 12) & hmask; };
  auto slot_claim = [&](int t, int slot) {
    const unsigned int key = (unsigned int)(t + 1) << 16;
    unsigned int h = h_of(t);
    for (;;) {
      const unsigned int old = atomicCAS(&htab[h], 0u, key | (unsigned int)slot);
      if (old == 0u) return slot;
      if ((old & 0xFFFF0000u) == key) return (int)(old & 0xFFFFu);
      h = (h + 1u) & hmask;
    }
  };
  auto slot_find = [&](int t) {
    const unsigned int key = (unsigned int)(t + 1) << 16;
    unsigned int h = h_of(t);
    for (;;) {
      const unsigned int e = htab[h];
      if ((e & 0xFFFF0000u) == key) return (int)(e & 0xFFFFu);
      h = (h + 1u) & hmask;
    }
  };
  // the value replayed for end-point slot o (< Bv) lives in the first 8 bytes of the slot's own 16-byte event list — the lane
  // that replays it has the events in registers by then, nobody else reads them — and a hot cell's (slot Bv + lane) in val_hot:
  // 8 bytes per beam less LDS than an array of its own, which is what lets FOUR workgroups share a CU when the box is small
  auto val_at = [&](int slot) -> double* { return slot < Bv ? reinterpret_cast<double*>(ev + slot * kEv) : val_hot + (slot - Bv); };
  __shared__ int bad, bx0, bx1, by0, by1, srx, sry, nocc_delta, n_ovf;
  __shared__ unsigned long long need_base;
  __shared__ unsigned int mt_id[kMapTilesMax];  // map tiles under the box: the tile the particle's table names (once written: its private tile) — where the update WRITES
  __shared__ unsigned int mt_src[kMapTilesMax]; // ... and the tile it named when the band began — where the update READS (round 5: a tile made private in this
                                                //     launch is not copied whole first; the cells the update writes go from the shared tile straight into the new one)
  __shared__ unsigned char mt_touch[kMapTilesMax];  // (bytes: every byte of LDS counts towards a fourth resident workgroup ...
  __shared__ unsigned int mt_priv_bits[2];          //  ... and bits: tile q of the box is private to the particle already)
  __shared__ int rc_delta[kBoxSideMax / kTS + 2];
  __shared__ unsigned short ovf[kWave];                 // slots whose event list overflowed (more than these: found by scanning)
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
      nocc_delta = 0; n_ovf = 0; robot_cnt = 0; mt_priv_bits[0] = 0u; mt_priv_bits[1] = 0u;
    }
  } else {
    uint4* t4 = reinterpret_cast<uint4*>(tile);
    for (int t = tid - kWave; t < tile_words / 4; t += nthr - kWave) t4[t] = uint4{0u, 0u, 0u, 0u};
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
  const unsigned int mty_m = (unsigned int)uni((int)udiv16_magic(mty));   // (uniform divisors: udiv16, rbpf_device.hpp)
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
    const int qi = udiv16(tq, mty_m), qj = tq - qi * mty;
    const unsigned int id0 = tab[(tx0 + qi) * M.TW + (ty0 + qj)];
    mt_id[tq] = id0; mt_src[tq] = id0;
  }
  auto map_tile = [&](int cx, int cy) { return __mul24((cx >> kTSh) - tx0, mty) + ((cy >> kTSh) - ty0); };
  auto cell_ptr = [&](int cx, int cy) -> const double* { return P.lo + (size_t)mt_src[map_tile(cx, cy)] * kTileCells + in_tile(cx, cy); };   // (reads: see mt_src)
  auto toggled = [&](int cx, int cy, bool now) {  // the cell crossed the occupied cut-off: its bit in the (private) tile, tile-row count, total
    atomicXor(&P.bm[(size_t)mt_id[map_tile(cx, cy)] * kTS + (cx & (kTS - 1))], 1u << (cy & (kTS - 1)));
    atomicAdd(&rc_delta[(cx >> kTSh) - tx0], now ? 1 : -1);
    atomicAdd(&nocc_delta, now ? 1 : -1);
  };
  auto record = [&](int o, int what) {  // an event for the flagged cell whose slot is o
    const unsigned int was = atomicAdd(reinterpret_cast<unsigned int*>(ecnt) + (o >> 1), (o & 1) ? 0x10000u : 1u);
    const int en = (int)((o & 1) ? was >> 16 : was & 0xFFFFu);
    if (en < kEv) ev[o * kEv + en] = (unsigned short)what;
  };
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
      // (the zeros are made HERE, from an opaque register: as a loop invariant they were hoisted above the band loop and — the kernel
      //  has 64 VGPRs — spilled there, a 16-byte scratch store per lane in EVERY launch for a path one-band scans never take)
      unsigned int z = 0u;
      asm volatile("" : "+v"(z));
      for (int t = tid; t < tile_words / 4; t += nthr) t4[t] = uint4{z, z, z, z};
      for (int b = tid; b < Bv; b += nthr) ecnt[b] = 0;
      for (int t = tid; t < kMapTilesMax; t += nthr) { mt_touch[t] = 0; mt_src[t] = mt_id[t]; }   // (what an earlier band made private is the tile to read now)
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
      if constexpr (C16) {
        const int t = cell_t(e);
        const int o = slot_claim(t, b);
        atomicOr(&tile[t >> 1], (t & 1) ? 0x80000000u : 0x8000u);   // (after the claim: whoever sees the flag finds the slot)
        record(o, (b << 1) | 1);
      } else {
        const unsigned int mine = kFlag | ((unsigned int)b << 16);
        const unsigned int old = atomicCAS(&tile[cell_t(e)], 0u, mine);
        record((int)(((old ? old : mine) >> 16) & 0x7FFFu), (b << 1) | 1);
      }
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
      if (id != 0u && rf == 1) atomicOr(&mt_priv_bits[tq >> 5], 1u << (tq & 31));
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
      const unsigned int S_m = (unsigned int)uni((int)udiv16_magic(S));
      constexpr int kCellBytes = C16 ? 2 : 4;
      const unsigned int band_bytes = (unsigned int)kCellBytes * (unsigned int)band_cells;
      int n_first = 0;
      for (int task = tid; task < kWave * G * S; task += nthr) {
        const int tb = udiv16(task, S_m), sgm = task - tb * S;
        const int b = __mul24(tb & (kWave - 1), G) + (tb >> 6);  // lanes of a wave take rays spread round the scan
        if (b >= Bv) continue;
        const int e = exy[b];
        const RayP pr = ray_packed(rx, ry, e & 0xFFFF, e >> 16);
        const int count = pr.dmaj, L = udiv16(count + S - 1, S_m);
        int n = __mul24(sgm, L);
        const int n1 = (n + L < count) ? n + L : count;
        if (n >= n1) continue;
        const int two_dmin = 2 * pr.dmin, two_dmaj = 2 * pr.dmaj;
        const int a0 = __mul24(two_dmin, n) - pr.dmaj;
        const int c0 = a0 > 0 ? floor_div_small(a0 + two_dmaj - 1, two_dmaj) : 0;  // operands < 2^24
        int rem = a0 - __mul24(two_dmaj, c0 - 1);
        const int sc = pr.neg ? -c0 : c0;
        // byte offset of the segment's first cell in the band's array, and the byte steps along / across the ray
        int at = kCellBytes * (__mul24((pr.ymajor ? pr.xa + sc : pr.xa + n) - x0, bw) + ((pr.ymajor ? pr.ya + n : pr.ya + sc) - miny));
        const int d_major = kCellBytes * (pr.ymajor ? 1 : bw);
        const int d_both = d_major + kCellBytes * (pr.ymajor ? bw : 1) * (pr.neg ? -1 : 1);
        // The error term in fixed point: x = rem / two_dmaj in (0, 1] is kept as acc = x * 2^32 - 1 - slack, a step adds
        // F = dmin / dmaj * 2^32 - slack', and the add's CARRY is the step across the ray (rem + two_dmin > two_dmaj) — an add, a select
        // and an add per cell instead of six instructions; with four workgroups per CU the kernel is bound by VALU issue.  Exact:
        // the integer condition holds by at least 1 / two_dmaj = 2^32 / 352 = 1.2e7 units (dmaj <= kBoxSideMax) or fails by at least
        // one, and the slack — R is 2^32 / dmaj from a reciprocal rounded DOWN by 2^-20: <= 2^13 in acc at the start, <= 2^12 per step —
        // stays below 2^20 over a ray and never pushes acc ABOVE its true value.
        const unsigned int R = (unsigned int)(__frcp_rn((float)pr.dmaj) * 4294963200.0f);  // 2^32 (1 - 2^-20) / dmaj, rounded down (dmaj >= 1 here)
        const unsigned int F = (unsigned int)pr.dmin * R;                                  // dmin <= dmaj: < 2^32
        unsigned int acc = (unsigned int)rem * (R >> 1) - 1u;                              // rem <= two_dmaj: < 2^32
        auto advance = [&]() {
          unsigned int nxt;
          const bool side = __builtin_add_overflow(acc, F, &nxt);
          acc = nxt;
          at += side ? d_both : d_major;
        };
        if (n == 0) { ++n_first; advance(); ++n; }  // position 0 is the robot's cell (or, for a reversed ray, the end point): counted below
        char* const tile_b = reinterpret_cast<char*>(tile);
        // Three adds are issued back to back and what they return is looked at together (below): three registers, no copies.
        unsigned int r0 = 0u, r1 = 0u, r2 = 0u;
        // (C16: what comes back is the cell's DWORD — this cell's half is picked when it is looked at — and the byte offset the add
        //  went to rides along, so that a flagged cell's slot can be looked up: a0..a2 take turns like r0..r2)
        int a0r = 0, a1r = 0, a2r = 0;
        auto look = [&](unsigned int& old, int where) {
          if constexpr (C16) {
            if ((old >> ((where & 2) << 3)) & 0x8000u) record(slot_find(where >> 1), b << 1);
          } else {
            if (old & kFlag) record((int)((old >> 16) & 0x7FFFu), b << 1);
          }
          old = 0u;
        };
        // (round 6) Three steps, THEN one look at what the three adds returned: a cell with an event is a few percent of the steps,
        // and the test per step was a compare, a saveexec, a branch and an exec restore each — with eight waves per SIMD a wave that
        // waits for its third add costs nothing, the scalar and vector issue slots it no longer takes are what the kernel is short of.
        auto flagged = [&](unsigned int old, int where) -> unsigned int {
          if constexpr (C16) return (old >> ((where & 2) << 3)) & 0x8000u; else return old & kFlag;
        };
        auto add_at = [&](auto clipped, unsigned int& fresh, int& fresh_at) {
          if (!decltype(clipped)::value || (unsigned int)at < band_bytes) {
            if constexpr (C16) { fresh = atomicAdd(reinterpret_cast<unsigned int*>(tile_b + (at & ~3)), (at & 2) ? 0x10000u : 1u); fresh_at = at; }
            else fresh = atomicAdd(reinterpret_cast<unsigned int*>(tile_b + at), 1u);
          } else fresh = 0u;
          advance();
        };
        auto walk = [&](auto clipped) {
          int m = n1 - n;
          for (; m >= 3; m -= 3) {
            add_at(clipped, r0, a0r); add_at(clipped, r1, a1r); add_at(clipped, r2, a2r);
            if (flagged(r0, a0r) | flagged(r1, a1r) | flagged(r2, a2r)) { look(r0, a0r); look(r1, a1r); look(r2, a2r); }
          }
          r0 = 0u; r1 = 0u; r2 = 0u;
          if (m >= 1) add_at(clipped, r0, a0r);
          if (m >= 2) add_at(clipped, r1, a1r);
        };
        if (clip) walk(std::true_type{}); else walk(std::false_type{});
        look(r0, a0r); look(r1, a1r);
      }
      // the robot's own cell is the first free cell of every ray that has a free cell at all
      n_first = wave_sum_dpp(n_first);
      if (lane == 0 && n_first && (unsigned int)(rx - x0) < (unsigned int)nr) {
        const int t = __mul24(rx - x0, bw) + (ry - miny);
        if constexpr (C16) atomicAdd(&tile[t >> 1], (unsigned int)n_first << ((t & 1) << 4));
        else atomicAdd(&tile[t], (unsigned int)n_first);
      }
    }
    TRACE_W(6);
    __syncthreads();  // every event is recorded
    TRACE_W(7);
    PHASE_STAMP_W(1);
    // 2. requests and marks in one pass.  The band as PAIRS of cells (16 bytes of a map tile's row, two tile words), a thread's
    //    work ITEM a column of kSl pairs: rows j * kSl .. + kSl - 1 of the band, pair column pc — consecutive lanes take consecutive
    //    pair columns, so a wave's request is whole cache lines, and down the column the address moves by one tile row (256 bytes):
    //    the map tile is looked up once per item (twice when the column crosses into the next tile row), not once per pair.
    //    (Pair tid + i * nthr, what this was until round 4, cost 38 of a pair's ~50 vector instructions in row / column
    //    bookkeeping and address arithmetic — twice per pair, here and in the last pass: 46 % of the kernel's VALU work was
    //    that pass.)  A counted or flagged pair marks its map tile as written and asks for its log-odds from whichever tile the
    //    particle's table names NOW (shared, private or the zero tile hold the same values: the loads fly while the tiles are
    //    made private).  Slots that overflowed are listed on the way.
    const int PW = bw >> 1;                                                   // pairs in a row of the box
    const unsigned int PW_m = (unsigned int)uni((int)udiv16_magic(PW));
    // (the columns are cut at ABSOLUTE rows that are multiples of kSl, which divides the tile side: a column never crosses into
    //  the next tile row; its first and last may stick out of the band)
    constexpr int kSl = 4, kSlSh = 2;  // pairs a thread holds across the passes
    static_assert((1 << kSlSh) == kSl && kTS % kSl == 0, "a column of pairs never crosses a tile row");
    const int A0 = x0 & ~(kSl - 1);
    const int n_items = __mul24(uni(((x0 + nr - 1) >> kSlSh) - (x0 >> kSlSh) + 1), PW);
    const uint2* tile2 = reinterpret_cast<const uint2*>(tile);
    double2 v[kSl];
#pragma unroll
    for (int i = 0; i < kSl; ++i) v[i] = double2{0.0, 0.0};
    auto pairs = [&](int first, auto&& fn) {  // fn(i, the pair's two tile words, cx, cy of its first cell, where its log-odds are, its map tile), i < kSl
      const int q = first + tid;
      if (q >= n_items) return;
      const int j = udiv16(q, PW_m), pc = q - __mul24(j, PW);                  // items < 2^16, PW <= 88
      const int cxb = A0 + kSl * j, cy = miny + 2 * pc;
      const int mt = map_tile(cxb, cy);
      double* const ptr = P.lo + (size_t)mt_id[mt] * kTileCells + in_tile(cxb, cy);
      const int pib = __mul24(cxb - x0, PW) + pc;
#pragma unroll
      for (int i = 0; i < kSl; ++i) {
        uint2 w = uint2{0u, 0u};
        if ((unsigned int)(cxb + i - x0) < (unsigned int)nr) {
          const int pi = pib + i * PW;
          if constexpr (C16) {  // the pair is ONE dword; widened to the 32-bit form's words (flag -> bit 31; the slot is looked up where needed)
            const unsigned int d = tile[pi];
            w.x = (d & 0x7FFFu) | ((d & 0x8000u) << 16);
            w.y = ((d >> 16) & 0x7FFFu) | (d & 0x80000000u);
          } else w = tile2[pi];
        }
        fn(i, w, cxb + i, cy, ptr + i * kTS, mt);
      }
    };
    pairs(0, [&](int i, uint2 w, int, int, double* ptr, int mt) {
      if (w.x | w.y) {
        mt_touch[mt] = 1;
        v[i] = *reinterpret_cast<const double2*>(ptr);
      }
    });
    for (int first = nthr; first < n_items; first += nthr)  // (bands of more than kSl * nthr pairs: marks only, their loads follow)
      pairs(first, [&](int, uint2 w, int, int, double*, int mt) { if (w.x | w.y) mt_touch[mt] = 1; });
    for (int o = tid; o < Bv; o += nthr) {
      if (ecnt[o] == 0) continue;
      const int e = exy[o];
      const bool robot_cell = (e & 0xFFFF) == rx && (e >> 16) == ry;  // an end point too: no events from the walk, replayed against every beam
      if (robot_cell) ecnt[o] = (unsigned short)(kEv + 1);
      if (robot_cell || ecnt[o] > kEv) { const int i = atomicAdd(&n_ovf, 1); if (i < kWave) ovf[i] = (unsigned short)o; }
    }
    TRACE_W(8);
    __syncthreads();
    PHASE_STAMP_W(2);
    TRACE_W(9);
    // C. make the written tiles private to the particle (first write after a resample, or first touch of the area): ONE pop
    //    of a free list for all of them (rbpf_device.hpp: sixteen lists, the workgroup's number names the first to try), then one wave per tile copies 8 KB.  Usually there is nothing to do — and every wave
    //    sees that for itself (one ballot over the at most 64 tiles under the box), without a barrier to agree on it.
    const unsigned long long need_m = __ballot(lane < mtn && mt_touch[lane < mtn ? lane : 0]) &
                                      ~((unsigned long long)mt_priv_bits[0] | ((unsigned long long)mt_priv_bits[1] << 32));
    if (need_m) {  // workgroup-uniform
      if (tid == 0) {
        const unsigned int cnt = (unsigned int)__popcll(need_m);
        const TileGrant g = tile_grab(P, cnt, blockIdx.x & (P.shards - 1u));
        unsigned long long nb = g.pos;
        if (g.n != cnt) {
          // the list was short: what it granted and the next lists' grants, tile by tile, the ids parked in mt_id (the clones below put
          // them there anyway; what they read — the tile's old id — is in mt_src and in the table).  If the pool runs out half way the
          // tiles taken are noted as shed by their entries (an entry that is not private has shed nothing since the last resample):
          // the next resample hands them back, nothing has been written.
          nb = kPoolScattered;
          TileTaker tk{g.pos, g.n, blockIdx.x, 1u};
          unsigned long long m = need_m, taken = 0ull;
          unsigned int left = cnt;
          while (m) {
            const int q = __ffsll((long long)m) - 1;
            m &= m - 1ull;
            const unsigned int nid = tile_take(P, tk, left);
            if (nid == 0u) { nb = ~0ull; break; }
            mt_id[q] = nid; taken |= 1ull << q; --left;
          }
          if (nb == ~0ull) while (taken) {
            const int q = __ffsll((long long)taken) - 1;
            taken &= taken - 1ull;
            const int qi = udiv16(q, mty_m), qj = q - qi * mty;
            P.ref[mt_id[q]] = 1; shed[(tx0 + qi) * M.TW + (ty0 + qj)] = mt_id[q];
          }
        }
        need_base = nb;
        if (nb == ~0ull) bad = 1;
      }
      __syncthreads();
      TRACE_W(15);
      if (bad) { if (tid == 0) atomicOr(&err[3], 8); return; }  // tile pool exhausted (nothing has been written if this is the first band)
      // (tried: every workgroup starting at another tile and another eighth of it — after a resampling hundreds of particles copy
      //  from the same few parents' tiles.  No change: the parents' lines are served by the L2s, tools/rbpf_cow_trace.py)
      for (int q = wid; q < mtn; q += nw) {
        if (!((need_m >> q) & 1ull)) continue;
        const int qi = udiv16(q, mty_m), qj = q - qi * mty;
        // (the ring position and the lane are taken afresh in every trip — an LDS read, an opaque copy: as loop invariants the position
        //  and the lane's bitmap address were kept live across the copy and spilled, 2 x 8 bytes of scratch per lane)
        const unsigned long long nb = *reinterpret_cast<volatile unsigned long long*>(&need_base);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int slot_q = __popcll(need_m & ((1ull << q) - 1ull));   // (every wave holds the same ballot: no table of slots in LDS)
        const unsigned int nid = nb == kPoolScattered ? mt_id[q] : tile_at(P, nb + (unsigned long long)slot_q);
        // The fresh tile takes from the shared one ONLY what this band's update will not write: the pairs of the tile outside the
        // band and the band's untouched pairs (and the occupancy bits).  A touched pair is read from the shared tile (mt_src) and
        // written to the new one by the passes below anyway — copying it first moved every byte of the tile twice: the scan that
        // follows a resampling spent 45 of its 85 us there (~16 tiles x 8 KB per particle, read and written).
        const int t_map = (tx0 + qi) * M.TW + (ty0 + qj);
        const unsigned int id = tab[t_map];
        double2* const dst = reinterpret_cast<double2*>(P.lo + (size_t)nid * kTileCells);
        const double2* const src = reinterpret_cast<const double2*>(P.lo + (size_t)id * kTileCells);  // id 0 = the zero tile
        const int gx0 = (tx0 + qi) << kTSh, gy0 = (ty0 + qj) << kTSh;   // the tile's first cell
#pragma unroll
        for (int i = 0; i < kTileCells / 2 / kWave; ++i) {
          const int pi = i * kWave + ln;                                   // pair pi of the tile: row pi / 16, cells 2 (pi % 16), + 1
          const int cx = gx0 + (pi >> (kTSh - 1)), cy = gy0 + 2 * (pi & (kTS / 2 - 1));
          bool written = false;
          if ((unsigned int)(cx - x0) < (unsigned int)nr && cy >= miny && cy < miny + bw) {
            const int bp = __mul24(cx - x0, bw >> 1) + ((cy - miny) >> 1);  // the pair's index in the band's array
            if constexpr (C16) written = tile[bp] != 0u;
            else { const uint2 wv = reinterpret_cast<const uint2*>(tile)[bp]; written = (wv.x | wv.y) != 0u; }
          }
          if (!written) dst[pi] = src[pi];
        }
        if (ln < kTS) P.bm[(size_t)nid * kTS + ln] = P.bm[(size_t)id * kTS + ln];
        if (ln == 0) {
          P.ref[nid] = 1;
          tab[t_map] = nid;
          if (id != 0u) shed[t_map] = id;  // a (p, t) entry leaves a shared tile at most once between two resamples
          mt_id[q] = nid; atomicOr(&mt_priv_bits[q >> 5], 1u << (q & 31));  // (mt_src keeps the shared tile: what the passes below read)
        }
      }
      __syncthreads();
    }
    PHASE_STAMP_W(3);
    TRACE_W(10);
    auto finish_end = [&](int slot, int cx, int cy, double v0o, double vv) {
      *val_at(slot) = vv;
      const bool was = v0o >= c.cut_occ, now = vv >= c.cut_occ;
      if (was != now) toggled(cx, cy, now);
    };
    // 3a. end-point cells whose slot holds every event: one lane each, the events applied in beam order
    for (int o = tid; o < Bv; o += nthr) {
      const int ne = ecnt[o];
      if (ne == 0 || ne > kEv) continue;
      const int e = exy[o], cx = e & 0xFFFF, cy = e >> 16;
      const double v0o = *cell_ptr(cx, cy);
      double vv = v0o;
      if constexpr (kEv == 4) {
        // four events: sorted by (beam, kind) in registers — five compare-exchanges on the 16-bit entries, an empty entry last — and
        // applied in that order (the reference's: ascending beam)
        const uint2 raw = *reinterpret_cast<const uint2*>(ev + o * kEv);
        unsigned int k0 = raw.x & 0xFFFFu, k1 = raw.x >> 16, k2 = raw.y & 0xFFFFu, k3 = raw.y >> 16;
        k1 = ne > 1 ? k1 : 0xFFFFu; k2 = ne > 2 ? k2 : 0xFFFFu; k3 = ne > 3 ? k3 : 0xFFFFu;
        auto cx2 = [](unsigned int& a, unsigned int& b) { const unsigned int lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; };
        cx2(k0, k1); cx2(k2, k3); cx2(k0, k2); cx2(k1, k3); cx2(k1, k2);
        vv += (k0 & 1u) ? c.d_occ : c.d_free;
        if (k1 != 0xFFFFu) vv += (k1 & 1u) ? c.d_occ : c.d_free;
        if (k2 != 0xFFFFu) vv += (k2 & 1u) ? c.d_occ : c.d_free;
        if (k3 != 0xFFFFu) vv += (k3 & 1u) ? c.d_occ : c.d_free;
      } else {
        // the events in beam order without sorting: every beam that reaches the cell lies within a few beams of the slot's
        // own beam b0, so bit (beam - b0 + 32) of a 64-bit mask per kind orders them (checked per event; a stray one sends
        // the slot to the exhaustive path).  Beam indices are circular: the bits are walked from the one that stands for the
        // lowest ABSOLUTE beam index.
        unsigned int w4[kEv / 2];
        if constexpr (kEv == 8) { const uint4 raw = *reinterpret_cast<const uint4*>(ev + o * kEv); w4[0] = raw.x; w4[1] = raw.y; w4[2] = raw.z; w4[3] = raw.w; }
        else { const uint2 raw = *reinterpret_cast<const uint2*>(ev + o * kEv); w4[0] = raw.x; w4[1] = raw.y; }
        const int base = o - 32;
        unsigned long long m_free = 0ull, m_occ = 0ull;
        bool stray = false;
#pragma unroll
        for (int q = 0; q < kEv; ++q) {
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
    //     flagged like end-point cells, and their group's owner takes the value from val_hot.  The few that take kVeryHot adds or
    //     more (the robot's neighbours: up to half the beams each) go to the last wave but one instead, which works them out
    //     without the chain (add_repeated: a few hundred integer instructions per binade, worth it from about a hundred adds);
    //     the two waves run side by side, so the phase lasts as long as a chain of kVeryHot adds, not of the longest.
    if ((wid == nw - 1 || wid == nw - 2) && lane < kHotSide * kHotSide) {
      const int hi = lane / kHotSide, hx = rx - kHotSide / 2 + hi, hy = ry - kHotSide / 2 + (lane - hi * kHotSide);
      if ((unsigned int)(hx - x0) < (unsigned int)nr && hy >= miny && hy < miny + bw) {
        int t = __mul24(hx - x0, bw) + (hy - miny);
        asm volatile("" : "+v"(t));   // held as ONE register across add_repeated (else hx - x0 and hy - miny both are: a 4-byte spill in the 16-bit form)
        unsigned int f;
        if constexpr (C16) { const unsigned int hh = tile16[t]; f = (hh & 0x7FFFu) | ((hh & 0x8000u) << 16); } else f = tile[t];
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
          if constexpr (C16) { slot_claim(t, Bv + lane); tile16[t] = (unsigned short)(0x8000u | (unsigned int)cnq); }   // (nobody else touches this cell in this phase)
          else tile[t] = kFlag | ((unsigned int)(Bv + lane) << 16);
          ++n_distinct;
          finish_end(Bv + lane, hx, hy, v0o, vv);
        }
      }
    }
    TRACE_W(12);
    __syncthreads();  // every replayed / hot value is in its slot
    TRACE_W(13);
    PHASE_STAMP_W(4);
    // 3c. the pairs: a plain cell adds its count, an end-point or hot cell takes the value worked out for it, an untouched one
    //     keeps its own; the pair goes back as one 16-byte store (the tile is private to the particle and nobody else writes
    //     these cells)
    for (int first = 0; first < n_items; first += nthr) {
      // v holds this trip's log-odds: asked for in phase 2 (first trip) or, slot by slot, while the PREVIOUS trip's pairs were
      // worked on — as soon as a pair is stored its register takes the request for the same slot of the thread's next item, so
      // no trip waits for memory with nothing to do
      const int qn = first + nthr + tid;
      const bool next_live = qn < n_items;
      const int jn = udiv16(next_live ? qn : 0, PW_m), pcn = (next_live ? qn : 0) - __mul24(jn, PW);
      const int cxbn = A0 + kSl * jn, cyn = miny + 2 * pcn;
      const double* const ptrn = P.lo + (size_t)mt_src[map_tile(cxbn, cyn)] * kTileCells + in_tile(cxbn, cyn);   // (reads: the tile named when the band began)
      const int pibn = __mul24(cxbn - x0, PW) + pcn;
      // (a slot whose next pair is untouched — or outside the band, or beyond the last item — keeps whatever it holds: the next trip
      //  looks at v[i] only under the same test of the same tile words.  Zeroing it cost ten moves per pair, round 6)
      auto request_next = [&](int i) {
        if (next_live && (unsigned int)(cxbn + i - x0) < (unsigned int)nr) {
          bool any;
          if constexpr (C16) any = tile[pibn + i * PW] != 0u; else { const uint2 wn = tile2[pibn + i * PW]; any = (wn.x | wn.y) != 0u; }
          if (any) v[i] = *reinterpret_cast<const double2*>(ptrn + i * kTS);
        }
      };
      pairs(first, [&](int i, uint2 w, int cx, int cy, double* ptr, int) {
        if (w.x | w.y) {
          const bool plain0 = w.x != 0u && !(w.x & kFlag), plain1 = w.y != 0u && !(w.y & kFlag);
          const int c0 = plain0 ? (int)(w.x & 0xFFFFu) : 0, c1 = plain1 ? (int)(w.y & 0xFFFFu) : 0;
          const double o0 = v[i].x, o1 = v[i].y;
          double n0 = o0, n1 = o1;
          // (one lean loop per cell — an add, a count, a compare — rather than one predicated loop for both: with four workgroups
          //  per CU the kernel is bound by VALU issue, not by the latency of a chain of adds: 43.8 -> 41.8 us per 1000 particles)
#pragma unroll 1
          for (int a = 0; a < c0; ++a) n0 += c.d_free;
#pragma unroll 1
          for (int a = 0; a < c1; ++a) n1 += c.d_free;
          if ((w.x | w.y) & kFlag) {  // an end-point or hot cell takes the value worked out for it
            const int t0c = __mul24(cx - x0, bw) + (cy - miny);   // the pair's first cell in the band's array
            if (w.x & kFlag) n0 = *val_at(C16 ? slot_find(t0c) : (int)((w.x >> 16) & 0x7FFFu));
            if (w.y & kFlag) n1 = *val_at(C16 ? slot_find(t0c + 1) : (int)((w.y >> 16) & 0x7FFFu));
          }
          n_distinct += (plain0 ? 1 : 0) + (plain1 ? 1 : 0);
          *reinterpret_cast<double2*>(ptr) = double2{n0, n1};
          const bool tog0 = plain0 && ((o0 >= c.cut_occ) != (n0 >= c.cut_occ)), tog1 = plain1 && ((o1 >= c.cut_occ) != (n1 >= c.cut_occ));
          if (tog0 | tog1) {
            if (tog0) toggled(cx, cy, n0 >= c.cut_occ);
            if (tog1) toggled(cx, cy + 1, n1 >= c.cut_occ);
          }
        }
        request_next(i);
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
template __global__ void rbpf_raycast_box<512, 6, false, 8>(ScanC, TilePool, MapT, const double2* __restrict__, const double* __restrict__, const double* __restrict__, int* __restrict__, int* __restrict__, int* __restrict__, int, unsigned long long* __restrict__, NormArgs, int* __restrict__, int* __restrict__, int, int);
template __global__ void rbpf_raycast_box<512, 6, true, 8>(ScanC, TilePool, MapT, const double2* __restrict__, const double* __restrict__, const double* __restrict__, int* __restrict__, int* __restrict__, int* __restrict__, int, unsigned long long* __restrict__, NormArgs, int* __restrict__, int* __restrict__, int, int);
template __global__ void rbpf_raycast_box<512, 8, false, 8>(ScanC, TilePool, MapT, const double2* __restrict__, const double* __restrict__, const double* __restrict__, int* __restrict__, int* __restrict__, int* __restrict__, int, unsigned long long* __restrict__, NormArgs, int* __restrict__, int* __restrict__, int, int);
template __global__ void rbpf_raycast_box<512, 8, false, 4>(ScanC, TilePool, MapT, const double2* __restrict__, const double* __restrict__, const double* __restrict__, int* __restrict__, int* __restrict__, int* __restrict__, int, unsigned long long* __restrict__, NormArgs, int* __restrict__, int* __restrict__, int, int);
template __global__ void rbpf_raycast_box<512, 8, true, 8>(ScanC, TilePool, MapT, const double2* __restrict__, const double* __restrict__, const double* __restrict__, int* __restrict__, int* __restrict__, int* __restrict__, int, unsigned long long* __restrict__, NormArgs, int* __restrict__, int* __restrict__, int, int);
template __global__ void rbpf_raycast_box<1024, 8, false, 8>(ScanC, TilePool, MapT, const double2* __restrict__, const double* __restrict__, const double* __restrict__, int* __restrict__, int* __restrict__, int* __restrict__, int, unsigned long long* __restrict__, NormArgs, int* __restrict__, int* __restrict__, int, int);

}  // namespace tbnav_rk

#ifdef TBNAV_PHASE_PROF
#include <cstdio>
#include <map>
#include <vector>
namespace tbnav_rk {
void rbpf_prof_print_raycast() {
    unsigned long long tr[2][16][16];
    if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_trace), sizeof(tr)) == hipSuccess && tr[0][0][0]) {
      for (int g = 0; g < 2; ++g) {
        std::fprintf(stderr, "[raycast_box trace of workgroup %d, us since its first stamp; columns: entry, pose barrier, end-point barrier, flags, flag barrier, own events, walk, walk barrier, requests, barrier, tiles private, replay, overflow+hot, barrier, stores, (ring position known: only when tiles are made private)]\n", g ? 900 : 100);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 16; ++w) if (tr[g][w][0] && tr[g][w][0] < t0) t0 = tr[g][w][0];
        for (int w = 0; w < 16; ++w) {
          std::fprintf(stderr, "  wave %2d:", w);
          for (int i = 0; i < 16; ++i) std::fprintf(stderr, " %5.2f", tr[g][w][i] ? (double)(tr[g][w][i] - t0) * 0.01 : -1.0);
          std::fprintf(stderr, "\n");
        }
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
}  // namespace tbnav_rk
#endif
