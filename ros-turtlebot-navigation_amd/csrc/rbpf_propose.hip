// rbpf_propose.hip — the proposal side of ParticleFilter::SLAM (particle_filter.cpp:158-231, :383-437, :504-599;
// grid_mapper.cpp:69-133; sensor_model.cpp:43-112): the handle's mixture table, the production noise source, the
// one-pose likelihood (bmapping::GridMapper::likelihoodFieldModel), the per-particle scan matcher (option N1) and
// rbpf_propose itself — one workgroup per particle: k samples, one lookup per beam at their centre, stable-beam collapse,
// Gaussian proposal, new pose, weight *= eta.
#include "rbpf_device.hpp"

namespace tbnav_rk {

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
  // reference-field mode: where a stored-field lookup that reads kCodePending leaves cell + 1 (NULL: nobody asks)
  int* pend;
};
// a stored code as the lookups read it: a pending cell (the reference-field mode's lazy brushfire has not written it) is reported
__device__ __forceinline__ uint16_t stored_code(const GridC& g, const DistSrc& d, int ci, int cj) {
  const uint16_t v = d.code[(size_t)ci * g.xsize + cj];
  if (v == kCodePending && d.pend) *d.pend = ci * g.xsize + cj + 1;
  return v;
}
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
  return stored_code(g, d, ci, cj);
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
// Box-Muller pair number P of scan `scan` under `seed`: normals 2P and 2P + 1 of the scan's stream (fp64: 53-bit uniforms, log, sqrt,
// sincospi).  What rbpf_sample_normals stores and what rbpf_propose draws in place: one definition.
__device__ __forceinline__ void normal_pair(unsigned long long seed, unsigned long long scan, size_t P, double& a_out, double& b_out) {
  unsigned int r[4];
  philox4x32_10((scan << 40) + P, seed, r);
  const unsigned long long a = ((unsigned long long)r[0] << 32) | r[1], b = ((unsigned long long)r[2] << 32) | r[3];
  const double u1 = ((double)(a >> 11) + 0.5) * 0x1.0p-53, u2 = ((double)(b >> 11) + 0.5) * 0x1.0p-53;
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  a_out = rad * cs; b_out = rad * sn;
}
// normals g, g + 1, g + 2 of the stream: two pairs whatever g's parity
__device__ __forceinline__ void normal3(const NoiseSrc& ns, size_t g, double& n0, double& n1, double& n2) {
  double a0, b0, a1, b1;
  normal_pair(ns.seed, ns.scan, g >> 1, a0, b0);
  normal_pair(ns.seed, ns.scan, (g >> 1) + 1, a1, b1);
  const bool odd = (g & 1) != 0;
  n0 = odd ? b0 : a0; n1 = odd ? a1 : b0; n2 = odd ? b1 : a1;
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
  return stored_code(g, d, ci, cj);
}


__global__ void rbpf_mix_lut(ScanC c, double* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < kMixLut) out[q] = beam_mixture(c, (uint16_t)q);
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
                                    size_t out_stride, size_t beam_stride, size_t base, size_t z_index,
                                    size_t z_slot) {
  scan += blockIdx.y; out += blockIdx.y * out_stride; host_beams += blockIdx.y * beam_stride; dev_beams += blockIdx.y * beam_stride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_copy; i += gridDim.x * blockDim.x) dev_beams[i] = host_beams[i];
  auto pair = [&](size_t P, double& a_out, double& b_out) { normal_pair(seed, scan, P, a_out, b_out); };
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
                                                                const double* __restrict__ mixlut, NoiseSrc ns, int* __restrict__ pend) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  if (gate_prev && *gate_prev) return;  // the scan before this one resamples: see rbpf_raycast_box
  // DN: the noise is drawn here and workgroup 0 carries the beam table over (NoiseSrc).  A template argument, not a launch-time
  // switch: with both forms in one body the stored-normals form — same source as round 4's — ran 3.3 us slower (30.7 -> 34.0 us per
  // 1000 particles by events: registers and scheduling of code it never executes)
  constexpr bool dn = DN;
  if (dn && blockIdx.x == 0) {
    // (two copies: the fine-grained one — uncached, so what the other workgroups of THIS launch read is what was stored, on whichever
    //  XCD they run, without a cache invalidate per wave (four thousand of those emptied the L2s and cost 50 us a scan) — and the
    //  ordinary one the map update, a later launch, reads through the caches)
    //  (system-scope stores and loads, word by word: a plain access to fine-grained memory may still be served by the XCD's L2)
    for (int b = threadIdx.x; b < c.Bv; b += NT) {
      const double2 v = ns.host_beams[b];
      __hip_atomic_store(&ns.fg_beams[b].x, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&ns.fg_beams[b].y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ns.dev_beams[b] = v;
    }
    if (threadIdx.x == 0) {
      double va, vb;
      normal_pair(ns.seed, ns.scan, ns.z_index >> 1, va, vb);
      *ns.z_out = (ns.z_index & 1) ? vb : va;
    }
    __builtin_amdgcn_s_waitcnt(0);   // this thread's stores have been acknowledged (vmcnt counts stores) ...
    __syncthreads();                 // ... and so have every other thread's, before the sequence number says so
    // kReadyCopies copies of the sequence number, a cache line apart: a thousand workgroups looking at ONE word queue at the memory
    // side (a single address retires ~90 requests per microsecond: the first form of this cost every scan 80 us); workgroup b looks
    // at copy b mod kReadyCopies
    // (a RELEASE store, round 6: the table's stores above are ordered before the number by the language's rules too, not only by the
    //  s_waitcnt + barrier — eight threads of one workgroup pay for it, once per scan)
    if (threadIdx.x < kReadyCopies) __hip_atomic_store(ns.ready + threadIdx.x * kReadyStride, ns.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  const unsigned int* const my_ready = dn ? ns.ready + (blockIdx.x & (kReadyCopies - 1)) * kReadyStride : nullptr;
  const int p = blockIdx.x - (dn ? 1 : 0);
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
  const double* z = normals + (size_t)p * c.stride_normals;   // (dn: never dereferenced)
  const size_t zg = ns.base + (size_t)p * c.stride_normals;   // dn: the same place in the ensemble's stream
  __shared__ double sh_zz[3];   // dn: the new pose's three normals, drawn by the thread that would be sample k
  const int nocc = n_occ[p];
  // a particle whose field is authoritative (injected / whole-field fresh) always reads it
  // (the tile pointers are set unconditionally — nW == 0 means "no tile" — so that the compiler can see they are LDS
  //  addresses and use ds_read instead of flat loads in the lookups)
  unsigned long long* const tile_bm = reinterpret_cast<unsigned long long*>(ulist + c.Bv);
  DistSrc ds{code, occ_of(P, M, trow_occ, p), win[p], skip[p] == skip_eq ? 0 : df_mode,
             tile_bm, reinterpret_cast<const int*>(tile_bm), 0, 0, 0, 0, nullptr, pend ? pend + p : nullptr};
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
  if (!dn && c.icp_ok && tid < k) { zj0 = z[3 * tid + 0]; zj1 = z[3 * tid + 1]; zj2 = z[3 * tid + 2]; }
  double zz0 = 0.0, zz1 = 0.0, zz2 = 0.0, w_old = 0.0;
  if (c.icp_ok && wid == 0) { if (!dn) { zz0 = z[3 * k + 0]; zz1 = z[3 * k + 1]; zz2 = z[3 * k + 2]; } w_old = weight[p]; }
  // (the beam table: has workgroup 0 published it already?  One look, no waiting — the workgroups dispatched after the first few
  //  microseconds find it and request their beams here, under everything else, as before; the others come back to it below)
  bool have_beams = true;
  if (dn) have_beams = __hip_atomic_load(my_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ns.seq;   // (fine-grained memory: no cache holds it)
  auto beam_in = [&](int b) -> double2 {
    if (!dn) return beams[b];
    return double2{__hip_atomic_load(&ns.fg_beams[b].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __hip_atomic_load(&ns.fg_beams[b].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)};
  };
  // (a table of at most NT entries — maps up to 512 x 512 cells at 256 threads — is requested WHOLE here, with the pose: which
  //  entries the window needs depends on the pose, and waiting for it made the staging below three dependent round trips)
  const int tt_all = ds.occ.TW * ds.occ.TW;
  const bool whole_table = ds.mode == 2 && occ_half > 0 && nocc && tt_all <= NT && tt_all <= 256;
  unsigned int my_id = 0u;
  if (whole_table && tid < tt_all) my_id = ds.occ.tab[tid];
  if (dn && c.icp_ok) {
    // drawn while the loads above are in flight (nothing here depends on them): the particle's 3k + 3 normals are (3k + 3) / 2 (+ 1)
    // Box-Muller PAIRS of the stream — one pair per thread, the first threads of the workgroup (k = 50: 77 or 78 threads, two waves
    // side by side; one thread per sample drew two pairs each and took twice as long on the workgroup's critical path) — straight
    // into the samples' LDS slots (smp[3j + q] is local normal 3j + q) and, the last three, the new pose's
    const int n_loc = 3 * k + 3;
    const size_t P0 = zg >> 1;
    const int npairs = (int)(((zg + (size_t)n_loc - 1) >> 1) - P0) + 1;
    for (int t = tid; t < npairs; t += NT) {
      double a, b;
      normal_pair(ns.seed, ns.scan, P0 + (size_t)t, a, b);
      const int l0 = (int)(2 * (P0 + (size_t)t) - zg + 1) - 1;   // local index of the pair's first normal: -1 .. n_loc - 1
      if (l0 >= 0) { if (l0 < 3 * k) smp[l0] = a; else sh_zz[l0 - 3 * k] = a; }
      if (l0 + 1 < n_loc) { if (l0 + 1 < 3 * k) smp[l0 + 1] = b; else sh_zz[l0 + 1 - 3 * k] = b; }
    }
  }
  TRACE_P(0);
  const double th0 = uniform_d(th0v), x0 = uniform_d(x0v), y0 = uniform_d(y0v);
  double mu0[3];
  if (!c.icp_ok) {
    // ICP failed: the pose moves by the odometry motion model (particle_filter.cpp:161-176, :295-322) — every thread works it
    // out (three draws, two sincos), and the LDS slice of the bitmap is staged round THAT pose's sensor
    double z0n, z1n, z2n;
    if (dn) normal3(ns, zg, z0n, z1n, z2n); else { z0n = z[0]; z1n = z[1]; z2n = z[2]; }
    const double w0 = c.Lm[0] * z0n, w1 = c.Lm[1] * z1n, w2 = c.Lm[2] * z2n;
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
  if (have_beams) for (int b = tid; b < c.Bv; b += NT) lbeams[b] = beam_in(b);  // visible after the next barrier
  // (not published yet when this wave looked: wait for it just before the barrier in front of step 1 — bounded: a table that never
  //  arrives raises err[3] bit 4 instead of hanging the device)
  // Returns true (wave-uniform) when the table never came: the workgroup then leaves WITHOUT writing a pose or a weight (the callers
  // turn the barrier that follows into a vote, round 6) — up to round 5 such a wave went on with whatever fg_beams held.
  // On the consumer side the loads stay relaxed: every one of them is a system-scope access to fine-grained memory, served by the
  // memory side and not by a cache, and a wave issues them in order after its lane 0 has SEEN the number — an acquire here is a cache
  // invalidate per wave, four thousand per scan, which is what this form was built to avoid (gfx950 only, like the rest of the library;
  // the default form — TBNAV_RBPF_OPT_NOISE_IN_KERNEL 0 — has no hand-over inside a launch at all).
  auto late_beams = [&]() -> bool {
    if (have_beams) return false;
    int dead = 0;
    if (lane == 0) {
      // (bounded by a count of looks, not by the clock: wall_clock64 is a message to a unit every wave of the chip shares — four
      //  thousand waves asking it once per look serialised there and cost every scan 50 us; a look is a trip to the memory side,
      //  ~1 us with the sleep: the bound is a few tenths of a second)
      int looks = 0;
      while (__hip_atomic_load(my_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != ns.seq) {
        if (++looks > 200000) { atomicOr(&err[3], 16); dead = 1; break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    dead = __builtin_amdgcn_readfirstlane(dead);
    __builtin_amdgcn_wave_barrier();
    if (dead) return true;
    for (int b = tid; b < c.Bv; b += NT) lbeams[b] = beam_in(b);   // (issued after lane 0 has SEEN the number; uncached: what was stored before it)
    return false;
  };
  // the barrier behind late_beams(): with in-kernel noise it also carries the waves' "the table never came" to everybody
  auto barrier_or_leave = [&](bool dead_wave) -> bool {
    if (!dn) { __syncthreads(); return false; }
    return __syncthreads_or(dead_wave ? 1 : 0) != 0;
  };
  for (int q = tid; q < kMixLds; q += NT) sh_mix[q] = mixlut[q];
  double Tc[4];  // sensor transform at the centre of the samples
  sensor_transform(c, mu0[0], mu0[1], mu0[2], Tc);
  Tc[0] = uniform_d(Tc[0]); Tc[1] = uniform_d(Tc[1]); Tc[2] = uniform_d(Tc[2]); Tc[3] = uniform_d(Tc[3]);
  // (the normals have arrived with the pose: parked in the samples' own LDS slots until wave 0 turns them into samples, so that
  //  they do not hold six registers through the staging)
  if (!dn && c.icp_ok && tid < k) { smp[3 * tid + 0] = zj0; smp[3 * tid + 1] = zj1; smp[3 * tid + 2] = zj2; }
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
    if (barrier_or_leave(late_beams())) return;   // (workgroup-uniform: nothing written yet)
    staged = true;
  }
  if (!staged) { if (barrier_or_leave(late_beams())) return; }  // lbeams / sh_mix / sh_def / sh_zz
  TRACE_P(3);
  if (dn && c.icp_ok) { zz0 = sh_zz[0]; zz1 = sh_zz[1]; zz2 = sh_zz[2]; }
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
      const bool parked = dn || j < NT;   // (dn: every sample's normals are in its slot)
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
template __global__ void rbpf_propose<kProposeThreads, false>(ScanC, const double2* __restrict__, const uint16_t* __restrict__, TilePool, MapT, const int* __restrict__, const int* __restrict__, int, int, int, int, const int* __restrict__, const int4* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, Trace, double* __restrict__, int* __restrict__, const int* __restrict__, const double* __restrict__, NoiseSrc, int* __restrict__);
template __global__ void rbpf_propose<kProposeThreads, true>(ScanC, const double2* __restrict__, const uint16_t* __restrict__, TilePool, MapT, const int* __restrict__, const int* __restrict__, int, int, int, int, const int* __restrict__, const int4* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, Trace, double* __restrict__, int* __restrict__, const int* __restrict__, const double* __restrict__, NoiseSrc, int* __restrict__);
template __global__ void rbpf_propose<2 * kProposeThreads, false>(ScanC, const double2* __restrict__, const uint16_t* __restrict__, TilePool, MapT, const int* __restrict__, const int* __restrict__, int, int, int, int, const int* __restrict__, const int4* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, Trace, double* __restrict__, int* __restrict__, const int* __restrict__, const double* __restrict__, NoiseSrc, int* __restrict__);
template __global__ void rbpf_propose<2 * kProposeThreads, true>(ScanC, const double2* __restrict__, const uint16_t* __restrict__, TilePool, MapT, const int* __restrict__, const int* __restrict__, int, int, int, int, const int* __restrict__, const int4* __restrict__, const double* __restrict__, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, Trace, double* __restrict__, int* __restrict__, const int* __restrict__, const double* __restrict__, NoiseSrc, int* __restrict__);

}  // namespace tbnav_rk

#ifdef TBNAV_PHASE_PROF
#include <cstdio>
#include <map>
#include <vector>
namespace tbnav_rk {
void rbpf_prof_print_propose() {
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
}
}  // namespace tbnav_rk
#endif
