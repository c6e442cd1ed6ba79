// rbpf_normalize.hpp — normalizeWeights / effectiveParticles / lowVarianceResampling (particle_filter.cpp:442-500) as device
// code: the order-sensitive sums by one lane in the reference's order (or, beyond one chunk, bit for bit WITHOUT the chain of
// dependent adds: chain_exact), the selection by bisection.  One body, two homes: rbpf_normalize (rbpf_resample.hip) and
// workgroup 0 of rbpf_raycast_box (rbpf_raycast.hip), where it rides beside the map update.
#ifndef TBNAV_RBPF_NORMALIZE_HPP
#define TBNAV_RBPF_NORMALIZE_HPP
#include "rbpf_device.hpp"

namespace tbnav_rk {
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

}  // namespace tbnav_rk
#endif  // TBNAV_RBPF_NORMALIZE_HPP
