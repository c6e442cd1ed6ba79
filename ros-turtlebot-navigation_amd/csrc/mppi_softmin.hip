// mppi_softmin.hip — the soft-min half of controller::MPPI::newControls (mppi.cpp:112-137) and the small kernels round it:
//   mppi_partials       grid (K-slices, T): per-time-step min / soft-min partial sums over one K-slice -> records[T][S][8]   (mppi.cpp:115-121)
//   mppi_merge_records  sharded small-K ticks: fold the fused kernel's fine records into the K-slice records (and publish them)
//   mppi_combine        any number of workgroups: merge the records of all slices / shards, update + clamp u, emit u(:,0); the
//                       shift is applied on read by the next tick                                                          (mppi.cpp:118-137)
//   mppi_direct_publish / _collect   the sharded tick's records as self-validating 8-byte words in the peers' buffers
//   mppi_sample_noise   Philox4x32-10 + Box-Muller, production replacement of mppi.cpp:173-184;  mppi_unpack_noise: reference draw
//                       order [K][T][2] -> duL / duR [T][K];  mppi_tick_advance / _set: the replayed graph's tick word;  debug kernels
// Shared types, the noise source and the kernel declarations: mppi_device.hpp.
#include "mppi_device.hpp"

namespace tbnav_mk {

__device__ __forceinline__ double block_min(double v, double* scratch) {
  v = tbnav::wave_min_dpp(v);
  const int wid = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) scratch[wid] = v;
  __syncthreads();
  double r = scratch[0];
  for (int w = 1; w < kSliceThreads / kWave; ++w) r = fmin(r, scratch[w]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = tbnav::wave_sum_dpp(v);
  const int wid = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) scratch[wid] = v;
  __syncthreads();
  double r = scratch[0];
  for (int w = 1; w < kSliceThreads / kWave; ++w) r += scratch[w];
  __syncthreads();
  return r;
}

// grid = (S, T).  Block (s, i) reduces time step i over rollouts [s*kSlice, (s+1)*kSlice).
// prefix_rows > 0 (mppi_rollout_prefix ran): rows i < prefix_rows of J hold the exclusive prefix E(i) and total[k] the
// rollout's whole cost S: J(i, k) = S - E(i) is formed here (the 8 * K bytes of `total` are re-read by every time step's
// workgroups: L2 hits).
__global__ __launch_bounds__(kSliceThreads) void mppi_partials(int T, int K, int S, Lam lam,
                                                               const double* __restrict__ J,
                                                               const double* __restrict__ duL,
                                                               const double* __restrict__ duR,
                                                               double* __restrict__ records, int prefix_rows,
                                                               const double* __restrict__ total) {
  __shared__ double scratch[kSliceThreads / kWave];
  // which rows the rollout kernel touched LAST are the ones still in L2 / the Infinity Cache: the backward suffix pass of the
  // round-2 kernels ends at row 0, the prefix-form kernel's forward pass at row T - 1 — start there (35.7 -> us at K = 65536)
  const int s = blockIdx.x, i = prefix_rows > 0 ? T - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int base = s * kSlice;
  const double inf = __builtin_huge_val();
  double j[kSliceItems], l[kSliceItems], r[kSliceItems], tot[kSliceItems];
  const bool pre = i < prefix_rows;
  double mn = inf;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < kSliceItems; ++it) {
    const int k = base + it * kSliceThreads + threadIdx.x;  // coalesced across lanes
    const bool ok = k < K;
    j[it] = ok ? J[(size_t)i * K + k] : inf;
    tot[it] = (ok && pre) ? total[k] : inf;   // (requested with the other loads; subtracted below, once everything is on its way)
    l[it] = ok ? duL[(size_t)i * K + k] : 0.0;
    r[it] = ok ? duR[(size_t)i * K + k] : 0.0;
    cnt += ok ? 1 : 0;
  }
#pragma unroll
  for (int it = 0; it < kSliceItems; ++it) {
    // J(i) = S - E(i).  A missing rollout (tot = +inf by construction) and a rollout whose total OVERFLOWED to +inf both end as
    // J = +inf, weight 0 — what the suffix-sum kernels give the latter (its prefix row alone would be a FINITE E(i) and look like
    // the cheapest rollout of the step; round-3 advisor finding)
    if (pre) j[it] = (tot[it] == inf) ? inf : tot[it] - j[it];
    mn = fmin(mn, j[it]);
  }
  mn = block_min(mn, scratch);
  double A = 0, B = 0, C = 0, D = 0, E = 0;
#pragma unroll
  for (int it = 0; it < kSliceItems; ++it) {
    // exp(-(J - min)/lambda) with the reference's association: (J - min) * -1.0 / lambda (mppi.cpp:117)
    const double e = (j[it] == inf) ? 0.0 : exp(div_lambda((j[it] - mn) * -1.0, lam));
    A += e;
    B += e * l[it];
    C += e * r[it];
    D += l[it];
    E += r[it];
  }
  A = block_sum(A, scratch);
  B = block_sum(B, scratch);
  C = block_sum(C, scratch);
  D = block_sum(D, scratch);
  E = block_sum(E, scratch);
  const double n = block_sum((double)cnt, scratch);
  if (threadIdx.x == 0) {
    double* rec = records + ((size_t)i * S + s) * TBNAV_MPPI_REC;
    rec[0] = mn; rec[1] = A; rec[2] = B; rec[3] = C; rec[4] = D; rec[5] = E; rec[6] = n; rec[7] = 0.0;
  }
}


// Fold the fused kernel's fine records ([T][Sf][8], one per R rollouts) into the K-slice records the sharding
// interface exchanges ([T][S][8], one per kSlice = 2048 rollouts): grid (S, T), one wave each.  Same algebra as the
// combine's first half — re-base every partial sum to the common minimum — so the result equals the partials
// kernel's record for that slice up to the association of the sums.
// (the direct exchange's sending side, when the records are produced here: see mppi_direct_publish further down)
__global__ __launch_bounds__(kWave) void mppi_merge_records(int T, int Sf, int per_slice, int S, Lam lam,
                                                            const double* __restrict__ fine, double* __restrict__ records, DirectPub pub) {
  const int s = blockIdx.x, i = blockIdx.y, lane = threadIdx.x;
  const int r0 = s * per_slice, r1 = min(Sf, r0 + per_slice);
  const double inf = __builtin_huge_val();
  double M = inf;
  for (int r = r0 + lane; r < r1; r += kWave) {
    const double* rec = fine + ((size_t)i * Sf + r) * TBNAV_MPPI_REC;
    if (rec[6] > 0.0) M = fmin(M, rec[0]);
  }
  M = tbnav::wave_min_dpp(M);
  double A = 0, B = 0, C = 0, D = 0, E = 0, n = 0;
  for (int r = r0 + lane; r < r1; r += kWave) {
    const double* rec = fine + ((size_t)i * Sf + r) * TBNAV_MPPI_REC;
    if (rec[6] > 0.0) {
      const double sc = exp(div_lambda((rec[0] - M) * -1.0, lam));
      A += sc * rec[1]; B += sc * rec[2]; C += sc * rec[3];
      D += rec[4]; E += rec[5]; n += rec[6];
    }
  }
  A = tbnav::wave_sum_dpp(A); B = tbnav::wave_sum_dpp(B); C = tbnav::wave_sum_dpp(C);
  D = tbnav::wave_sum_dpp(D); E = tbnav::wave_sum_dpp(E); n = tbnav::wave_sum_dpp(n);
  if (lane == 0) {
    double* out = records + ((size_t)i * S + s) * TBNAV_MPPI_REC;
    out[0] = M; out[1] = A; out[2] = B; out[3] = C; out[4] = D; out[5] = E; out[6] = n; out[7] = 0.0;
  }
  if (pub.peers) {
    // direct exchange: this record goes straight into every rank's buffer as tagged words (mppi_direct_publish's layout) — the
    // fold and the publish are one launch
    __shared__ double rec8[TBNAV_MPPI_REC];
    if (lane == 0) { rec8[0] = M; rec8[1] = A; rec8[2] = B; rec8[3] = C; rec8[4] = D; rec8[5] = E; rec8[6] = n; rec8[7] = 0.0; }
    __syncthreads();
    const size_t nrec = (size_t)T * S * TBNAV_MPPI_REC, base = ((size_t)i * S + s) * TBNAV_MPPI_REC;
    const unsigned long long tag = (unsigned long long)pub.seq << 32;
    for (int j = lane; j < pub.P * 2 * TBNAV_MPPI_REC; j += kWave) {
      const int q = j / (2 * TBNAV_MPPI_REC), w = j - q * 2 * TBNAV_MPPI_REC, f = w >> 1;
      if (pub.only_self && q != pub.me) continue;
      const unsigned long long b = (unsigned long long)__double_as_longlong(rec8[f]);
      unsigned long long* dst = pub.peers[q] + ((size_t)(pub.parity * pub.P + pub.me) * nrec + base + f) * 2 + (w & 1);
      __hip_atomic_store(dst, tag | ((w & 1) ? (b >> 32) : (b & 0xFFFFFFFFull)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Merge the G*S partial records of every time step (records: [G][T][S][8]) and update u(:,i)
// (mppi.cpp:118-125).  Each time step gets a group of `tpr` lanes (the power of two >= the record count, at
// most a wave): 64/tpr steps per wave, xor-shuffle reductions inside the group.  Any number of workgroups:
// the updated controls are written UNSHIFTED to u_out and the shift is applied on read by the next tick
// (USrc), u(:,0) goes to `out` (mppi.cpp:129-131).
__device__ __forceinline__ double direct_load(const DirectSrc& d, size_t idx, bool& failed) {
  unsigned long long* w = const_cast<unsigned long long*>(d.w0) + 2 * idx;
  unsigned long long lo = 0ull, hi = 0ull;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    lo = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    hi = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned int)(lo >> 32) == d.seq && (unsigned int)(hi >> 32) == d.seq) break;
    if (wall_clock64() - t0 > d.budget) {
      __hip_atomic_fetch_or(d.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_or(d.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      lo = hi = 0ull;
      failed = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
}

// NR whole records (7 fields = 14 consecutive words each) at once: every word is requested before any is looked at — the
// buffer is fine-grained memory, every load a trip to the fabric, and one field after the other would be fourteen of them
// in a row per record; only records whose words do not all carry the tick's number yet are asked for again.
template <int NR>
__device__ __forceinline__ bool direct_load_records(const DirectSrc& d, const size_t (&idx)[NR], const bool (&have)[NR], double (&out)[NR][7]) {
  unsigned long long w[NR][14];
  bool done[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    done[q] = !have[q];
#pragma unroll
    for (int f = 0; f < 7; ++f) out[q][f] = 0.0;
  }
  const unsigned long long t0 = wall_clock64();
  for (;;) {
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (!done[q]) {
        unsigned long long* p = const_cast<unsigned long long*>(d.w0) + 2 * idx[q];
#pragma unroll
        for (int k = 0; k < 14; ++k) w[q][k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    bool all = true;
#pragma unroll
    for (int q = 0; q < NR; ++q)
      if (!done[q]) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 14; ++k) ok = ok && (unsigned int)(w[q][k] >> 32) == d.seq;
        if (ok) {
          done[q] = true;
#pragma unroll
          for (int f = 0; f < 7; ++f) out[q][f] = __longlong_as_double((long long)((w[q][2 * f + 1] << 32) | (w[q][2 * f] & 0xFFFFFFFFull)));
        } else all = false;
      }
    if (all) break;
    if (wall_clock64() - t0 > d.budget) {
      __hip_atomic_fetch_or(d.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_fetch_or(d.err_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;  // (the records that never came stay zero; the caller leaves its time step's controls as they were)
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

template <int kKeep, int MODE>
__global__ __launch_bounds__(256) void mppi_combine(int T, int G, int S, Lam lam, double umax, USrc u,
                                                    const double* __restrict__ records, double* __restrict__ u_out,
                                                    double* __restrict__ out, double* __restrict__ out_host, double seq, DirectSrc ds) {
  // MODE 0 — one group of records, the single-GPU tick — carries none of the exchange's failure handling: a template argument, not a
  // launch-time test (as one body the headline tick ran 0.2 us slower than round 4's: 7.95 against 7.75 us).
  constexpr bool DIRECT = MODE == 2, GATHERED = MODE == 1;
  // (DIRECT: `records` is not read — field f of record (g, i, sl) is polled for in the exchange buffer, same index)
  // (an exchange that has timed out once stays dead: the ticks queued behind it must not each wait the whole bound again)
  const bool dead = DIRECT && __hip_atomic_load(ds.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  // (a time step whose peers' records did not all arrive — now or in an earlier tick — is NOT updated: its controls go out as they
  //  came in, never a soft-min over this rank's shard alone; the error word reaches the host with the next enqueue / last_controls)
  // (the other exchange, an all-gather: a rank whose own rollouts failed joins it with records whose count is NEGATIVE (kPoison) —
  //  the same rule: a time step that sees one is not updated, the error words are raised, every rank returns the error from its
  //  next enqueue / last_controls.  Round-4 advisor finding: NaN records used to stand for this and came out of the clamp as
  //  u = -max_wheel_vel on every healthy rank.)
  bool failed = dead;
  auto field = [&](const double* rec, int f) { return DIRECT ? (dead ? 0.0 : direct_load(ds, (size_t)(rec - records) + f, failed)) : rec[f]; };
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave, nw = blockDim.x / kWave;
  const int R = G * S;
  int tpr = 1;
  while (tpr < R && tpr < kWave) tpr <<= 1;
  const int spw = kWave / tpr, sub = lane / tpr, l = lane - sub * tpr;
  const int i = (blockIdx.x * nw + wid) * spw + sub;
  const bool valid = i < T;
  // the warm-start controls do not depend on the records: fetch them first, under the record loads
  const double u_l = valid ? u.get(0, i, T) : 0.0, u_r = valid ? u.get(1, i, T) : 0.0;
  // Up to kKeep records per lane stay in registers (2: at most 128 records per step — the K = 1024 tick, whose critical
  // path should not carry idle slots; 4 / 8: up to 256 / 512 — the fused kernel with 16 rollouts per workgroup up to K = 4096 / 8192);
  // beyond that the second pass re-reads them (L1/L2 hits).
  const bool keep = R <= kKeep * tpr;
  double rk[kKeep][7];
  if constexpr (DIRECT) {
    size_t idx[kKeep];
    bool hv[kKeep];
#pragma unroll
    for (int q = 0; q < kKeep; ++q) {
      const int r = l + q * tpr;
      hv[q] = valid && keep && r < R && !dead;
      const int g = (hv[q] && G > 1) ? r / S : 0, sl = hv[q] ? r - g * S : 0;
      idx[q] = (((size_t)g * T + (valid ? i : 0)) * S + sl) * TBNAV_MPPI_REC;
    }
    if constexpr (kKeep <= 2) failed = !direct_load_records<kKeep>(ds, idx, hv, rk) || failed;   // (the K = 1024 tick: both records' words in flight together)
    else {
#pragma unroll
      for (int q = 0; q < kKeep; ++q) {
        const size_t i1[1] = {idx[q]};
        const bool h1[1] = {hv[q]};
        double o1[1][7];
        failed = !direct_load_records<1>(ds, i1, h1, o1) || failed;
#pragma unroll
        for (int f = 0; f < 7; ++f) rk[q][f] = o1[0][f];
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < kKeep; ++q) {
      const int r = l + q * tpr;
      const bool have = valid && keep && r < R;
      const int g = (have && G > 1) ? r / S : 0, sl = have ? r - g * S : 0;  // (one group — every single-GPU tick: no division)
      const double* rec = records + (((size_t)g * T + (valid ? i : 0)) * S + sl) * TBNAV_MPPI_REC;
#pragma unroll
      for (int f = 0; f < 7; ++f) rk[q][f] = have ? rec[f] : 0.0;  // n == 0 marks "no record"
    }
  }
  double M = __builtin_huge_val();
  if (keep) {
#pragma unroll
    for (int q = 0; q < kKeep; ++q) { if (rk[q][6] > 0.0) M = fmin(M, rk[q][0]); if constexpr (GATHERED) failed = failed || rk[q][6] < 0.0; }
  } else if (valid) {
    for (int r = l; r < R; r += tpr) {
      const int g = r / S, sl = r - g * S;
      const double* rec = records + (((size_t)g * T + i) * S + sl) * TBNAV_MPPI_REC;
      const double rn = field(rec, 6);
      if (rn > 0.0) M = fmin(M, field(rec, 0));
      if constexpr (GATHERED) failed = failed || rn < 0.0;
    }
  }
  if (tpr == kWave) M = tbnav::wave_min_dpp(M);  // a whole wave per time step: reductions on the DPP network
  else for (int off = tpr >> 1; off > 0; off >>= 1) M = fmin(M, __shfl_xor(M, off, kWave));
  double W = 0, NL = 0, NR = 0, SD = 0, SE = 0, SN = 0;
  if (keep) {
#pragma unroll
    for (int q = 0; q < kKeep; ++q)
      if (rk[q][6] > 0.0) {
        const double sc = exp(div_lambda((rk[q][0] - M) * -1.0, lam));
        W += sc * rk[q][1]; NL += sc * rk[q][2]; NR += sc * rk[q][3];
        SD += rk[q][4]; SE += rk[q][5]; SN += rk[q][6];
      }
  } else if (valid) {
    for (int r = l; r < R; r += tpr) {
      const int g = r / S, sl = r - g * S;
      const double* rec = records + (((size_t)g * T + i) * S + sl) * TBNAV_MPPI_REC;
      const double rn = field(rec, 6);
      if (rn > 0.0) {
        const double sc = exp(div_lambda((field(rec, 0) - M) * -1.0, lam));
        W += sc * field(rec, 1); NL += sc * field(rec, 2); NR += sc * field(rec, 3);
        SD += field(rec, 4); SE += field(rec, 5); SN += rn;
      }
    }
  }
  if (tpr == kWave) {
    W = tbnav::wave_sum_dpp(W); NL = tbnav::wave_sum_dpp(NL); NR = tbnav::wave_sum_dpp(NR);
    SD = tbnav::wave_sum_dpp(SD); SE = tbnav::wave_sum_dpp(SE); SN = tbnav::wave_sum_dpp(SN);
  } else {
    for (int off = tpr >> 1; off > 0; off >>= 1) {
      W += __shfl_xor(W, off, kWave); NL += __shfl_xor(NL, off, kWave); NR += __shfl_xor(NR, off, kWave);
      SD += __shfl_xor(SD, off, kWave); SE += __shfl_xor(SE, off, kWave); SN += __shfl_xor(SN, off, kWave);
    }
  }
  if constexpr (MODE != 0) {  // any lane of the time step's group
    int fl = failed ? 1 : 0;
    for (int off = tpr >> 1; off > 0; off >>= 1) fl |= __shfl_xor(fl, off, kWave);
    failed = fl != 0;
  }
  if (valid && l == 0) {
    W += 1e-8 * SN;  // the reference adds 1e-8 to every weight before normalising (mppi.cpp:117)
    double ul = u_l + (NL + 1e-8 * SD) / W;
    double ur = u_r + (NR + 1e-8 * SE) / W;
    // std::clamp(u, -max, max), mppi.cpp:124-125 — by its definition (v < lo ? lo : hi < v ? hi : v), so that a NaN goes through
    // as the reference's does instead of coming out as -max (fmin / fmax return the other operand)
    ul = (ul < -umax) ? -umax : ((umax < ul) ? umax : ul);
    ur = (ur < -umax) ? -umax : ((umax < ur) ? umax : ur);
    if (MODE != 0 && failed) {
      ul = u_l; ur = u_r;
      if (!DIRECT && ds.err) {   // (the polling loads have raised them already)
        __hip_atomic_fetch_or(ds.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_fetch_or(ds.err_dev, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    u_out[i] = ul;
    u_out[T + i] = ur;
    if (i == 0) {
      out[0] = ul; out[1] = ur;
      if (out_host) {
        // synchronous ticks: the answer also lands in mapped pinned host memory, without a copy.  The tick number goes
        // last, behind a system-scope fence, so a host that sees it also sees the two values.  (Not done for
        // enqueue-only ticks: the fence and the write over the fabric sit on the kernel's critical path.)
        out_host[0] = ul; out_host[1] = ur;
        __threadfence_system();
        out_host[2] = seq;
      }
    }
  }
}

// The single-GPU combine for MANY records per time step (256 < records <= 1024: the fused kernel's 16-rollout records at K = 4097 ...
// 8192, round 5): a workgroup of FOUR waves per time step, at most four records a lane.  mppi_combine gives such a step ONE wave holding
// eight records of seven doubles per lane (198 registers, 56 loads in a chain of latency): 5.7-6.1 us at K = 8192, a quarter of that
// tick; here the step's loads are spread over four times the lanes and the two reductions (min, then the six sums) meet in LDS.
// Same algebra, same epilogue (mppi.cpp:117-137); the sums associate differently (per-wave trees, then four partials in wave order):
// the controls agree with the one-wave form to rounding (asserted at 1e-9 against the oracle like every kernel).
constexpr int kWideWaves = 4;
__global__ __launch_bounds__(kWave * kWideWaves) void mppi_combine_wide(int T, int R, Lam lam, double umax, USrc u, const double* __restrict__ records,
                                                                       double* __restrict__ u_out, double* __restrict__ out, double* __restrict__ out_host, double seq) {
  __shared__ double s_min[kWideWaves], s_sum[kWideWaves][6];
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
  constexpr int kPer = 4, nthr = kWave * kWideWaves;
  const double u_l = u.get(0, i, T), u_r = u.get(1, i, T);   // (do not depend on the records: requested first)
  double rk[kPer][7];
#pragma unroll
  for (int q = 0; q < kPer; ++q) {
    const int r = tid + q * nthr;
    const double* rec = records + ((size_t)i * R + (r < R ? r : 0)) * TBNAV_MPPI_REC;
#pragma unroll
    for (int f = 0; f < 7; ++f) rk[q][f] = r < R ? rec[f] : 0.0;   // n == 0 marks "no record"
  }
  double M = __builtin_huge_val();
#pragma unroll
  for (int q = 0; q < kPer; ++q) if (rk[q][6] > 0.0) M = fmin(M, rk[q][0]);
  M = tbnav::wave_min_dpp(M);
  if (lane == 0) s_min[wid] = M;
  __syncthreads();
  M = fmin(fmin(s_min[0], s_min[1]), fmin(s_min[2], s_min[3]));
  double acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < kPer; ++q)
    if (rk[q][6] > 0.0) {
      const double sc = exp(div_lambda((rk[q][0] - M) * -1.0, lam));
      acc[0] += sc * rk[q][1]; acc[1] += sc * rk[q][2]; acc[2] += sc * rk[q][3];
      acc[3] += rk[q][4]; acc[4] += rk[q][5]; acc[5] += rk[q][6];
    }
#pragma unroll
  for (int f = 0; f < 6; ++f) { acc[f] = tbnav::wave_sum_dpp(acc[f]); if (lane == 0) s_sum[wid][f] = acc[f]; }
  __syncthreads();
  if (tid == 0) {
    double W = ((s_sum[0][0] + s_sum[1][0]) + s_sum[2][0]) + s_sum[3][0], NL = ((s_sum[0][1] + s_sum[1][1]) + s_sum[2][1]) + s_sum[3][1];
    const double NR = ((s_sum[0][2] + s_sum[1][2]) + s_sum[2][2]) + s_sum[3][2], SD = ((s_sum[0][3] + s_sum[1][3]) + s_sum[2][3]) + s_sum[3][3];
    const double SE = ((s_sum[0][4] + s_sum[1][4]) + s_sum[2][4]) + s_sum[3][4], SN = ((s_sum[0][5] + s_sum[1][5]) + s_sum[2][5]) + s_sum[3][5];
    W += 1e-8 * SN;  // the reference adds 1e-8 to every weight before normalising (mppi.cpp:117)
    double ul = u_l + (NL + 1e-8 * SD) / W;
    double ur = u_r + (NR + 1e-8 * SE) / W;
    ul = (ul < -umax) ? -umax : ((umax < ul) ? umax : ul);   // std::clamp (mppi.cpp:124-125): a NaN stays a NaN
    ur = (ur < -umax) ? -umax : ((umax < ur) ? umax : ur);
    u_out[i] = ul;
    u_out[T + i] = ur;
    if (i == 0) {
      out[0] = ul; out[1] = ur;
      if (out_host) {   // (synchronous ticks: see mppi_combine)
        out_host[0] = ul; out_host[1] = ur;
        __threadfence_system();
        out_host[2] = seq;
      }
    }
  }
}

__global__ void mppi_debug_div_lambda(int n, const double* __restrict__ x, Lam lam, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = div_lambda(x[i], lam);
}
// raw[(k*T + i)*2 + c]  ->  duL[i*K + k], duR[i*K + k]
__global__ void mppi_debug_sincos(int n, const double* __restrict__ x, double* __restrict__ sn, double* __restrict__ cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { double a, b; fast_sincos(x[i], a, b); sn[i] = a; cs[i] = b; }
}

// raw[(k*T + i)*2 + c]  ->  duL[i*K + k], duR[i*K + k]
__global__ void mppi_unpack_noise(int T, int K, const double* __restrict__ raw,
                                  double* __restrict__ duL, double* __restrict__ duR) {
  const size_t n = (size_t)T * K;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx / T), i = (int)(idx % T);
    const double2 v = reinterpret_cast<const double2*>(raw)[idx];
    duL[(size_t)i * K + k] = v.x;
    duR[(size_t)i * K + k] = v.y;
  }
}

// ---- Philox4x32-10 (Salmon et al., SC'11) ------------------------------------------------------
// last node of a captured chunk of ticks: the next replay's first tick
__global__ void mppi_tick_advance(uint64_t* __restrict__ tick0, uint64_t n) { *tick0 += n; }
// the first tick of a replay that does not continue the previous one (the value rides in the launch arguments: no host buffer
// has to outlive the call)
__global__ void mppi_tick_set(uint64_t* __restrict__ tick0, uint64_t v) { *tick0 = v; }

// ---- direct exchange of the sharded tick's records between the ranks of ONE node (multi-process communicators) ---------------
// An RCCL all-gather of a few KB costs tens of microseconds per call on eight GPUs — several 9 us ticks.  Here every rank
// stores its records straight into every peer's gather buffer (mapped through hipIpcMemHandle, fine-grained memory, xGMI) as
// self-validating 8-byte words — (sequence number << 32) | 32 bits of payload, two words a double: a naturally aligned
// 8-byte store is atomic, so a word is either the old tick's or the new one's and no flag, fence or ordering between stores
// is needed — and the receiver polls its OWN buffer's words until they carry the tick's number (system-scope loads; bounded:
// a peer that never delivers raises an error word instead of hanging the device).  Two buffers take turns by the tick's
// parity: a rank can be at most one tick ahead of a peer still reading (it needs that peer's records to get further).
__global__ __launch_bounds__(256) void mppi_direct_publish(const double* __restrict__ mine, int n, unsigned long long* const* __restrict__ peers,
                                                            int me, int P, int parity, unsigned int seq, int only_self) {
  if (only_self && (int)blockIdx.y != me) return;  // (fault injection for the tests of the bound: the peers never see this tick's records)
  unsigned long long* dst = peers[blockIdx.y] + (size_t)(parity * P + me) * 2 * n;
  const unsigned long long tag = (unsigned long long)seq << 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(mine[i]);
    __hip_atomic_store(dst + 2 * i, tag | (b & 0xFFFFFFFFull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 2 * i + 1, tag | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(256) void mppi_direct_collect(unsigned long long* __restrict__ words, int n, int P, int parity, unsigned int seq,
                                                            double* __restrict__ out, int* __restrict__ err, unsigned long long budget_ticks) {
  const size_t total = (size_t)P * n;
  unsigned long long* w0 = words + (size_t)parity * P * 2 * n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long lo = 0ull, hi = 0ull;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      lo = __hip_atomic_load(w0 + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      hi = __hip_atomic_load(w0 + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if ((unsigned int)(lo >> 32) == seq && (unsigned int)(hi >> 32) == seq) break;
      if (wall_clock64() - t0 > budget_ticks) {  // (100 MHz ticks) the records never came: report, deliver zeros
        __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lo = hi = 0ull;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    out[i] = __longlong_as_double((long long)((hi << 32) | (lo & 0xFFFFFFFFull)));
  }
}

__global__ void mppi_sample_noise(int T, int K, uint64_t seed, uint64_t base, double sig_l, double sig_r, int wide,
                                  double* __restrict__ duL, double* __restrict__ duR) {
  const size_t n = (size_t)T * K;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / K), k = (int)(idx % K);  // k fastest: coalesced stores
    const RngArgs g{seed, base, sig_l, sig_r};
    if (wide) device_noise<true>(g, T, i, k, duL[idx], duR[idx]);   // (launch-uniform)
    else device_noise<false>(g, T, i, k, duL[idx], duR[idx]);
  }
}

#define TBNAV_INST_COMBINE(KEEP, MODE) template __global__ void mppi_combine<KEEP, MODE>(int, int, int, Lam, double, USrc, const double* __restrict__, double* __restrict__, double* __restrict__, double* __restrict__, double, DirectSrc);
TBNAV_INST_COMBINE(2, 0) TBNAV_INST_COMBINE(4, 0) TBNAV_INST_COMBINE(8, 0)
TBNAV_INST_COMBINE(2, 1) TBNAV_INST_COMBINE(4, 1) TBNAV_INST_COMBINE(8, 1)
TBNAV_INST_COMBINE(2, 2) TBNAV_INST_COMBINE(4, 2) TBNAV_INST_COMBINE(8, 2)
#undef TBNAV_INST_COMBINE

}  // namespace tbnav_mk
