// rbpf_batch.hip — tbnav_rbpf_slam_batch (a logged run replayed with two scans in the stream) and the shard-level entry points
// the Python-driven exchange (rtn_amd.sharded) calls: slam_local, resample_global*, gather_local, weights.
#include "rbpf_host.hpp"

namespace tbnav_rh {

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

}  // namespace tbnav_rh

extern "C" {

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
  int chunk_first = 0;  // the first scan of the chunk prepared last: scan s of it uses slot s - chunk_first of the rings
  auto prepare = [&](int first) -> int {
    // The call's FIRST chunk is two scans: the host's share of a chunk (the scans' beam tables, ~5 us each) sits in front of the
    // call's first launch, where nothing hides it — a call of 6 scans cost 40 us on top of its scans, one of 3 cost 21.  Later chunks
    // are prepared while two scans are in the stream.  (Two, not one: the pinned staging of a chunk is rewritten when the next is
    // prepared, during the iteration of its last scan — by then the scan before that has been waited for, and with it the launch
    // that read the staging, only if the chunk had two scans at least.)
    const int m = std::min(first == 0 ? 2 : chunk, n_scans - first);
    chunk_first = first;
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
                        T_icp + 3 * s, nullptr, out + s, false, s % kScanSlots, gate_prev, t, ahead ? &pre[s - chunk_first] : nullptr);
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
    if (!h->last_z_ptr) return TBNAV_ERR_INVALID_ARG;
    TBNAV_HIP(hipMemcpyAsync(h->d_gz, h->last_z_ptr, sizeof z, hipMemcpyDeviceToDevice, st));
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

}  // extern "C"
