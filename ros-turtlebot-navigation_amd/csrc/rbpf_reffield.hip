// rbpf_reffield.hip — the reference distance-field mode's device plumbing (TBNAV_RBPF_DF_REFERENCE; host side: ref_field.hpp).
// In: the scan's log of occupied-set changes (the beam-ordered raycast kernel writes it; one packed copy to the host).  Out: the
// journal of field cells each state's pass has written, applied to the particles' field slots (rbpf_field_journal), whole images
// where a slot's content is unknown.  Round the proposal: the kept particle state and the pending-lookup flags (ref_field_settle).
#include "rbpf_host.hpp"

namespace tbnav_rh {

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
// slot p takes the field slot src[p] holds (src[p] == p: keep) — the particles that share a state with one whose whole image was uploaded
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
// The journal of field slot p (ref_field.hpp, plan_flush): optionally "everything pending" first, then `count` (cell, code) pairs —
// every cell at most once per launch.  One workgroup per slot.
__global__ __launch_bounds__(256) void rbpf_field_journal(uint16_t* __restrict__ code, size_t G, const uint3* __restrict__ jobs, const uint2* __restrict__ entries) {
  const int p = blockIdx.x;
  const uint3 j = jobs[p];
  if (!j.y && !j.z) return;
  uint16_t* const slot = code + (size_t)p * G;
  if (j.z) {
    const unsigned int fill = (unsigned int)kCodePending * 0x10001u;
    if ((G & 7) == 0) {
      uint4* d = reinterpret_cast<uint4*>(slot);
      for (size_t i = threadIdx.x; i < G / 8; i += blockDim.x) d[i] = make_uint4(fill, fill, fill, fill);
    } else {
      for (size_t i = threadIdx.x; i < G; i += blockDim.x) slot[i] = kCodePending;
    }
    __threadfence();
    __syncthreads();
  }
  for (unsigned int e = threadIdx.x; e < j.y; e += blockDim.x) {
    const uint2 v = entries[j.x + e];
    slot[v.x] = (uint16_t)v.y;
  }
}
// Bring the device's field slots to where the host's states are: whole images for slots whose content is unknown (or whose state
// has become exact and complete), journal ranges for the others.  Synchronises the stream (the plan's host buffers are read).
int ref_field_flush(tbnav_rbpf* h) {
  UsTimer ut(h->ref_us[3]);
  const int N = h->N;
  hipStream_t st = h->stream;
  tbnav::RefField::Flush& f = h->ref_flush;
  h->ref->plan_flush(f);   // (the plan already counts the slots as brought up to date ...)
  if (f.dense_slot.empty() && !f.any_job) return TBNAV_OK;
  // ... so a copy or launch that fails below leaves them unknown: the next flush sends whole images
  struct Guard { tbnav::RefField* r; bool done = false; ~Guard() { if (!done) r->forget_all_slots(); } } guard{h->ref};
  bool any_copy = false;
  std::vector<int> src;
  for (size_t q = 0; q < f.dense_slot.size(); ++q) {
    if (f.dense_img[q] >= 0)
      TBNAV_HIP(hipMemcpyAsync(h->d_code[h->cur] + (size_t)f.dense_slot[q] * h->G, f.images[f.dense_img[q]].data(), sizeof(uint16_t) * h->G, hipMemcpyHostToDevice, st));
    else {
      if (src.empty()) { src.resize(N); for (int p = 0; p < N; ++p) src[p] = p; }
      src[f.dense_slot[q]] = f.dense_src[q];
      any_copy = true;
    }
  }
  if (any_copy) {
    if (!h->d_code_src) TBNAV_HIP(hipMalloc((void**)&h->d_code_src, sizeof(int) * N));
    TBNAV_HIP(hipMemcpyAsync(h->d_code_src, src.data(), sizeof(int) * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rbpf_copy_codes, dim3(64, N), dim3(256), 0, st, h->d_code[h->cur], h->G, h->d_code_src);
    TBNAV_HIP(hipGetLastError());
  }
  if (f.any_job) {
    static_assert(sizeof(tbnav::RefField::JEntry) == sizeof(uint2) && sizeof(tbnav::RefField::Flush::Job) == sizeof(uint3), "the plan's records are what the kernel reads");
    if (f.entries.size() > h->jentries_cap) {
      TBNAV_HIP(hipStreamSynchronize(st));
      (void)hipFree(h->d_jentries); h->d_jentries = nullptr; h->jentries_cap = 0;
      const size_t cap = f.entries.size() + f.entries.size() / 2 + 4096;
      TBNAV_HIP(hipMalloc((void**)&h->d_jentries, sizeof(uint2) * cap));
      h->jentries_cap = cap;
    }
    if (!h->d_jjobs) TBNAV_HIP(hipMalloc((void**)&h->d_jjobs, sizeof(uint3) * N));
    if (!f.entries.empty()) TBNAV_HIP(hipMemcpyAsync(h->d_jentries, f.entries.data(), sizeof(uint2) * f.entries.size(), hipMemcpyHostToDevice, st));
    TBNAV_HIP(hipMemcpyAsync(h->d_jjobs, f.jobs.data(), sizeof(uint3) * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rbpf_field_journal, dim3(N), dim3(256), 0, st, h->d_code[h->cur], h->G, h->d_jjobs, h->d_jentries);
    TBNAV_HIP(hipGetLastError());
  }
  TBNAV_HIP(hipStreamSynchronize(st));
  guard.done = true;
  return TBNAV_OK;
}
// The whole field of one particle as the reference holds it, on the device (exports, the one-particle entry points): the pass is run
// to the end, stale cells are recovered by replaying the lineage where they are not known (ref_field.hpp).
int ref_field_materialize(tbnav_rbpf* h, int particle) {
  if (!h->ref->codes(particle)) {
    tbnav::last_hip_error_slot() = "reference-field mode: a whole field was asked for whose stale cells need history beyond the history budget";
    return TBNAV_ERR_UNSUPPORTED;
  }
  const int rc = ref_field_flush(h);
  if (rc != TBNAV_OK) return rc;
  const int two = 2;
  TBNAV_HIP(hipMemcpy(h->d_fstate + particle, &two, sizeof two, hipMemcpyHostToDevice));
  h->fstate_dirty = true;
  return TBNAV_OK;
}
// Before the proposal of a scan: the slots in step with the states (imports and exports since the last scan), the particle state
// kept for a second run, the pending flags cleared.
int ref_field_before_propose(tbnav_rbpf* h) {
  UsTimer ut(h->ref_us[4]);
  const int N = h->N;
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  if (!h->d_pend) {
    TBNAV_HIP(hipMalloc((void**)&h->d_pend, sizeof(int) * N));
    TBNAV_HIP(hipHostMalloc((void**)&h->h_pend, sizeof(int) * N, hipHostMallocDefault));
    TBNAV_HIP(hipMalloc((void**)&h->d_state_snap, sizeof(double) * 7 * N));
  }
  if (h->sm_on) {   // the per-particle scan matcher reads whole fields: every pass to the end (the option is not the reference's filter)
    for (int p = 0; p < N; ++p) if (!h->ref->codes(p)) return TBNAV_ERR_UNSUPPORTED;
  }
  { const int rc = ref_field_flush(h); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_fstate), 2, N, h->stream));
  h->fstate_dirty = true;
  TBNAV_HIP(hipMemcpyAsync(h->d_state_snap, h->d_state[h->cur], sizeof(double) * 7 * N, hipMemcpyDeviceToDevice, h->stream));
  TBNAV_HIP(hipMemsetAsync(h->d_pend, 0, sizeof(int) * N, h->stream));
  return TBNAV_OK;
}
// After the proposal: did a lookup land on a cell its particle's pass has not written?  Then exactly those states are resumed on the
// host (RefField::ensure), the new cells go to the device, and the proposal runs again from the kept particle state — until none
// does.  (A closed room never gets here: its beams end within a cell or two of the obstacles the last scans integrated.)
int ref_field_settle(tbnav_rbpf* h, int* h_err, const std::function<int()>& relaunch) {
  const int N = h->N;
  hipStream_t st = h->stream;
  std::vector<int> ps, cs;
  for (int round = 0; round < 4096; ++round) {
    {
      UsTimer ut(h->ref_us[5]);
      TBNAV_HIP(hipMemcpyAsync(h->h_pend, h->d_pend, sizeof(int) * N, hipMemcpyDeviceToHost, st));
      TBNAV_HIP(hipStreamSynchronize(st));
    }
    ps.clear(); cs.clear();
    for (int p = 0; p < N; ++p) if (h->h_pend[p]) { ps.push_back(p); cs.push_back(h->h_pend[p] - 1); }
    if (ps.empty()) return TBNAV_OK;
    const int rc = h->ref->ensure(ps.data(), cs.data(), (int)ps.size(), std::max(h->host_threads, h->ref_threads_used));
    if (rc == -1) {
      tbnav::last_hip_error_slot() = "reference-field mode: a lookup needs a stale cell whose history is beyond the history budget";
      return TBNAV_ERR_UNSUPPORTED;
    }
    if (rc != 0) { tbnav::last_hip_error_slot() = "reference-field mode: a replayed pass differs from the pass it re-ran (internal error)"; return TBNAV_ERR_HIP; }
    { const int rf = ref_field_flush(h); if (rf != TBNAV_OK) return rf; }
    TBNAV_HIP(hipMemcpyAsync(h->d_state[h->cur], h->d_state_snap, sizeof(double) * 7 * N, hipMemcpyDeviceToDevice, st));
    TBNAV_HIP(hipMemsetAsync(h->d_pend, 0, sizeof(int) * N, st));
    for (int q = 0; q < 4; ++q) h_err[q] = 0;   // (mapped; the stream is idle)
    ++h->ref_reruns;
    const int rl = relaunch();
    if (rl != TBNAV_OK) return rl;
  }
  return TBNAV_ERR_UNSUPPORTED;
}
int ref_field_after_scan(tbnav_rbpf* h, bool resampled, int p_first, int p_count) {
  const int N = h->N;
  if (p_count < 0) p_count = N;
  { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
  hipStream_t st = h->stream;
  TBNAV_HIP(hipStreamSynchronize(st));
  auto t_log0 = std::chrono::steady_clock::now();
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
  h->ref_us[0] += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_log0).count();
  h->ref->set_reach(h->ref_reach);
  // how many threads: the quota's own count, or — automatic mode — as many as keep the CPU time per scan period under kRefCpuShare of
  // the quota (rbpf_host.hpp): with W thread-seconds of passes per scan and t_other seconds of everything else, T threads use
  // W / (t_other + W / T) CPUs on average
  int threads = h->host_threads;
  const double now_s = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  if (h->host_threads_auto && h->ref_prev_step_start_s >= 0.0 && h->ref_threads_used > 0 && p_count == N) {
    constexpr double kRefCpuShare = 0.85;
    const double W = (double)h->ref->last_step_busy_us() * 1e-6, period = now_s - h->ref_prev_step_start_s;
    const double t_other = std::max(period - h->ref_last_step_wall_s, 0.0), budget = kRefCpuShare * h->host_quota_cpus;
    const int cap = std::min(std::min(h->host_affinity, 4 * h->host_threads), 128);
    const double denom = W / budget - t_other;
    const double t_max = denom <= 0.0 ? (double)cap : W / denom;
    threads = std::max(h->host_threads, std::min(cap, (int)t_max));
  }
  h->ref_prev_step_start_s = now_s;
  {
    UsTimer ut(h->ref_us[1]);
    const auto t0 = std::chrono::steady_clock::now();
    h->ref->step(p_first, p_count, threads, all.data(), off.data());
    h->ref_last_step_wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    h->ref_threads_used = threads;
  }
  if (resampled) {
    UsTimer ut(h->ref_us[2]);
    h->h_parent.resize(N);
    TBNAV_HIP(hipMemcpy(h->h_parent.data(), h->d_parent, sizeof(int) * N, hipMemcpyDeviceToHost));
    h->ref->resample(h->h_parent.data());   // (the device's gather has moved the field slots the same way: resample_on_device)
  }
  // to the device: what each pass wrote, as a journal on top of the parent's image the slot holds
  { const int rc = ref_field_flush(h); if (rc != TBNAV_OK) return rc; }
  TBNAV_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->d_fstate), 2, N, st));
  TBNAV_HIP(hipStreamSynchronize(st));
  h->fstate_dirty = true;
  return TBNAV_OK;
}

}  // namespace tbnav_rh

extern "C" {

int tbnav_rbpf_reference_field_counts(tbnav_rbpf* h, int32_t* distinct_states, int32_t* last_brushfires, int64_t* total_brushfires) {
  if (!h || !h->ref_field || !h->ref) return TBNAV_ERR_INVALID_ARG;
  if (distinct_states) *distinct_states = h->ref->distinct_states();
  if (last_brushfires) *last_brushfires = h->ref->last_step_brushfires();
  if (total_brushfires) *total_brushfires = h->ref->total_brushfires();
  return TBNAV_OK;
}

int tbnav_rbpf_reference_field_stats(tbnav_rbpf* h, int64_t out[18]) {
  if (!h || !out || !h->ref_field || !h->ref) return TBNAV_ERR_INVALID_ARG;
  const tbnav::RefField::Counters& k = h->ref->counters();
  out[0] = k.passes; out[1] = k.pops; out[2] = k.resumes; out[3] = k.completions; out[4] = k.replays; out[5] = k.replay_generations;
  out[6] = h->ref->history_bytes(); out[7] = h->ref_reruns;
  for (int q = 0; q < 6; ++q) out[8 + q] = h->ref_us[q];
  out[14] = k.us_group; out[15] = k.us_bury;
  out[16] = k.us_busy; out[17] = h->ref_threads_used;
  return TBNAV_OK;
}

}  // extern "C"
