// rbpf_api.hip — queries and options of the RBPF handle: trace, best state / map (getRobotState / newMap), the per-particle scan
// matcher's switch, the one-particle GridMapper calls (integrateScan, likelihoodFieldModel, gridMap), tbnav_rbpf_set_option and the
// measurement hooks.
#include "rbpf_host.hpp"

namespace tbnav_rh {

int one_particle_consts(tbnav_rbpf* h, int32_t particle, const float* scan, int32_t n_beams, ScanC& c) {
  const double zero[3] = {0.0, 0.0, 0.0};
  std::vector<double2> beams;
  int rc = build_scan_consts(h, c, scan, n_beams, zero, zero, zero, 1, zero, beams);
  if (rc != TBNAV_OK) return rc;
  c.p0 = particle;
  return upload_beams(h, beams, n_beams, c.Bv);
}

}  // namespace tbnav_rh

extern "C" {

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
  if (h->ref_field) { rc = ref_field_materialize(h, particle); if (rc != TBNAV_OK) return rc; }  // (a lookup anywhere: the whole field)
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
        if (h->xsize > tbnav::RefField::kMaxSide || (long)h->radius * h->radius >= 65534) return TBNAV_ERR_UNSUPPORTED;  // (the host queue's nodes: 12-bit coordinates, 16-bit squared distances)
        { const int rc = ensure_codes(h); if (rc != TBNAV_OK) return rc; }
        delete h->ref;
        h->ref = new (std::nothrow) tbnav::RefField(h->N, h->xsize, h->radius);
        if (!h->ref) return TBNAV_ERR_INVALID_ARG;
        h->ref->set_reach(h->ref_reach);
        // (no scan yet: the maps are empty, and the slots were allocated holding "unreached" everywhere — the initial state's image)
        TBNAV_HIP(hipStreamSynchronize(h->stream));
        TBNAV_HIP(hipMemset(h->d_code[h->cur], 0xFF, sizeof(uint16_t) * h->G * (size_t)h->N));
        h->ref->slots_hold_initial_image();
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
    case TBNAV_RBPF_OPT_REF_REACH:   // reference-field mode: cells a scan's brushfire runs out to before it stops (0: to the end, as up to round 5)
      if (value < 0 || value > 65535) return TBNAV_ERR_INVALID_ARG;
      h->ref_reach = value;
      if (h->ref) h->ref->set_reach(value);
      return TBNAV_OK;
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
    case TBNAV_RBPF_OPT_RAYCAST_CELL16:
      if (value < 0 || value > 2) return TBNAV_ERR_INVALID_ARG;
      h->raycast_cell16 = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_ADAPT:
      if (value < 0 || value > 3) return TBNAV_ERR_INVALID_ARG;
      h->raycast_adapt = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_BATCH_PIPELINE:
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->batch_pipeline = value;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_HOST_THREADS:
      if (value < 0 || value > 256) return TBNAV_ERR_INVALID_ARG;
      h->host_threads = value ? value : default_host_threads();
      h->host_threads_auto = value == 0;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_RAYCAST_FORM:
      // retired: 1 named round 2's tile kernel (removed in round 4) and for a while silently meant the much slower beam-ordered
      // kernel instead — that one has its own switch, _RAYCAST_ORDERED (round-4 advisor finding)
      return value == 0 ? TBNAV_OK : TBNAV_ERR_INVALID_ARG;
    case TBNAV_RBPF_OPT_COUNT_CELLS:
      h->count_touched = value != 0;
      return TBNAV_OK;
    case TBNAV_RBPF_OPT_NOISE_IN_KERNEL:
      if (value != 0 && value != 1) return TBNAV_ERR_INVALID_ARG;
      h->noise_in_kernel = value;
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

int tbnav_rbpf_set_timing(tbnav_rbpf* h, int32_t enable) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  h->timing = enable != 0;
  return TBNAV_OK;
}

int tbnav_rbpf_last_kernel_names(const tbnav_rbpf* h, char* propose, int32_t propose_cap, char* raycast, int32_t raycast_cap, int32_t* raycast_workgroups) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (propose && propose_cap > 0) { if (h->lk_propose) snprintf(propose, (size_t)propose_cap, "rbpf_propose<%d, %s>", h->lk_propose, h->lk_propose_dn ? "true" : "false"); else propose[0] = 0; }
  if (raycast && raycast_cap > 0) {
    if (h->lk_raycast > 0) snprintf(raycast, (size_t)raycast_cap, "rbpf_raycast_box<%d, %d, %s, %d>", h->lk_raycast, h->lk_raycast_wps, h->lk_raycast_c16 ? "true" : "false", h->lk_raycast_ev);
    else if (h->lk_raycast == 0) snprintf(raycast, (size_t)raycast_cap, "rbpf_raycast");
    else raycast[0] = 0;
  }
  if (raycast_workgroups) *raycast_workgroups = h->lk_raycast_grid;
  return TBNAV_OK;
}

int tbnav_rbpf_raycast_box_cells(const tbnav_rbpf* h, int32_t* need_cells, int32_t* array_cells) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (need_cells) *need_cells = h->lk_box_need;
  if (array_cells) *array_cells = h->lk_box_cap;
  return TBNAV_OK;
}

int tbnav_rbpf_last_kernel_ms(tbnav_rbpf* h, float ms[TBNAV_RBPF_NKERNELS]) {
  if (!h || !ms) return TBNAV_ERR_INVALID_ARG;
  for (int i = 0; i < TBNAV_RBPF_NKERNELS; ++i) ms[i] = h->last_ms[i];
  return TBNAV_OK;
}

}  // extern "C"
