// rbpf.hip — bmapping::ParticleFilter::SLAM on MI355X (gfx950) behind the C-ABI of include/tbnav_rbpf.h: the launch sequence of
// one scan, create / destroy and tbnav_rbpf_slam.  The handle itself and the other host files (batch pipeline, sharded scan, blobs,
// queries / options, reference-field plumbing): rbpf_host.hpp.  Reference (paths relative to the reference tree):
//   bmapping/src/bmapping/particle_filter.cpp:141-251 (SLAM), :295-322, :383-437, :442-500, :504-599
//   bmapping/src/bmapping/grid_mapper.cpp:69-182 (likelihood field, integrateScan), :549-898
//   bmapping/src/bmapping/sensor_model.cpp:43-112 (laserEndPoints)
// The kernels live in their families' files (all -ffp-contract=off; DESIGN.md section 4 has the reasoning and the numbers):
//   rbpf_propose.hip   rbpf_mix_lut, rbpf_sample_normals, rbpf_likelihood_one, rbpf_field_by_query, rbpf_scanmatch, rbpf_propose
//   rbpf_raycast.hip   rbpf_raycast_box<512 / 1024> (default map update), rbpf_raycast (beam-ordered), rbpf_add_repeated_test
//   rbpf_field.hip     rbpf_densify, rbpf_window, rbpf_edt<C>, rbpf_edt_compact<R> (stored-field modes, on-demand fields)
//   rbpf_resample.hip  rbpf_normalize, rbpf_resample_apply, rbpf_pool_init, dense <-> tiles, rbpf_argmax, rbpf_export_map
//   rbpf_migrate.hip   particle blobs for the sharded filter's cross-rank resample
//   rbpf_device.hpp    what they share: launch-argument structs, map accessors, reductions, kernel declarations
//   rbpf_normalize.hpp the normalise / selection body (a kernel of its own and workgroup 0 of the map update)
#include "rbpf_host.hpp"

namespace tbnav_rh {

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
  // (first: a scan whose beam table never arrived has computed nothing — whatever else is raised would be an artefact of that)
  if (err[3] & 16) { tbnav::last_hip_error_slot() = "rbpf_propose: the scan's beam table never reached the device (the launch's leading workgroup never published it)"; return TBNAV_ERR_HIP; }
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
                      h->d_fstate, h->d_fstate_alt, h->d_code[h->cur], h->d_code[nxt], (h->df_mode != 2 || h->ref_field) ? 1 : 0};
  hipLaunchKernelGGL(rbpf_resample_apply, dim3(blocks + N * chunks), dim3(kResampleThreads), 0, st, N, h->TT, h->d_parent, h->d_parent + N,
                     h->d_table[h->cur], h->d_table[nxt], h->d_shed, h->pool, blocks, chunks, ga);
  TBNAV_HIP(hipGetLastError());
  std::swap(h->d_fstate, h->d_fstate_alt);
  h->cur = nxt;
  return TBNAV_OK;
}

// GridMapper::integrateScan's map update (grid_mapper.cpp:140-178) for particles [c.p0, c.p0 + count) at their poses.
// sens: the sensor transforms the proposal kernel left for exactly these poses (NULL: the raycast derives them).
// nz (optional): the weights' normalise / select step to run with this update — inside the box-counter kernel's launch as
// workgroup 0 (no second stream, no event), behind the other map-update kernels as a launch of its own.
int launch_raycast(tbnav_rbpf* h, const ScanC& c, int count, const double* sens, const NormArgs* nz, int* err,
                   const double2* beams_dev) {
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
  long cap_win = 0, cap4 = 0;
  if (h->tile_cap > 0) {
    // every end point lies within `reach` of the robot's position: at most floor(2 reach / res) + 2 rows or columns (+1 spare);
    // along y the box is padded to whole pairs of cells
    const double reach = c.rmax + std::hypot(h->p.Trs[1], h->p.Trs[2]);
    const long side = (long)std::floor(2.0 * reach / h->p.resolution) + 3;
    cap_win = (side * ((side + 2) & ~1L) + 7) & ~7L;
    // ... but no more than lets TWO workgroups share a CU's 160 KB (the kernel works a larger box through in bands of rows;
    // at least one padded row must fit)
    const long cap_fit = ((78L * 1024 - (long)box_lds_bytes(0, (size_t)bvn) - (long)kBoxStaticLds) / 4) & ~7L;
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
      // FOUR 512-thread workgroups per CU — every one of 1000 particles resident at once instead of 768 and a second, partial
      // round (round 4; measured 24.6 / 30.8 / 37.1 us with 1 / 2 / 3 workgroups per CU, 48.3 for 1000 particles on 768 slots) —
      // when the boxes' need plus a margin of three rows or so fits a quarter of the CU's LDS.  The margin is tighter than the
      // three-per-CU form's 1/8 + 512: a box that outgrows it costs its particle a second band, never correctness.
      cap4 = std::max(((long)need + 256 + 7) & ~7L, (side + 9) & ~7L);
    }
  }
  // Residency the launch gets: R workgroups per CU need R x (dynamic + static LDS) <= 160 KB; 4 and 3 per CU are 512-thread
  // workgroups (64 / 80 registers a lane), 2 per CU 1024 threads (64).  Two cell formats: 32-bit words (slot in the word) and,
  // when that buys a higher residency, 16-bit words + a slot table (C16: a few more instructions per walk step).  Measured at
  // cfg3, N = 1000 / 4000, 32-bit words: 1024 x 2: 55.8 / 191 us; 512 x 2: 54.4 / 199; 512 x 3: 47.9 / 163; 512 x 4 (four
  // events a slot instead of eight: what lets the bench room's 8272-cell boxes in): 41.7 / 143.
  const size_t nz_lds = nz ? sizeof(double) * 2 * kNormChunk : (size_t)0;
  auto fits = [&](int R, size_t bytes) { return (size_t)R * (std::max(bytes, nz_lds) + kBoxStaticLds) <= (size_t)kMaxLds; };
  const bool may4 = h->raycast_adapt != 2 && cap4 > 0 && cap4 <= cap_win;
  int wps = 8;
  bool c16 = false;
  const bool pick = nt_auto || nt == 512;
  const bool c16_ok = h->raycast_cell16 != 0 && cap_win > 0 && cap_win < 65528 && c.Bv + 64 < 32768;
  // (measured, N = 1000: the 16-bit form costs ~15 % at EQUAL residency — 57.2 against 49.8 us at three per CU: twice the
  //  same-dword collisions of the LDS adds where rays converge, the sub-word arithmetic of every step — so going from three to
  //  four per CU with it loses, 51.5 against 49.8 us, and it is used only where the 32-bit form is stuck at TWO per CU: the
  //  SURVEY room's 13 860-cell boxes, 63.7 -> 60.7 us)
  // (four per CU: with eight events a slot where they fit, with four where only they do — measured in a 3 x 1.9 m room whose cells take
  //  5-10 events: 39.4 us with four-event slots against 37.7 with three workgroups per CU and eight)
  size_t ev_slot = kBoxEv;  // (what the instantiation launched below holds per slot)
  const bool force4 = h->raycast_adapt == 3;  // (tests: four-event slots wherever four workgroups fit)
  if (pick && may4 && !force4 && fits(4, box_lds_bytes((size_t)cap4, (size_t)bvn, kBoxEv))) { nt = 512; cap_win = cap4; }
  else if (pick && may4 && fits(4, box_lds_bytes((size_t)cap4, (size_t)bvn, kBoxEvFour))) { nt = 512; cap_win = cap4; ev_slot = kBoxEvFour; }
  else if (pick && cap_win > 0 && fits(3, box_lds_bytes((size_t)cap_win, (size_t)bvn))) { nt = 512; wps = 6; }
  else if (pick && may4 && c16_ok && fits(4, box16_lds_bytes((size_t)cap4, (size_t)bvn, kBoxEv))) { nt = 512; cap_win = cap4; c16 = true; }
  // (no 16-bit form with four-event slots: its niche — boxes that fit four per CU only with BOTH economies — is a few hundred cells wide, and
  //  the instantiation was the one map-update kernel left with a spill; such boxes run three per CU in <512, 6, true, 8>)
  else if (pick && c16_ok && fits(3, box16_lds_bytes((size_t)cap_win, (size_t)bvn))) { nt = 512; wps = 6; c16 = true; }
  else if (nt == 512) wps = 6;
  if (h->raycast_cell16 == 2 && c16_ok && nt == 512) c16 = true;   // (tests / A-B: the 16-bit form wherever it can run)
  if (c16) ev_slot = kBoxEv;   // (the 16-bit form has eight-event slots only)
  const size_t lds_win = c16 ? box16_lds_bytes((size_t)cap_win, (size_t)bvn, ev_slot) : box_lds_bytes((size_t)cap_win, (size_t)bvn, ev_slot);
  if (cap_win > 0 && !h->ref_field && c.Bv < 32768 - kWave && nt >= 512 && lds_win <= (size_t)kMaxLds - 4096) {
    // default: box counters (rbpf_raycast_box)
    unsigned long long* touched = h->count_touched ? h->d_touched : nullptr;
    const NormArgs na = nz ? *nz : NormArgs{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, nullptr};
    const int blocks = count + (nz ? 1 : 0);
    const size_t lds_launch = nz ? std::max(lds_win, sizeof(double) * 2 * kNormChunk) : lds_win;  // (workgroup 0's two arrays)
    const int need_slot = (int)(h->rc_launches++ % 3u);
    h->lk_raycast = nt == 512 ? 512 : 1024; h->lk_raycast_wps = wps; h->lk_raycast_c16 = c16 ? 1 : 0; h->lk_raycast_ev = (int)ev_slot; h->lk_raycast_grid = blocks;
    h->lk_box_cap = (int)cap_win; h->lk_box_need = (h->raycast_adapt && h->h_box_need) ? *reinterpret_cast<volatile int*>(h->h_box_need) : 0;
    const int hash_words = c16 ? (int)box16_hash_words((size_t)bvn) : 0;
#define TBNAV_BOX(NT_, WPS_, C16_, EV_) hipLaunchKernelGGL((rbpf_raycast_box<NT_, WPS_, C16_, EV_>), dim3(blocks), dim3(NT_), lds_launch, st, c, h->pool, M, beams_dev, sp.pose, sens, \
                                                h->d_trow[h->cur], h->d_nocc[h->cur], err, (int)cap_win, touched, na, h->d_box_need, h->d_box_need_host, need_slot, hash_words)
    const bool ev4 = ev_slot == (size_t)kBoxEvFour;
    if (nt == 512 && wps == 8 && c16) TBNAV_BOX(512, 8, true, 8);
    else if (nt == 512 && wps == 8 && ev4) TBNAV_BOX(512, 8, false, 4);
    else if (nt == 512 && wps == 8) TBNAV_BOX(512, 8, false, 8);
    else if (nt == 512 && c16) TBNAV_BOX(512, 6, true, 8);
    else if (nt == 512) TBNAV_BOX(512, 6, false, 8);
    else TBNAV_BOX(1024, 8, false, 8);
#undef TBNAV_BOX
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
int upload_beams(tbnav_rbpf* h, const std::vector<double2>& beams, int n_beams, int Bv, bool stage_only, int slot) {
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

int scan_enqueue(tbnav_rbpf* h, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* normals,
                 tbnav_rbpf_stats* out, bool local_only, int slot, const int* gate_prev, ScanTicket& tk,
                 const Prefetched* pre, hipEvent_t weights_ready) {
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
  // device noise: drawn inside the proposal kernel, whose leading workgroup also carries the beam table over (NoiseSrc) — unless a
  // kernel in front of it needs the table on the device (the per-particle scan matcher), the scan comes prepared with its chunk
  // (tbnav_rbpf_slam_batch), or the option is off: then rbpf_sample_normals stores the same values first, as up to round 4
  // (... or the reference-field mode may have to run the proposal twice: the stored stream is there for the second run)
  bool dn = !pre && !normals && !(h->sm_on && c.icp_ok) && h->noise_in_kernel == 1 && !h->ref_field;
  NoiseSrc ns{};
  if (dn) {
    // (the hand-over needs fine-grained device memory: where the platform does not give any, the option switches itself off for good —
    //  the stored-first form computes the same values)
    hipError_t ea = hipSuccess;
    if (!h->d_zslot) {
      ea = hipMalloc((void**)&h->d_zslot, sizeof(double));
      if (ea == hipSuccess) ea = hipExtMallocWithFlags((void**)&h->d_beam_ready, sizeof(unsigned int) * kReadyCopies * kReadyStride, hipDeviceMallocFinegrained);
      if (ea == hipSuccess) ea = hipMemsetAsync(h->d_beam_ready, 0, sizeof(unsigned int) * kReadyCopies * kReadyStride, st);
      h->beam_seq = 0;
    }
    if (ea == hipSuccess && h->fg_beams_cap < h->max_beams) {
      ea = hipStreamSynchronize(st);
      (void)hipFree(h->d_beams_fg); h->d_beams_fg = nullptr; h->fg_beams_cap = 0;
      if (ea == hipSuccess) ea = hipExtMallocWithFlags((void**)&h->d_beams_fg, sizeof(double2) * h->max_beams, hipDeviceMallocFinegrained);
      if (ea == hipSuccess) h->fg_beams_cap = h->max_beams;
    }
    if (ea != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(h->d_zslot); (void)hipFree(h->d_beam_ready); (void)hipFree(h->d_beams_fg);
      h->d_zslot = nullptr; h->d_beam_ready = nullptr; h->d_beams_fg = nullptr; h->fg_beams_cap = 0;
      h->noise_in_kernel = 0;
      dn = false;
    }
  }
  if (dn) {
    if (++h->beam_seq == 0u) ++h->beam_seq;   // (0 is the cleared word)
    ns.seed = h->seed; ns.scan = h->scan_index;
    ns.base = h->rng_n_global ? (size_t)h->rng_first * c.stride_normals : 0;
    ns.z_index = h->rng_n_global ? (size_t)h->rng_n_global * c.stride_normals : (size_t)h->N * c.stride_normals;
    ns.z_out = h->d_zslot;
    ns.host_beams = (const double2*)(h->h_beams + (size_t)slot * h->max_beams); ns.dev_beams = h->d_beams; ns.fg_beams = h->d_beams_fg;
    ns.ready = h->d_beam_ready; ns.seq = h->beam_seq;
  }
  h->last_drawn.valid = false;
  if (dn) {
    h->last_drawn.seed = ns.seed; h->last_drawn.scan = ns.scan; h->last_drawn.base = ns.base; h->last_drawn.z_index = ns.z_index;
    h->last_drawn.n = n_norm; h->last_drawn.valid = true;
  } else {
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
  }
  ++h->scan_index;
  const double2* const beams_dev = pre ? pre->d_beams : h->d_beams;
  const double* const normals_dev = dn ? nullptr : (pre ? pre->d_normals : h->d_normals);
  h->last_normals = normals_dev;
  h->last_z_index = (size_t)h->N * c.stride_normals;
  h->last_z_ptr = dn ? h->d_zslot : normals_dev + h->last_z_index;
  for (int q = 0; q < 4; ++q) h_err[q] = 0;  // mapped: the scan that last owned the slot has been waited for
  h->h_norm[slot] = NormOut{};
  StatePtrs sp = state_ptrs(h->d_state[h->cur], h->N);

  // ---- distance-field refresh for this call's lookups (windowed), then the particle update
  if (h->ref_field) { rc = ref_field_before_propose(h); if (rc != TBNAV_OK) { out->status = rc; return rc; } }
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
  h->lk_propose_dn = dn ? 1 : 0;
  int* const pend = h->ref_field ? h->d_pend : nullptr;
#define TBNAV_PROPOSE(NT_, DN_) hipLaunchKernelGGL((rbpf_propose<NT_, DN_>), dim3(h->N + (DN_ ? 1 : 0)), dim3(NT_), propose_lds, st, c, beams_dev,                    \
                     h->d_code[h->cur], h->pool, map_of(h), h->d_trow[h->cur], skip_arr, skip_eq, h->df_mode, h->radius, occ_half,                       \
                     h->d_nocc[h->cur], h->d_win, normals_dev, center, sp.pose, sp.prev, sp.weight, h->tr, h->d_sens, d_err, gate_prev, h->d_mixlut, ns, pend)
  auto launch_propose = [&]() -> int {
    if (propose_lds + 3072 > (size_t)kMaxLds / 4) { if (dn) TBNAV_PROPOSE(2 * kProposeThreads, true); else TBNAV_PROPOSE(2 * kProposeThreads, false); }
    else { if (dn) TBNAV_PROPOSE(kProposeThreads, true); else TBNAV_PROPOSE(kProposeThreads, false); }
    TBNAV_HIP(hipGetLastError());
    return TBNAV_OK;
  };
#undef TBNAV_PROPOSE
  rc = launch_propose();
  if (rc != TBNAV_OK) return rc;
  if (h->ref_field) {  // lookups that met cells the lazy brushfire has not written: resume those states, run the proposal again
    rc = ref_field_settle(h, h_err, launch_propose);
    if (rc != TBNAV_OK) { out->status = rc; return rc; }
  }
  if (weights_ready) TBNAV_HIP(hipEventRecord(weights_ready, st));  // (sharded filter: the exchange starts here, beside the map update)
  if (h->timing) TBNAV_HIP(hipEventRecord(h->ev[2], st));
  // normalise / select needs only the weights the proposal kernel left: it rides in the map update's launch as one extra
  // workgroup (the chain of adds it is made of would otherwise sit on the critical path, and a second stream costs an
  // event and a dependent boundary).  With event timing on it is a launch of its own, so that the intervals mean what
  // they say.
  const double* z_norm = h->last_z_ptr;
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

// host threads for the reference-field mode: the cores this process may run on — its affinity mask, and under a cgroup CPU quota
// (cpu.max: a container that sees 128 CPUs but may use 32 of them) no more than that — at most 128; TBNAV_RBPF_OPT_HOST_THREADS
// overrides.  (Round 4 capped this at 32: the bench box has more.)
int host_cpu_budget(double& quota_cpus) {
  int n = 0;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  quota_cpus = (double)n;
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
    long long quota = 0, period = 0;
    if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) quota_cpus = std::min((double)n, (double)quota / (double)period);
    std::fclose(f);
  }
  return n;
}
int default_host_threads() {
  double q = 1.0;
  const int n = host_cpu_budget(q);
  const int t = std::min(n, (int)std::ceil(q));
  return t < 1 ? 1 : (t > 128 ? 128 : t);
}
// free lists: rbpf_device.hpp.  (TBNAV_POOL_SHARD_MIN: a test hook — the GPU suite is run once with every pool of 32 tiles or
//  more on sixteen lists, so that the small parity cases pop from lists a few tiles long, move on and gather all the time)
unsigned int pool_lists_for(size_t cap_tiles) {
  size_t shard_min = kPoolShardMin;
  if (const char* ev = std::getenv("TBNAV_POOL_SHARD_MIN")) { const long v = std::atol(ev); if (v >= kPoolShards) shard_min = (size_t)v; }
  return cap_tiles >= shard_min ? (unsigned int)kPoolShards : 1u;
}
int pool_free_tiles(tbnav_rbpf* h, uint64_t* free_tiles) {
  unsigned long long ctr[kPoolCtrWords];
  TBNAV_HIP(hipMemcpy(ctr, h->pool.ctr, sizeof ctr, hipMemcpyDeviceToHost));
  uint64_t f = 0;
  for (unsigned int s = 0; s < h->pool.shards; ++s) f += ctr[s * kPoolCtrStride + 1] - ctr[s * kPoolCtrStride];
  *free_tiles = f;
  return TBNAV_OK;
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
  h->host_affinity = host_cpu_budget(h->host_quota_cpus);
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
    h->pool.shards = pool_lists_for(cap);
    h->pool.shard_cap = (unsigned int)((cap + h->pool.shards - 1) / h->pool.shards);
    A((void**)&h->pool.lo, sizeof(double) * kTileCells * cap);
    A((void**)&h->pool.bm, sizeof(unsigned int) * kTS * cap);
    A((void**)&h->pool.ref, sizeof(int) * cap);
    A((void**)&h->pool.ring, sizeof(unsigned int) * (size_t)h->pool.shard_cap * h->pool.shards);
    A((void**)&h->pool.ctr, sizeof(unsigned long long) * kPoolCtrWords);
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
    if (e == hipSuccess) {  // the queue's scratch memory, before the first map update that needs it (see the kernel)
      hipLaunchKernelGGL(rbpf_warm_scratch, dim3(1024), dim3(512), 0, h->stream, reinterpret_cast<int*>(h->d_touched), 0);
      e = hipGetLastError();
    }
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
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 6, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 6, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 8, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 8, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<512, 8, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast_box<1024, 8, false, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 4096);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_raycast), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 1024);
  // the proposal / scan-match kernels carry the scan, the per-sample data and the bitmap slice: more than the 64 KB
  // default for long scans or many samples
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<kProposeThreads, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);  // (2.3 KB static)
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<2 * kProposeThreads, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<kProposeThreads, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rbpf_propose<2 * kProposeThreads, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds - 3072);
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

}  // namespace tbnav_rh

extern "C" {

int tbnav_rbpf_create(const tbnav_rbpf_params* P, tbnav_rbpf** out) { return create_impl(P, 0, out); }
int tbnav_rbpf_create_pool(const tbnav_rbpf_params* P, uint64_t max_pool_bytes, tbnav_rbpf** out) { return create_impl(P, max_pool_bytes, out); }

int tbnav_rbpf_pool_stats(tbnav_rbpf* h, uint64_t* capacity_tiles, uint64_t* free_tiles, uint64_t* tile_bytes) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  uint64_t free_now = 0;
  { const int rc = tbnav_rh::pool_free_tiles(h, &free_now); if (rc != TBNAV_OK) return rc; }
  if (capacity_tiles) *capacity_tiles = h->pool.cap - 1;  // tile 0 is the shared zero tile
  if (free_tiles) *free_tiles = free_now;
  if (tile_bytes) *tile_bytes = sizeof(double) * kTileCells;
  return TBNAV_OK;
}

void tbnav_rbpf_destroy(tbnav_rbpf* h) {
#ifdef TBNAV_PHASE_PROF
  rbpf_prof_print_propose();
  rbpf_prof_print_raycast();
#endif
  if (!h) return;
  DeviceGuard guard(h->device);
  for (int b = 0; b < 2; ++b) { (void)hipFree(h->d_state[b]); (void)hipFree(h->d_table[b]); (void)hipFree(h->d_code[b]); (void)hipFree(h->d_nocc[b]); (void)hipFree(h->d_trow[b]); }
  (void)hipFree(h->d_bm_dense); (void)hipFree(h->d_rc_dense);
  (void)hipFree(h->pool.lo); (void)hipFree(h->pool.bm); (void)hipFree(h->pool.ref); (void)hipFree(h->pool.ring); (void)hipFree(h->pool.ctr);
  (void)hipFree(h->d_sens); (void)hipFree(h->d_shed); (void)hipFree(h->d_dense); (void)hipFree(h->d_cs); (void)hipFree(h->d_touched); (void)hipFree(h->d_box_need); if (h->h_box_need) (void)hipHostFree(h->h_box_need); (void)hipFree(h->d_fstate_alt);
  (void)hipFree(h->d_log_ev); (void)hipFree(h->d_log_cnt); (void)hipFree(h->d_tile_scratch); (void)hipFree(h->d_log_pack); (void)hipFree(h->d_log_off); (void)hipFree(h->d_code_src);
  (void)hipFree(h->d_pend); (void)hipHostFree(h->h_pend); (void)hipFree(h->d_state_snap); (void)hipFree(h->d_jentries); (void)hipFree(h->d_jjobs);
  (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); (void)hipFree(h->d_gz);
  (void)hipFree(h->d_gw_raw); (void)hipFree(h->d_sendbuf); (void)hipFree(h->d_recvbuf); (void)hipFree(h->d_sizes); (void)hipFree(h->d_status);
  if (h->ev_w) (void)hipEventDestroy(h->ev_w);
  if (h->ev_g) (void)hipEventDestroy(h->ev_g);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  (void)hipFree(h->d_zslot); (void)hipFree(h->d_beam_ready); (void)hipFree(h->d_beams_fg);
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
  if (!h || !out || n <= 0) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  if (h->last_drawn.valid) {
    // the last scan drew its noise inside the proposal kernel and stored none of it: the same counters through rbpf_sample_normals
    // give the same values (one definition: normal_pair) — what the kernel used, regenerated for whoever asks
    if ((size_t)n > h->last_drawn.n) return TBNAV_ERR_INVALID_ARG;
    const size_t nn = h->last_drawn.n;
    if (nn > h->normals_cap) {
      (void)hipFree(h->d_normals); h->d_normals = nullptr; h->normals_cap = 0;
      TBNAV_HIP(hipMalloc((void**)&h->d_normals, sizeof(double) * nn));
      h->normals_cap = nn;
    }
    const int blocks = (int)std::min<size_t>((nn / 2 + 255) / 256, 4096);
    const bool sharded = h->last_drawn.z_index != nn - 1 || h->last_drawn.base != 0;
    if (sharded)
      hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, h->stream, nn - 1, h->last_drawn.seed, h->last_drawn.scan, h->d_normals,
                         (const double2*)nullptr, (double2*)nullptr, 0, (size_t)0, (size_t)0, h->last_drawn.base, h->last_drawn.z_index, nn - 1);
    else
      hipLaunchKernelGGL(rbpf_sample_normals, dim3(blocks), dim3(256), 0, h->stream, nn, h->last_drawn.seed, h->last_drawn.scan, h->d_normals,
                         (const double2*)nullptr, (double2*)nullptr, 0, (size_t)0, (size_t)0, (size_t)0, ~(size_t)0, (size_t)0);
    TBNAV_HIP(hipGetLastError());
    TBNAV_HIP(hipStreamSynchronize(h->stream));
    TBNAV_HIP(hipMemcpy(out, h->d_normals, sizeof(double) * n, hipMemcpyDeviceToHost));
    return TBNAV_OK;
  }
  if ((size_t)n > std::max(h->normals_cap, h->norm_ring_stride)) return TBNAV_ERR_INVALID_ARG;
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

}  // extern "C"
