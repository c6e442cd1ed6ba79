// rbpf_io.hip — particles in and out: blobs of one or many particles (tiles + state [+ stored field]) for the cross-rank resample,
// tbnav_rbpf_copy_particle, and the parity hooks (particles, log-odds, distance fields, occupied counts).
#include "rbpf_host.hpp"

namespace tbnav_rh {

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

}  // namespace tbnav_rh

extern "C" {

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
    uint64_t free_now = 0;
    { const int rc = tbnav_rh::pool_free_tiles(h, &free_now); if (rc != TBNAV_OK) return rc; }
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
    dst->ref->copy_slot(dst_slot, *src->ref, src_slot);   // (the slot's device content counts as unknown: the next flush uploads the state's image)
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
    h->ref->forget_slot(particle);
  }
  return TBNAV_OK;
}

int tbnav_rbpf_get_dist_code(tbnav_rbpf* h, int32_t particle, uint16_t* out) {
  if (!h || !out || particle < 0 || particle >= h->N) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  if (h->ref_field) { const int rc = ref_field_materialize(h, particle); if (rc != TBNAV_OK) return rc; }  // the pass to its end, stale cells by replay
  else { const int rc = ensure_full_field(h, particle); if (rc != TBNAV_OK) return rc; }  // whole field on demand
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
  if (h->ref_field) { h->ref->set_codes(particle, code.data()); h->ref->forget_slot(particle); }
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

}  // extern "C"
