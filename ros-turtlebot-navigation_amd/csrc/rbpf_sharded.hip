// rbpf_sharded.hip — the sharded filter inside the library (include/tbnav_comm.h): sharded_scan (local scans,
// ONE all-gather of the weights, the global normalise / selection beside the map update, tile-blob migration when resampling fires,
// status agreements), tbnav_rbpf_attach_comm and the one-process group tbnav_rbpf_group_*.
#include "rbpf_host.hpp"

namespace tbnav_rh {

int ensure_shard_state(tbnav_rbpf* h) {
  const int P = tbnav::comm_size(h->comm);
  const size_t ng = (size_t)P * h->N;
  if (!h->stream2) TBNAV_HIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  if (!h->ev_w) TBNAV_HIP(hipEventCreateWithFlags(&h->ev_w, hipEventDisableTiming));
  if (!h->ev_g) TBNAV_HIP(hipEventCreateWithFlags(&h->ev_g, hipEventDisableTiming));
  if (ng > h->g_cap || !h->d_gw_raw) {
    TBNAV_HIP(hipStreamSynchronize(h->stream));
    TBNAV_HIP(hipStreamSynchronize(h->stream2));
    (void)hipFree(h->d_gw); (void)hipFree(h->d_gcs); (void)hipFree(h->d_gparent); (void)hipFree(h->d_gw_raw); (void)hipFree(h->d_sizes); (void)hipFree(h->d_status);
    h->d_gw = h->d_gcs = h->d_gw_raw = nullptr; h->d_gparent = nullptr; h->d_sizes = nullptr; h->d_status = nullptr; h->g_cap = 0;
    TBNAV_HIP(hipMalloc((void**)&h->d_gw, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gcs, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gw_raw, sizeof(double) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_gparent, sizeof(int) * ng));
    TBNAV_HIP(hipMalloc((void**)&h->d_sizes, sizeof(unsigned long long) * ((size_t)h->N + ng)));
    TBNAV_HIP(hipMalloc((void**)&h->d_status, sizeof(int) * (1 + (size_t)P)));
    h->g_cap = ng;
  }
  return TBNAV_OK;
}

int grow(char*& buf, size_t& cap, size_t need) {
  if (need <= cap) return TBNAV_OK;
  (void)hipFree(buf); buf = nullptr; cap = 0;
  const size_t want = need + need / 4 + 4096;
  TBNAV_HIP(hipMalloc((void**)&buf, want));
  cap = want;
  return TBNAV_OK;
}

// ParticleFilter::SLAM over the members' shards.  n == 1: this process's rank of a multi-process filter; n > 1: every member of a
// one-process group, in rank order.  Per scan and member, on the device:
//   main stream   noise -> propose -> [event: weights final] -> map update ........................ -> (resample: migration)
//   second stream                      wait -> ONE all-gather of the raw weights -> the reference's sequential normalise /
//                                      Neff / selection over the GLOBAL vector (identical on every rank) -> own slice back
// so the chain of adds of the global normalise (which grows with the ensemble, not with the shard) runs BESIDE the local map
// update, and the host waits once, for both streams.  Only when resampling fires do particles move: one all-gather of blob
// sizes, one batched export per rank, one message per (source, destination) pair, one batched import (tbnav_rbpf_export_batch_*
// / _import_batch_dev), and an all-gather of the ranks' statuses so that a rank whose pool is exhausted stops everybody.
int sharded_scan(int n, tbnav_rbpf* const* hs, const float* scan, int n_beams, const double u[3], const double cur_odom[3],
                 const double prev_odom[3], int icp_ok, const double T_icp[3], const double* const* normals, tbnav_rbpf_stats* out,
                 tbnav_rbpf_stats* local_out) {
  if (n <= 0 || !hs || !scan || n_beams <= 0 || !u || !cur_odom || !prev_odom || !T_icp || !out) return TBNAV_ERR_INVALID_ARG;
  for (int r = 0; r < n; ++r) if (!hs[r] || !hs[r]->comm || hs[r]->N != hs[0]->N || hs[r]->ref_field) return TBNAV_ERR_INVALID_ARG;
  // (made at attach and never resized while attached — tbnav_rbpf_attach_comm; a handle without them was never attached)
  for (int r = 0; r < n; ++r) if (!hs[r]->d_gw_raw || !hs[r]->d_status || !hs[r]->stream2 || (size_t)tbnav::comm_size(hs[r]->comm) * hs[r]->N > hs[r]->g_cap) return TBNAV_ERR_INVALID_ARG;
  const int P = tbnav::comm_size(hs[0]->comm), nl = hs[0]->N;
  const size_t ng = (size_t)P * nl;
  if (ng > ((size_t)1 << 24)) return TBNAV_ERR_UNSUPPORTED;
  // (everything above is a function of arguments every rank shares: all ranks return together.  From here on a failure that
  //  only THIS rank sees — a launch that fails, an allocation, a pool that runs dry — must not make it leave while its peers
  //  wait in a collective that has no timeout: the rank notes the code in lerr[], skips its own work, KEEPS JOINING the
  //  collectives, and the ranks agree on a status before anyone acts on data that may be missing.  Round-3 advisor finding.)
  std::vector<tbnav_comm*> comms(n);
  std::vector<hipStream_t> s1(n), s2(n);
  std::vector<ScanTicket> tk(n);
  std::vector<tbnav_rbpf_stats> lst(n);
  std::vector<int> lerr(n, TBNAV_OK);
  auto note = [&](int r, int rc) { if (rc != TBNAV_OK && lerr[r] == TBNAV_OK) lerr[r] = rc; };
  auto hipok = [&](int r, hipError_t e, const char* what, int line) { if (e != hipSuccess) note(r, tbnav::hip_fail(e, what, __FILE__, line)); return e == hipSuccess; };
#define TBNAV_L(r, call) hipok(r, (call), #call, __LINE__)
  // the ranks' codes -> one status, the same on every rank (the lowest rank's failure); collective when ranks live elsewhere
  auto agree = [&](const std::vector<int>& codes, int& status) -> int {
    status = TBNAV_OK;
    if (n == P) { for (int r = 0; r < n; ++r) if (codes[r] != TBNAV_OK && status == TBNAV_OK) status = codes[r]; return TBNAV_OK; }
    // (the same rule inside the agreement itself: a copy that fails on this rank is a code this rank contributes — if its word
    //  cannot even be uploaded, the word it holds is whatever the last agreement left, and the rank still reports its own
    //  failure below — never a return before the all-gather its peers are entering)
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    int local_fail = TBNAV_OK;
    for (int r = 0; r < n; ++r) {
      DeviceGuard guard(hs[r]->device);
      const hipError_t e = hipMemcpyAsync(hs[r]->d_status, &codes[r], sizeof(int), hipMemcpyHostToDevice, hs[r]->stream);
      if (e != hipSuccess && local_fail == TBNAV_OK) local_fail = tbnav::hip_fail(e, "agree: status upload", __FILE__, __LINE__);
      send[r] = hs[r]->d_status; recv[r] = hs[r]->d_status + 1;
    }
    { const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(int), s1.data()); if (rc != TBNAV_OK) return rc; }
    std::vector<int> all(P, TBNAV_OK);
    { DeviceGuard guard(hs[0]->device);
      hipError_t e = hipMemcpyAsync(all.data(), hs[0]->d_status + 1, sizeof(int) * P, hipMemcpyDeviceToHost, hs[0]->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(hs[0]->stream);
      if (e != hipSuccess && local_fail == TBNAV_OK) local_fail = tbnav::hip_fail(e, "agree: status download", __FILE__, __LINE__); }
    for (int q = 0; q < P; ++q) if (all[q] != TBNAV_OK) { status = all[q]; break; }
    // A failure of the agreement's own copies on THIS rank is not this scan's status (round-5 advisor finding: the rank returned
    // while its peers, who agreed on OK, went on into the scan's next collectives and waited for it).  It is latched: the rank goes
    // on with the agreed status, keeps joining this scan's collectives, and contributes the failure to the NEXT scan's first
    // agreement, where every rank stops with it.
    if (local_fail != TBNAV_OK) for (int r = 0; r < n; ++r) if (hs[r]->shard_latched == TBNAV_OK) hs[r]->shard_latched = local_fail;
    return TBNAV_OK;
  };
  std::memset(out, 0, sizeof *out);
  // ---- A: every member's local scan (no normalise tail), its "weights are final" event recorded behind the proposal kernel
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    comms[r] = h->comm; s1[r] = h->stream; s2[r] = h->stream2;
    note(r, h->shard_latched);
    if (lerr[r] == TBNAV_OK)
      note(r, scan_enqueue(h, scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals ? normals[r] : nullptr, &lst[r], true, 0, nullptr, tk[r], nullptr, h->ev_w));
    if (lerr[r] == TBNAV_OK) TBNAV_L(r, hipStreamWaitEvent(h->stream2, h->ev_w, 0));
  }
  // ---- B: the ONE collective of the update + the global normalise / selection, on the second streams
  //         (a member that failed above still takes part — with whatever its weight buffer holds: nobody will use the result)
  {
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    for (int r = 0; r < n; ++r) { send[r] = state_ptrs(hs[r]->d_state[hs[r]->cur], nl).weight; recv[r] = hs[r]->d_gw_raw; }
    const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(double) * nl, s2.data());
    if (rc != TBNAV_OK) return rc;   // (the communicator itself failed: it reports on every rank)
  }
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    if (lerr[r] != TBNAV_OK) continue;
    DeviceGuard guard(h->device);
    h->h_norm[1] = NormOut{};
    // the resampling offset: the scan's last normal — with tbnav_rbpf_set_rng_shard (device noise) the ENSEMBLE's, identical on every rank
    const double* zp = h->last_z_ptr;
    hipLaunchKernelGGL(rbpf_normalize, dim3(1), dim3(256), 0, h->stream2, (int)ng, zp, h->d_gw_raw, h->d_gw, h->d_gcs, h->d_gparent, h->d_norm + 1,
                       nullptr, nullptr, nullptr, 0u);
    (void)TBNAV_L(r, hipGetLastError());
    // (the normalised weights go back into the shard only once the ranks have AGREED that this scan succeeded everywhere — below:
    //  a failed rank's slice of the gathered vector is whatever its buffer held)
  }
  // ---- C: the host waits once per member (the reference's SLAM() is synchronous)
  std::vector<int> lstat(n, TBNAV_OK);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    TBNAV_L(r, hipStreamSynchronize(h->stream2));
    // what the reference reports by throwing (a particle left the world, eta is 0 ...) happens to the rank that holds the particle
    lstat[r] = lerr[r] != TBNAV_OK ? lerr[r] : scan_finish(h, tk[r], &lst[r]);
    if (local_out) local_out[r] = lst[r];
  }
  // Every rank must stop at the SAME scan with the same status — a rank that went on alone would sit in the next scan's
  // all-gather for ever: one all-gather of the ranks' statuses per scan (4 bytes each; ~1 % of a scan) when ranks live elsewhere.
  int status = TBNAV_OK;
  { const int rc = agree(lstat, status); if (rc != TBNAV_OK) return rc; }
  const NormOut no = hs[0]->h_norm[1];
  out->sum_w = no.sum_w; out->sq_sum = no.sq_sum; out->neff = no.neff; out->resampled = no.resampled;
  out->n_valid_beams = lst[0].n_valid_beams;
  out->status = status;
  if (status != TBNAV_OK) return status;
  // the scan stands on every rank: each shard takes its slice of the globally normalised weights (on its main stream — the host
  // has waited for the second one above; whatever the main stream does next sees them)
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const size_t off = (size_t)tbnav::comm_rank(h->comm) * nl;
    TBNAV_L(r, hipMemcpyAsync(state_ptrs(h->d_state[h->cur], nl).weight, h->d_gw + off, sizeof(double) * nl, hipMemcpyDeviceToDevice, h->stream));
  }
  // (a copy that could not even be enqueued: the ranks have already agreed on this scan — the code is latched and stops every rank
  //  at the next scan's agreement, or at this one's if a resampling follows)
  for (int r = 0; r < n; ++r) if (lerr[r] != TBNAV_OK) hs[r]->shard_latched = lerr[r];
  if (!no.resampled) return TBNAV_OK;
  // ---- D: lowVarianceResampling's copies across shards.  Slot m (global) takes particle parents[m].
  std::vector<int> parents(ng);
  { DeviceGuard guard(hs[0]->device); if (!TBNAV_L(0, hipMemcpy(parents.data(), hs[0]->d_gparent, sizeof(int) * ng, hipMemcpyDeviceToHost))) std::fill(parents.begin(), parents.end(), 0); }
  struct Plan { std::vector<std::pair<int, int>> sends, recvs; std::vector<int32_t> send_slots; std::vector<uint64_t> send_sizes, send_offs; std::vector<unsigned long long> sizes_local; };
  std::vector<Plan> plan(n);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const int me = tbnav::comm_rank(h->comm), lo = me * nl;
    Plan& pl = plan[r];
    for (size_t m = 0; m < ng; ++m) {  // (dst, q): every particle of mine some other rank's slot chose — once per destination
      const int q = parents[m], dst = (int)(m / nl);
      if (q / nl == me && dst != me) pl.sends.emplace_back(dst, q);
    }
    std::sort(pl.sends.begin(), pl.sends.end());
    pl.sends.erase(std::unique(pl.sends.begin(), pl.sends.end()), pl.sends.end());
    for (int m = lo; m < lo + nl; ++m) { const int q = parents[m]; if (q / nl != me) pl.recvs.emplace_back(q / nl, q); }
    std::sort(pl.recvs.begin(), pl.recvs.end());
    pl.recvs.erase(std::unique(pl.recvs.begin(), pl.recvs.end()), pl.recvs.end());
    pl.send_slots.resize(pl.sends.size());
    for (size_t i = 0; i < pl.sends.size(); ++i) pl.send_slots[i] = pl.sends[i].second - lo;
    pl.send_sizes.assign(pl.sends.size(), 0);
    if (lerr[r] == TBNAV_OK) note(r, tbnav_rbpf_export_batch_sizes(h, (int32_t)pl.sends.size(), pl.send_slots.data(), pl.send_sizes.data()));
    if (lerr[r] != TBNAV_OK) std::fill(pl.send_sizes.begin(), pl.send_sizes.end(), 0);
    // what a particle of mine weighs, for whoever receives it (a particle sent to several ranks weighs the same for each)
    pl.sizes_local.assign(nl, 0ull);
    for (size_t i = 0; i < pl.sends.size(); ++i) pl.sizes_local[pl.sends[i].second - lo] = pl.send_sizes[i];
    TBNAV_L(r, hipMemcpyAsync(h->d_sizes, pl.sizes_local.data(), sizeof(unsigned long long) * nl, hipMemcpyHostToDevice, h->stream));
  }
  {
    std::vector<const void*> send(n);
    std::vector<void*> recv(n);
    for (int r = 0; r < n; ++r) { send[r] = hs[r]->d_sizes; recv[r] = hs[r]->d_sizes + nl; }
    const int rc = tbnav::comm_all_gather(n, comms.data(), send.data(), recv.data(), sizeof(unsigned long long) * nl, s1.data());
    if (rc != TBNAV_OK) return rc;
  }
  std::vector<std::vector<tbnav::P2P>> p_send(n), p_recv(n);
  std::vector<std::vector<uint64_t>> recv_offs(n);
  std::vector<unsigned long long> sizes_all(ng);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    Plan& pl = plan[r];
    if (!(TBNAV_L(r, hipMemcpyAsync(sizes_all.data(), h->d_sizes + nl, sizeof(unsigned long long) * ng, hipMemcpyDeviceToHost, h->stream)) &&
          TBNAV_L(r, hipStreamSynchronize(h->stream)))) std::fill(sizes_all.begin(), sizes_all.end(), 0ull);
    // everything this rank sends: ONE export, the blobs back to back in (destination, particle) order
    uint64_t total = 0;
    for (uint64_t b : pl.send_sizes) total += b;
    pl.send_offs.assign(pl.sends.size() + 1, 0);
    if (lerr[r] == TBNAV_OK) note(r, grow(h->d_sendbuf, h->send_cap, (size_t)total));
    if (lerr[r] == TBNAV_OK) note(r, tbnav_rbpf_export_batch_dev(h, (int32_t)pl.sends.size(), pl.send_slots.data(), h->d_sendbuf, total, pl.send_offs.data()));
    for (size_t i = 0; i < pl.sends.size();) {  // one message per destination
      size_t j = i;
      while (j < pl.sends.size() && pl.sends[j].first == pl.sends[i].first) ++j;
      p_send[r].push_back(tbnav::P2P{pl.sends[i].first, h->d_sendbuf + pl.send_offs[i], (size_t)(pl.send_offs[j] - pl.send_offs[i])});
      i = j;
    }
    // everything it receives: one buffer, the blobs in (source, particle) order
    recv_offs[r].assign(pl.recvs.size() + 1, 0);
    for (size_t i = 0; i < pl.recvs.size(); ++i) recv_offs[r][i + 1] = recv_offs[r][i] + sizes_all[pl.recvs[i].second];
    if (lerr[r] == TBNAV_OK) note(r, grow(h->d_recvbuf, h->recv_cap, (size_t)recv_offs[r].back()));
    for (size_t i = 0; i < pl.recvs.size();) {
      size_t j = i;
      while (j < pl.recvs.size() && pl.recvs[j].first == pl.recvs[i].first) ++j;
      p_recv[r].push_back(tbnav::P2P{pl.recvs[i].first, h->d_recvbuf + recv_offs[r][i], (size_t)(recv_offs[r][j] - recv_offs[r][i])});
      i = j;
    }
  }
  // is every rank ready to send what the sizes promised and to receive it?  A rank whose export or allocation failed cannot
  // honour its messages (its peers would wait for bytes that never come): agree BEFORE the exchange; nobody has touched a slot yet
  { const int rc = agree(lerr, status); if (rc != TBNAV_OK) return rc; }
  if (status != TBNAV_OK) { out->status = status; return status; }
  { const int rc = tbnav::comm_exchange(n, comms.data(), p_send.data(), p_recv.data(), s1.data()); if (rc != TBNAV_OK) return rc; }
  // local parents inside the handle (tile tables + reference counts), then the imported ones; weights are NOT reset by the
  // reference: every slot carries its parent's normalised weight
  std::vector<int> mstat(n, TBNAV_OK);
  for (int r = 0; r < n; ++r) {
    tbnav_rbpf* h = hs[r];
    DeviceGuard guard(h->device);
    const int me = tbnav::comm_rank(h->comm), lo = me * nl;
    Plan& pl = plan[r];
    std::vector<int32_t> local_parent(nl), imp_slots;
    std::vector<uint64_t> imp_offs;
    for (int m = 0; m < nl; ++m) {
      const int q = parents[lo + m];
      if (q / nl == me) local_parent[m] = q - lo;
      else {
        local_parent[m] = -1;
        const auto it = std::lower_bound(pl.recvs.begin(), pl.recvs.end(), std::make_pair(q / nl, q));
        imp_slots.push_back(m);
        imp_offs.push_back(recv_offs[r][(size_t)(it - pl.recvs.begin())]);
      }
    }
    int rc = tbnav_rbpf_gather_local(h, local_parent.data());
    if (rc == TBNAV_OK && !imp_slots.empty())
      rc = tbnav_rbpf_import_batch_dev(h, (int32_t)imp_slots.size(), imp_slots.data(), h->d_recvbuf, recv_offs[r].back(), imp_offs.data());
    if (rc == TBNAV_OK) rc = tbnav_rbpf_set_weights_from_global_dev(h, parents.data() + lo);
    mstat[r] = rc;
  }
  // a rank that failed (tile pool exhausted) must not leave the others waiting in the next scan's collective: agree on it
  { const int rc = agree(mstat, status); if (rc != TBNAV_OK) return rc; }
#undef TBNAV_L
  out->status = status;
  return status;
}

}  // namespace tbnav_rh

extern "C" {

int tbnav_rbpf_attach_comm(tbnav_rbpf* h, tbnav_comm* comm) {
  if (!h) return TBNAV_ERR_INVALID_ARG;
  if (comm && (h->ref_field || tbnav_comm_device(comm) != h->device)) return TBNAV_ERR_INVALID_ARG;
  DeviceGuard guard(h->device);
  TBNAV_HIP(hipStreamSynchronize(h->stream));
  h->comm = comm;
  h->shard_latched = TBNAV_OK;
  if (!comm) { h->rng_first = 0; h->rng_n_global = 0; return TBNAV_OK; }
  // equal shards: this rank's particles are [rank * N, (rank + 1) * N) of nranks * N — also for the device noise source
  h->rng_first = (uint64_t)tbnav::comm_rank(comm) * (uint64_t)h->N;
  h->rng_n_global = (uint64_t)tbnav::comm_size(comm) * (uint64_t)h->N;
  // the buffers every collective of a scan works on exist from here on (nranks * N is fixed for the attachment): a handle
  // whose shard state cannot be made is NOT attached — sharded_scan's precondition, so that no rank finds itself without
  // something to join a collective with in the middle of a scan (round-4 advisor finding)
  const int rc = ensure_shard_state(h);
  if (rc != TBNAV_OK) { h->comm = nullptr; h->rng_first = 0; h->rng_n_global = 0; }
  return rc;
}

void tbnav_rbpf_group_destroy(tbnav_rbpf_group* g) {
  if (!g) return;
  for (int r = 0; r < g->n; ++r) {
    if (r < (int)g->m.size()) tbnav_rbpf_destroy(g->m[r]);
    if (r < (int)g->c.size()) tbnav_comm_destroy(g->c[r]);
  }
  delete g;
}

int tbnav_rbpf_group_create(const tbnav_rbpf_params* params, int32_t n_gpus, const int32_t* devices, uint64_t max_pool_bytes_per_member,
                            tbnav_rbpf_group** out) {
  if (!params || !out || n_gpus <= 0 || params->num_particles <= 0 || params->num_particles % n_gpus != 0) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  tbnav_rbpf_group* g = new (std::nothrow) tbnav_rbpf_group();
  if (!g) return TBNAV_ERR_INVALID_ARG;
  g->n = n_gpus; g->n_global = params->num_particles;
  g->m.assign(n_gpus, nullptr); g->c.assign(n_gpus, nullptr); g->normals.resize(n_gpus);
  int rc = tbnav_comm_create_local(n_gpus, devices, g->c.data());
  for (int r = 0; r < n_gpus && rc == TBNAV_OK; ++r) {
    tbnav_rbpf_params p = *params;
    p.num_particles = params->num_particles / n_gpus;
    p.device = tbnav_comm_device(g->c[r]);
    rc = create_impl(&p, max_pool_bytes_per_member, &g->m[r]);
    if (rc == TBNAV_OK) {
      // initParticleSet gives every particle weight 1 / N of the WHOLE filter (particle_filter.cpp:134)
      std::vector<double> w((size_t)p.num_particles, 1.0 / params->num_particles);
      rc = tbnav_rbpf_set_particles(g->m[r], nullptr, nullptr, w.data());
    }
    if (rc == TBNAV_OK) rc = tbnav_rbpf_attach_comm(g->m[r], g->c[r]);
  }
  if (rc != TBNAV_OK) { tbnav_rbpf_group_destroy(g); return rc; }
  *out = g;
  return TBNAV_OK;
}

int tbnav_rbpf_group_size(const tbnav_rbpf_group* g) { return g ? g->n : -1; }
int tbnav_rbpf_group_member(tbnav_rbpf_group* g, int32_t rank, tbnav_rbpf** out) {
  if (!g || !out || rank < 0 || rank >= g->n) return TBNAV_ERR_INVALID_ARG;
  *out = g->m[rank];
  return TBNAV_OK;
}
int tbnav_rbpf_group_set_seed(tbnav_rbpf_group* g, uint64_t seed) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_rbpf* h : g->m) { const int rc = tbnav_rbpf_set_seed(h, seed); if (rc != TBNAV_OK) return rc; }  // one seed: the members draw disjoint slices of its stream
  return TBNAV_OK;
}
int tbnav_rbpf_group_set_option(tbnav_rbpf_group* g, int32_t option, int32_t value) {
  if (!g) return TBNAV_ERR_INVALID_ARG;
  for (tbnav_rbpf* h : g->m) { const int rc = tbnav_rbpf_set_option(h, option, value); if (rc != TBNAV_OK) return rc; }
  return TBNAV_OK;
}
int64_t tbnav_rbpf_group_num_normals(const tbnav_rbpf_group* g, int32_t icp_ok) {
  if (!g) return -1;
  return (int64_t)g->n_global * (icp_ok ? 3 * g->m[0]->k + 3 : 3) + 1;
}

// normals: the ENSEMBLE's draw stream in the reference's order (tbnav_rbpf_group_num_normals values: particle-major, the
// resampling offset last) or NULL (device noise: every member draws its slice of one stream).
int tbnav_rbpf_group_slam(tbnav_rbpf_group* g, const float* scan, int32_t n_beams, const double u[3], const double cur_odom[3],
                          const double prev_odom[3], int32_t icp_ok, const double T_icp[3], const double* normals, tbnav_rbpf_stats* out) {
  if (!g || !out) return TBNAV_ERR_INVALID_ARG;
  std::vector<const double*> nr(g->n, nullptr);
  if (normals) {
    const size_t stride = icp_ok ? 3 * (size_t)g->m[0]->k + 3 : 3, nl = (size_t)g->m[0]->N;
    for (int r = 0; r < g->n; ++r) {
      std::vector<double>& v = g->normals[r];
      v.resize(nl * stride + 1);
      std::memcpy(v.data(), normals + (size_t)r * nl * stride, sizeof(double) * nl * stride);
      v[nl * stride] = normals[(size_t)g->n_global * stride];
      nr[r] = v.data();
    }
  }
  return sharded_scan(g->n, g->m.data(), scan, n_beams, u, cur_odom, prev_odom, icp_ok, T_icp, normals ? nr.data() : nullptr, out, nullptr);
}

// ParticleFilter::getRobotState over the ensemble: strict >, first wins (particle_filter.cpp:255-274) — members in rank order
int tbnav_rbpf_group_best_state(tbnav_rbpf_group* g, double pose[3], int32_t* best_index) {
  if (!g || !pose) return TBNAV_ERR_INVALID_ARG;
  double best_w = 0.0; int best_r = 0, best_i = 0; double best_pose[3] = {0, 0, 0};
  bool have = false;
  for (int r = 0; r < g->n; ++r) {
    double p[3]; int32_t idx = 0;
    int rc = tbnav_rbpf_best_state(g->m[r], p, &idx);
    if (rc != TBNAV_OK) return rc;
    double w = 0.0;
    { DeviceGuard guard(g->m[r]->device); TBNAV_HIP(hipMemcpy(&w, state_ptrs(g->m[r]->d_state[g->m[r]->cur], g->m[r]->N).weight + idx, sizeof w, hipMemcpyDeviceToHost)); }
    // (a member whose weights are all <= 0.0 reports its slot 0, as the reference's loop would keep index 0)
    if (!have || w > best_w) { best_w = w; best_r = r; best_i = idx; std::memcpy(best_pose, p, sizeof p); have = true; }
  }
  std::memcpy(pose, best_pose, sizeof best_pose);
  if (best_index) *best_index = best_r * g->m[0]->N + best_i;
  return TBNAV_OK;
}
int tbnav_rbpf_group_best_map(tbnav_rbpf_group* g, int8_t* map) {
  if (!g || !map) return TBNAV_ERR_INVALID_ARG;
  double pose[3]; int32_t idx = 0;
  const int rc = tbnav_rbpf_group_best_state(g, pose, &idx);
  if (rc != TBNAV_OK) return rc;
  const int nl = g->m[0]->N;
  return tbnav_rbpf_particle_map(g->m[idx / nl], idx % nl, map);
}

}  // extern "C"
