// rbpf_migrate.hip — a particle as ONE device buffer (state, its tiles' log-odds and occupancy bits, tile-row counts, the
// stored field where it is authoritative), singly and in batches: what the sharded filter's cross-rank resample moves
// (SURVEY.md section 8-e).
#include "rbpf_device.hpp"

namespace tbnav_rk {

// ---- particle migration between handles (sharded filter): a particle travels as its state, its
//      per-tile-row counts and ONLY the tiles (log-odds + occupancy bits) it does not share with the zero tile ------------------------------------
__global__ __launch_bounds__(256) void rbpf_pack_tiles(TilePool P, const unsigned int* __restrict__ ids, double* __restrict__ out,
                                                       unsigned int* __restrict__ out_bm) {
  const double2* src = reinterpret_cast<const double2*>(P.lo + (size_t)ids[blockIdx.x] * kTileCells);
  double2* dst = reinterpret_cast<double2*>(out + (size_t)blockIdx.x * kTileCells);
  for (int i = threadIdx.x; i < kTileCells / 2; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x < kTS) out_bm[(size_t)blockIdx.x * kTS + threadIdx.x] = P.bm[(size_t)ids[blockIdx.x] * kTS + threadIdx.x];
}
// one workgroup per received tile: take a free tile, name it in the (released) slot's table, fill it
__global__ __launch_bounds__(256) void rbpf_unpack_tiles(TilePool P, MapT M, int p, const unsigned int* __restrict__ tidx,
                                                         const double* __restrict__ in, const unsigned int* __restrict__ in_bm,
                                                         int* __restrict__ err) {
  __shared__ unsigned int sid;
  if (threadIdx.x == 0) {
    const unsigned int id = tile_pop(P, blockIdx.x);
    if (id) { P.ref[id] = 1; M.table[(size_t)p * M.TT + tidx[blockIdx.x]] = id; } else atomicOr(&err[3], 8);
    sid = id;
  }
  __syncthreads();
  if (sid == 0u) return;
  const double2* src = reinterpret_cast<const double2*>(in + (size_t)blockIdx.x * kTileCells);
  double2* dst = reinterpret_cast<double2*>(P.lo + (size_t)sid * kTileCells);
  for (int i = threadIdx.x; i < kTileCells / 2; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x < kTS) P.bm[(size_t)sid * kTS + threadIdx.x] = in_bm[(size_t)blockIdx.x * kTS + threadIdx.x];
}
// tiles named by each listed slot's table, and the slot's field state
__global__ __launch_bounds__(256) void rbpf_count_tiles(MapT M, const int* __restrict__ slots, const int* __restrict__ fstate, int2* __restrict__ out) {
  __shared__ int tot;
  if (threadIdx.x == 0) tot = 0;
  __syncthreads();
  const int slot = slots[blockIdx.x];
  const unsigned int* tab = M.table + (size_t)slot * M.TT;
  int c = 0;
  for (int t = threadIdx.x; t < M.TT; t += blockDim.x) c += tab[t] != 0u ? 1 : 0;
  if (c) atomicAdd(&tot, c);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = int2{tot, fstate[slot]};
}
// one workgroup per exported particle: header, state, tile indices (ascending), tile payloads, tile-row counts, stored field
__global__ __launch_bounds__(256) void rbpf_pack_batch(TilePool P, MapT M, const double* __restrict__ pose, const double* __restrict__ prev,
                                                       const double* __restrict__ weight, const int* __restrict__ trow, const int* __restrict__ nocc,
                                                       const int* __restrict__ fstate, const uint16_t* __restrict__ codes, size_t G, int xsize,
                                                       const BatchItem* __restrict__ items, char* __restrict__ buf) {
  const BatchItem it = items[blockIdx.x];
  const int slot = it.slot, tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
  const BlobLayout L = blob_layout_hd(M.TW, G, it.n_tiles, it.has_codes != 0);
  char* b = buf + it.off;
  const unsigned int* tab = M.table + (size_t)slot * M.TT;
  unsigned int* tidx = reinterpret_cast<unsigned int*>(b + L.tidx);
  if (tid == 0) {
    *reinterpret_cast<BlobHeader*>(b) = BlobHeader{kBlobMagic, it.n_tiles, it.has_codes ? 1u : 0u, nocc[slot], fstate[slot], (uint32_t)xsize, (uint32_t)M.TT};
    double* bs = reinterpret_cast<double*>(b + L.state);
    for (int q = 0; q < 3; ++q) { bs[q] = pose[(size_t)slot * 3 + q]; bs[3 + q] = prev[(size_t)slot * 3 + q]; }
    bs[6] = weight[slot];
  }
  __shared__ int base, wcnt[4];
  if (tid == 0) base = 0;
  __syncthreads();
  for (int t0 = 0; t0 < M.TT; t0 += 256) {
    const int t = t0 + tid;
    const bool f = t < M.TT && tab[t] != 0u;
    const unsigned long long m = __ballot(f);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wv; ++w) off += wcnt[w];
    if (f) tidx[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)t;
    __syncthreads();
    if (tid == 0) base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  __threadfence_block();  // tidx was written by this workgroup: visible to all of it after the fence + the barrier above
  __syncthreads();
  double2* dst = reinterpret_cast<double2*>(b + L.tiles);
  for (size_t i = tid; i < (size_t)it.n_tiles * (kTileCells / 2); i += 256) {
    const unsigned int id = tab[tidx[i / (kTileCells / 2)]];
    dst[i] = reinterpret_cast<const double2*>(P.lo + (size_t)id * kTileCells)[i % (kTileCells / 2)];
  }
  unsigned int* dbm = reinterpret_cast<unsigned int*>(b + L.tile_bm);
  for (size_t i = tid; i < (size_t)it.n_tiles * kTS; i += 256) dbm[i] = P.bm[(size_t)tab[tidx[i / kTS]] * kTS + (i % kTS)];
  int* dtr = reinterpret_cast<int*>(b + L.trow);
  for (int r = tid; r < M.TW; r += 256) dtr[r] = trow[(size_t)slot * M.TW + r];
  if (it.has_codes) {
    uint16_t* dc = reinterpret_cast<uint16_t*>(b + L.codes);
    const uint16_t* sc = codes + (size_t)slot * G;
    for (size_t i = tid; i < G; i += 256) dc[i] = sc[i];
  }
}
__global__ __launch_bounds__(256) void rbpf_blob_headers(const BatchItem* __restrict__ items, const char* __restrict__ buf, BlobHeader* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = *reinterpret_cast<const BlobHeader*>(buf + items[i].off);
}
__global__ __launch_bounds__(256) void rbpf_release_slots(TilePool P, MapT M, const BatchItem* __restrict__ items) {
  const int p = items[blockIdx.x].slot;
  unsigned int* tab = M.table + (size_t)p * M.TT;
  unsigned int* shed = M.shed + (size_t)p * M.TT;
  for (int t = threadIdx.x; t < M.TT; t += blockDim.x) {
    const unsigned int id = tab[t], sh = shed[t];
    if (id && atomicSub(&P.ref[id], 1) == 1) tile_push(P, id);
    if (sh && atomicSub(&P.ref[sh], 1) == 1) tile_push(P, sh);
    tab[t] = 0u; shed[t] = 0u;
  }
}
// one workgroup per imported particle (its slot was released by the launch before).  Its tiles are popped in runs of 64 (more for
// maps of more than 16 384 tiles: at most 256 runs), one thread a run, each from the free list its number names first
// (rbpf_device.hpp: a list supplies a run or passes).  A run no list can supply is popped tile by tile from whichever list has
// any, the ids going straight into the slot's table.  If the pool runs out the particle is not installed: the tiles taken so far
// are noted as SHED by the slot — the next resample, or the slot's release, hands them back (pops and pushes never share a
// launch) — and the slot keeps the empty map.
__global__ __launch_bounds__(256) void rbpf_unpack_batch(TilePool P, MapT M, double* __restrict__ pose, double* __restrict__ prev,
                                                         double* __restrict__ weight, int* __restrict__ trow, int* __restrict__ nocc,
                                                         int* __restrict__ fstate, uint16_t* __restrict__ codes, size_t G,
                                                         const BatchItem* __restrict__ items, const char* __restrict__ buf, int* __restrict__ err) {
  const BatchItem it = items[blockIdx.x];
  const int slot = it.slot, tid = threadIdx.x;
  const char* b = buf + it.off;
  const BlobHeader hd = *reinterpret_cast<const BlobHeader*>(b);
  const BlobLayout L = blob_layout_hd(M.TW, G, hd.n_tiles, hd.has_codes != 0);
  __shared__ unsigned long long sbase[256];
  __shared__ int s_fail;
  int rsh = 6;                                                   // log2 of a run's length
  while (((hd.n_tiles + (1u << rsh) - 1u) >> rsh) > 256u) ++rsh;
  const unsigned int run = 1u << rsh, n_runs = (hd.n_tiles + run - 1u) >> rsh;
  const unsigned int* tidx = reinterpret_cast<const unsigned int*>(b + L.tidx);
  unsigned int* const tab = M.table + (size_t)slot * M.TT;
  if (tid == 0) s_fail = 0;
  __syncthreads();
  if ((unsigned int)tid < n_runs) {
    const unsigned int first = (unsigned int)tid << rsh, n = hd.n_tiles - first < run ? hd.n_tiles - first : run;
    const unsigned int list = (blockIdx.x * 5u + (unsigned int)tid) & (P.shards - 1u);
    const TileGrant g = tile_grab(P, n, list);
    unsigned long long pos = g.pos;
    if (g.n != n) {   // the list was short: its grant and the next lists', tile by tile; the ids wait in the table (the entries are zero: the slot was released)
      pos = kPoolScattered;
      TileTaker tk{g.pos, g.n, list, 1u};
      for (unsigned int j = first; j < first + n; ++j) {
        const unsigned int id = tile_take(P, tk, first + n - j);
        if (!id) { s_fail = 1; break; }
        tab[tidx[j]] = id;
      }
    }
    sbase[tid] = pos;
  }
  __syncthreads();
  auto nth = [&](size_t j) { const unsigned long long sb = sbase[j >> rsh]; return sb == kPoolScattered ? tab[tidx[j]] : tile_at(P, sb + (j & (run - 1u))); };
  if (s_fail) {  // pool exhausted
    for (unsigned int j = tid; j < hd.n_tiles; j += 256) {
      const unsigned long long sb = sbase[j >> rsh];
      const unsigned int id = nth(j);     // (a scattered run that stopped half way: zero from there on)
      if (sb == kPoolScattered) tab[tidx[j]] = 0u;
      if (id) { P.ref[id] = 1; M.shed[(size_t)slot * M.TT + tidx[j]] = id; }
    }
    if (tid == 0) atomicOr(&err[3], 8);
    return;
  }
  const double2* src = reinterpret_cast<const double2*>(b + L.tiles);
  for (size_t i = tid; i < (size_t)hd.n_tiles * (kTileCells / 2); i += 256)
    reinterpret_cast<double2*>(P.lo + (size_t)nth(i / (kTileCells / 2)) * kTileCells)[i % (kTileCells / 2)] = src[i];
  const unsigned int* sbm = reinterpret_cast<const unsigned int*>(b + L.tile_bm);
  for (size_t i = tid; i < (size_t)hd.n_tiles * kTS; i += 256) P.bm[(size_t)nth(i / kTS) * kTS + (i % kTS)] = sbm[i];
  __syncthreads();   // (a scattered run's ids are read from the table above: every reader is through before the entries are rewritten)
  for (unsigned int j = tid; j < hd.n_tiles; j += 256) {
    const unsigned int id = nth(j);
    P.ref[id] = 1;
    tab[tidx[j]] = id;
  }
  const int* str = reinterpret_cast<const int*>(b + L.trow);
  for (int r = tid; r < M.TW; r += 256) trow[(size_t)slot * M.TW + r] = str[r];
  if (tid == 0) {
    const double* bs = reinterpret_cast<const double*>(b + L.state);
    for (int q = 0; q < 3; ++q) { pose[(size_t)slot * 3 + q] = bs[q]; prev[(size_t)slot * 3 + q] = bs[3 + q]; }
    weight[slot] = bs[6];
    nocc[slot] = hd.nocc;
    fstate[slot] = hd.has_codes ? 2 : 0;
  }
  if (hd.has_codes) {
    const uint16_t* sc = reinterpret_cast<const uint16_t*>(b + L.codes);
    uint16_t* dc = codes + (size_t)slot * G;
    for (size_t i = tid; i < G; i += 256) dc[i] = sc[i];
  }
}

}  // namespace tbnav_rk
