// comm.hip — include/tbnav_comm.h: communicators for the sharded MPPI tick and RBPF scan.
// RCCL (librccl, loaded on first use) carries every exchange between distinct devices; ranks of a one-process group that
// share a device (tests on a one-GPU box) exchange by event-ordered device-to-device copies with the same layout.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "comm.hpp"
#include "common.hpp"

namespace {

// ---- librccl, resolved at run time ------------------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};
Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy the process has mapped already (PyTorch ships its own librccl) is reused; otherwise the system's
    for (const char* name : {"librccl.so", "librccl.so.1"}) { r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break; }
    if (!r.lib) for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) { r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
    if (!r.lib) return;
#define TBNAV_SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f))
    TBNAV_SYM(GetUniqueId); TBNAV_SYM(CommInitRank); TBNAV_SYM(CommInitAll); TBNAV_SYM(CommDestroy); TBNAV_SYM(AllGather);
    TBNAV_SYM(Send); TBNAV_SYM(Recv); TBNAV_SYM(GroupStart); TBNAV_SYM(GroupEnd); TBNAV_SYM(GetErrorString);
#undef TBNAV_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllGather && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
  });
  return r;
}
int rccl_fail(ncclResult_t e, const char* what) {
  char buf[384];
  const Rccl& r = rccl();
  std::snprintf(buf, sizeof buf, "rccl: %s failed: %s", what, r.GetErrorString ? r.GetErrorString(e) : "?");
  tbnav::last_hip_error_slot() = buf;
  return TBNAV_ERR_HIP;
}
#define TBNAV_NCCL(call) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return rccl_fail(e_, #call); } while (0)

// shared state of the communicators one process made with tbnav_comm_create_local
struct LocalGroup {
  int n = 0, alive = 0;
  bool use_rccl = false;                 // distinct devices: RCCL; repeated devices: in-process copies
  std::vector<hipEvent_t> ready, done;   // copy transport: "my send buffer is written" / "I have read everybody's"
  std::vector<int> device;
};

struct DevGuard {
  int prev = -1;
  explicit DevGuard(int d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; (void)hipSetDevice(d); }
  ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// ---- the IPC transport: processes of one node whose ranks may SHARE a device (RCCL refuses two ranks on one device) ----------
// A POSIX shared-memory segment carries a barrier and, per rank, the table of what it sends this round (destination, bytes);
// every rank owns one device MAILBOX, exported once (hipIpcMemHandle) and mapped once by every peer when the communicator is
// made.  An exchange: the sender copies its messages, back to back, into its own mailbox; the ranks meet; every receiver
// copies what is addressed to it out of the sender's mailbox (device to device); they meet again.  More than a mailbox holds
// goes in several such rounds.  Host-synchronous — every call drains the caller's stream and returns when the data has
// arrived — which is more than the stream order the callers rely on.  What it is for: running the multi-PROCESS code paths (one
// rank per process, exactly as under RCCL) on a one-GPU box — tests/test_comm_processes_gpu.py, bench.py's one-GPU dev switch.
// An all-zero segment is the initial state.
constexpr unsigned char kIpcMagic[8] = {'T', 'B', 'N', 'A', 'V', 'I', 'P', 'C'};
constexpr int kIpcMaxRanks = 16, kIpcMaxMsgs = kIpcMaxRanks + 1;
constexpr size_t kIpcMailbox = (size_t)16 << 20;
constexpr double kIpcTimeoutS = 120.0;   // a peer that never arrives (it failed) is an error here, not a hang
struct IpcMsg { unsigned long long bytes; int dst, pad; };   // dst -1: every other rank reads it (an all-gather's block)
struct IpcSlot { hipIpcMemHandle_t mailbox; unsigned int n_msgs, pad; IpcMsg msg[kIpcMaxMsgs]; };
struct IpcShared {
  std::atomic<unsigned int> arrived, generation, failed;
  unsigned int pad;
  IpcSlot slot[kIpcMaxRanks];
};
struct IpcState {
  IpcShared* sh = nullptr;
  char* mailbox = nullptr;            // mine
  char* peer[kIpcMaxRanks] = {};      // the others', mapped
  bool i_failed = false;
  // a failure is final for the whole communicator: the flag in the segment stops every rank at its next barrier
  int fail(const char* what) {
    if (sh) sh->failed.store(1u);
    if (!i_failed) tbnav::last_hip_error_slot() = std::string("ipc transport: ") + what;   // (the first cause stays)
    i_failed = true;
    return TBNAV_ERR_HIP;
  }
  int peer_failed() {
    if (!i_failed) tbnav::last_hip_error_slot() = "ipc transport: another rank reported a failure";
    return TBNAV_ERR_HIP;
  }
  // every rank arrives; TBNAV_ERR_HIP if some rank has failed or a peer does not arrive within kIpcTimeoutS
  int barrier(int nranks) {
    if (sh->failed.load()) return peer_failed();
    const unsigned int gen = sh->generation.load();
    if (sh->arrived.fetch_add(1u) + 1u == (unsigned int)nranks) { sh->arrived.store(0u); sh->generation.fetch_add(1u); }
    else {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned int spins = 0; sh->generation.load() == gen; ++spins) {
        if (spins > 2000) usleep(50);
        if ((spins & 255u) == 255u) {
          if (sh->failed.load()) return peer_failed();
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kIpcTimeoutS) return fail("a rank did not arrive at the barrier");
        }
      }
    }
    return sh->failed.load() ? peer_failed() : TBNAV_OK;
  }
};

}  // namespace

struct tbnav_comm {
  int rank = 0, nranks = 1, device = 0;
  ncclComm_t nccl = nullptr;
  LocalGroup* group = nullptr;  // non-null: member of a one-process group
  IpcState* ipc = nullptr;      // non-null: rank of a multi-process job on the IPC transport
};

namespace {
// One exchange on the IPC transport (collective over ALL ranks).  sends: (peer, pointer, bytes), peer -1 = every other rank reads
// it; recvs: (peer, pointer, bytes); message i from q to me = the i-th entry of q's table naming me (or everybody) and my i-th
// receive naming q.  A rank's messages form one byte stream in table order; round i moves the stream's window
// [i * kIpcMailbox, (i + 1) * kIpcMailbox) through the mailbox.
int ipc_round(tbnav_comm* c, const std::vector<tbnav::P2P>& sends, const std::vector<tbnav::P2P>& recvs, hipStream_t st) {
  IpcState& I = *c->ipc;
  DevGuard dg(c->device);
  const int P = c->nranks, me = c->rank;
  int rc = TBNAV_OK;
  auto note = [&](bool ok, const char* what) { if (!ok && rc == TBNAV_OK) rc = I.fail(what); return ok; };
  note(hipStreamSynchronize(st) == hipSuccess, "hipStreamSynchronize");   // what I send is written
  IpcSlot& mine = I.sh->slot[me];
  if (note(sends.size() <= (size_t)kIpcMaxMsgs, "too many messages in one exchange")) {
    mine.n_msgs = (unsigned int)sends.size();
    for (size_t i = 0; i < sends.size(); ++i) { mine.msg[i].bytes = sends[i].bytes; mine.msg[i].dst = sends[i].peer; mine.msg[i].pad = 0; }
  } else mine.n_msgs = 0;
  if (I.barrier(P) != TBNAV_OK || rc != TBNAV_OK) return TBNAV_ERR_HIP;   // tables are up
  // where each of my receives sits in its sender's stream
  struct Piece { const char* src_base; unsigned long long off, bytes; char* dst; };
  std::vector<Piece> in;
  unsigned long long longest = 0;
  for (int q = 0; q < P; ++q) {
    const IpcSlot& sl = I.sh->slot[q];
    unsigned long long total = 0;
    unsigned int is = 0;
    std::vector<unsigned long long> off(sl.n_msgs + 1, 0);
    for (unsigned int i = 0; i < sl.n_msgs; ++i) off[i + 1] = off[i] + sl.msg[i].bytes;
    total = off[sl.n_msgs];
    longest = total > longest ? total : longest;
    if (q == me) continue;
    for (const tbnav::P2P& rv : recvs) {
      if (rv.peer != q) continue;
      while (is < sl.n_msgs && sl.msg[is].dst != me && sl.msg[is].dst != -1) ++is;
      if (!note(is < sl.n_msgs && sl.msg[is].bytes == rv.bytes, "the two sides of an exchange disagree")) break;
      if (rv.bytes) in.push_back(Piece{I.peer[q], off[is], rv.bytes, static_cast<char*>(rv.ptr)});
      ++is;
    }
  }
  // a message to myself (the self-test of a one-rank communicator) never touches the mailbox
  for (const tbnav::P2P& rv : recvs) {
    if (rv.peer != me || !rv.bytes) continue;
    const tbnav::P2P* sv = nullptr;
    for (const tbnav::P2P& x : sends) if (x.peer == me && x.bytes == rv.bytes) { sv = &x; break; }
    if (note(sv != nullptr, "a message to itself without its send")) note(hipMemcpyAsync(rv.ptr, sv->ptr, rv.bytes, hipMemcpyDeviceToDevice, st) == hipSuccess, "local copy");
  }
  std::vector<unsigned long long> my_off(sends.size() + 1, 0);
  for (size_t i = 0; i < sends.size(); ++i) my_off[i + 1] = my_off[i] + sends[i].bytes;
  const unsigned long long rounds = (longest + kIpcMailbox - 1) / kIpcMailbox;
  for (unsigned long long r = 0; r < rounds; ++r) {
    const unsigned long long w0 = r * kIpcMailbox, w1 = w0 + kIpcMailbox;
    for (size_t i = 0; i < sends.size() && rc == TBNAV_OK; ++i) {     // my stream's window into my mailbox
      if (sends[i].peer == me) continue;
      const unsigned long long a0 = my_off[i] > w0 ? my_off[i] : w0, a1 = my_off[i + 1] < w1 ? my_off[i + 1] : w1;
      if (a0 < a1) note(hipMemcpyAsync(I.mailbox + (a0 - w0), static_cast<const char*>(sends[i].ptr) + (a0 - my_off[i]), (size_t)(a1 - a0), hipMemcpyDeviceToDevice, st) == hipSuccess, "copy into the mailbox");
    }
    note(hipStreamSynchronize(st) == hipSuccess, "hipStreamSynchronize");
    if (I.barrier(P) != TBNAV_OK || rc != TBNAV_OK) return TBNAV_ERR_HIP;   // mailboxes are full
    for (const Piece& pc : in) {
      const unsigned long long a0 = pc.off > w0 ? pc.off : w0, a1 = pc.off + pc.bytes < w1 ? pc.off + pc.bytes : w1;
      if (a0 < a1 && rc == TBNAV_OK) note(hipMemcpyAsync(pc.dst + (a0 - pc.off), pc.src_base + (a0 - w0), (size_t)(a1 - a0), hipMemcpyDeviceToDevice, st) == hipSuccess, "copy out of a peer's mailbox");
    }
    note(hipStreamSynchronize(st) == hipSuccess, "hipStreamSynchronize");
    if (I.barrier(P) != TBNAV_OK || rc != TBNAV_OK) return TBNAV_ERR_HIP;   // mailboxes are free again
  }
  if (rounds == 0) note(hipStreamSynchronize(st) == hipSuccess, "hipStreamSynchronize");
  return rc;
}

// the mailboxes: allocate and export mine, map everybody else's (called once, by tbnav_comm_create)
int ipc_open_mailboxes(tbnav_comm* c) {
  IpcState& I = *c->ipc;
  DevGuard dg(c->device);
  int rc = TBNAV_OK;
  if (hipMalloc((void**)&I.mailbox, kIpcMailbox) != hipSuccess) rc = I.fail("hipMalloc of the mailbox");
  if (rc == TBNAV_OK && hipIpcGetMemHandle(&I.sh->slot[c->rank].mailbox, I.mailbox) != hipSuccess) rc = I.fail("hipIpcGetMemHandle");
  if (I.barrier(c->nranks) != TBNAV_OK || rc != TBNAV_OK) return TBNAV_ERR_HIP;
  for (int q = 0; q < c->nranks && rc == TBNAV_OK; ++q) {
    if (q == c->rank) { I.peer[q] = I.mailbox; continue; }
    void* base = nullptr;
    if (hipIpcOpenMemHandle(&base, I.sh->slot[q].mailbox, hipIpcMemLazyEnablePeerAccess) != hipSuccess) rc = I.fail("hipIpcOpenMemHandle");
    I.peer[q] = static_cast<char*>(base);
  }
  if (I.barrier(c->nranks) != TBNAV_OK || rc != TBNAV_OK) return TBNAV_ERR_HIP;
  return TBNAV_OK;
}
void ipc_close(tbnav_comm* c) {
  IpcState* I = c->ipc;
  if (!I) return;
  DevGuard dg(c->device);
  (void)hipDeviceSynchronize();
  for (int q = 0; q < c->nranks; ++q) if (q != c->rank && I->peer[q]) (void)hipIpcCloseMemHandle(I->peer[q]);
  if (I->mailbox) (void)hipFree(I->mailbox);
  if (I->sh) munmap(I->sh, sizeof(IpcShared));
  delete I;
  c->ipc = nullptr;
}
}  // namespace

namespace tbnav {

int comm_rank(const tbnav_comm* c) { return c ? c->rank : 0; }
int comm_size(const tbnav_comm* c) { return c ? c->nranks : 1; }

namespace {
// the members handed in are either one rank of a multi-process job, or every member of one local group in rank order
int check_members(int n, tbnav_comm* const* comms) {
  if (n <= 0 || !comms || !comms[0]) return TBNAV_ERR_INVALID_ARG;
  if (n == 1) return (comms[0]->group && comms[0]->nranks != 1) ? TBNAV_ERR_INVALID_ARG : TBNAV_OK;
  LocalGroup* g = comms[0]->group;
  if (!g || g->n != n) return TBNAV_ERR_INVALID_ARG;
  for (int r = 0; r < n; ++r) if (!comms[r] || comms[r]->group != g || comms[r]->rank != r) return TBNAV_ERR_INVALID_ARG;
  return TBNAV_OK;
}
}  // namespace

int comm_all_gather(int n, tbnav_comm* const* comms, const void* const* send, void* const* recv, size_t bytes, hipStream_t const* streams) {
  { const int rc = check_members(n, comms); if (rc != TBNAV_OK) return rc; }
  if (bytes == 0) return TBNAV_OK;
  LocalGroup* g = comms[0]->group;
  if (comms[0]->ipc) {  // (n == 1: check_members) my block is ONE entry every other rank reads; the own block is a local copy
    tbnav_comm* c = comms[0];
    std::vector<P2P> sends{P2P{-1, const_cast<void*>(send[0]), bytes}}, recvs;
    for (int q = 0; q < c->nranks; ++q) if (q != c->rank) recvs.push_back(P2P{q, static_cast<char*>(recv[0]) + (size_t)q * bytes, bytes});
    char* own = static_cast<char*>(recv[0]) + (size_t)c->rank * bytes;
    if (own != send[0]) { DevGuard dg(c->device); TBNAV_HIP(hipMemcpyAsync(own, send[0], bytes, hipMemcpyDeviceToDevice, streams[0])); }
    return ipc_round(c, sends, recvs, streams[0]);
  }
  if (!g || g->use_rccl) {
    Rccl& R = rccl();
    if (!R.ok) return TBNAV_ERR_UNSUPPORTED;
    if (n > 1) TBNAV_NCCL(R.GroupStart());
    for (int r = 0; r < n; ++r) {
      DevGuard dg(comms[r]->device);
      const ncclResult_t e = R.AllGather(send[r], recv[r], bytes, ncclChar, comms[r]->nccl, streams[r]);
      if (e != ncclSuccess) { if (n > 1) (void)R.GroupEnd(); return rccl_fail(e, "ncclAllGather"); }
    }
    if (n > 1) TBNAV_NCCL(R.GroupEnd());
    return TBNAV_OK;
  }
  // copy transport: rank q's block goes to every rank's recv + q * bytes, on the RECEIVER's stream behind the sender's "ready"
  for (int r = 0; r < n; ++r) { DevGuard dg(g->device[r]); TBNAV_HIP(hipEventRecord(g->ready[r], streams[r])); }
  for (int d = 0; d < n; ++d) {
    DevGuard dg(g->device[d]);
    for (int q = 0; q < n; ++q) {
      char* dst = static_cast<char*>(recv[d]) + (size_t)q * bytes;
      if (dst == send[q]) continue;  // in place
      if (q != d) TBNAV_HIP(hipStreamWaitEvent(streams[d], g->ready[q], 0));
      TBNAV_HIP(hipMemcpyAsync(dst, send[q], bytes, hipMemcpyDeviceToDevice, streams[d]));
    }
    TBNAV_HIP(hipEventRecord(g->done[d], streams[d]));
  }
  // nobody rewrites its send buffer before every reader is through with it
  for (int r = 0; r < n; ++r) {
    DevGuard dg(g->device[r]);
    for (int d = 0; d < n; ++d) if (d != r) TBNAV_HIP(hipStreamWaitEvent(streams[r], g->done[d], 0));
  }
  return TBNAV_OK;
}

int comm_exchange(int n, tbnav_comm* const* comms, const std::vector<P2P>* sends, const std::vector<P2P>* recvs, hipStream_t const* streams) {
  { const int rc = check_members(n, comms); if (rc != TBNAV_OK) return rc; }
  LocalGroup* g = comms[0]->group;
  const int world = comms[0]->nranks;
  for (int r = 0; r < n; ++r) {
    for (const P2P& m : sends[r]) if (m.peer < 0 || m.peer >= world || (m.bytes && !m.ptr)) return TBNAV_ERR_INVALID_ARG;
    for (const P2P& m : recvs[r]) if (m.peer < 0 || m.peer >= world || (m.bytes && !m.ptr)) return TBNAV_ERR_INVALID_ARG;
  }
  if (comms[0]->ipc) return ipc_round(comms[0], sends[0], recvs[0], streams[0]);  // (collective on this transport: every rank calls, also with nothing to send)
  if (!g || g->use_rccl) {
    Rccl& R = rccl();
    if (!R.ok) return TBNAV_ERR_UNSUPPORTED;
    bool any = false;
    for (int r = 0; r < n; ++r) any |= !sends[r].empty() || !recvs[r].empty();
    if (!any) return TBNAV_OK;
    TBNAV_NCCL(R.GroupStart());
    ncclResult_t e = ncclSuccess;
    for (int r = 0; r < n && e == ncclSuccess; ++r) {
      DevGuard dg(comms[r]->device);
      for (const P2P& m : sends[r]) if (m.bytes && e == ncclSuccess) e = R.Send(m.ptr, m.bytes, ncclChar, m.peer, comms[r]->nccl, streams[r]);
      for (const P2P& m : recvs[r]) if (m.bytes && e == ncclSuccess) e = R.Recv(m.ptr, m.bytes, ncclChar, m.peer, comms[r]->nccl, streams[r]);
    }
    const ncclResult_t e2 = R.GroupEnd();
    if (e != ncclSuccess) return rccl_fail(e, "ncclSend / ncclRecv");
    if (e2 != ncclSuccess) return rccl_fail(e2, "ncclGroupEnd");
    return TBNAV_OK;
  }
  // copy transport: message i from q to d = the i-th send of q naming d and the i-th receive of d naming q
  for (int r = 0; r < n; ++r) { DevGuard dg(g->device[r]); TBNAV_HIP(hipEventRecord(g->ready[r], streams[r])); }
  for (int d = 0; d < n; ++d) {
    DevGuard dg(g->device[d]);
    for (int q = 0; q < n; ++q) {
      size_t is = 0;
      bool waited = q == d;
      for (const P2P& rv : recvs[d]) {
        if (rv.peer != q) continue;
        while (is < sends[q].size() && sends[q][is].peer != d) ++is;
        if (is == sends[q].size() || sends[q][is].bytes != rv.bytes) return TBNAV_ERR_INVALID_ARG;  // the two sides disagree
        if (rv.bytes) {
          if (!waited) { TBNAV_HIP(hipStreamWaitEvent(streams[d], g->ready[q], 0)); waited = true; }
          TBNAV_HIP(hipMemcpyAsync(rv.ptr, sends[q][is].ptr, rv.bytes, hipMemcpyDeviceToDevice, streams[d]));
        }
        ++is;
      }
      for (; is < sends[q].size(); ++is) if (sends[q][is].peer == d) return TBNAV_ERR_INVALID_ARG;  // a send nobody receives
    }
    TBNAV_HIP(hipEventRecord(g->done[d], streams[d]));
  }
  for (int r = 0; r < n; ++r) {
    DevGuard dg(g->device[r]);
    for (int d = 0; d < n; ++d) if (d != r) TBNAV_HIP(hipStreamWaitEvent(streams[r], g->done[d], 0));
  }
  return TBNAV_OK;
}

bool comm_is_multiprocess(const tbnav_comm* c) { return c && !c->group && c->nranks > 1; }

int comm_all_gather_host(tbnav_comm* c, const void* send_host, void* recv_host, size_t bytes) {
  if (!c || !send_host || !recv_host || bytes == 0 || (c->group && c->nranks != 1)) return TBNAV_ERR_INVALID_ARG;
  DevGuard dg(c->device);
  hipStream_t st = nullptr;
  char *d_send = nullptr, *d_recv = nullptr;
  auto done = [&](int code) { if (st) (void)hipStreamDestroy(st); (void)hipFree(d_send); (void)hipFree(d_recv); return code; };
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&d_send, bytes) != hipSuccess ||
      hipMalloc((void**)&d_recv, bytes * (size_t)c->nranks) != hipSuccess) return done(TBNAV_ERR_HIP);
  if (hipMemcpyAsync(d_send, send_host, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return done(TBNAV_ERR_HIP);
  const void* sp = d_send; void* rp = d_recv;
  const int rc = comm_all_gather(1, &c, &sp, &rp, bytes, &st);
  if (rc != TBNAV_OK) return done(rc);
  if (hipMemcpyAsync(recv_host, d_recv, bytes * (size_t)c->nranks, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return done(TBNAV_ERR_HIP);
  return done(TBNAV_OK);
}

}  // namespace tbnav

extern "C" {

int tbnav_comm_unique_id(uint8_t id[TBNAV_COMM_ID_BYTES]) {
  if (!id) return TBNAV_ERR_INVALID_ARG;
  Rccl& R = rccl();
  if (!R.ok) { tbnav::last_hip_error_slot() = "librccl could not be loaded"; return TBNAV_ERR_UNSUPPORTED; }
  static_assert(sizeof(ncclUniqueId) == TBNAV_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  TBNAV_NCCL(R.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return TBNAV_OK;
}

int tbnav_comm_unique_id_ipc(uint8_t id[TBNAV_COMM_ID_BYTES]) {
  if (!id) return TBNAV_ERR_INVALID_ARG;
  std::memset(id, 0, TBNAV_COMM_ID_BYTES);
  std::memcpy(id, kIpcMagic, sizeof kIpcMagic);
  const int fd = open("/dev/urandom", O_RDONLY);
  const bool ok = fd >= 0 && read(fd, id + 8, 16) == 16;
  if (fd >= 0) close(fd);
  if (!ok) { tbnav::last_hip_error_slot() = "ipc transport: /dev/urandom"; return TBNAV_ERR_HIP; }
  return TBNAV_OK;
}

int tbnav_comm_create(const uint8_t id[TBNAV_COMM_ID_BYTES], int32_t nranks, int32_t rank, int32_t device, tbnav_comm** out) {
  if (!id || !out || nranks <= 0 || rank < 0 || rank >= nranks) return TBNAV_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  { const hipError_t e = hipGetDeviceCount(&ndev); if (e != hipSuccess || ndev <= 0) return tbnav::hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount", __FILE__, __LINE__); }
  if (device < 0) TBNAV_HIP(hipGetDevice(&device));
  if (device >= ndev) return TBNAV_ERR_INVALID_ARG;
  if (std::memcmp(id, kIpcMagic, sizeof kIpcMagic) == 0) {  // an id drawn by tbnav_comm_unique_id_ipc: the IPC transport
    if (nranks > kIpcMaxRanks) return TBNAV_ERR_INVALID_ARG;
    char name[48] = "/tbnav_";
    for (int i = 0; i < 16; ++i) std::snprintf(name + 7 + 2 * i, 3, "%02x", id[8 + i]);
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { tbnav::last_hip_error_slot() = "ipc transport: shm_open"; return TBNAV_ERR_HIP; }
    void* mem = MAP_FAILED;
    if (ftruncate(fd, (off_t)sizeof(IpcShared)) == 0) mem = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) { tbnav::last_hip_error_slot() = "ipc transport: mmap"; return TBNAV_ERR_HIP; }
    tbnav_comm* c = new (std::nothrow) tbnav_comm();
    IpcState* st = new (std::nothrow) IpcState();
    if (!c || !st) { delete c; delete st; munmap(mem, sizeof(IpcShared)); return TBNAV_ERR_INVALID_ARG; }
    st->sh = static_cast<IpcShared*>(mem);
    c->rank = rank; c->nranks = nranks; c->device = device; c->ipc = st;
    int rc = st->barrier(nranks);   // collective, as ncclCommInitRank is
    if (rank == 0) (void)shm_unlink(name);  // every rank has it mapped (or has given up): the name can go
    if (rc == TBNAV_OK) rc = ipc_open_mailboxes(c);
    if (rc != TBNAV_OK) { ipc_close(c); delete c; return rc; }
    *out = c;
    return TBNAV_OK;
  }
  Rccl& R = rccl();
  if (!R.ok) { tbnav::last_hip_error_slot() = "librccl could not be loaded"; return TBNAV_ERR_UNSUPPORTED; }
  tbnav_comm* c = new (std::nothrow) tbnav_comm();
  if (!c) return TBNAV_ERR_INVALID_ARG;
  c->rank = rank; c->nranks = nranks; c->device = device;
  DevGuard dg(device);
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  const ncclResult_t e = R.CommInitRank(&c->nccl, nranks, u, rank);
  if (e != ncclSuccess) { delete c; return rccl_fail(e, "ncclCommInitRank"); }
  *out = c;
  return TBNAV_OK;
}

int tbnav_comm_create_local(int32_t n, const int32_t* devices, tbnav_comm** out) {
  if (n <= 0 || !out) return TBNAV_ERR_INVALID_ARG;
  for (int r = 0; r < n; ++r) out[r] = nullptr;
  int ndev = 0;
  { const hipError_t e = hipGetDeviceCount(&ndev); if (e != hipSuccess || ndev <= 0) return tbnav::hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount", __FILE__, __LINE__); }
  std::vector<int> dev(n);
  std::set<int> distinct;
  for (int r = 0; r < n; ++r) {
    dev[r] = devices ? devices[r] : r;
    if (dev[r] < 0 || dev[r] >= ndev) return TBNAV_ERR_INVALID_ARG;
    distinct.insert(dev[r]);
  }
  LocalGroup* g = new (std::nothrow) LocalGroup();
  if (!g) return TBNAV_ERR_INVALID_ARG;
  g->n = n; g->alive = n; g->device = dev;
  g->use_rccl = (int)distinct.size() == n;
  std::vector<ncclComm_t> nc(n, nullptr);
  if (g->use_rccl) {
    Rccl& R = rccl();
    if (!R.ok) { delete g; tbnav::last_hip_error_slot() = "librccl could not be loaded"; return TBNAV_ERR_UNSUPPORTED; }
    const ncclResult_t e = R.CommInitAll(nc.data(), n, dev.data());
    if (e != ncclSuccess) { delete g; return rccl_fail(e, "ncclCommInitAll"); }
  } else {
    g->ready.resize(n); g->done.resize(n);
    for (int r = 0; r < n; ++r) {
      DevGuard dg(dev[r]);
      if (hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g->done[r], hipEventDisableTiming) != hipSuccess) {
        delete g;  // (events created so far leak with a failing runtime: nothing else can be done about them here)
        return TBNAV_ERR_HIP;
      }
    }
  }
  for (int r = 0; r < n; ++r) {
    tbnav_comm* c = new (std::nothrow) tbnav_comm();
    if (!c) return TBNAV_ERR_INVALID_ARG;
    c->rank = r; c->nranks = n; c->device = dev[r]; c->nccl = nc[r]; c->group = g;
    out[r] = c;
  }
  return TBNAV_OK;
}

void tbnav_comm_destroy(tbnav_comm* c) {
  if (!c) return;
  if (c->nccl) { DevGuard dg(c->device); (void)rccl().CommDestroy(c->nccl); }
  ipc_close(c);
  if (LocalGroup* g = c->group) {
    if (--g->alive == 0) {
      for (size_t r = 0; r < g->ready.size(); ++r) { DevGuard dg(g->device[r]); (void)hipEventDestroy(g->ready[r]); (void)hipEventDestroy(g->done[r]); }
      delete g;
    }
  }
  delete c;
}

// Every transport call this library makes, once, on `bytes` of pattern data, checked on the host: an all-gather (each rank's
// block is its rank number repeated) and a ring of point-to-point messages (rank r sends to r + 1, receives from r - 1; with one
// rank that is a message to itself — RCCL supports it — so ncclSend / ncclRecv execute even on a one-GPU box).  Collective:
// every rank of the communicator calls it (a one-process group: pass member 0, the group's members are walked here).
int tbnav_comm_selftest(tbnav_comm* c, uint64_t bytes) {
  if (!c || bytes == 0 || bytes > (1ull << 28)) return TBNAV_ERR_INVALID_ARG;
  const int P = c->nranks;
  // a one-process group is driven as a whole from its member 0 — which this library never keeps a list of: only the multi-process
  // form and groups of one are tested here (the groups' exchanges are exercised through tbnav_mppi_group_* / tbnav_rbpf_group_*)
  if (c->group && P != 1) return TBNAV_ERR_UNSUPPORTED;
  DevGuard dg(c->device);
  hipStream_t st = nullptr;
  TBNAV_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned char *d_send = nullptr, *d_all = nullptr, *d_recv = nullptr;
  int rc = TBNAV_OK;
  std::vector<unsigned char> h((size_t)bytes * P);
  auto done = [&](int code) { (void)hipStreamDestroy(st); (void)hipFree(d_send); (void)hipFree(d_all); (void)hipFree(d_recv); return code; };
  if (hipMalloc((void**)&d_send, bytes) != hipSuccess || hipMalloc((void**)&d_all, bytes * P) != hipSuccess || hipMalloc((void**)&d_recv, bytes) != hipSuccess) return done(TBNAV_ERR_HIP);
  if (hipMemsetAsync(d_send, 0x40 + c->rank, bytes, st) != hipSuccess || hipMemsetAsync(d_all, 0, bytes * P, st) != hipSuccess || hipMemsetAsync(d_recv, 0, bytes, st) != hipSuccess) return done(TBNAV_ERR_HIP);
  const void* sp = d_send; void* rp = d_all;
  rc = tbnav::comm_all_gather(1, &c, &sp, &rp, (size_t)bytes, &st);
  if (rc != TBNAV_OK) return done(rc);
  std::vector<tbnav::P2P> sends{tbnav::P2P{(c->rank + 1) % P, d_send, (size_t)bytes}}, recvs{tbnav::P2P{(c->rank + P - 1) % P, d_recv, (size_t)bytes}};
  rc = tbnav::comm_exchange(1, &c, &sends, &recvs, &st);
  if (rc != TBNAV_OK) return done(rc);
  if (hipMemcpyAsync(h.data(), d_all, bytes * P, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return done(TBNAV_ERR_HIP);
  for (int q = 0; q < P; ++q) for (uint64_t i = 0; i < bytes; ++i) if (h[(size_t)q * bytes + i] != (unsigned char)(0x40 + q)) return done(TBNAV_ERR_HIP);
  if (hipMemcpy(h.data(), d_recv, bytes, hipMemcpyDeviceToHost) != hipSuccess) return done(TBNAV_ERR_HIP);
  for (uint64_t i = 0; i < bytes; ++i) if (h[i] != (unsigned char)(0x40 + (c->rank + P - 1) % P)) return done(TBNAV_ERR_HIP);
  return done(TBNAV_OK);
}

int tbnav_comm_rank(const tbnav_comm* c) { return c ? c->rank : -1; }
int tbnav_comm_size(const tbnav_comm* c) { return c ? c->nranks : -1; }
int tbnav_comm_device(const tbnav_comm* c) { return c ? c->device : -1; }
int tbnav_comm_uses_rccl(const tbnav_comm* c) { return c ? (c->nccl ? 1 : 0) : -1; }

}  // extern "C"
