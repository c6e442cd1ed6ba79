// comm.hpp — internal face of include/tbnav_comm.h for the kernel files: the two exchanges the sharded paths need, over
// whichever transport the communicator has (RCCL, or in-process copies for local-group ranks that share a device).
// Every call only ENQUEUES on the given streams; nothing here waits on the host — except on the IPC transport (ranks in separate
// processes that may share a device: tbnav_comm_unique_id_ipc), which drains the stream and returns when the data has arrived.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <vector>

#include "tbnav_comm.h"

namespace tbnav {

// One collective for `n` members: n == 1 — this process's rank of a multi-process job (or a group of one); n > 1 — ALL members
// of one local group, in rank order (one host thread drives them).  recv[r] receives size(comm) * bytes: rank q's
// contribution at offset q * bytes.  send[r] may lie inside recv[r] at its own offset (in place).
int comm_all_gather(int n, tbnav_comm* const* comms, const void* const* send, void* const* recv, size_t bytes, hipStream_t const* streams);

// Point-to-point exchange.  Per member: what it sends and what it receives, as (peer rank, pointer, bytes); between one pair of
// ranks the messages match in list order.  Zero-byte messages are skipped on both sides.
struct P2P { int peer; void* ptr; size_t bytes; };
int comm_exchange(int n, tbnav_comm* const* comms, const std::vector<P2P>* sends, const std::vector<P2P>* recvs, hipStream_t const* streams);

// A small HOST blob from every rank (IPC handles, status words): recv_host receives size(comm) * bytes, rank q's at q * bytes.
// Collective and host-synchronous; communicators of multi-process jobs (and groups of one).
int comm_all_gather_host(tbnav_comm* c, const void* send_host, void* recv_host, size_t bytes);
// a rank of a multi-process job of more than one rank (not a member of a one-process group)
bool comm_is_multiprocess(const tbnav_comm* c);

int comm_rank(const tbnav_comm* c);
int comm_size(const tbnav_comm* c);

}  // namespace tbnav
