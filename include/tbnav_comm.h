/* tbnav_comm.h — the communicator the sharded MPPI tick and the sharded RBPF scan exchange through (RCCL over xGMI).
 *
 * The reference is one process per ROS node on one CPU (no parallelism, no collectives),
 * so nothing here replaces a reference interface: it is the ensemble sharding of SURVEY.md section 8-e behind the same
 * C-ABI as the single-GPU paths.  Two ways to get a communicator:
 *
 *   one process per GPU  (bench.py under torch.distributed.run, any MPI-style launcher):
 *       rank 0: tbnav_comm_unique_id(id)  ->  the job's own channel carries the 128 bytes to every rank  ->
 *       every rank: tbnav_comm_create(id, nranks, rank, device, &comm)          (ncclCommInitRank)
 *
 *   one process driving several GPUs  (a ROS node is ONE process: controller::MPPI / bmapping::ParticleFilter built
 *   with n_gpus > 1 use this through tbnav_mppi_group_* / tbnav_rbpf_group_*):
 *       tbnav_comm_create_local(n, devices, comms)                               (ncclCommInitAll)
 *       Devices may repeat (tests on a one-GPU box: {0, 0, ...}): RCCL refuses two ranks on one device, so ranks that
 *       share a device exchange by device-to-device copies ordered with events instead — same layout, same stream
 *       order, no RCCL call.  A list of DISTINCT devices always goes through RCCL.
 *
 * What travels: MPPI — one all-gather of the per-time-step soft-min records per tick ([T][S][8] doubles per rank,
 * include/tbnav_mppi.h); RBPF — one all-gather of the raw weights per scan (N doubles) and, when a resample moves
 * particles across ranks, one all-gather of blob sizes and point-to-point sends of the particles' tile blobs
 * (include/tbnav_rbpf.h).  Everything is enqueued on the handle's stream: no host round trip inside a tick.
 *
 * librccl is loaded on the first tbnav_comm_* call (dlopen), not when libtbnav_hip.so is: single-GPU users never touch it.
 */
#ifndef TBNAV_COMM_H
#define TBNAV_COMM_H

#include <stdint.h>
#include "tbnav_status.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TBNAV_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct tbnav_comm tbnav_comm; /* opaque */

/* A fresh job-wide identifier (ncclGetUniqueId); call on one rank and hand the bytes to the others. */
int tbnav_comm_unique_id(uint8_t id[TBNAV_COMM_ID_BYTES]);
/* The same for the IPC transport: the ranks are processes of ONE node and may share a device, which RCCL refuses.  They meet in
 * a POSIX shared-memory segment named by the id and copy device to device out of each other's allocations (hipIpcMemHandle).
 * Host-synchronous (every exchange drains the stream it is given and returns when the data has arrived) and every exchange is
 * collective over all ranks; a peer that does not show up within two minutes is an error, not a hang.  What it is for: running
 * the multi-process code paths — one rank per process, exactly as under RCCL — on a one-GPU box (tests, bench.py's dev
 * switch).  tbnav_comm_create recognises such an id. */
int tbnav_comm_unique_id_ipc(uint8_t id[TBNAV_COMM_ID_BYTES]);
/* This process's rank of an nranks-rank job; device = HIP ordinal (-1: current).  Collective: returns when every
 * rank has called it. */
int tbnav_comm_create(const uint8_t id[TBNAV_COMM_ID_BYTES], int32_t nranks, int32_t rank, int32_t device, tbnav_comm** out);
/* n communicators of one process, rank r on devices[r] (NULL: device r).  out receives n pointers. */
int tbnav_comm_create_local(int32_t n, const int32_t* devices, tbnav_comm** out);
/* Destroys one communicator (a local group's shared state goes with its last member). */
void tbnav_comm_destroy(tbnav_comm* c);
/* Runs every transport call the library makes — one all-gather, one ring of point-to-point messages (a message to itself when
 * there is one rank: ncclSend / ncclRecv execute) — on `bytes` of pattern data and checks the result on the host.  Collective over
 * the communicator's ranks; communicators made by tbnav_comm_create (and local groups of one). */
int tbnav_comm_selftest(tbnav_comm* c, uint64_t bytes);
int tbnav_comm_rank(const tbnav_comm* c);
int tbnav_comm_size(const tbnav_comm* c);
int tbnav_comm_device(const tbnav_comm* c);
/* 1 if this communicator's exchanges are RCCL calls, 0 if it is a member of a local group whose ranks share devices
 * (in-process copies). */
int tbnav_comm_uses_rccl(const tbnav_comm* c);

#ifdef __cplusplus
}
#endif
#endif /* TBNAV_COMM_H */
