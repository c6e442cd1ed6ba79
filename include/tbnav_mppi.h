/* tbnav_mppi.h — C-ABI of the MI355X MPPI rollout path.
 *
 * Drop-in boundary for controller::MPPI (reference controller/include/controller/mppi.hpp:119-185,
 * controller/src/controller/mppi.cpp:28-184).  Plain pointers and sizes only; the handle owns all
 * device memory; one handle per controller object; not thread-safe (the reference is not either:
 * it draws from a process-global RNG, rigid2d/src/rigid2d/utilities.cpp:12-24).
 *
 * State vector order is (x, y, theta) as in mppi.cpp:75-76 — NOT rigid2d::Pose's (theta, x, y).
 *
 * Device data layout (all fp64):
 *   duL, duR : [T][K]  time-major, rollout index fastest (lanes = rollouts, coalesced)
 *              — the transpose of the reference's column-major duL/duR(T,K), mppi.hpp:182-183
 *   J        : [T][K]  cost-to-go, same layout                           (mppi.hpp:181)
 *   u        : [2][T]  warm-start controls, resident on the device       (mppi.hpp:180)
 *   records  : [T][R][TBNAV_MPPI_REC]  per-time-step soft-min partial sums (see below)
 *
 * Sharded soft-min (one shard = one GPU, or one K-slice inside a GPU).  For time step i and shard
 * g over its rollouts k:
 *   m = min_k J(i,k);  e_k = exp(((J(i,k) - m) * -1.0) / lambda)
 *   rec = { m, A=sum e_k, B=sum e_k*duL(i,k), C=sum e_k*duR(i,k), D=sum duL(i,k), E=sum duR(i,k),
 *           n=count, 0 }
 * Combining R records reproduces mppi.cpp:112-126 including its "+1e-8" weight floor:
 *   M = min m_r;  s_r = exp(((m_r - M) * -1.0) / lambda)
 *   W  = sum s_r*A_r + 1e-8 * sum n_r
 *   uL(i) += (sum s_r*B_r + 1e-8 * sum D_r) / W     (then clamp to +-max_wheel_vel, mppi.cpp:124-125)
 */
#ifndef TBNAV_MPPI_H
#define TBNAV_MPPI_H

#include <stdint.h>
#include "tbnav_status.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TBNAV_MPPI_REC 8 /* doubles per partial record: m, A, B, C, D, E, n, pad */

/* Constructor arguments of controller::CartModel (mppi.hpp:33), controller::LossFunc
 * (mppi.hpp:63-65) and controller::MPPI (mppi.hpp:133-141), flattened. */
typedef struct tbnav_mppi_params {
  double wheel_radius;   /* CartModel::wheel_radius                      */
  double wheel_base;     /* CartModel::wheel_base                        */
  double lambda;         /* temperature                                  */
  double max_wheel_vel;  /* clamp on the updated controls                */
  double ul_var;         /* sampling variance, left wheel                */
  double ur_var;         /* sampling variance, right wheel               */
  double horizon;        /* seconds; steps = (int)(horizon / dt), mppi.cpp:47 */
  double dt;             /* RK4 step                                     */
  double Q[3];           /* diag state cost (x, y, theta)                */
  double R[2];           /* diag control cost (uL, uR)                   */
  double P1[3];          /* diag terminal cost                           */
  int32_t rollouts;      /* K handled by THIS handle (a shard's slice when sharded) */
  int32_t device;        /* HIP device ordinal, -1 = current device      */
} tbnav_mppi_params;

typedef struct tbnav_mppi tbnav_mppi; /* opaque */

/* ---- lifetime -------------------------------------------------------------------------------- */

/* MPPI::MPPI + initController (mppi.cpp:28-51,157-170): u = 0, uinit = 0, xd = 0, J/duL/duR = 0. */
int tbnav_mppi_create(const tbnav_mppi_params* params, tbnav_mppi** out);
/* Waits for the handle's device.  A handle that is still attached to a communicator detaches first (tbnav_mppi_attach_comm(h, NULL)):
 * destroy handles BEFORE their communicators; where the attachment is a multi-process direct exchange the detach — hence this call —
 * is collective (every rank of the communicator destroys or detaches). */
void tbnav_mppi_destroy(tbnav_mppi* h);

int tbnav_mppi_steps(const tbnav_mppi* h);    /* T = (int)(horizon / dt)   */
int tbnav_mppi_rollouts(const tbnav_mppi* h); /* K of this handle           */
/* Which rollout kernel a tick of this handle launches: 0 = mppi_rollout_cost (one lane per rollout, sequential in
 * time), n > 0 = mppi_rollout_scan with n time steps per thread (64 rollouts x ceil(T/n) waves), n < 0 =
 * mppi_rollout_fused with -n rollouts per workgroup (one wave per rollout, lanes over time, partial records formed
 * in the same launch; tbnav_mppi_shard_partials then folds those fine records into the K-slice records). */
int tbnav_mppi_rollout_variant(const tbnav_mppi* h);
/* Which form of the sequential (one lane per rollout) kernel a variant-0 handle launches: 0 = mppi_rollout_cost (general: any T,
 * either dynamics), 2 = mppi_rollout_prefix (the streaming shape: exclusive prefixes written as the rollout goes, exact suffix
 * sums for the horizon's last steps).  (1 was round 2's mppi_rollout_cost_reg, removed in round 4: superseded by the prefix form.) */
int tbnav_mppi_streaming_form(const tbnav_mppi* h);

/* Options (explicit setters; nothing in the library reads the environment).
 *  TBNAV_MPPI_OPT_KERNEL   which rollout kernel the handle launches instead of the automatic choice: 0 = mppi_rollout_cost,
 *                          n > 0 = mppi_rollout_scan with n steps per thread, -4 / -8 / -16 = mppi_rollout_fused with that
 *                          many rollouts per workgroup (A/B measurements, variant parity tests).
 *  TBNAV_MPPI_OPT_TRIG     sincos evaluations per RK4 step: 1 (default: angle addition, a fresh sincos every 4th step),
 *                          2 (fresh every step), 3 (the reference's three evaluations).
 *  TBNAV_MPPI_OPT_NO_LDS_STAGING  1 = mppi_rollout_cost stages every per-step loss through J (development).
 *  TBNAV_MPPI_OPT_KEEP_J   1 = the fused kernel also stores the cost-to-go J[T][K] (410 KB at K=1024, T=50) so that
 *                          tbnav_mppi_get_cost_to_go can return it; off by default — the update needs only the records.
 *  TBNAV_MPPI_OPT_REG_TAIL (retired: the kernel it switched was removed in round 4; accepted and ignored.)
 *  TBNAV_MPPI_OPT_PREFIX_FORM 0 = do not use mppi_rollout_prefix (exclusive prefixes of the losses to J as the rollout goes, exact
 *                          suffix sums only for the horizon's last steps, J = total - prefix formed by the consumers) even where
 *                          it applies — the large-K default; mppi_rollout_cost is the general fallback. */
enum { TBNAV_MPPI_OPT_KERNEL = 1, TBNAV_MPPI_OPT_TRIG = 2, TBNAV_MPPI_OPT_NO_LDS_STAGING = 3, TBNAV_MPPI_OPT_KEEP_J = 4, TBNAV_MPPI_OPT_REG_TAIL = 5,
       TBNAV_MPPI_OPT_BATCH_GRAPH = 6 /* 0: tbnav_mppi_enqueue_rng_batch launches every tick by itself instead of replaying captured hipGraphs (chunks of 100 ticks; one of the batch's own length for 8-99) */,
       TBNAV_MPPI_OPT_PREFIX_FORM = 7,
       TBNAV_MPPI_OPT_DIRECT_EXCHANGE = 8 /* 0: a handle attached to a multi-process communicator always exchanges through the communicator's
                                             all-gather (default 1: directly into the peers' buffers when every rank can; takes effect at the next attach; 2: as 1 with a fault injected for the tests of the
                                             bound — after the self-test this rank's records never reach its peers, and the bound is 0.3 s instead of 2 s) */,
       TBNAV_MPPI_OPT_SAMPLER = 9 /* the device noise source's width (replacement of utilities.cpp:20-24 / mppi.cpp:173-184; the Philox counters are the same either way):
                                     1 (default since round 6): fp64 Box-Muller on 52-bit uniforms — what std::normal_distribution<double> is in width, out to
                                        8.57 sigma (drawn inside the fused kernel wherever the fp32 one is);
                                     0: fp32 Box-Muller on 24-bit uniforms — normals on a 2^-24 grid out to 5.9 sigma (about 2 % faster at K = 1024) */,
       TBNAV_MPPI_OPT_WIDE_COMBINE = 11 /* 1 (default): a single-GPU tick whose time steps have more than 256 soft-min records (the fused kernel at K = 4097 ... 8192) combines
                                            them with four waves per step (mppi_combine_wide); 0: always one wave per step (A-B measurements) */,
       TBNAV_MPPI_OPT_FAULT_INJECT = 10 /* tests: 1 = the local half of this handle's next sharded tick reports a failure (its rollouts are not launched) */ };
int tbnav_mppi_set_option(tbnav_mppi* h, int32_t option, int32_t value);

/* Rollout dynamics.  TBNAV_MPPI_DYN_RK4 (default) is the reference MPPI: CartModel + RK4 (controller/include/
 * controller/mppi.hpp:41-48, controller/src/controller/rk4.cpp:95-115).  TBNAV_MPPI_DYN_ARC is an OPTION the
 * reference's controller does not have: every rollout step is the plant's own update,
 * rigid2d::DiffDrive::feedforward(wheelsToTwist(u) * dt) — exact arcs through Transform2D::integrateTwist, heading
 * normalised to (-pi, pi] each step (rigid2d/src/rigid2d/diff_drive.cpp:79-94,153-195, rigid2d.cpp:239-303).  It
 * changes the controller's results (model == plant, no RK4 truncation), so it has its own oracle, pinned against
 * the reference's DiffDrive class. */
#define TBNAV_MPPI_DYN_RK4 0
#define TBNAV_MPPI_DYN_ARC 1
int tbnav_mppi_set_dynamics(tbnav_mppi* h, int32_t model);

/* ---- controller state ------------------------------------------------------------------------ */

/* MPPI::setInitialControls (mppi.cpp:54-61): uinit = (uL,uR) and every column of u = uinit. */
int tbnav_mppi_set_initial_controls(tbnav_mppi* h, double uL, double uR);
/* MPPI::setWaypoint (mppi.cpp:64-69): xd = (x, y, theta). */
int tbnav_mppi_set_waypoint(tbnav_mppi* h, double x, double y, double theta);
/* Read / overwrite the warm-start control matrix u, host buffer of 2*T doubles laid out [2][T]. */
int tbnav_mppi_get_controls(tbnav_mppi* h, double* u_host);
int tbnav_mppi_set_controls(tbnav_mppi* h, const double* u_host);

/* ---- one control tick: MPPI::newControls (mppi.cpp:72-140) ---------------------------------- */

/* Noise supplied by the caller on the HOST in the reference's draw order
 * (mppi.cpp:81-89,173-184): noise[(k*T + i)*2 + c], c=0 left, c=1 right, already scaled by
 * sqrt(var).  Uploads, transposes on the device, runs the tick, returns u(:,0) before the shift. */
int tbnav_mppi_new_controls(tbnav_mppi* h, const double x0[3], const double* noise_host,
                            double u_out[2]);

/* Noise already resident in HBM in the device layout (duL[T][K], duR[T][K]).  `stream` is a
 * hipStream_t (NULL = default stream).  Synchronous: waits for the tick and returns u(:,0). */
int tbnav_mppi_new_controls_dev(tbnav_mppi* h, const double x0[3], const double* d_duL,
                                const double* d_duR, void* stream, double u_out[2]);

/* Same tick, enqueue only (no host wait).  u(:,0) of the most recent tick is fetched with
 * tbnav_mppi_last_controls, which synchronises `stream`. */
int tbnav_mppi_enqueue_dev(tbnav_mppi* h, const double x0[3], const double* d_duL,
                           const double* d_duR, void* stream);
int tbnav_mppi_last_controls(tbnav_mppi* h, void* stream, double u_out[2]);

/* Noise drawn ON the device (production mode; replaces MPPI::pertubations, mppi.cpp:173-184):
 * Philox4x32-10 keyed by `seed`, counter = tick*K*T + k*T + i (see tbnav_mppi_set_rng_shard for sharded ensembles), one
 * Box-Muller pair per (k,i) — evaluated on the fp32 transcendental units: normals on a 2^-24 grid out to 5.9 sigma —
 * gives (duL, duR) scaled by sqrt(ul_var), sqrt(ur_var).  Fills the handle's own duL/duR buffers, which
 * tbnav_mppi_*_dev accept when d_duL == d_duR == NULL. */
int tbnav_mppi_sample_noise(tbnav_mppi* h, uint64_t seed, uint64_t tick, void* stream);
/* Sharded ensembles: this handle's K rollouts are rollouts [first_rollout, first_rollout + K) of rollouts_global.  The
 * device noise counter becomes tick*T*rollouts_global + (first_rollout + k)*T + i, so ranks that share a seed draw
 * DISJOINT perturbations — the same ones the unsharded ensemble of rollouts_global rollouts would draw.  Without this
 * call a handle numbers its rollouts from 0 and ranks sharing a seed would all draw the same K perturbations. */
int tbnav_mppi_set_rng_shard(tbnav_mppi* h, uint64_t first_rollout, uint64_t rollouts_global);
/* Copy the handle's own noise buffers to the host ([T][K] each) — for statistical tests. */
int tbnav_mppi_get_noise(tbnav_mppi* h, double* duL_host, double* duR_host);

/* ---- sharded tick (multi-GPU: rollouts split across ranks; mppi.cpp:81-126 split in two) ----- */

/* Number of K-slices this handle cuts its own rollouts into (records per time step it emits). */
int tbnav_mppi_records_per_step(const tbnav_mppi* h);

/* Rollouts + per-time-step partial records of THIS shard.  d_records_out is a device buffer of
 * T * records_per_step * TBNAV_MPPI_REC doubles laid out [T][S][REC].  Enqueue only. */
int tbnav_mppi_shard_partials(tbnav_mppi* h, const double x0[3], const double* d_duL,
                              const double* d_duR, void* stream, double* d_records_out);

/* Same with the shard's perturbations drawn on the device (production mode; call tbnav_mppi_set_rng_shard once so that
 * the shards of one ensemble draw disjoint perturbations from one seed). */
int tbnav_mppi_shard_partials_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream,
                                  double* d_records_out);

/* Combine `n_shards` record sets (device buffer [n_shards][T][S][REC], e.g. the output of one
 * all-gather), update u, clamp, emit u(:,0), shift.  Every rank runs this on the same input and so
 * holds the same u afterwards.  Enqueue only; fetch with tbnav_mppi_last_controls. */
int tbnav_mppi_shard_combine(tbnav_mppi* h, const double* d_records_all, int32_t n_shards,
                             void* stream);

/* ---- the sharded tick behind the ordinary entry points (include/tbnav_comm.h) -------------------------------------
 * One process per GPU: after tbnav_mppi_attach_comm(h, comm) — comm on the handle's device; params.rollouts is this
 * rank's share, every rank the same — EVERY tick entry point of the handle (tbnav_mppi_new_controls*, _enqueue_dev,
 * _enqueue_rng, _enqueue_rng_batch) runs rollouts + records of its shard -> ONE ncclAllGather of the records ->
 * the combine of all shards' records, all enqueued on the tick's stream: no host round trip inside a tick, and every
 * rank ends with the same warm start (mppi.cpp:112-137 over the whole ensemble).  The device noise counter space is
 * set for rank * K ... of nranks * K (tbnav_mppi_set_rng_shard).  comm = NULL detaches.  The communicator must outlive
 * the handle's last tick; the handle does not own it.
 * Failures.  A rank whose own rollouts fail still JOINS the tick's all-gather, with records that say so; a combine that meets
 * such a record leaves that time step's controls as they were (shifted, not updated) and latches the error on its rank — so every
 * rank returns it from its next enqueue / last_controls / synchronize, and all warm starts stay identical.  A latched rank keeps
 * joining the all-gather on every further tick (round 6: the latch is a word each rank sees at its own time, so no rank may stay
 * away from a collective its peers could already be in) and keeps returning the error; detach and re-attach to go on.  The direct
 * exchange (kind 2) has a bound instead: a rank that fails publishes nothing and latches itself, its peers latch when the bound
 * expires. */
struct tbnav_comm;
int tbnav_mppi_attach_comm(tbnav_mppi* h, struct tbnav_comm* comm);
/* How an attached handle's ticks exchange their records: 0 = no communicator attached; 1 = the communicator's all-gather (RCCL; copies
 * inside a one-process group); 2 = DIRECT: ranks in separate processes of one node store their records straight into every peer's
 * gather buffer (hipIpcMemHandle mappings of fine-grained memory over xGMI; self-validating 8-byte words, the receiver polls its own
 * buffer with a bound) — an RCCL all-gather of a few KB costs several 9 us ticks.  Chosen at attach, collectively: every rank maps
 * every rank's buffer and runs a self-test through the tick's own kernels; if any rank cannot, all ranks use 1. */
int tbnav_mppi_exchange_kind(const tbnav_mppi* h);
/* `rounds` exchanges of the handle's record block ([T][S][8] doubles per rank) and nothing else — through the direct path if that is on,
 * else the communicator's all-gather — timed with HIP events on `stream`: microseconds per exchange on the node at hand.  Collective:
 * every rank of the (multi-process) communicator calls it with the same count.  Between ticks only (it overwrites the gather buffer). */
int tbnav_mppi_exchange_probe(tbnav_mppi* h, int32_t rounds, void* stream, double* us_per_round);

/* One process driving n_gpus devices (what controller::MPPI(..., n_gpus) holds: a ROS node is one process): the
 * ensemble of params->rollouts rollouts (a multiple of n_gpus) split evenly over devices[0..n_gpus) (NULL: 0, 1, ...;
 * a device may repeat — the members sharing it then exchange by copies instead of RCCL, see tbnav_comm.h).  Each member
 * is a tbnav_mppi handle with its own stream; a tick enqueues every member's rollouts, ONE grouped all-gather, every
 * member's combine.  Setters apply to every member; results are member 0's (all members hold the same controls). */
typedef struct tbnav_mppi_group tbnav_mppi_group;
int tbnav_mppi_group_create(const tbnav_mppi_params* params, int32_t n_gpus, const int32_t* devices, tbnav_mppi_group** out);
void tbnav_mppi_group_destroy(tbnav_mppi_group* g);
int tbnav_mppi_group_size(const tbnav_mppi_group* g);
int tbnav_mppi_group_member(tbnav_mppi_group* g, int32_t rank, tbnav_mppi** out);  /* borrowed: parity hooks of one shard */
int tbnav_mppi_group_set_waypoint(tbnav_mppi_group* g, double x, double y, double theta);
int tbnav_mppi_group_set_initial_controls(tbnav_mppi_group* g, double uL, double uR);
int tbnav_mppi_group_set_controls(tbnav_mppi_group* g, const double* u_host);
int tbnav_mppi_group_get_controls(tbnav_mppi_group* g, double* u_host);
int tbnav_mppi_group_set_dynamics(tbnav_mppi_group* g, int32_t model);
int tbnav_mppi_group_set_option(tbnav_mppi_group* g, int32_t option, int32_t value);
/* MPPI::newControls over the whole ensemble.  noise_host: the ENSEMBLE's perturbations in the reference's draw order,
 * noise[(k*T + i)*2 + c] for k < params->rollouts (member r takes rollouts [r*K/n, (r+1)*K/n)). */
int tbnav_mppi_group_new_controls(tbnav_mppi_group* g, const double x0[3], const double* noise_host, double u_out[2]);
/* Production ticks, perturbations drawn on the devices (each member in its slice of the ensemble's counter space). */
int tbnav_mppi_group_new_controls_rng(tbnav_mppi_group* g, const double x0[3], uint64_t seed, uint64_t tick, double u_out[2]);
int tbnav_mppi_group_enqueue_rng(tbnav_mppi_group* g, const double x0[3], uint64_t seed, uint64_t tick);
int tbnav_mppi_group_enqueue_rng_batch(tbnav_mppi_group* g, const double* x0s, int32_t x0_stride, uint64_t seed, uint64_t first_tick,
                                       int32_t n_ticks);
int tbnav_mppi_group_last_controls(tbnav_mppi_group* g, double u_out[2]);
int tbnav_mppi_group_synchronize(tbnav_mppi_group* g);

/* ---- parity / debug hooks --------------------------------------------------------------------- */

/* Cost-to-go J of the last tick BEFORE the per-step min subtraction, host buffer [T][K]
 * (reference J(i,k) after cumSumCost, mppi.cpp:109). */
int tbnav_mppi_get_cost_to_go(tbnav_mppi* h, double* J_host);

/* The rollout kernel's own sin/cos (Cody-Waite reduction + fdlibm kernel polynomials; ocml's sincos
 * beyond |x| = 1e5) evaluated on n host values — lets the tests bound its error against libm. */
int tbnav_mppi_debug_sincos(const double* x_host, int32_t n, double* sin_host, double* cos_host);

/* The soft-min kernels' x / lambda (three dependent instructions from lambda's correctly rounded reciprocal instead of the
 * division's own iterations) evaluated on n host values — lets the tests hold it against the IEEE division bit for bit.
 * *used_reciprocal = 0 when this lambda takes the plain division (reciprocal not normal, or lambda's significand all ones). */
int tbnav_mppi_debug_div_lambda(const double* x_host, int32_t n, double lambda, double* out_host, int32_t* used_reciprocal);

/* ---- measurement hook -------------------------------------------------------------------------- */

/* One tick with a hipEvent pair around each kernel, recorded on `stream` (the stream the kernels
 * run on); waits, then returns the three durations in milliseconds:
 *   ms[0] mppi_rollout_cost, ms[1] mppi_partials, ms[2] mppi_combine.
 * Same launches and same state update as tbnav_mppi_enqueue_dev; bench.py averages it over the
 * timed number of steps to price the dominant kernel against the HBM roofline. */
#define TBNAV_MPPI_NKERNELS 3
int tbnav_mppi_profile_tick(tbnav_mppi* h, const double x0[3], const double* d_duL,
                            const double* d_duR, void* stream, float ms[TBNAV_MPPI_NKERNELS]);

/* Production tick with the perturbations drawn on the device (Philox4x32-10 + Box-Muller, the values
 * tbnav_mppi_sample_noise(h, seed, tick) would write — MPPI::pertubations, mppi.cpp:173-184, in production mode): for
 * the fused small-K kernel they are generated inside it and never touch HBM; other configurations sample into the
 * handle's buffers first.  Either way the result equals tbnav_mppi_sample_noise + tbnav_mppi_new_controls_dev(NULL
 * noise) bit for bit.  _enqueue_ is asynchronous on `stream`, _new_controls_ waits and returns (ul, ur). */
int tbnav_mppi_enqueue_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream);
int tbnav_mppi_new_controls_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream,
                                double u_out[2]);
/* n_ticks production ticks in a row, enqueued from C: tick i starts from x0s + i * x0_stride (doubles; x0_stride = 0: the
 * same state every tick — replay / throughput runs, the warm start is carried from tick to tick as always) with the
 * perturbations of (seed, first_tick + i).  Exactly n_ticks tbnav_mppi_enqueue_rng calls, without a trip through the
 * caller's language per tick (from Python one enqueue costs as much as the tick takes on the device).  With one state for
 * all ticks (x0_stride = 0) on a non-default stream and the fused kernel, whole chunks of 100 ticks are replayed from a captured
 * hipGraph (same kernels, same arguments but for the tick number, which the kernel then reads from device memory): same result.
 * What is left of a batch, or a batch of 8 to 99 ticks, is replayed from ONE graph of its own length (less one tick if odd) once
 * a second batch in a row asks for the same length: a caller that synchronises every 20 ticks submits one graph per block. */
int tbnav_mppi_enqueue_rng_batch(tbnav_mppi* h, const double* x0s, int32_t x0_stride, uint64_t seed, uint64_t first_tick,
                                 int32_t n_ticks, void* stream);

/* How many of the ticks enqueued through tbnav_mppi_enqueue_rng_batch so far went out as replays of a captured graph (the rest
 * were plain launches): lets a benchmark line say what actually ran. */
int64_t tbnav_mppi_graph_replayed_ticks(const tbnav_mppi* h);

/* Per-kernel durations priced without the events' own cost: each kernel of the tick is launched `reps` (even, >= 2)
 * times back to back between one event pair; ms[i] = elapsed / reps (ms[1] = 0 when rollout and partials are one
 * kernel).  The controller state advances as if `reps` ticks had run on the same inputs. */
int tbnav_mppi_profile_kernels(tbnav_mppi* h, const double x0[3], const double* d_duL, const double* d_duR, void* stream,
                               int32_t reps, float ms[TBNAV_MPPI_NKERNELS]);

/* The same for the PRODUCTION tick (tbnav_mppi_enqueue_rng): where that tick draws its perturbations inside the fused kernel,
 * the kernel launched here is that very instantiation (in-kernel Philox, nothing read from the noise buffers), with the
 * perturbations of (seed, tick + r) in repetition r.  Other configurations: sample once into the handle's buffers, then
 * tbnav_mppi_profile_kernels on them (what their production tick runs). */
int tbnav_mppi_profile_kernels_rng(tbnav_mppi* h, const double x0[3], uint64_t seed, uint64_t tick, void* stream, int32_t reps,
                                   float ms[TBNAV_MPPI_NKERNELS]);

/* Names of the kernel instantiations the handle's LAST rollout and combine launches were, spelled as a profiler prints
 * them ("mppi_rollout_fused<2, 8, 1, true>", "mppi_combine<2, false>"): lets a benchmark line point at one row of a
 * rocprofv3 --kernel-trace --stats summary.  Empty strings before the first launch. */
int tbnav_mppi_last_kernel_names(const tbnav_mppi* h, char* rollout, int32_t rollout_cap, char* combine, int32_t combine_cap);

#ifdef __cplusplus
}
#endif
#endif /* TBNAV_MPPI_H */
