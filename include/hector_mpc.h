/*
 * hector_mpc.h -- C ABI of libhector_mpc_hip.so: MI355X (gfx950) batched force-and-moment MPC QP solver for
 * HECTOR's convex-MPC loop.  Plain pointers and sizes only; no torch / Eigen / C++ types cross this boundary.
 *
 * Two surfaces:
 *  (1) the reference's own interface, byte-for-byte the same signatures, so hector_control links against this
 *      library instead of its ConvexMPC/convexMPC_interface.cpp + SolverMPC.cpp + qpOASES and nothing else changes:
 *        setup_problem / update_problem_data / get_solution / update_solver_settings
 *            -> Hector_ROS_Simulation/hector_control/ConvexMPC/convexMPC_interface.h:39-43
 *        solve_mpc(update_data_t*, problem_setup*), get_q_soln()   (C++ linkage in the reference)
 *            -> ConvexMPC/SolverMPC.h:56,63    (exported here additionally as C symbols hmpc_solve_mpc / hmpc_get_q_soln
 *               and, under the north-star name, solveDenseMPC)
 *        problem_setup, update_data_t PODs -> ConvexMPC/convexMPC_interface.h:11-37
 *  (2) a handle-based, re-entrant BATCHED interface (names are ours) over packed records
 *      (hector_simulation_amd/records.py documents the layout; hmpc_pack_record builds one from the reference's
 *      argument list).
 *
 * Every function returns 0 on success or a negative hmpc_error; nothing throws across this boundary
 * (the reference throws std::runtime_error for horizon > 19, SolverMPC.cpp:140-143: here setup returns/records
 * HMPC_E_HORIZON and get_solution keeps returning the previous values).
 */
#ifndef HECTOR_MPC_H
#define HECTOR_MPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define K_MAX_GAIT_SEGMENTS 36 /* convexMPC_interface.h:3 */
#define HMPC_MAX_HORIZON 20    /* device scratch is sized for this (reference: 10 hard-coded, cap 19) */
#define HMPC_MAX_VARS 120      /* reduced QP variables (6 per stance leg-step) the fast on-chip variants hold (two contacts; 180 with three) */
#define HMPC_MAX_VARS_WIDE 240 /* ... the wide variant (double support over h = 11 .. 20): picked when the batch needs it -- from the gait
                                  tables of host-uploaded records, hmpc_set_max_reduced_vars, or per instance on the device */

/* ---- reference PODs (convexMPC_interface.h:11-37), same field order and types ---- */
struct problem_setup {
  float dt;
  float mu; /* ignored by the reference (mu = 2.0 is hard-coded, SolverMPC.cpp:488); ignored here too */
  float f_max;
  int horizon;
};

struct update_data_t {
  float p[3];
  float v[3];
  float q[4];
  float w[3];
  float r[6];
  float joint_angles[10];
  float yaw;
  float weights[12];
  float traj[12 * K_MAX_GAIT_SEGMENTS];
  float Alpha_K[12];
  unsigned char gait[K_MAX_GAIT_SEGMENTS];
  unsigned char hack_pad[1000];
  int max_iterations;
  double rho, sigma, solver_alpha, terminate;
};

/* ---- (1) reference interface, convexMPC_interface.h:39-43 ---- */
void setup_problem(double dt, int horizon, double mu, double f_max);
void update_problem_data(double *p, double *v, double *q, double *w, double *r, double *joint_angles, double yaw,
                         double *weights, double *state_trajectory, double *Alpha_K, int *gait);
double get_solution(int index);
void update_solver_settings(int max_iter, double rho, double sigma, double solver_alpha, double terminate,
                            double use_jcqp);
/* SolverMPC.h:56,63 under C names (the C++-linkage solve_mpc/get_q_soln are exported too, see hmpc_capi) */
void hmpc_solve_mpc(struct update_data_t *update, struct problem_setup *setup);
void solveDenseMPC(struct update_data_t *update, struct problem_setup *setup);
double *hmpc_get_q_soln(void);
/* extra: status word of the last legacy solve (see HMPC_STATUS_*), never part of the reference.
 * The process-global solver behind the reference interface runs on the device named by the environment variable
 * HMPC_DEVICE (default 0).  On a flagged solve it prints the reference's "failed to solve!" line and, like the
 * reference (SolverMPC.cpp:714-732), still scatters the last iterate; HMPC_S_OK_RELAXED counts as solved. */
uint32_t hmpc_last_status(void);

/* ---- (2) batched interface ---- */
typedef struct hmpc_handle hmpc_handle;

enum hmpc_error {
  HMPC_OK = 0,
  HMPC_E_ARG = -1,      /* null pointer / bad size */
  HMPC_E_HORIZON = -2,  /* horizon < 1 or > HMPC_MAX_HORIZON */
  HMPC_E_BATCH = -3,    /* batch > max_batch */
  HMPC_E_HIP = -4,      /* HIP runtime error (hmpc_last_hip_error) */
  HMPC_E_NO_DEVICE = -5 /* no gfx950 device visible: the library has NO CPU fallback */
};

/* per-instance status word written by the kernel: bits 0-7 code, 8-19 active-set iterations, 20-31 final |W| */
#define HMPC_STATUS_CODE(s) ((s) & 0xffu)
#define HMPC_STATUS_ITERS(s) (((s) >> 8) & 0xfffu)
#define HMPC_STATUS_NACTIVE(s) (((s) >> 20) & 0xfffu)
enum hmpc_status_code {
  HMPC_S_OK = 0,
  HMPC_S_MAXITER = 1,     /* iteration cap hit (reference analogue: nWSR = 500 exhausted) */
  HMPC_S_INFEASIBLE = 2,  /* constraints inconsistent */
  HMPC_S_TOO_LARGE = 3,   /* more reduced variables than the variant the batch was launched with holds (only when
                             hmpc_set_max_reduced_vars named a smaller size than the batch really contains) */
  HMPC_S_KKT = 4,         /* final KKT check outside tolerance (rows outside the working set, the residual of the rows in it,
                             multiplier signs; relative to the force scale) */
  HMPC_S_WORKSET = 5,     /* more simultaneously active constraints than the FAST variant's on-chip working set holds (64 rows; 96
                             with three contacts, 152 in the wide variant).  The safe pass holds as many rows as there are
                             variables -- in LDS for 120 variables, in global memory for 180 / 240 -- and cannot overflow */
  HMPC_S_SWEEP_MISMATCH = 7, /* hmpc_solve_command_sweep: the record differs from its chunk's first record outside the trajectory; not solved */
  HMPC_S_INDEFINITE = 8,  /* the reduced Hessian, assembled in binary32 as the reference assembles it (SolverMPC.cpp:560-570), is NOT
                             positive definite: a sweep pivot of the safe pass came out <= 0 (seen with 20-step horizons at 10x the
                             nominal input ranges: rounding at 6e-8 |H| against a smallest eigenvalue of ~2 alpha).  An intermediate
                             state: the safe pass -- hmpc_resolve_failed / hmpc_download, or (horizons > 10) the device-side chain of
                             hmpc_set_device_repair(h, 1) -- answers it the way the reference's qpOASES run does (H + rho I, then one
                             step with g - rho x_1: QProblem.cpp:1753-1860) and reports HMPC_S_OK; seen by a caller only where those
                             two QPs fail as well (forces zeroed) */
  HMPC_S_REG_STEP = 9,    /* (internal to hmpc_resolve_failed: between the two regularised QPs of an HMPC_S_INDEFINITE instance) */
  HMPC_S_OK_RELAXED = 6   /* solved only after every bound was moved outward by <= 2e-5 (relative for the Fz cap) AND the exact
                             re-solve on the working set so found did not pass the exact KKT check: the last-resort pass of
                             hmpc_resolve_failed for instances cycling at a degenerate vertex.  (When the exact re-solve passes --
                             every case seen up to 10x the nominal input ranges -- the instance is HMPC_S_OK, exact.) */
};

size_t hmpc_record_stride(int horizon);  /* bytes per packed record: (54+12h)*4 + 2h rounded up to 16 */
/* packs one record from the reference's update_problem_data argument list (double -> float, int -> u8 narrowing
 * exactly as convexMPC_interface.cpp:83-103) */
int hmpc_pack_record(void *record, int horizon, const double *p, const double *v, const double *q, const double *w,
                     const double *r, const double *joint_angles, double yaw, const double *weights,
                     const double *state_trajectory, const double *Alpha_K, const int *gait);

int hmpc_create(hmpc_handle **out, const struct problem_setup *setup, int max_batch, int device);

/* Robot and contact constants of the formulation.  The reference hard-codes every one of them in its solver source (and ignores
 * problem_setup.mu): a payload, friction or foot-geometry sweep needs a recompile there.  Here they are data of the handle; the
 * defaults are the reference's literals, and a handle that was never given anything else assembles bit for bit the QP it
 * assembled before the struct existed (tests/test_gpu_assembly.py).  The CPU oracle takes the same struct (orc_set_params), so
 * non-default values are checked bitwise as well. */
struct hmpc_params {
  float mass;       /* 9.0                      SolverMPC.cpp:423  (B_ct: v' += F / mass) */
  float inertia[3]; /* 0.5413, 0.5200, 0.0691   RobotState.cpp:45  (body inertia, diagonal) */
  float mu;         /* 2.0                      SolverMPC.cpp:488  (friction pyramid rows 0-3 are (-+mu, 0, 1) F >= 0, i.e. |F_t| <= F_z / mu: the
                                                 reference's 2.0 is a friction coefficient of 0.5 -- convention kept; problem_setup.mu is ignored, as there) */
  float lt, lh;     /* 0.09, 0.06               SolverMPC.cpp:489-490 (toe / heel lever arms of the line-contact rows 5, 6) */
  float gravity;    /* 9.81                     SolverMPC.cpp:420  (the constant 13th state) */
};
void hmpc_default_params(struct hmpc_params *p);
/* takes effect with the next solve of the handle; p == NULL restores the defaults.  mass, inertia, mu > 0 and finite, else HMPC_E_ARG */
int hmpc_set_params(hmpc_handle *h, const struct hmpc_params *p);
int hmpc_get_params(const hmpc_handle *h, struct hmpc_params *p);
/* TERRAIN SWEEPS: a friction parameter per INSTANCE -- device_mu[batch] floats in HBM, caller-owned, read by every later solve of the
 * handle in place of hmpc_params.mu until it is set to NULL again (instance i of the current batch uses device_mu[i]; values must
 * be > 0).  H and its inverse do not depend on mu -- only the friction rows (-+mu, 0, 1) of the constraint block do -- so the
 * records of one hmpc_solve_command_sweep group may differ in their mu as well as in their trajectory: one state under many commands
 * on many floors, one inverse. */
int hmpc_set_instance_mu(hmpc_handle *h, const float *device_mu);
/* ... for the process-global solver behind the reference interface (applies from the next setup_problem / solve on) */
int hmpc_legacy_set_params(const struct hmpc_params *p);

/* ---- extension beyond the reference: a third (hand) contact per horizon step -- BASELINE.json config 5, the
 * loco-manipulation shape 180 variables x 240 rows at h = 10.  The reference has no code for it;
 * the formulation is the reference's own with one more contact: B_ct gains the hand's force / moment columns
 * (SolverMPC.cpp:312-331 with a third r), the hand gets the left foot's 8-row block (SolverMPC.cpp:488-548) expressed
 * in its contact frame Rhand (body frame, row-major) with its own force cap, gait gets a third flag per step.
 * Component order within a step: [F_left F_right F_hand M_left M_right M_hand]; r[3*axis + contact]; Alpha_K has 18
 * entries in that order; gait[3*step + contact].  h <= 10.  Rows f1-f3 (tick builder, wrench, torques) are two-foot only. */
#define HMPC_MAX_VARS_3C 180
int hmpc_create_ex(hmpc_handle **out, const struct problem_setup *setup, int max_batch, int device, int n_contacts);
int hmpc_contacts(const hmpc_handle *h);
size_t hmpc_record_stride_ex(int horizon, int n_contacts); /* n_contacts 3: (73+12h)*4 + 3h rounded up to 16 */
int hmpc_pack_record_ex(void *record, int horizon, int n_contacts, const double *p, const double *v, const double *q,
                        const double *w, const double *r, const double *joint_angles, double yaw, const double *weights,
                        const double *state_trajectory, const double *Alpha_K, const int *gait, const double *Rhand,
                        double f_max_hand);
int hmpc_destroy(hmpc_handle *h);
/* host records -> device (synchronous copy on the handle's stream) */
int hmpc_upload_records(hmpc_handle *h, const void *host_records, int batch);
/* use records that already live in HBM (no copy; pointer must stay valid until the solve finishes) */
int hmpc_set_device_records(hmpc_handle *h, const void *device_records, int batch);
/* optional: write forces/status into caller-owned device buffers instead of the handle's own */
int hmpc_set_device_outputs(hmpc_handle *h, float *device_forces, uint32_t *device_status);
/* asynchronous: enqueue assembly+solve of the current batch on `stream` (a hipStream_t, NULL = default stream) */
int hmpc_solve(hmpc_handle *h, void *stream);
/* waits for `stream` work, copies forces [batch][12h] float and status [batch] to host (either may be NULL) */
int hmpc_download(hmpc_handle *h, float *forces, uint32_t *status);
/* COMMAND SWEEPS (round 6): the current batch is B = G x group_size records in G groups of group_size CONSECUTIVE records that
 * share the robot state, foot positions, joint angles, weights and gait table and differ in the reference trajectory only -- one
 * state under many commands: ConvexMPCLocomotion.cpp:351-406 builds state_trajectory from the velocity / yaw-rate commands, while
 * A_qp, B_qp, H = 2(B'SB + alpha) and the constraint block depend on the state and the gait alone (SolverMPC.cpp:398-447, 488-570),
 * so only g = 2 B'S (A_qp x0 - X_d) changes inside a group and M = H^-1 is a property of the GROUP.  Two launches on `stream`:
 * one workgroup per group forms M once and leaves it in HBM (74 KB per group; a per-handle buffer grown on demand), then one
 * workgroup per instance -- the chip as full as for independent solves -- assembles its own g, takes M from its group's slot
 * instead of assembling and inverting H (55 % of an independent solve) and runs the block start and the active-set iteration as
 * they stand: forces and status words are BIT-IDENTICAL to hmpc_solve's.  A record that differs from its group's first record
 * anywhere but in the trajectory is not solved: status HMPC_S_SWEEP_MISMATCH, forces 0 (checked on the device, word by word).
 * Flagged instances are repaired as independent ones (hmpc_download / hmpc_set_device_repair).  Two-contact handles, horizon <= 10;
 * batch % group_size must be 0; group_size 1 is hmpc_solve.  A per-call CPU solver has no counterpart: the reference assembles and
 * factorises H for every command. */
int hmpc_solve_command_sweep(hmpc_handle *h, int group_size, void *stream);
/* hint for device-resident records: the widest reduced QP (6 x stance leg-steps) in the batch: one launch of the variant
 * that holds it.  -1 = unknown (the state after hmpc_set_device_records / hmpc_build_records_device): two-contact handles
 * then route every instance on the device -- its stance leg-steps are counted there (by the record builder, or from the
 * records' gait bytes at the head of the solve) and each variant of the family (60 / 120 / at h > 10 also 240 variables)
 * is launched over the whole batch, a workgroup leaving at once when its instance belongs to another variant.
 * hmpc_upload_records derives the hint from the gait tables itself. */
int hmpc_set_max_reduced_vars(hmpc_handle *h, int n_reduced);
/* Working-set start of the active-set solver.  on = 1 (default): every moment / line-contact row violated at the
 * unconstrained minimiser enters at once (block warm start); on = 0: cold start from the empty set, one row per
 * iteration, as the reference's qpOASES call does.  Same optimum either way (strictly convex QP). */
int hmpc_set_warm_start(hmpc_handle *h, int on);
/* Warm start ACROSS ticks (the reference cold-starts every tick, SolverMPC.cpp:702).
 * Default off.  When on, every solve leaves each instance's final working set in HBM (one signed byte per constraint
 * row) and the next solve of the same handle starts instance k from the set instance k ended with: its Schur matrix
 * is formed and inverted at once, rows whose multiplier comes out negative are released, and the dual active-set
 * iteration continues from there.  horizon_shift = how many horizon steps the gait table advanced between the two
 * solves (rows of step i start from saved step i + shift).  Same optimum as a cold start (the QP is strictly convex);
 * only the iteration count changes.  A saved set that has become rank deficient under the new data is discarded for
 * that instance.  hmpc_reset_tick_warm_start forgets all saved sets (e.g. before an unrelated batch). */
int hmpc_set_tick_warm_start(hmpc_handle *h, int on, int horizon_shift);
int hmpc_reset_tick_warm_start(hmpc_handle *h);
/* Safe pass: waits for the last solve, then re-solves every instance whose status is working-set-full / max-iter /
 * infeasible / KKT and overwrites its forces and status in place.  The passes, each over what the one before left flagged
 * (round 6): (1) instances that handed their state over (hmpc_set_handover) are CONTINUED on the 96-row variant; (2) the
 * large-working-set variant (capacity = number of variables: cannot overflow; the Schur inverse is rebuilt from scratch every
 * 48 working-set changes, so that hundreds of them do not add up round-off), cold, with every bound moved outward by a relative
 * 1e-6 (a different amount per row: separates the coinciding vertices at which the exact problem makes Goldfarb-Idnani cycle --
 * the instances that reach this pass are the degenerate ones), ending with an exact re-solve on the working set it found:
 * HMPC_S_OK when that passes the exact KKT check -- every one of the 8 192 at 6x the nominal input ranges --, HMPC_S_OK_RELAXED
 * otherwise; (3) the same variant on the EXACT bounds, started by the block start, for relaxed and flagged instances alike;
 * (4) instances whose reduced Hessian is not positive definite (HMPC_S_INDEFINITE, found by the sweeps of (2)): the reference's
 * regularisation, two more launches -- H + rho I with rho = |H|_F sqrt(1e3 * 2.221e-16), then the same QP with the gradient
 * g - rho x_1 (qpOASES QProblem.cpp:1753-1860, QProblemB.cpp:1418-1431, 1999-2031 under Options::setToMPC) -- within 6e-8 of what
 * the reference returns for them; (5) up to three last-resort passes with the bounds moved by 1e-7, 1e-6, 1e-5, cold.  Measured at 1x .. 10x the nominal input
 * ranges (scripts/stress.py, profiles/r06/stress.txt, 28 shapes x ranges of 1 024 instances): every instance qpOASES solves ends
 * HMPC_S_OK.  On 4 096 per row (stress_4096.txt): the same up to 6x; at 10x five rows keep ONE instance flagged KKT, and two
 * instances end HMPC_S_OK_RELAXED 3e-3 / 8e-3 from qpOASES' forces (7 of 114 688; an ok-relaxed answer satisfies the bounds to 2e-5
 * but need not be close to the exact optimum where the problem is that degenerate: check for the code where that matters).
 * *n_resolved (may be NULL) =
 * how many were re-solved.  hmpc_download does this automatically unless hmpc_set_auto_resolve(h, 0). */
int hmpc_resolve_failed(hmpc_handle *h, int *n_resolved);
/* Dispatch order of the workgroups of a solve.  mode 1 (default), longest first: the instances are started in the order of
 * the active-set iterations their PREVIOUS solve took (most first; read on the device from the status words the previous
 * solve of this handle left, one small sorting launch at the head of each solve).  An MPC tick resembles the tick before it,
 * so the solves that will run longest start first and the short ones fill the last, partly occupied round of workgroup
 * slots -- what bounds small and medium batches is that tail (2 048 three-contact instances are four rounds of 512 resident
 * workgroups and one 87-iteration straggler: 1.18 M solves/s in natural order, 1.66 M with a hint one tick old).  mode 0:
 * instance b runs in workgroup b.  Results do not depend on the mode, nor on whether the assumption holds (instance i of
 * this solve = instance i of the previous one, as for hmpc_set_tick_warm_start): any status words give a valid order, a
 * stale one merely stops helping.  The first solve of a handle and the first after a change of the batch size -- no previous
 * solve to go by -- are ordered by a COST PREDICTED FROM THE RECORDS (round 5): the forward acceleration the tick asks for,
 * (v_x commanded - v_x) + 2 x mean foot x, and the body's tilt, which is what the iteration count of this QP follows
 * (correlation 0.8 on the bench's sets; hmpc_builder.h predicted_cost_bucket) -- so a cold handle, and the first tick of a
 * device-built pipeline, are ordered too.  mode 2: always by the predictor (never by a previous solve).  Batches of at
 * most 512 instances (they fit the chip's workgroup slots at once), of more than 32 768, and batches known to hold
 * single-support QPs only (<= 60 reduced variables: one or two iterations each, nothing to sort) run in natural order.
 * The reference has no counterpart (one QP per call). */
int hmpc_set_dispatch_order(hmpc_handle *h, int mode);

/* Cap on the active-set iterations of every later solve of the handle -- the analogue of the reference's nWSR = 500
 * (SolverMPC.cpp:706).  0 (default) = the kernel variant's own bound.  Block rounds and switch passes of the block start
 * count as one iteration each and always complete; the cap is tested before every single-row iteration after them.  An
 * instance that would need more ends as HMPC_S_MAXITER with its last iterate in the force buffer and is NOT re-solved by
 * the safe pass -- neither by hmpc_download's host-driven one nor by the device-side one (hmpc_set_device_repair).
 * hmpc_legacy_set_max_iterations sets the same cap for the process-global solver behind setup_problem / update_problem_data.
 * The legacy update_solver_settings(max_iter, ...) does NOT: as in the reference (convexMPC_interface.cpp:112-118 stores its
 * arguments, nothing reads them; the qpOASES path uses a fixed nWSR, SolverMPC.cpp:706) it is inert, so that a drop-in caller
 * passing a small JCQP-style max_iter still gets full solves. */
int hmpc_set_max_iterations(hmpc_handle *h, int max_iter);
int hmpc_legacy_set_max_iterations(int max_iter);
int hmpc_set_auto_resolve(hmpc_handle *h, int on);
/* Hand-over of a full working set (round 6; default on).  A fast two-contact variant (<= 120 reduced variables) whose working
 * set is full (HMPC_STATUS_NACTIVE = 64) when another row is violated no longer throws its work away: it writes its live
 * Goldfarb-Idnani state -- x, multipliers, working set, the packed Schur inverse and the register-resident H^-1 -- to the
 * instance's slot of a per-handle buffer in HBM (101 KB per instance of max_batch, allocated at the first solve; at most
 * 32 768 slots) and flags the instance HMPC_S_WORKSET as before; the safe pass -- the device-side one of hmpc_set_device_repair or
 * the host-driven one of hmpc_resolve_failed / hmpc_download -- then CONTINUES that solve on the variant whose working set cannot
 * overflow instead of re-solving it cold without the block start: same optimum (the QP is strictly convex), a fraction of the
 * iterations (the reference's qpOASES run needs up to 190 working-set changes for such an instance, SolverMPC.cpp:699-712; the
 * continuation the 20-40 that were still missing).  on = 0: the round-5 behaviour (flag, re-solve cold).  Three-contact handles
 * and the wide variant (> 120 reduced variables) always take the cold path. */
int hmpc_set_handover(hmpc_handle *h, int on);
/* Device-side safe pass (default off for a plain handle): when on, hmpc_solve enqueues, behind the fast launch and on the
 * same stream, the safe variant over the list of instances the fast launch flagged (the list and its length stay on the
 * device; workgroups beyond the length leave at once) -- device-resident outputs, hmpc_download_async and the group
 * exchange then see repaired forces/status without any host involvement.  Costs one 4-byte memset and one (normally
 * empty) extra launch per solve; repairs every flagged instance of batches up to 65 536 (the wide variant's instances, whose safe
 * pass keeps 231 KB of global scratch per workgroup: the first 4 096 positions of the flagged list, 0.95 GB of scratch allocated with
 * the first such solve), the rest stay flagged for
 * hmpc_resolve_failed / hmpc_download.  Instances whose working set merely outgrew the fast variant are CONTINUED, not
 * re-solved (hmpc_set_handover): first the continuation variant over the flagged list (96-row working set, two workgroups per
 * CU), then the safe variant over what is still flagged: cold, bounds moved outward by a relative 1e-6 and an exact re-solve
 * on the working set found (pass (2) of hmpc_resolve_failed: HMPC_S_OK, or HMPC_S_OK_RELAXED when the exact KKT check fails
 * on that set: a solved instance whose bounds were off by <= 2e-6 relative; not seen on any measured set), and -- for horizons beyond
 * 10 steps, where such Hessians occur -- two more short launches for the instances whose Hessian is not positive definite (pass (4)
 * of hmpc_resolve_failed; up to 256 per solve, the rest and any such instance of a shorter-horizon handle stay HMPC_S_INDEFINITE for
 * hmpc_resolve_failed).  Every launch of the chain is trimmed on the device by a counter: with nothing flagged their workgroups leave
 * at once, and a nominal solve pays 11-15 us of dispatch latency for the chain (scripts/dev/chain_overhead.py).  on = 2: the continuation pass only -- what it does not finish
 * (~0.5 % of the instances at 6x the nominal input ranges: degenerate vertices, working sets beyond 96 rows) stays FLAGGED in
 * the status word for hmpc_resolve_failed / hmpc_download; a cold re-solve of such an instance takes hundreds of iterations on a
 * single workgroup, milliseconds at the tail of the stream, which a device-resident pipeline may prefer not to wait for. */
int hmpc_set_device_repair(hmpc_handle *h, int on);
/* NOTE (device repair): the list of flagged instances and its counter belong to the handle -- keep the solves of ONE handle
 * on one stream at a time (use one handle per stream to overlap launches, as bench.py does). */
/* Stream-ordered variants for pipelining host batches (two handles on two streams: the copies of one overlap the solve
 * of the other).  The host buffers should be pinned (hipHostMalloc / hipHostRegister) for the copies to be asynchronous
 * and must stay valid until the stream reaches them.  hmpc_download_async does not run the safe pass: check the status
 * words after synchronising and call hmpc_resolve_failed + hmpc_download for a batch that has flagged instances. */
int hmpc_upload_records_async(hmpc_handle *h, const void *host_records, int batch, void *stream);
/* ... every pitch_bytes-th record of a larger host array (pitch_bytes >= hmpc_record_stride and a multiple of 4, else HMPC_E_ARG): record k of the
 * batch is read at host_records + k * pitch_bytes -- one strided copy, no host staging (what a striped device group uses). */
int hmpc_upload_records_strided_async(hmpc_handle *h, const void *host_records, int batch, size_t pitch_bytes, void *stream);
int hmpc_download_async(hmpc_handle *h, float *forces, uint32_t *status, void *stream);
int hmpc_get_device_outputs(hmpc_handle *h, float **device_forces, uint32_t **device_status);
int hmpc_batch(const hmpc_handle *h);
int hmpc_horizon(const hmpc_handle *h);
/* average kernel time in ms of `reps` back-to-back launches of the current batch measured with HIP events on
 * `stream` (used by bench.py for the roofline object) */
int hmpc_time_solve(hmpc_handle *h, void *stream, int reps, float *ms_per_launch);


/* ---- the rows either side of the solve, batched on the device ----
 * hmpc_tick_inputs = everything ConvexMPCLocomotion::updateMPCIfNeeded reads to build one MPC tick
 * (ConvexMPC/ConvexMPCLocomotion.cpp:283-406): state estimate, leg joint angles, foot positions, commands, the
 * persistent world_position_desired, and the Gait parameters of ConvexMPC/GaitGenerator.cpp:85-113. */
struct hmpc_tick_inputs {
  double position[3], vWorld[3], omegaWorld[3], orientation[4], rpy[3];
  double rBody[9];        /* row-major, world -> body (orientation_tools.h:182-200) */
  /* Joint angles, left 0-4, right 5-9, AS updateMPCIfNeeded READS THEM: data[leg].q of the LegController
   * (ConvexMPCLocomotion.cpp:288-296).  On the reference's live path that is NOT the motor angle: updateData() hands
   * data[leg].q BY REFERENCE to computeLegJacobianAndPosition, which adds (+0.3, -0.6, +0.3) * 3.14159 to joints 2-4 in place
   * (common/LegController.cpp:48-52, 108-113) before the MPC runs; the MPC then adds its own offsets (:302-308) and the
   * solver a third time (SolverMPC.cpp:382-388) -- SURVEY A.9 "triple offset", reproduced, not fixed.  So pass either
   *   - data[leg].q after the LegController update (flags = 0), or
   *   - the raw motor angles state->motorState[].q with HMPC_TICK_LEG_Q_MOTOR set in `flags`: the builder then applies the
   *     LegController's first offset itself (same constant 3.14159, same operation order).
   * hmpc_leg_torques takes the Jacobian's argument, i.e. the MOTOR angle before that offset (LegController.cpp:108-113). */
  double leg_q[10];
  double pFoot[6];        /* world foot positions, [leg][axis] */
  double v_des_robot[2];  /* stateDes[6], stateDes[7] */
  double yaw_rate_des;    /* stateDes[11] */
  double roll_des, pitch_des; /* stateDes[3], stateDes[4] */
  double world_position_desired[2];
  int gait_offsets[2], gait_durations[2], gait_iteration;
  int flags;              /* HMPC_TICK_* bits (was padding: 0 keeps the previous meaning) */
};
#define HMPC_TICK_LEG_Q_MOTOR 1 /* leg_q holds raw motor angles: apply LegController.cpp:111-113's offset first */
/* f1+f2: builds the packed records of `batch` ticks on the device into the handle's own record buffer (which becomes
 * the current batch) and returns the clamped world_position_desired (ConvexMPCLocomotion.cpp:336-346) per instance.
 * host_ticks / wpd_out are host pointers (wpd_out may be NULL); the _device form takes device pointers and a stream. */
int hmpc_build_records(hmpc_handle *h, const struct hmpc_tick_inputs *host_ticks, int batch, double dtMPC, double *wpd_out);
int hmpc_build_records_device(hmpc_handle *h, const void *device_ticks, int batch, double dtMPC, double *device_wpd_out,
                              void *stream);
/* f3: f_ff[batch][2][6] = -rBody [GRF; GRM] from the forces of the last solve (ConvexMPCLocomotion.cpp:419-440) */
int hmpc_body_wrench(hmpc_handle *h, const double *host_rBody, double *host_f_ff);
int hmpc_body_wrench_device(hmpc_handle *h, const double *device_rBody, double *device_f_ff, void *stream);
/* f3, with the joint torques: f_ff as above, then tau[batch][2][5] = J_fm' f_ff with the force-moment Jacobian of
 * common/LegController.cpp:108-167 evaluated at leg_q[batch][10] (LegController.cpp:57-61).  f_ff may be NULL.
 * leg_q here is what computeLegJacobianAndPosition RECEIVES: the raw motor angles (it adds its 3.14159-based offset
 * itself) -- not the hmpc_tick_inputs.leg_q of a flags = 0 tick, which already carries that offset. */
int hmpc_leg_torques(hmpc_handle *h, const double *host_rBody, const double *host_leg_q, double *host_f_ff, double *host_tau);
int hmpc_leg_torques_device(hmpc_handle *h, const double *device_rBody, const double *device_leg_q, double *device_f_ff,
                            double *device_tau, void *stream);
/* f1+f2 -> solve -> f3 in one call, device-resident end to end: builds the records of `batch` ticks (device_ticks:
 * hmpc_tick_inputs[batch] in HBM) into the handle, solves them -- every instance on the smallest kernel variant that holds
 * its reduced QP, decided on the device from the gait table the builder has just generated: a walking sweep runs on the
 * 60-variable variant with no host hint -- and writes f_ff[batch][2][6] (may be NULL) and the stance feed-forward torques
 * tau[batch][2][5] (rBody and the joint angles are read from the tick structs; leg_q per the tick's HMPC_TICK_LEG_Q_MOTOR
 * flag).  device_wpd_out (may be NULL) receives the clamped world_position_desired.  Three or four launches on `stream`, no
 * host synchronisation: forces, status, f_ff and tau are valid once the stream reaches them.  With hmpc_set_device_repair
 * the torques are computed from repaired forces.  Two-contact handles only. */
int hmpc_tick_solve_device(hmpc_handle *h, const void *device_ticks, int batch, double dtMPC, double *device_wpd_out,
                           double *device_f_ff, double *device_tau, void *stream);
/* copies the current batch's packed records device -> host (parity hook for f1/f2) */
int hmpc_download_records(hmpc_handle *h, void *host_records);

/* Parity hook: runs the ASSEMBLY stage only for instance `index` of the current batch (same device code the
 * solve kernel runs) and returns the reduced QP exactly as the solver sees it: n, m, var_ind[n] (original
 * variable index, SolverMPC.cpp:644-658), H[n*n] (float, row-major, symmetric), g[n], the 16x12 constraint
 * block Fc, lb/ub[16h] (unreduced) and x0[13], Acd[169], Bcd[156].  Any output pointer may be NULL. */
int hmpc_debug_assemble(hmpc_handle *h, int index, int *n, int *m, int *var_ind, float *H, float *g, float *Fc,
                        float *lb, float *ub, float *x0, float *Acd, float *Bcd);
/* Parity hook: double-precision primal solution [batch][12h] and per-instance dual objective of the last solve.
 * hmpc_enable_f64_output (before the solve) makes every later solve write them; without it hmpc_download_f64 enables
 * the copy-out and runs the current batch once more (flagged instances get the safe pass of hmpc_download). */
int hmpc_enable_f64_output(hmpc_handle *h);
/* Parity hook: runs the SOLVER stages of the kernel (inverse, block start, dual active set, scatter) for the current batch
 * on QP data handed in from outside instead of the kernel's own assembly: per instance the reduced Hessian H[ld][ld] and
 * gradient g[ld] in the reference's reduced order (SolverMPC.cpp:644-697; binary32 values -- the reference's H_red / g_red
 * are widened floats; the upper triangle of H is read) and the per-step constraint block Fc[16][12] (fmat,
 * SolverMPC.cpp:466-548; [24][18] with three contacts).  The records of the current batch still supply the gait tables and
 * f_max.  Forces/status are written as by hmpc_solve.  tests/test_reference_source.py feeds it the reference's own data. */
int hmpc_debug_solve_external_qp(hmpc_handle *h, const float *H, const float *g, const float *Fc, int ld);
int hmpc_download_f64(hmpc_handle *h, double *x, double *obj);

/* Developer hook (only in builds with -DHMPC_PROFILE, scripts/phase_profile.py): per-phase shader-clock cycles of
 * one more launch of the current batch, [batch][32] (phase ids: hmpc_kernel.h P_*). */
int hmpc_debug_phase_cycles(hmpc_handle *h, long long *cycles);

/* ---- device groups: one process, several GPUs of a node (csrc/hmpc_group.hip) ----
 * Every MPC instance is an independent QP: a batch is cut into contiguous slices [lo, hi), one per member device
 * (hmpc_shard_bounds: the first batch % n members carry one extra instance), each solved by its own handle on its own
 * stream with no data-path collective.  The single exchange step is the gather of what the controller reads,
 * get_solution(0..11) of every instance (ConvexMPCLocomotion.cpp:419-440): the step-0 wrench [F_L F_R M_L M_R] and the
 * status word.  hmpc_group_post_gather packs them on each solve stream and all-gathers the packed slices on each
 * member's communication stream (RCCL: ncclCommInitAll + grouped ncclAllGather, librccl loaded on first use; or
 * hipMemcpyPeerAsync copies), so the exchange of solve k runs under solve k+1; afterwards EVERY member holds the
 * gathered block in HBM (hmpc_group_device_gathered) and hmpc_group_gather_wrench also hands it to the host. */
typedef struct hmpc_group hmpc_group;
enum hmpc_group_deal {
  HMPC_DEAL_CONTIGUOUS = 0, /* member i holds the contiguous slice of hmpc_shard_bounds (default) */
  HMPC_DEAL_STRIPED = 1     /* member i holds instances i, i + G, i + 2 G, ...: an ORDERED sweep's hard instances are spread over all members */
};
enum hmpc_group_transport {
  HMPC_GROUP_AUTO = 0, /* RCCL when all listed devices are distinct, else P2P */
  HMPC_GROUP_RCCL = 1,
  HMPC_GROUP_P2P = 2   /* the only transport that accepts a device listed twice (one-GPU test of the multi-member path) */
};
int hmpc_shard_bounds(int global_batch, int n_shards, int index, int *lo, int *hi);
/* devices == NULL: devices 0 .. n_devices-1.  max_batch is the GLOBAL batch the group can hold. */
int hmpc_group_create(hmpc_group **out, const struct problem_setup *setup, const int *devices, int n_devices,
                      int max_batch, int transport);
/* the same for handles of n_contacts = 2 or 3 (BASELINE config 5 runs the three-contact extension on 4 GPUs): records of
 * hmpc_record_stride_ex(h, n_contacts) bytes, forces [batch][6 n_contacts h], and the exchange carries the step-0 wrench of
 * every contact -- 6 n_contacts values [F_0 .. F_{nc-1}, M_0 .. M_{nc-1}] + the status word per instance. */
int hmpc_group_create_ex(hmpc_group **out, const struct problem_setup *setup, const int *devices, int n_devices,
                         int max_batch, int transport, int n_contacts);
int hmpc_group_contacts(const hmpc_group *g);
int hmpc_group_destroy(hmpc_group *g);
int hmpc_group_size(const hmpc_group *g);
int hmpc_group_transport(const hmpc_group *g);
int hmpc_group_batch(const hmpc_group *g);
/* member's handle (for the per-handle switches), device, slice of the current batch and solve stream; any may be NULL */
int hmpc_group_member(hmpc_group *g, int member, hmpc_handle **handle, int *device, int *lo, int *n, void **solve_stream);
int hmpc_group_upload_records(hmpc_group *g, const void *host_records, int batch);
/* How the next batch is dealt to the members (enum hmpc_group_deal).  Host-facing results (hmpc_group_gather_wrench,
 * hmpc_group_download) are in instance order either way (and the same bits, as long as every member's handle picks the same
 * kernel variant under both deals -- a member that is dealt single-support instances only runs the 60-variable variant, whose
 * answers agree with the 120-variable one's to solver precision); member sizes are the same either way; in striped mode row r of slot s
 * of the device-resident gathered block is instance s + r G, and hmpc_group_member_step returns G (1 for contiguous slices). */
int hmpc_group_set_deal(hmpc_group *g, int deal);
/* the same robot / contact constants on every member (hmpc_set_params) */
int hmpc_group_set_params(hmpc_group *g, const struct hmpc_params *p);
int hmpc_group_deal(const hmpc_group *g);
int hmpc_group_member_step(const hmpc_group *g, int member);
/* device_records[i] = member i's first record, resident on member i's device (slice sizes from hmpc_shard_bounds) */
int hmpc_group_set_device_records(hmpc_group *g, const void *const *device_records, int batch, int max_reduced_vars);
int hmpc_group_solve(hmpc_group *g);       /* asynchronous on every member */
/* hmpc_solve_command_sweep on every member: every member's slice must consist of whole groups (contiguous deal, and
 * slice sizes that are multiples of group_size -- e.g. batch = members x k x group_size), else HMPC_E_ARG and nothing is enqueued */
int hmpc_group_solve_command_sweep(hmpc_group *g, int group_size);
int hmpc_group_post_gather(hmpc_group *g); /* asynchronous: the exchange step for the solves enqueued so far */
int hmpc_group_wait_gather(hmpc_group *g);
/* member's gathered copy in HBM: [group size][slot_rows][6 nc + 1] 32-bit words (13 for two contacts), slot s = member s's
 * slice, row = the 6 nc floats of the step-0 wrench + the status word.  Valid from hmpc_group_wait_gather until the next hmpc_group_post_gather. */
int hmpc_group_device_gathered(hmpc_group *g, int member, const uint32_t **gathered, int *slot_rows);
/* blocking form of the exchange step: collects the exchange hmpc_group_post_gather posted if it has not been collected
 * yet (by hmpc_group_wait_gather or an earlier hmpc_group_gather_wrench) -- the pipelined pattern post(k), solve(k+1),
 * gather_wrench() returns solve k's results -- and otherwise posts one now for the solves enqueued so far; waits, and
 * returns host copies in instance order of the batch the exchange was posted for (its slices are remembered at post
 * time): wrench [batch][6 nc] (12 for two contacts), status [batch] (either may be NULL).
 * The exchange carries REPAIRED rows: every member's solve is followed on its stream, without a host round trip, by the
 * safe variant over the instances the fast variant flagged (hmpc_set_device_repair, on by default for group members).  hmpc_group_set_exchange_repair(g, 0) turns that off: the exchange then carries the fast
 * pass's results as they are (a flagged instance shows in its status word, its wrench is not valid) and only
 * hmpc_group_download repairs.  Instances that defeat even the safe variant (degenerate vertices, < 0.1 % at 6x the
 * nominal input ranges) stay flagged in the status word either way; hmpc_group_download's relaxed passes handle them. */
int hmpc_group_gather_wrench(hmpc_group *g, float *host_wrench, uint32_t *host_status);
int hmpc_group_set_exchange_repair(hmpc_group *g, int on);
/* all 12h forces of every instance to the host (no collective; the members' safe pass included) */
int hmpc_group_download(hmpc_group *g, float *forces, uint32_t *status);
int hmpc_group_synchronize(hmpc_group *g);
const char *hmpc_group_last_error(void);

const char *hmpc_last_hip_error(void);
const char *hmpc_version(void);

#ifdef __cplusplus
}
#endif
#endif
