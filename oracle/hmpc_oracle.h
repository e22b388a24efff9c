/*
 * hmpc_oracle.h -- CPU ORACLE for the HECTOR convex-MPC QP path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm
 *   (Hector_ROS_Simulation/hector_control/ConvexMPC/SolverMPC.cpp:65-89,120-131,133-193,302-342,371-738,
 *    RobotState.cpp:9-53, convexMPC_interface.cpp:42-110)
 * used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
 * The product (hector_simulation_amd/) never includes, links or calls anything in oracle/.
 *
 * PARITY STATUS: the reference ships no tests, golden vectors or fixtures for this path.
 * The SOLVER half is the reference itself: its vendored qpOASES 3.2.0 compiled unmodified into oracle/_ref/.
 * The ASSEMBLY half (this file's restatement) is pinned against an EXECUTION OF THE REFERENCE'S OWN SOURCE:
 * oracle/_ref/libsolvempc_ref.so = SolverMPC.cpp + RobotState.cpp + convexMPC_interface.cpp compiled unmodified
 * against the Eigen stand-in oracle/mini_eigen (Eigen3 is an un-vendored, un-pinned dependency that is not installed
 * here).  tests/test_reference_source.py: reduced structure, elimination pattern and bounds identical; with the two
 * study switches orc_set_unfused_chain / orc_set_libm_trig on, x_0, A_ct, A_qp, B_qp, F_control, g and the upper triangle
 * of H equal the reference's source bit for bit; under the default contract the data stay within binary32 round-off.
 * Not pinned: the association inside Eigen's own kernels (see the header of oracle/mini_eigen/eigen3/Eigen/Dense).
 * The CALLER-SIDE restatements at the end of this header (orc_mpc_gait, orc_build_record, orc_body_wrench,
 * orc_leg_jacobian, orc_leg_torques) are pinned the same way since round 3: against
 * oracle/_ref/libcaller_ref.so = GaitGenerator.cpp + ConvexMPCLocomotion.cpp + LegController.cpp (+ FootSwingTrajectory.cpp,
 * DesiredCommand.cpp) compiled unmodified (tests/test_caller_reference.py: tables and records bit for bit, wrench bit for
 * bit, Jacobian within one ulp of 1.0).
 *
 * Pinned arithmetic ("HMPC-A1"): IEEE binary32, round-to-nearest-even, no implicit
 * contraction; every matrix contraction is a k-ascending fmaf chain started at +0; trigonometry is evaluated
 * by the deterministic binary64 routines in this file and rounded once to binary32.
 */
#ifndef HMPC_ORACLE_H
#define HMPC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_HORIZON 36 /* K_MAX_GAIT_SEGMENTS, convexMPC_interface.h:3 */

/* POD inputs, same fields as update_data_t / problem_setup (convexMPC_interface.h:11-37). */
typedef struct {
  float p[3], v[3], q[4], w[3];
  float r[9];            /* contact positions relative to the body, r[nc*axis + contact] (reference: r[2*axis+leg]) */
  float joint_angles[10], yaw, weights[12];
  float traj[12 * ORC_MAX_HORIZON];
  float Alpha_K[18];     /* [F of each contact (3 nc), M of each contact (3 nc)] */
  unsigned char gait[3 * ORC_MAX_HORIZON]; /* gait[nc*step + contact] */
  /* ---- ORACLE EXTENSION (BASELINE config 5, no reference code: SURVEY.md section 8d "cfg-5 default extension") ----
   * nc = 3 adds a hand contact (index 2): B_ct gains its force/moment columns, the hand gets the same 8-row block as
   * the LEFT foot expressed in its own contact frame Rhand (body frame, row-major), its own force cap and stance flag.
   * nc = 0 or 2 is the reference formulation. */
  int nc;
  float Rhand[9];
  float f_max_hand;
} orc_update_t;

typedef struct {
  float dt, mu, f_max;
  int horizon;
} orc_setup_t;

/* Everything the reference forms before calling qpOASES, in the reference's own (unreduced) indexing. */
typedef struct {
  int horizon;
  float qj[10];        /* offset + fmod'ed joint angles, SolverMPC.cpp:374-393 */
  float R[9];          /* body rotation, RobotState.cpp:30 */
  float rpy[3];        /* SolverMPC.cpp:333-342 */
  float x0[13];        /* SolverMPC.cpp:420 */
  float Acd[169];      /* I + dt*A_ct, row-major */
  int nc;              /* contacts per step (2 = reference) */
  float Bcd[13 * 18];  /* dt*B_ct, row-major 13 x 6nc */
  float Rfoot[3][9];   /* SolverMPC.cpp:426-433 (index 2: the hand frame of the extension) */
  float Fc[24 * 18];   /* F_control, SolverMPC.cpp:511-548, 8nc x 6nc */
  float *Phi;          /* [h][13][6nc]  Acd^k * Bcd */
  float *Apow;         /* [h+1][13][13] Acd^k */
  float *H;            /* [6nc h][6nc h] row-major, exactly symmetric */
  float *g;            /* [6nc h] */
  float *lb, *ub;      /* [8nc h] */
} orc_qp_t;

/* Reduced QP exactly as handed to qpOASES (SolverMPC.cpp:589-697), binary64 row-major. */
typedef struct {
  int n, m;            /* new_vars, new_cons */
  int *var_ind;        /* [n] original variable index */
  int *con_ind;        /* [m] original constraint index */
  double *H, *g, *A, *lb, *ub;
} orc_red_t;

orc_qp_t *orc_qp_alloc(int horizon);
void orc_qp_free(orc_qp_t *);
orc_red_t *orc_red_alloc(int horizon);
void orc_red_free(orc_red_t *);

/* deterministic binary64 trig (exported so tests can pin them against libm) */
void orc_sincos(double x, double *s, double *c);
double orc_atan2(double y, double x);
double orc_asin(double v);

/* SolverMPC.cpp:371-577 (assembly) */
void orc_assemble(const orc_update_t *u, const orc_setup_t *s, orc_qp_t *out);
/* SolverMPC.cpp:589-697 (swing-leg elimination) */
void orc_reduce(const orc_qp_t *qp, orc_red_t *red);

/* SolverMPC.cpp:699-732: solve with the reference's vendored qpOASES (oracle/_ref), scatter to q_soln[12h].
 * returns qpOASES' init() return value (0 = SUCCESSFUL_RETURN); *nwsr receives the working-set changes used;
 * *obj the optimal objective value 0.5 x'Hx + g'x of the reduced QP.  y_red (may be NULL) receives the n+m duals. */
int orc_solve_reduced(const orc_red_t *red, double *x_red, double *y_red, int *nwsr, double *obj);
int orc_solve_mpc(const orc_update_t *u, const orc_setup_t *s, double *q_soln /*[12h]*/, int *nwsr, double *obj,
                  int *n_red, int *m_red);

/* The reference's legacy C interface (convexMPC_interface.h:39-43), prefixed so both libraries can be loaded. */
void orc_setup_problem(double dt, int horizon, double mu, double f_max);
void orc_update_problem_data(double *p, double *v, double *q, double *w, double *r, double *joint_angles, double yaw,
                             double *weights, double *state_trajectory, double *Alpha_K, int *gait);
double orc_get_solution(int index);

/* robot / contact constants, laid out as include/hector_mpc.h struct hmpc_params (defaults = the reference's literals) */
typedef struct {
  float mass, inertia[3], mu, lt, lh, gravity;
} orc_params_t;
void orc_set_params(const orc_params_t *p); /* NULL: defaults */
void orc_get_params(orc_params_t *p);
void orc_set_dense_chain(int on); /* 1: run every cost chain over all 13h rows (test of zero-block neutrality) */
void orc_set_unfused_chain(int on); /* study switch, see hmpc_oracle.c */
void orc_set_libm_trig(int on);     /* study switch, see hmpc_oracle.c */
void orc_unpack_record(const unsigned char *rec, int horizon, orc_update_t *u);
void orc_unpack_record3(const unsigned char *rec, int horizon, orc_update_t *u); /* extension records (nc = 3) */
void orc_set_records_nc(int nc); /* record flavour orc_solve_records parses: 2 (reference) or 3 (extension) */

/* Batched CPU baseline over packed records (layout: hector_simulation_amd/records.py; 54+12h floats then 2h gait bytes,
 * record stride `stride` bytes).  Solves records [first, first+count) and writes 12h doubles each.
 * Returns the number of instances whose qpOASES status != 0.  t_assemble/t_solve (may be NULL) accumulate seconds. */
int orc_solve_records(const unsigned char *records, int stride, int first, int count, int horizon, float dt,
                      float f_max, double *q_soln, int *nwsr_out, double *obj_out, double *t_assemble, double *t_solve);
int orc_solve_records_ex(const unsigned char *records, int stride, int first, int count, int horizon, float dt,
                         float f_max, double *q_soln, int *nwsr_out, double *obj_out, double *t_assemble,
                         double *t_solve, int *rv_out /* qpOASES' return value per instance, may be NULL */);


/* ---- the caller-side code either side of the solve ----
 * f1  updateMPCIfNeeded input builder, ConvexMPC/ConvexMPCLocomotion.cpp:283-406 (+ narrowing of
 *     convexMPC_interface.cpp:83-103) -> one packed record;  f2  Gait::mpc_gait, ConvexMPC/GaitGenerator.cpp:85-103;
 * f3  body-frame feed-forward wrench f_ff[leg] = -rBody [GRF; GRM], ConvexMPCLocomotion.cpp:419-440.
 * All binary64, evaluated in the order written there. */
typedef struct {
  double position[3], vWorld[3], omegaWorld[3], orientation[4], rpy[3];
  double rBody[9];        /* row-major, world -> body */
  double leg_q[10];       /* data[leg].q as updateMPCIfNeeded reads it (flags = 0) or raw motor angles (flags & 1) */
  double pFoot[6];        /* world foot positions, [leg][axis] */
  double v_des_robot[2];  /* stateDes[6], stateDes[7] */
  double yaw_rate_des;    /* stateDes[11] */
  double roll_des, pitch_des; /* stateDes[3], stateDes[4] */
  double world_position_desired[2];
  int gait_offsets[2], gait_durations[2], gait_iteration, flags;
} orc_tick_t;

void orc_mpc_gait(int n_segments, const int offsets[2], const int durations[2], int iteration, int *table /*[2n]*/);
/* builds the packed record (layout: hector_simulation_amd/records.py) and the clamped world_position_desired */
void orc_build_record(const orc_tick_t *t, int horizon, double dtMPC, unsigned char *record, double wpd_out[2]);
/* q_soln (>= 12 doubles) + rBody -> f_ff[2][6] */
void orc_body_wrench(const double *q_soln, const double *rBody, double *f_ff);
/* f3, second half: leg Jacobian (common/LegController.cpp:108-167) and joint torques tau = J' f_ff (:57-61) */
void orc_leg_jacobian(const double q[5], int leg, double J[30]);
void orc_leg_torques(const double *f_ff, const double *leg_q, double *tau);

/* provided by oracle/_ref/libqpoases_ref.so (oracle/qpoases_shim.cpp, built from the reference's own sources) */
int ref_qpoases_solve(int nV, int nC, const double *H, const double *g, const double *A, const double *lbA,
                      const double *ubA, int nWSR_max, double *x, double *y, double *obj, int *nWSR_used);

#ifdef __cplusplus
}
#endif
#endif
