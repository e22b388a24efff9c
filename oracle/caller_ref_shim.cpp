// caller_ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/hmpc_oracle.h).
//
// Drives the reference's OWN caller-side code for SURVEY.md section 8(f) rows f1-f3, compiled UNMODIFIED from
// /root/reference by oracle/Makefile into oracle/_ref/libcaller_ref.so:
//   ConvexMPC/GaitGenerator.cpp          Gait::setIterations / mpc_gait / get*SubPhase        (row f2)
//   ConvexMPC/ConvexMPCLocomotion.cpp    ConvexMPCLocomotion::run / updateMPCIfNeeded         (rows f1, f3: f_ff = -rBody [GRF; GRM])
//   src/common/LegController.cpp         computeLegJacobianAndPosition, updateData, updateCommand (row f3: tau = J' f)
//   src/common/FootSwingTrajectory.cpp, src/common/DesiredCommand.cpp   (what run() and the data structs link against)
// against the Eigen stand-in oracle/mini_eigen and the placeholder boost/lcm headers oracle/ref_stubs.
//
// What this file supplies instead of reference code, and why:
//   * setup_problem / update_problem_data / get_solution (convexMPC_interface.h:39-43): CAPTURE the arguments the
//     reference's updateMPCIfNeeded passes (ConvexMPCLocomotion.cpp:410-415) and either forward them to a back end
//     (the reference's real solver in _ref/libsolvempc_ref.so, handed in as function pointers) or answer
//     get_solution from a vector the test supplies;
//   * swingLegController::initSwingLegController / updateSwingLeg: empty.  The swing-leg IK (SwingLegController.cpp) is
//     outside SURVEY section 8 and run() only needs the symbols;
//   * the object graph of ControlFSMData (Biped, LegController, StateEstimatorContainer, DesiredStateCommand,
//     LowlevelState/Cmd) that the ROS main.cpp builds.
// Nothing here computes a quantity under test, with one labelled exception (refc_leg_tau_f64).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include <unistd.h>

// ConvexMPCLocomotion keeps updateMPCIfNeeded, pFoot, world_position_desired, f_ff, iterationCounter private
// (ConvexMPCLocomotion.h:47-100).  The reference's translation units are compiled without this; only this window sees
// the members (access specifiers do not change layout).
#define private public
#include "ConvexMPCLocomotion.h"
#undef private
#include "convexMPC_interface.h"

// ---- capture of the convex-MPC interface --------------------------------------------------------------------
namespace {
struct Capture {
  int n_setup = 0, n_update = 0;
  double dt = 0, mu = 0, f_max = 0;
  int horizon = 0;
  double p[3], v[3], q[4], w[3], r[6], joint_angles[10], yaw, weights[12], traj[12 * K_MAX_GAIT_SEGMENTS], Alpha_K[12];
  int gait[2 * K_MAX_GAIT_SEGMENTS];
} g_cap;
double g_solution[12 * K_MAX_GAIT_SEGMENTS];
typedef void (*setup_fn)(double, int, double, double);
typedef void (*update_fn)(double *, double *, double *, double *, double *, double *, double, double *, double *, double *, int *);
typedef double (*getsol_fn)(int);
setup_fn g_be_setup = nullptr;
update_fn g_be_update = nullptr;
getsol_fn g_be_get = nullptr;
}  // namespace

extern "C" void setup_problem(double dt, int horizon, double mu, double f_max) {
  g_cap.n_setup++;
  g_cap.dt = dt, g_cap.horizon = horizon, g_cap.mu = mu, g_cap.f_max = f_max;
  if (g_be_setup) g_be_setup(dt, horizon, mu, f_max);
}
extern "C" void update_problem_data(double *p, double *v, double *q, double *w, double *r, double *joint_angles, double yaw,
                                    double *weights, double *state_trajectory, double *Alpha_K, int *gait) {
  g_cap.n_update++;
  memcpy(g_cap.p, p, sizeof g_cap.p), memcpy(g_cap.v, v, sizeof g_cap.v), memcpy(g_cap.q, q, sizeof g_cap.q);
  memcpy(g_cap.w, w, sizeof g_cap.w), memcpy(g_cap.r, r, sizeof g_cap.r);
  memcpy(g_cap.joint_angles, joint_angles, sizeof g_cap.joint_angles);
  g_cap.yaw = yaw;
  memcpy(g_cap.weights, weights, sizeof g_cap.weights), memcpy(g_cap.Alpha_K, Alpha_K, sizeof g_cap.Alpha_K);
  memcpy(g_cap.traj, state_trajectory, sizeof(double) * 12 * g_cap.horizon);
  memcpy(g_cap.gait, gait, sizeof(int) * 2 * g_cap.horizon);
  if (g_be_update) g_be_update(p, v, q, w, r, joint_angles, yaw, weights, state_trajectory, Alpha_K, gait);
}
extern "C" double get_solution(int index) { return g_be_get ? g_be_get(index) : g_solution[index]; }
extern "C" void update_solver_settings(int, double, double, double, double, double) {}

// ---- the two swing-leg symbols run() links against (SwingLegController.cpp is out of scope) ---------------------
void swingLegController::initSwingLegController(ControlFSMData *d, Gait *g, double dtSwing) {
  data = d, gait = g, _dtSwing = dtSwing;
}
void swingLegController::updateSwingLeg() {}

// ---- the object graph main.cpp builds ---------------------------------------------------------------------------
struct RefCaller {
  Biped biped;
  LowlevelState lowState;
  LowlevelCmd lowCmd;
  StateEstimate se;
  LegController *legs;
  StateEstimatorContainer *estimator;
  DesiredStateCommand *command;
  ControlFSMData fsm;
  ConvexMPCLocomotion *mpc;
};

extern "C" {

void refc_set_backend(void *setup, void *update, void *get) {
  g_be_setup = (setup_fn)setup, g_be_update = (update_fn)update, g_be_get = (getsol_fn)get;
}
void refc_set_solution(const double *sol, int n) { memcpy(g_solution, sol, sizeof(double) * n); }

// dt / iterations_between_mpc as FSMState_Walking.cpp:5 passes them (0.001, 40).  The constructor opens "foot_pos.txt"
// in the working directory (ConvexMPCLocomotion.cpp:26): construct inside `scratch_dir`.
RefCaller *refc_create(double dt, int iterations_between_mpc, const char *scratch_dir) {
  char cwd[4096];
  if (!getcwd(cwd, sizeof cwd)) return nullptr;
  if (scratch_dir && chdir(scratch_dir) != 0) return nullptr;
  RefCaller *c = new RefCaller();
  memset(&c->se, 0, sizeof c->se);
  c->legs = new LegController(c->biped);
  c->estimator = new StateEstimatorContainer(&c->lowState, c->legs->data, &c->se);
  c->command = new DesiredStateCommand(&c->se, dt);
  c->fsm._biped = &c->biped;
  c->fsm._stateEstimator = c->estimator;
  c->fsm._legController = c->legs;
  c->fsm._desiredStateCommand = c->command;
  c->fsm._interface = nullptr;
  c->fsm._lowCmd = &c->lowCmd;
  c->fsm._lowState = &c->lowState;
  c->mpc = new ConvexMPCLocomotion(dt, iterations_between_mpc);
  if (scratch_dir && chdir(cwd) != 0) return nullptr;
  return c;
}
void refc_destroy(RefCaller *c) {
  if (!c) return;
  delete c->mpc;
  delete c->command;
  delete c->estimator;
  delete c->legs;
  delete c;
}

// StateEstimate fields the MPC path reads (StateEstimatorContainer.h:44-56); rBody row-major, world -> body
void refc_set_state(RefCaller *c, const double *position, const double *vWorld, const double *omegaWorld,
                    const double *orientation, const double *rpy, const double *rBody) {
  for (int i = 0; i < 3; ++i) {
    c->se.position[i] = position[i], c->se.vWorld[i] = vWorld[i], c->se.omegaWorld[i] = omegaWorld[i], c->se.rpy[i] = rpy[i];
    for (int j = 0; j < 3; ++j) c->se.rBody(i, j) = rBody[3 * i + j];
  }
  for (int i = 0; i < 4; ++i) c->se.orientation[i] = orientation[i];
  c->se.vBody = c->se.rBody * c->se.vWorld;  // PositionVelocityEstimator.cpp:10-11 (read by run() for v_abs only)
}
void refc_set_command(RefCaller *c, const double *stateDes12) {
  for (int i = 0; i < 12; ++i) c->command->data.stateDes[i] = stateDes12[i];
}
// the reference's own LegController::updateData (LegController.cpp:42-55) on motor angles (MotorState::q is float)
void refc_update_leg_data_from_motors(RefCaller *c, const float *q10, const float *dq10) {
  for (int i = 0; i < 10; ++i) c->lowState.motorState[i].q = q10[i], c->lowState.motorState[i].dq = dq10 ? dq10[i] : 0.f;
  c->legs->updateData(&c->lowState);
}
// same call updateData makes (LegController.cpp:51), on binary64 angles: data[leg].q is passed BY REFERENCE and comes
// back with the 3.14159-based offsets added (LegController.cpp:111-113)
void refc_set_leg_q(RefCaller *c, const double *q10) {
  for (int leg = 0; leg < 2; ++leg) {
    for (int j = 0; j < 5; ++j) c->legs->data[leg].q(j) = q10[5 * leg + j];
    computeLegJacobianAndPosition(c->biped, c->legs->data[leg].q, &(c->legs->data[leg].J_force_moment),
                                  &(c->legs->data[leg].J_force), &(c->legs->data[leg].p), leg);
  }
}
// writes data[leg].q WITHOUT the Jacobian call (a tick whose leg_q is "data[leg].q as updateMPCIfNeeded reads it")
void refc_poke_leg_q(RefCaller *c, const double *q10) {
  for (int leg = 0; leg < 2; ++leg)
    for (int j = 0; j < 5; ++j) c->legs->data[leg].q(j) = q10[5 * leg + j];
}
void refc_get_leg(RefCaller *c, int leg, double *q5, double *J_fm30, double *J_f15, double *p3) {
  const LegControllerData &d = c->legs->data[leg];
  for (int j = 0; j < 5; ++j) q5[j] = d.q(j);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 5; ++j) J_fm30[5 * i + j] = d.J_force_moment(i, j);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 5; ++j) J_f15[5 * i + j] = d.J_force(i, j);
  for (int i = 0; i < 3; ++i) p3[i] = d.p(i);
}

// private state of ConvexMPCLocomotion that updateMPCIfNeeded reads (ConvexMPCLocomotion.h:62-99)
void refc_set_members(RefCaller *c, const double *world_position_desired3, const double *pFoot6, int iterationCounter,
                      int gaitNumber, int firstRun) {
  for (int i = 0; i < 3; ++i) c->mpc->world_position_desired[i] = world_position_desired3[i];
  for (int leg = 0; leg < 2; ++leg)
    for (int i = 0; i < 3; ++i) c->mpc->pFoot[leg][i] = pFoot6[3 * leg + i];
  c->mpc->iterationCounter = iterationCounter;
  c->mpc->gaitNumber = gaitNumber;
  c->mpc->firstRun = firstRun != 0;
}
void refc_get_members(RefCaller *c, double *world_position_desired3, double *pFoot6, double *f_ff12, int *iterationCounter) {
  for (int i = 0; i < 3; ++i) world_position_desired3[i] = c->mpc->world_position_desired[i];
  for (int leg = 0; leg < 2; ++leg) {
    for (int i = 0; i < 3; ++i) pFoot6[3 * leg + i] = c->mpc->pFoot[leg][i];
    for (int i = 0; i < 6; ++i) f_ff12[6 * leg + i] = c->mpc->f_ff[leg][i];
  }
  *iterationCounter = c->mpc->iterationCounter;
}

// The three statements of run() between the foot-placement heuristics and the swing/stance dispatch
// (ConvexMPCLocomotion.cpp:157 setIterations, :174 mpc_gait, :177 updateMPCIfNeeded), on the gait object run() would
// pick (:39-43), executed on the reference's objects -- the part of run() that is rows f1/f2.
void refc_update_mpc(RefCaller *c) {
  Gait *gait = (c->mpc->gaitNumber == 2) ? &c->mpc->walking : &c->mpc->standing;
  gait->setIterations(c->mpc->iterationsBetweenMPC, c->mpc->iterationCounter);
  int *mpcTable = gait->mpc_gait();
  c->mpc->updateMPCIfNeeded(mpcTable, c->fsm, false);
}
// the whole public entry point, as FSMState_Walking::run calls it (FSMState_Walking.cpp:36-37)
void refc_run(RefCaller *c, int gaitNumber) {
  c->mpc->setGaitNum(gaitNumber);
  c->mpc->run(c->fsm);
}

struct refc_capture {
  int n_setup, n_update, horizon, pad;
  double dt, mu, f_max;
  double p[3], v[3], q[4], w[3], r[6], joint_angles[10], yaw, weights[12], Alpha_K[12];
  double traj[12 * K_MAX_GAIT_SEGMENTS];
  int gait[2 * K_MAX_GAIT_SEGMENTS];
};
void refc_get_capture(refc_capture *out) {
  out->n_setup = g_cap.n_setup, out->n_update = g_cap.n_update, out->horizon = g_cap.horizon, out->pad = 0;
  out->dt = g_cap.dt, out->mu = g_cap.mu, out->f_max = g_cap.f_max;
  memcpy(out->p, g_cap.p, sizeof g_cap.p), memcpy(out->v, g_cap.v, sizeof g_cap.v), memcpy(out->q, g_cap.q, sizeof g_cap.q);
  memcpy(out->w, g_cap.w, sizeof g_cap.w), memcpy(out->r, g_cap.r, sizeof g_cap.r);
  memcpy(out->joint_angles, g_cap.joint_angles, sizeof g_cap.joint_angles);
  out->yaw = g_cap.yaw;
  memcpy(out->weights, g_cap.weights, sizeof g_cap.weights), memcpy(out->Alpha_K, g_cap.Alpha_K, sizeof g_cap.Alpha_K);
  memcpy(out->traj, g_cap.traj, sizeof g_cap.traj), memcpy(out->gait, g_cap.gait, sizeof g_cap.gait);
}
int refc_sizeof_capture(void) { return (int)sizeof(refc_capture); }

// stance feed-forward: commands[leg].feedforwardForce = f (what run() does for a stance foot, ConvexMPCLocomotion.cpp:263),
// then the reference's LegController::updateCommand (LegController.cpp:57-99): legtau = J_fm' f, narrowed into
// LowlevelCmd::motorCmd[].tau (float)
void refc_update_command(RefCaller *c, const double *f_ff12, float *tau10) {
  for (int leg = 0; leg < 2; ++leg) {
    c->legs->commands[leg].zero();
    for (int i = 0; i < 6; ++i) c->legs->commands[leg].feedforwardForce(i) = f_ff12[6 * leg + i];
  }
  c->legs->updateCommand(&c->lowCmd);
  for (int i = 0; i < 10; ++i) tau10[i] = c->lowCmd.motorCmd[i].tau;
}
// LABELLED RESTATEMENT (one expression): updateCommand keeps legtau only as float.  This evaluates the expression of
// LegController.cpp:60 on the reference's own J_force_moment so the binary64 value can be compared as well.
void refc_leg_tau_f64(RefCaller *c, const double *f_ff12, double *tau10) {
  for (int leg = 0; leg < 2; ++leg) {
    Vec6<double> footForce;
    for (int i = 0; i < 6; ++i) footForce(i) = f_ff12[6 * leg + i];
    Vec5<double> legtau = c->legs->data[leg].J_force_moment.transpose() * footForce;
    for (int j = 0; j < 5; ++j) tau10[5 * leg + j] = legtau(j);
  }
}

// Row f2 on its own: a Gait of the reference (GaitGenerator.cpp:6-16), setIterations (:109-113), mpc_gait (:85-104) and
// the two sub-phase functions (:29-80)
void refc_gait(int nMPC_segments, int off0, int off1, int dur0, int dur1, int iterationsPerMPC, int currentIteration,
               int *table, double *contact2, double *swing2, int *stance_swing2) {
  Gait g(nMPC_segments, Vec2<int>(off0, off1), Vec2<int>(dur0, dur1), "probe");
  g.setIterations(iterationsPerMPC, currentIteration);
  int *t = g.mpc_gait();
  for (int i = 0; i < 2 * nMPC_segments; ++i) table[i] = t[i];
  Vec2<double> cs = g.getContactSubPhase(), ss = g.getSwingSubPhase();
  contact2[0] = cs[0], contact2[1] = cs[1], swing2[0] = ss[0], swing2[1] = ss[1];
  stance_swing2[0] = g._stance, stance_swing2[1] = g._swing;
}
}  // extern "C"
