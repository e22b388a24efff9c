// Source-level drop-in proof (tests/test_source_dropin.py): this caller includes the REFERENCE'S OWN headers
//   ConvexMPC/convexMPC_interface.h  (C interface, EXTERNC)            -- via -I/root/reference/.../ConvexMPC
//   ConvexMPC/SolverMPC.h            (C++ solve_mpc / get_q_soln; pulls in Eigen -> oracle/mini_eigen stands in)
// and NOT include/hector_mpc.h, and is linked against libhector_mpc_hip.so only.  It drives the same sequence as
// ConvexMPCLocomotion::updateMPCIfNeeded (ConvexMPCLocomotion.cpp:410-429), then the C++-linkage entry with the
// reference's PODs.  Exit codes: 0 solved, 3 = the library reported that no GPU is present (prints to stderr), 1 wrong.
#include <cstdio>
#include <cstring>

#include "SolverMPC.h"            // the reference's header (declares solve_mpc, get_q_soln, resize_qp_mats ...)
#include "convexMPC_interface.h"  // the reference's header

int main() {
  const int horizon = 10;
  double p[3] = {0.0, 0.0, 0.55}, v[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, w[3] = {0, 0, 0};
  double r[6] = {0.0, 0.0, 0.06, -0.06, -0.55, -0.55};
  double ja[10] = {0}, Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1};
  double Alpha[12] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
  double traj[12 * horizon] = {0};
  int gait[2 * horizon];
  for (int i = 0; i < horizon; ++i) traj[12 * i + 5] = 0.55, gait[2 * i] = gait[2 * i + 1] = 1;

  if (get_solution(2) != 0.0) return 1;  // convexMPC_interface.cpp:107: 0 before the first solve
  setup_problem(0.04, horizon, 0.25, 500.0);
  update_problem_data(p, v, q, w, r, ja, 0.0, Q, traj, Alpha, gait);
  update_solver_settings(100, 1e-7, 1e-8, 1.5, 0.1, 0);
  const double fz = get_solution(2) + get_solution(5);
  std::printf("C interface: Fz_L + Fz_R = %.4f\n", fz);
  if (fz == 0.0) return 3;  // no device: nothing was solved (the library has printed why)

  // C++ linkage of SolverMPC.h:56,63 with the reference's own PODs (convexMPC_interface.h:11-37)
  problem_setup ps;
  ps.dt = 0.04f, ps.mu = 0.25f, ps.f_max = 500.f, ps.horizon = horizon;
  static update_data_t ud;
  std::memset(&ud, 0, sizeof ud);
  ud.p[2] = 0.55f, ud.q[0] = 1.f;
  ud.r[2] = 0.06f, ud.r[3] = -0.06f, ud.r[4] = ud.r[5] = -0.55f;
  for (int i = 0; i < 12; ++i) ud.weights[i] = (float)Q[i], ud.Alpha_K[i] = (float)Alpha[i];
  for (int i = 0; i < horizon; ++i) ud.traj[12 * i + 5] = 0.55f, ud.gait[2 * i] = ud.gait[2 * i + 1] = 1;
  solve_mpc(&ud, &ps);
  mfp *qs = get_q_soln();
  std::printf("C++ interface: Fz_L + Fz_R = %.4f\n", qs[2] + qs[5]);
  const double d = (qs[2] + qs[5]) - fz;
  return (fz > 80.0 && fz < 110.0 && d < 1e-9 && d > -1e-9) ? 0 : 1;
}
