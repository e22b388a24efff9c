// The reference's own call sequence for one robot, from C++, against libhector_mpc_hip.so -- exactly what
// ConvexMPCLocomotion::updateMPCIfNeeded does (ConvexMPC/ConvexMPCLocomotion.cpp:410-429): setup_problem,
// update_problem_data (blocking), get_solution(0..11).  Build (no source of the reference changes):
//   g++ -std=c++17 -Iinclude examples/legacy_tick.cpp -Lhector_simulation_amd -lhector_mpc_hip
//       -Wl,-rpath,$PWD/hector_simulation_amd -o legacy_tick
// Prints the step-0 wrench [F_L F_R M_L M_R] of a nominal stand-in-place tick and the solver status.
#include <cstdio>
#include <vector>

#include "hector_mpc.h"

int main() {
  const int horizon = 10;
  const double dtMPC = 0.001 * 40, f_max = 500.0;
  double p[3] = {0.0, 0.0, 0.55}, v[3] = {0, 0, 0}, q[4] = {1, 0, 0, 0}, w[3] = {0, 0, 0};
  double r[6] = {0.0, 0.0, 0.06, -0.06, -0.55, -0.55};  // r[2*axis + leg]
  double joint_angles[10] = {0};
  double Q[12] = {100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1};
  double Alpha[12] = {1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2};
  std::vector<double> traj(12 * horizon, 0.0);
  for (int i = 0; i < horizon; ++i) traj[12 * i + 5] = 0.55;  // hold the nominal height
  std::vector<int> gait(2 * horizon, 1);                      // both feet in stance over the horizon

  if (get_solution(2) != 0.0) return 4;  // convexMPC_interface.cpp:107: 0 before the first solve of the process
  setup_problem(dtMPC, horizon, 0.25, f_max);
  if (get_solution(2) != 0.0) return 4;
  update_problem_data(p, v, q, w, r, joint_angles, 0.0, Q, traj.data(), Alpha, gait.data());
  const unsigned st = hmpc_last_status();
  std::printf("status code %u, active-set iterations %u\n", HMPC_STATUS_CODE(st), HMPC_STATUS_ITERS(st));
  for (int i = 0; i < 12; ++i) std::printf("%s%.4f", i ? " " : "u0 = ", get_solution(i));
  std::printf("\n");
  const double fz = get_solution(2) + get_solution(5);
  return (HMPC_STATUS_CODE(st) == HMPC_S_OK && fz > 80.0 && fz < 110.0) ? 0 : 1;  // ~ m g = 88.3 N plus the height error
}
