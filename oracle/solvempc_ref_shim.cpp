// solvempc_ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/hmpc_oracle.h).
//
// Read-only window onto the file-scope state of the reference's OWN assembly code
//   ConvexMPC/SolverMPC.cpp:19-63, 365-368 (rs, A_qp, B_qp, S, X_d, U_b, L_b, fmat, qH, qg, x_0, I_world, A_ct, B_ct_r,
//   H_qpoases ... q_red, var_elim, con_elim)
// after a call of its solve_mpc().  oracle/Makefile compiles SolverMPC.cpp, RobotState.cpp and
// convexMPC_interface.cpp UNMODIFIED from /root/reference against oracle/mini_eigen (the Eigen stand-in) and links
// them with this file and the reference's vendored qpOASES into oracle/_ref/libsolvempc_ref.so.  Nothing here computes:
// every function copies one of the reference's globals out.  The reference's C interface itself
// (setup_problem / update_problem_data / get_solution, convexMPC_interface.h:39-43) is exported by the library as is.
#include <string.h>

#include "RobotState.h"
#include "SolverMPC.h"
#include "common_types.h"
#include "../third_party/qpOASES/include/qpOASES.hpp"

using Eigen::Dynamic;

extern RobotState rs;
extern Matrix<fpt, Dynamic, 13> A_qp;
extern Matrix<fpt, Dynamic, Dynamic> B_qp;
extern Matrix<fpt, Dynamic, Dynamic> S;
extern Matrix<fpt, Dynamic, 1> X_d;
extern Matrix<fpt, Dynamic, 1> U_b;
extern Matrix<fpt, Dynamic, 1> L_b;
extern Matrix<fpt, Dynamic, Dynamic> fmat;
extern Matrix<fpt, Dynamic, Dynamic> qH;
extern Matrix<fpt, Dynamic, 1> qg;
extern Matrix<fpt, 13, 1> x_0;
extern Matrix<fpt, 3, 3> I_world;
extern Matrix<fpt, 13, 13> A_ct;
extern Matrix<fpt, 13, 12> B_ct_r;
extern qpOASES::real_t *H_qpoases, *g_qpoases, *A_qpoases, *lb_qpoases, *ub_qpoases, *q_soln;
extern qpOASES::real_t *H_red, *g_red, *A_red, *lb_red, *ub_red, *q_red;
extern char var_elim[2000];
extern char con_elim[2000];
extern problem_setup problem_configuration;
extern update_data_t update;

namespace {
template <int R, int C>
int dump(const Matrix<fpt, R, C> &m, float *out, int cap, int *rows, int *cols) {
  *rows = m.rows(), *cols = m.cols();
  if (m.rows() * m.cols() > cap) return -2;
  for (int i = 0; i < m.rows(); ++i) /* row-major out */
    for (int j = 0; j < m.cols(); ++j) out[(size_t)i * m.cols() + j] = m(i, j);
  return 0;
}
}  // namespace

// copies the named float matrix (row-major) into out[cap]; returns 0, -1 unknown name, -2 too small
extern "C" int ref_get_matrix(const char *name, float *out, int cap, int *rows, int *cols) {
#define M(n, expr) \
  if (!strcmp(name, n)) return dump(expr, out, cap, rows, cols);
  M("A_qp", A_qp)
  M("B_qp", B_qp)
  M("S", S)
  M("X_d", X_d)
  M("U_b", U_b)
  M("L_b", L_b)
  M("fmat", fmat)
  M("qH", qH)
  M("qg", qg)
  M("x_0", x_0)
  M("I_world", I_world)
  M("A_ct", A_ct)
  M("B_ct_r", B_ct_r)
  M("R", rs.R)
  M("R_yaw", rs.R_yaw)
  M("I_body", rs.I_body)
  M("r_feet", rs.r_feet)
#undef M
  return -1;
}

// the binary64 arrays of SolverMPC.cpp:47-59 (pointers into the reference's own buffers; valid until the next
// setup_problem) and the elimination marks of :62-63
extern "C" const double *ref_get_array(const char *name) {
#define A(n) \
  if (!strcmp(name, #n)) return n;
  A(H_qpoases) A(g_qpoases) A(A_qpoases) A(lb_qpoases) A(ub_qpoases) A(q_soln)
  A(H_red) A(g_red) A(A_red) A(lb_red) A(ub_red) A(q_red)
#undef A
  return nullptr;
}
extern "C" const char *ref_var_elim(void) { return var_elim; }
extern "C" const char *ref_con_elim(void) { return con_elim; }
extern "C" int ref_horizon(void) { return problem_configuration.horizon; }
// the float-typed update_data_t as the reference's own narrowing (convexMPC_interface.cpp:83-103) left it
extern "C" const update_data_t *ref_update(void) { return &update; }
// direct entry to SolverMPC.cpp:371 with caller-supplied PODs (bypasses only the double->float narrowing)
extern "C" void ref_solve_mpc(update_data_t *u, problem_setup *s) { solve_mpc(u, s); }
extern "C" int ref_sizeof_update(void) { return (int)sizeof(update_data_t); }
