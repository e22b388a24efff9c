// qpoases_shim.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/hmpc_oracle.h).
//
// Thin extern "C" wrapper around the reference's OWN vendored qpOASES 3.2.0
// (/root/reference/Hector_ROS_Simulation/hector_control/third_party/qpOASES, compiled unmodified by
// oracle/Makefile into oracle/_ref/libqpoases_ref.so).  The call sequence restates
// ConvexMPC/SolverMPC.cpp:584,702-712: fresh QProblem(nV,nC), Options::setToMPC(), printLevel PL_NONE,
// init(H,g,A,NULL,NULL,lbA,ubA,nWSR=500) (cold start), getPrimalSolution.
#include <qpOASES.hpp>

extern "C" int ref_qpoases_solve(int nV, int nC, const double *H, const double *g, const double *A, const double *lbA,
                                 const double *ubA, int nWSR_max, double *x, double *y, double *obj, int *nWSR_used) {
  qpOASES::int_t nWSR = nWSR_max;
  qpOASES::QProblem problem(nV, nC);
  qpOASES::Options op;
  op.setToMPC();
  op.printLevel = qpOASES::PL_NONE;
  problem.setOptions(op);
  int rval = (int)problem.init(const_cast<double *>(H), g, const_cast<double *>(A), NULL, NULL, lbA, ubA, nWSR);
  int rval2 = (int)problem.getPrimalSolution(x);
  if (y) problem.getDualSolution(y);
  if (obj) *obj = problem.getObjVal();
  if (nWSR_used) *nWSR_used = (int)nWSR;
  if (rval2 != (int)qpOASES::SUCCESSFUL_RETURN) return rval2 ? rval2 : -1;
  return rval;
}

extern "C" int ref_qpoases_sizes(void) { return (int)sizeof(qpOASES::real_t) * 100 + (int)sizeof(qpOASES::int_t); }
