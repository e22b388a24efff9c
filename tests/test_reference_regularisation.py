"""What the reference's qpOASES run does with a reduced Hessian that is not positive definite -- pinned on the CPU against the
vendored qpOASES itself (oracle/_ref), so that the GPU's regularisation steps (KernelArgs::reg_step, hmpc_resolve_failed) have a
stated behaviour to match:

  * the Cholesky factorisation of H fails at a pivot p < 0 (LAPACKReplacement.cpp:85-105 "tunnels" p back in R[0]);
  * options.epsRegularisation = min(-p + eps, sqrt(eps)), eps = 1e3 * 2.221e-16 (QProblemB.cpp:1418-1431, Options.cpp:144);
  * H += rho I with rho = |H|_F * epsRegularisation -- IN PLACE, in the caller's array (QProblemB.cpp:1999-2031);
  * the regularised QP is solved, then once more with the gradient g - rho x_1 (numRegularisationSteps = 1 in Options::setToMPC,
    which SolverMPC.cpp:703 selects; QProblem.cpp:1753-1860); the second answer is what getPrimalSolution returns.
Both regularised QPs are strictly convex, so the answer does not depend on qpOASES' path through them."""
import numpy as np

from hector_simulation_amd import records, synthetic

EPS_REG = 1.0e3 * 2.221e-16


def failing_pivot(H):
    """Upper Cholesky in the reference's loop order; the first pivot sum that is not positive (None: positive definite)."""
    n = H.shape[0]
    R = H.copy()
    for j in range(n):
        s = R[j, j] - np.dot(R[:j, j], R[:j, j])
        if not s > 0.0:
            return s
        R[j, j] = np.sqrt(s)
        R[j, j + 1:] = (R[j, j + 1:] - R[:j, j] @ R[:j, j + 1:]) / R[j, j]
    return None


def test_qpoases_regularises_an_indefinite_hessian_and_takes_one_proximal_step(oracle):
    h = 20
    rec = records.pack_records(synthetic.hard_batch(96, h, "standing", 17, 10), h)
    found = 0
    for i in range(56, 96):
        a = oracle.assemble_record(rec[i], h, synthetic.DT_MPC, synthetic.F_MAX)
        H, g, A, lb, ub = a["H_red"], a["g_red"], a["A_red"], a["lb_red"], a["ub_red"]
        piv = failing_pivot(H)
        if piv is None:
            continue
        found += 1
        assert piv < 0.0 and np.linalg.eigvalsh(H)[0] < 0.0 and np.abs(H - H.T).max() == 0.0
        Hc = H.copy()
        x_ref, _, _, _, st = oracle.qpoases_solve(Hc, g, A, lb, ub)  # the reference's call (its H array is modified in place)
        assert st == 0
        eps_r = min(-piv + EPS_REG, np.sqrt(EPS_REG))
        rho = np.sqrt((H * H).sum()) * eps_r
        assert abs((Hc[0, 0] - H[0, 0]) - rho) < 1e-9 * rho + 1e-15  # exactly what qpOASES added to the diagonal
        Hr = H + rho * np.eye(H.shape[0])
        assert failing_pivot(Hr) is None
        x1, _, _, _, st1 = oracle.qpoases_solve(Hr.copy(), g, A, lb, ub)
        x2, _, _, _, st2 = oracle.qpoases_solve(Hr.copy(), g - rho * x1, A, lb, ub)
        assert st1 == 0 and st2 == 0
        scale = max(1.0, np.abs(x_ref).max())
        assert np.abs(x2 - x_ref).max() / scale < 1e-7, np.abs(x2 - x_ref).max()
        assert np.abs(x1 - x_ref).max() / scale > 1e-3  # (the proximal step matters: x_1 alone is not the reference's answer)
    assert found >= 1
