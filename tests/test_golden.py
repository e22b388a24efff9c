"""Golden fixtures (tests/golden/mpc_golden.npz; provenance in tests/golden/make_golden.py: our oracle's assembly +
the reference's own qpOASES).  CPU: the oracle reproduces them (bit-exact assembly).  GPU: the HIP path matches them
without needing /root/reference on the box."""
import os

import numpy as np
import pytest

from hector_simulation_amd import interface, synthetic

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpc_golden.npz")
NAMES = ["stand_nominal", "stand_rand", "walk_rand", "mixed_rand", "single_h20"]


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(oracle, gold, name):
    rec, h = gold[f"{name}/records"], int(gold[f"{name}/horizon"])
    sol = oracle.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
    assert sol["n_bad"] == 0
    np.testing.assert_allclose(sol["q_soln"], gold[f"{name}/q_soln"], rtol=0, atol=1e-9)
    np.testing.assert_array_equal(sol["nwsr"], gold[f"{name}/nwsr"])
    for k in range(rec.shape[0]):
        a = oracle.assemble_record(rec[k], h, synthetic.DT_MPC, synthetic.F_MAX)
        np.testing.assert_array_equal(a["var_ind"], gold[f"{name}/{k}/var_ind"])
        np.testing.assert_array_equal(a["H_red"].astype(np.float32).view(np.uint32), gold[f"{name}/{k}/H_red"].view(np.uint32))
        np.testing.assert_array_equal(a["g_red"].astype(np.float32).view(np.uint32), gold[f"{name}/{k}/g_red"].view(np.uint32))
        np.testing.assert_array_equal(a["Fc"].view(np.uint32), gold[f"{name}/{k}/Fc"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_golden(gold, name):
    rec, h = gold[f"{name}/records"], int(gold[f"{name}/horizon"])
    nb = rec.shape[0]
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.upload(rec)
    mpc.solve()
    forces, status = mpc.download()
    x64, obj = mpc.download_f64()
    assert (interface.status_code(status) == 0).all()
    q = gold[f"{name}/q_soln"]
    err = np.abs(forces - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    assert err.max() < 1e-4  # north_star tolerance
    assert (np.abs(x64 - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))).max() < 1e-7
    assert (np.abs(obj - gold[f"{name}/obj"]) / np.maximum(1.0, np.abs(gold[f"{name}/obj"]))).max() < 1e-4
    for k in range(nb):
        d = mpc.debug_assemble(k)
        np.testing.assert_array_equal(d["var_ind"], gold[f"{name}/{k}/var_ind"])
        np.testing.assert_array_equal(d["H"].view(np.uint32), gold[f"{name}/{k}/H_red"].view(np.uint32))
        np.testing.assert_array_equal(d["g"].view(np.uint32), gold[f"{name}/{k}/g_red"].view(np.uint32))
        np.testing.assert_array_equal(d["Fc"].view(np.uint32), gold[f"{name}/{k}/Fc"].view(np.uint32))
        np.testing.assert_array_equal(d["x0"].view(np.uint32), gold[f"{name}/{k}/x0"].view(np.uint32))
    mpc.close()
