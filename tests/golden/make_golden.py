#!/usr/bin/env python3
"""Generates tests/golden/mpc_golden.npz.

PROVENANCE: the reference ships no golden vectors for this path and its Eigen assembly cannot
be built here, so these goldens are produced by OUR oracle (oracle/hmpc_oracle.c: pinned-arithmetic restatement of
SolverMPC.cpp:371-697) followed by the REFERENCE'S OWN vendored qpOASES 3.2.0 (oracle/_ref, built from
/root/reference/.../third_party/qpOASES by oracle/Makefile).  They pin (a) the oracle against regressions and (b) the
HIP path on the GPU box, where /root/reference does not exist.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hector_simulation_amd import records, synthetic  # noqa: E402
from oracle import oracle_py  # noqa: E402

CASES = [  # name, gait, horizon, batch, seed, kwargs
    ("stand_nominal", "standing", 10, 1, 1, dict(randomize=False)),
    ("stand_rand", "standing", 10, 3, 6, dict()),
    ("walk_rand", "walking", 10, 4, 3, dict(phase="random")),
    ("mixed_rand", "mixed", 10, 2, 11, dict(phase="random", yaw_rate_cmd=True)),
    ("single_h20", "single", 20, 2, 4, dict(phase="random")),
]


def main():
    out = {}
    for name, gait, h, nb, seed, kw in CASES:
        f = synthetic.make_batch(nb, h, gait, seed=seed, **kw)
        rec = records.pack_records(f, h)
        sol = oracle_py.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
        assert sol["n_bad"] == 0
        out[f"{name}/records"] = rec
        out[f"{name}/horizon"] = np.int32(h)
        out[f"{name}/q_soln"] = sol["q_soln"]
        out[f"{name}/obj"] = sol["obj"]
        out[f"{name}/nwsr"] = sol["nwsr"]
        for k in range(nb):
            a = oracle_py.assemble_record(rec[k], h, synthetic.DT_MPC, synthetic.F_MAX)
            out[f"{name}/{k}/var_ind"] = a["var_ind"]
            out[f"{name}/{k}/H_red"] = a["H_red"].astype(np.float32)  # exact: the doubles are widened floats
            out[f"{name}/{k}/g_red"] = a["g_red"].astype(np.float32)
            out[f"{name}/{k}/Fc"] = a["Fc"]
            out[f"{name}/{k}/ub"] = a["ub"]
            out[f"{name}/{k}/x0"] = a["x0"]
    path = os.path.join(ROOT, "tests", "golden", "mpc_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    main_three_contacts()
    main_ticks()


CASES3 = [  # three-contact EXTENSION (no reference code; oracle nc = 3 branch): name, gait, hand, batch, seed
    ("stand_hand", "standing", "contact", 2, 5),
    ("walk_window", "walking", "window", 2, 6),
]


def main_three_contacts():
    out = {}
    for name, gait, hand, nb, seed in CASES3:
        f = synthetic.make_batch3(nb, 10, gait, seed=seed, hand=hand, phase="random")
        rec = records.pack_records(f, 10, 3)
        sol = oracle_py.solve_records(rec, 10, synthetic.DT_MPC, synthetic.F_MAX, nc=3)
        assert sol["n_bad"] == 0
        out[f"{name}/records"] = rec
        out[f"{name}/q_soln"] = sol["q_soln"]
        out[f"{name}/obj"] = sol["obj"]
        for k in range(nb):
            a = oracle_py.assemble_record(rec[k], 10, synthetic.DT_MPC, synthetic.F_MAX, nc=3)
            out[f"{name}/{k}/var_ind"] = a["var_ind"]
            out[f"{name}/{k}/H_red"] = a["H_red"].astype(np.float32)
            out[f"{name}/{k}/g_red"] = a["g_red"].astype(np.float32)
            out[f"{name}/{k}/Fc"] = a["Fc"]
    path = os.path.join(ROOT, "tests", "golden", "mpc_golden_3c.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main_ticks():
    """Rows f1-f3: tick inputs -> packed record (+ clamped world_position_desired), solved forces -> body-frame wrench
    and stance joint torques (oracle restatements of ConvexMPCLocomotion.cpp:283-440, GaitGenerator.cpp:85-103,
    LegController.cpp:57-61,108-167)."""
    out = {}
    for name, gait, h, nb, seed in (("walk", "walking", 10, 4, 77), ("stand", "standing", 10, 3, 78)):
        t = synthetic.make_ticks(nb, h, gait, seed=seed)
        rec, wpd = oracle_py.build_records(t, h, synthetic.DT_MPC)
        sol = oracle_py.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
        f_ff = oracle_py.body_wrench(sol["q_soln"].astype(np.float32).astype(np.float64), t["rBody"])
        tau = oracle_py.leg_torques(f_ff, t["leg_q"])
        out[f"{name}/ticks"] = t.view(np.uint8).reshape(nb, -1)
        out[f"{name}/horizon"] = np.int32(h)
        out[f"{name}/records"] = rec
        out[f"{name}/wpd"] = wpd
        out[f"{name}/q_soln"] = sol["q_soln"]
        out[f"{name}/f_ff"] = f_ff
        out[f"{name}/tau"] = tau
    path = os.path.join(ROOT, "tests", "golden", "tick_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
