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


if __name__ == "__main__":
    main()
