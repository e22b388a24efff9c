#!/usr/bin/env python3
"""Generates tests/golden/indefinite_golden.npz: two double-support h = 20 records at 10x the nominal input ranges whose reduced
Hessian -- assembled in binary32 as the contract demands -- is NOT positive definite, and what the reference's solver returns for
them: its vendored qpOASES 3.2.0 (oracle/_ref, built from /root/reference/.../third_party/qpOASES by oracle/Makefile) regularises
such a QP and reports success (QProblem.cpp:1753-1860, QProblemB.cpp:1418-1431, 1999-2031 under Options::setToMPC).

PROVENANCE as for mpc_golden.npz (make_golden.py): our oracle's pinned-arithmetic assembly + the REFERENCE'S OWN qpOASES.  h = 20
lies one step beyond the reference's own horizon limit (19): an oracle-extension shape, like the `single_h20` case there.
Run from the repo root:  python tests/golden/make_indefinite_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hector_simulation_amd import records, synthetic  # noqa: E402
from oracle import oracle_py  # noqa: E402

H, PICK = 20, (60, 61, 90)  # of synthetic.hard_batch(96, 20, "standing", 17, 10): 61 and 90 are indefinite, 60 is not (control)


def main():
    rec = records.pack_records(synthetic.hard_batch(96, H, "standing", 17, 10), H)[list(PICK)]
    sol = oracle_py.solve_records(rec, H, synthetic.DT_MPC, synthetic.F_MAX)
    assert sol["n_bad"] == 0
    eig = np.array([np.linalg.eigvalsh(oracle_py.assemble_record(r, H, synthetic.DT_MPC, synthetic.F_MAX)["H_red"])[0] for r in rec])
    assert eig[0] > 0 and (eig[1:] < 0).all(), eig
    path = os.path.join(ROOT, "tests", "golden", "indefinite_golden.npz")
    np.savez_compressed(path, records=rec, horizon=np.int32(H), q_soln=sol["q_soln"], nwsr=sol["nwsr"], obj=sol["obj"], min_eig=eig)
    print("wrote", path, os.path.getsize(path), "bytes; min eigenvalues", eig, "nWSR", sol["nwsr"])
    # the same data as raw arrays for tests/src/host_api_sweep.c (C has no npz reader): records [3][stride] bytes, forces [3][240] doubles
    rec.tofile(os.path.join(ROOT, "tests", "golden", "indefinite_records.bin"))
    sol["q_soln"].astype("<f8").tofile(os.path.join(ROOT, "tests", "golden", "indefinite_forces.bin"))


if __name__ == "__main__":
    main()
