#!/usr/bin/env python3
"""Developer tool: for a stress case, every instance reported solved whose forces differ from qpOASES by more than 1e-4 --
objective and constraint violation of BOTH answers on the oracle's reduced QP (who is right?)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch; torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic
from oracle import oracle_py, pool
import importlib.util
spec = importlib.util.spec_from_file_location("stress_mod", os.path.join(ROOT, "scripts", "stress.py"))
gait, h, scale, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1024
hard_batch = synthetic.hard_batch  # (nb, h, gait, seed, scale): the off-nominal stress rows
f = hard_batch(nb, h, gait, 17, scale); rec = records.pack_records(f, h)
mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb); mpc.upload(rec); mpc.solve(); forces, status = mpc.download()
x64, obj64 = mpc.download_f64(); mpc.close()
ref = pool.solve_records_parallel(rec, h, synthetic.DT_MPC, synthetic.F_MAX)
code = interface.status_code(status); q = ref["q_soln"]
err = np.abs(forces - q).max(axis=1) / np.maximum(1, np.abs(q).max(axis=1))
bad = np.nonzero(((code == 0) | (code == 6)) & (err > 1e-4))[0]
print("instances reported solved with err > 1e-4:", bad, "codes", code[bad])
for k in bad[:8]:
    o = oracle_py.assemble_record(rec[k], h, synthetic.DT_MPC, synthetic.F_MAX)
    H, g, A, lb, ub = o["H_red"], o["g_red"], o["A_red"], o["lb_red"], o["ub_red"]
    for name, x in (("hip", x64[k][o["var_ind"]]), ("qpoases", q[k][o["var_ind"]])):
        ax = A @ x
        viol = max(0.0, (lb - ax).max(), (ax - ub).max())
        print(f"  inst {k} {name:8s} obj {0.5 * x @ H @ x + g @ x:.10e}  max row violation {viol:.3e}  |x|max {np.abs(x).max():.3e}")
    print(f"  inst {k}: err {err[k]:.3e} nWSR {ref['nwsr'][k]} iters {interface.status_iters(status)[k]} |W| {interface.status_nactive(status)[k]} qp obj {ref['obj'][k]:.10e} hip obj64 {obj64[k]:.10e}")
