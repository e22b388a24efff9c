#!/usr/bin/env python3
"""CPU-only experiment: the reference's own qH = B'(S B) + ... is NOT symmetric in binary32 ((i,j) and (j,i)
round differently, relative asymmetry 2e-8); qpOASES reads the full matrix in products and the upper triangle in its Cholesky.
Our contract mirrors H from the UPPER triangle.  Which symmetrisation of the reference's own H_red -- mirror-upper, mirror-lower,
average -- best reproduces the reference's own q_soln (qpOASES on its unsymmetric matrix)?
    python scripts/dev/symmetrisation.py [count]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hector_simulation_amd import synthetic  # noqa: E402
from oracle import oracle_py, ref_py  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 128
h = 10
f = synthetic.make_batch(count, h, "standing", seed=6, phase="random")
res = {k: [] for k in ("as_is", "mirror_upper", "mirror_lower", "average")}
asym = []
with ref_py.quiet():
    for k in range(count):
        row = {key: np.asarray(v)[k] for key, v in f.items()}
        t = ref_py.tick(row, h, synthetic.DT_MPC, 0.25, synthetic.F_MAX)
        H, g, A, lb, ub = t["H_red"], t["g_red"], t["A_red"], t["lb_red"], t["ub_red"]
        q_ref = t["q_red"] if "q_red" in t else None
        if q_ref is None:
            q_full = t["q_soln"]
            q_ref = q_full[t["var_ind"]]
        asym.append(np.abs(H - H.T).max() / np.abs(H).max())
        up, lo = np.triu(H) + np.triu(H, 1).T, np.tril(H) + np.tril(H, -1).T
        for name, Hs in (("as_is", H), ("mirror_upper", up), ("mirror_lower", lo), ("average", 0.5 * (H + H.T))):
            x, _, _, _, st = oracle_py.qpoases_solve(Hs, g, A, lb, ub)
            res[name].append(np.abs(x - q_ref).max() / max(1.0, np.abs(q_ref).max()))
print(f"{count} instances of the metric's 2-contact case; relative asymmetry of the reference's H_red: max {max(asym):.2e} median {np.median(asym):.2e}")
for name, e in res.items():
    e = np.array(e)
    print(f"  qpOASES on H {name:13s}: forces vs the reference's own q_soln: max {e.max():.2e} median {np.median(e):.2e} above 1e-4: {(e > 1e-4).mean():.2f}")
