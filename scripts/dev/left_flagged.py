#!/usr/bin/env python3
"""Developer tool: the instances of a stress row that stay flagged (or end ok-relaxed) after hmpc_resolve_failed -- status, iterations, |W|
and how far the forces left in the buffer are from qpOASES' (a flagged answer that is right anyway points at the final check, one that
is far off at the iteration).    python scripts/dev/left_flagged.py [nb] [scale] [h] [gait]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402
from oracle import oracle_py  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h = int(sys.argv[3]) if len(sys.argv) > 3 else 10
gait = sys.argv[4] if len(sys.argv) > 4 else "standing"
rec = records.pack_records(synthetic.hard_batch(nb, h, gait, 17, scale), h)
m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
m.set_auto_resolve(False)
m.upload(rec)
m.solve()
_, s0 = m.download()
m.resolve_failed()
f1, s1 = m.download()
x64, obj = m.download_f64()
m.close()
c1 = interface.status_code(s1)
left = np.nonzero(c1 != 0)[0]
print(f"{gait} h={h} x{scale} nb={nb}: final codes {dict(zip(*np.unique(c1, return_counts=True)))}")
for i in left:
    ref = oracle_py.solve_records(rec, h, synthetic.DT_MPC, synthetic.F_MAX, first=int(i), count=1)
    q = ref["q_soln"][0]
    err = np.abs(f1[i] - q).max() / max(1.0, np.abs(q).max())
    print(f"  inst {i}: fast code {interface.status_code(s0)[i]} it {interface.status_iters(s0)[i]} |W| {interface.status_nactive(s0)[i]} -> final code {c1[i]} it {interface.status_iters(s1)[i]} |W| {interface.status_nactive(s1)[i]}"
          f"  err vs qpOASES {err:.2e} (qpOASES {'BAD' if ref['bad'][0] else 'ok'}, nWSR {int(ref['nwsr'][0])}, obj {ref['obj'][0]:.6e}; ours obj {obj[i]:.6e})")
