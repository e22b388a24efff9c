#!/usr/bin/env python3
"""Developer tool: active-set iteration statistics of the three-contact extension (BASELINE config 5 shape)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

nb = 2048
f = synthetic.make_batch3(nb, 10, "standing", seed=5, hand="contact")
rec = records.pack_records(f, 10, 3)
m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb, contacts=3)
m.upload(rec)
m.solve()
fo, st = m.download()
it = interface.status_iters(st)
ms = m.time_solve(5)
print("3-contact: mean", it.mean(), "median", np.median(it), "p90", np.percentile(it, 90), "p99", np.percentile(it, 99), "max", it.max(),
      "failed", int((interface.status_code(st) != 0).sum()), "kernel ms", ms)
