import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import torch; torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic
nb = 2048
f = synthetic.make_batch3(nb, 10, "standing", seed=5, hand="contact")
rec = records.pack_records(f, 10, 3)
m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb, contacts=3)
m.upload(rec); m.solve(); fo, st = m.download()
it = interface.status_iters(st); na = interface.status_nactive(st)
o = np.argsort(-it)[:12]
print("top iters", it[o], "nactive", na[o])
print("corr iters~nactive", np.corrcoef(it, na)[0, 1], "nactive mean", na.mean(), "max", na.max())
for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 200)):
    s = (it >= lo) & (it < hi)
    print(f"iters [{lo},{hi}): {s.sum():5d} instances, nactive mean {na[s].mean() if s.any() else 0:.1f}")
m.set_warm_start(False); m.solve(); _, st0 = m.download()
it0 = interface.status_iters(st0)
print("cold start: mean", it0.mean(), "max", it0.max(), " top-warm instances cold:", it0[o])
