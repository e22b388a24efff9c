import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch; torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic
for nb in (64, 256, 512, 768, 1024, 1536, 2304):
    f = synthetic.make_batch(nb, 10, "standing", seed=2, phase="random")
    m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb)
    m.upload(records.pack_records(f, 10)); m.solve(); m.download()
    ms = min(m.time_solve(20) for _ in range(3))
    print(f"b{nb:5d} {ms:.4f} ms  {nb/ms/1e3:.3f} M/s")
    m.close()
