import sys, numpy as np
sys.path.insert(0,'.')
from hector_simulation_amd import interface, records, synthetic
f=synthetic.make_batch(8192,10,"standing",seed=6,phase="random")
rec=records.pack_records(f,10)
m=interface.BatchedMPC(synthetic.DT_MPC,10,synthetic.F_MAX,8192); m.upload(rec); m.solve(); fo,st=m.download()
it=interface.status_iters(st)
print("mean",it.mean(),"median",np.median(it),"p90",np.percentile(it,90),"p99",np.percentile(it,99),"max",it.max())
print(np.bincount(np.minimum(it,40))[:41].tolist())
np.save("gpurun_out/iters_%s.npy"%sys.argv[1], it)
