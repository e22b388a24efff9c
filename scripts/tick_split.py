import sys, time, numpy as np
sys.path.insert(0,'.')
from hector_simulation_amd import interface, records, synthetic
f=synthetic.make_batch(1,10,"standing",seed=6)
rec=records.pack_records(f,10)
m=interface.BatchedMPC(synthetic.DT_MPC,10,synthetic.F_MAX,1); m.upload(rec); m.solve(); m.download()
print("kernel ms (batch 1, 20 reps back to back):", m.time_solve(20))
t=[]
for _ in range(200):
    t0=time.perf_counter(); m.solve(); m.L.hmpc_download(m.h, None, None); t.append(time.perf_counter()-t0)
print("launch+sync ms:", 1e3*np.median(t[20:]))
