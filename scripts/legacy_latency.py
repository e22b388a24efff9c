import sys, time, numpy as np
sys.path.insert(0, '.')
from hector_simulation_amd import interface, synthetic
f = synthetic.make_batch(4, 10, "standing", seed=6)
row = {k: np.asarray(v)[0] for k, v in f.items()}
lat=[]
for rep in range(200):
    t0=time.perf_counter()
    interface.setup_problem(synthetic.DT_MPC, 10, 0.25, synthetic.F_MAX)
    interface.update_problem_data(row["p"], row["v"], row["q"], row["w"], row["r"], row["joint_angles"], float(row["yaw"]), row["weights"], row["traj"], row["Alpha_K"], row["gait"])
    u0=[interface.get_solution(i) for i in range(12)]
    lat.append(time.perf_counter()-t0)
print("legacy tick median %.4f ms min %.4f ms  status %d iters %d" % (1e3*np.median(lat[20:]), 1e3*min(lat[20:]), interface.last_status()&0xff, (interface.last_status()>>8)&0xfff))
