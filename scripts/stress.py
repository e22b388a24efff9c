import sys, numpy as np
sys.path.insert(0,'/root/repo')
from hector_simulation_amd import interface, records, synthetic
from oracle import oracle_py as O
def hard_batch(nb, h, gait, seed, scale):
    f = synthetic.make_batch(nb, h, gait, seed=seed, phase="random", yaw_rate_cmd=True)
    rng = np.random.default_rng(seed+1)
    rpy = rng.uniform(-0.1*scale, 0.1*scale, (nb,3))
    f["q"] = synthetic.quat_from_rpy(rpy[:,0], rpy[:,1], rpy[:,2])
    f["v"] = rng.uniform(-0.3*scale, 0.3*scale, (nb,3))
    f["w"] = rng.uniform(-0.5*scale, 0.5*scale, (nb,3))
    f["joint_angles"] = rng.uniform(-0.15*scale, 0.15*scale, (nb,10))
    tr = f["traj"].reshape(nb,h,12); tr[:,:,9] *= scale; f["traj"]=tr.reshape(nb,-1)
    return f
for gait,h in (("standing",10),("walking",10),("mixed",10),("single",20)):
  for scale in (1,3,6):
    nb=256
    f = hard_batch(nb,h,gait,17,scale); rec = records.pack_records(f,h)
    mpc = interface.BatchedMPC(synthetic.DT_MPC,h,synthetic.F_MAX,nb); mpc.set_auto_resolve(False); mpc.upload(rec); mpc.solve(); forces0,status0 = mpc.download(); nres = mpc.resolve_failed(); forces,status = mpc.download(); mpc.close()
    print('   fast-pass codes', dict(zip(*np.unique(interface.status_code(status0),return_counts=True))), 're-solved', nres)
    code = interface.status_code(status)
    ref = O.solve_records(rec,h,synthetic.DT_MPC,synthetic.F_MAX)
    ok = (code==0) | (code==6)
    rel = (code==6)
    q = ref["q_soln"]; err = np.abs(forces-q).max(axis=1)/np.maximum(1,np.abs(q).max(axis=1))
    # per-instance qpOASES status unknown (n_bad total); report
    print(gait,h,"scale",scale,"gpu codes",dict(zip(*np.unique(code,return_counts=True))),"qpoases bad",ref["n_bad"],"nwsr max",ref["nwsr"].max(),"iters max",interface.status_iters(status).max(),"act max",interface.status_nactive(status).max(),"max err(ok)",err[ok].max() if ok.any() else None, "n err>1e-4", int((err[ok]>1e-4).sum()), "max err(relaxed)", err[rel].max() if rel.any() else None)
