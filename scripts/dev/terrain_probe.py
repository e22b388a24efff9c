import sys, numpy as np, torch
sys.path.insert(0, '.')
torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic
h=10; groups=128; kk=64; floors=4; bb=groups*kk
rng2=np.random.default_rng(13)
f0 = synthetic.make_batch(groups, h, "standing", seed=12, phase="random")
fs = {k: np.repeat(np.asarray(v), kk, axis=0) for k, v in f0.items()}
vx, vy, yr = rng2.uniform(-0.5, 0.5, bb), rng2.uniform(-0.2, 0.2, bb), rng2.uniform(-0.3, 0.3, bb)
tr = fs["traj"].reshape(bb, h, 12).copy(); stp=np.arange(h)[None,:]
tr[:, :, 9], tr[:, :, 10], tr[:, :, 8] = vx[:, None], vy[:, None], yr[:, None]
tr[:, :, 3] = fs["p"][:, 0:1] + stp * synthetic.DT_MPC * vx[:, None]
tr[:, :, 4] = fs["p"][:, 1:2] + stp * synthetic.DT_MPC * vy[:, None]
tr[:, 1:, 2] = tr[:, 0:1, 2] + stp[:, 1:] * synthetic.DT_MPC * yr[:, None]
fs["traj"] = tr.reshape(bb, 12*h)
recs = records.pack_records(fs, h)
d_mu = torch.from_numpy(np.array([0.8,1.25,2.0,3.0],dtype=np.float32)[np.arange(bb)%floors]).cuda()
out={}
for name in ("ind","sweep"):
    m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, bb)
    m.set_instance_mu(d_mu.data_ptr(), keepalive=d_mu)
    m.set_auto_resolve(False)
    m.upload(recs)
    if name=="ind": m.solve()
    else: m.solve_command_sweep(kk)
    _, s0 = m.download()
    m.set_auto_resolve(True)
    f, s1 = m.download()
    out[name]=(s0.copy(), f.copy(), s1.copy()); m.close()
s0i,fi,s1i=out["ind"]; s0s,fsw,s1s=out["sweep"]
print("fast-pass codes ind", dict(zip(*np.unique(interface.status_code(s0i),return_counts=True))), "sweep", dict(zip(*np.unique(interface.status_code(s0s),return_counts=True))))
print("fast-pass status words differing", int((s0i!=s0s).sum()), "final differing", int((s1i!=s1s).sum()), "forces bit-identical", bool(np.array_equal(fi.view(np.uint32), fsw.view(np.uint32))))
d=np.nonzero(s1i!=s1s)[0][:8]
for i in d: print(i, hex(s0i[i]), hex(s0s[i]), hex(s1i[i]), hex(s1s[i]))
