"""GPU tests at BASELINE.json's full sizes through size-independent properties: determinism, batch-order invariance, exact
zeros on swing legs, warm-vs-cold start agreement, primal feasibility against the oracle's constraint data on a sample,
oracle parity on a sample.  (Every-instance oracle parity at the same sizes -- all 65 536 walking instances of config 3 and
all 8 192 three-contact instances of config 5, as a pool of oracle processes -- lives in tests/test_gpu_full_batch.py.)"""
import numpy as np
import pytest

from hector_simulation_amd import interface, records, synthetic

pytestmark = pytest.mark.gpu


def solve(rec, h, warm=True, device_batch=None):
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, rec.shape[0])
    mpc.set_warm_start(warm)
    mpc.upload(rec)
    mpc.solve()
    forces, status = mpc.download()
    mpc.close()
    return forces, status


@pytest.mark.parametrize("name,gait,h,nb,seed,phase", [
    ("cfg2 walking 1024 fixed phase", "walking", 10, 1024, 2, 0),
    ("cfg3 walking sweep 65536", "walking", 10, 65536, 3, "random"),
    ("cfg4 h20 single support 4096", "single", 20, 4096, 4, "random"),
    ("metric 2-contact 8192", "standing", 10, 8192, 6, "random"),
])
def test_full_size_properties(oracle, name, gait, h, nb, seed, phase):
    f = synthetic.make_batch(nb, h, gait, seed=seed, phase=phase)
    rec = records.pack_records(f, h)
    forces, status = solve(rec, h)
    assert (interface.status_code(status) == 0).all(), np.unique(interface.status_code(status), return_counts=True)
    assert np.isfinite(forces).all()
    # determinism: a second launch reproduces every bit
    forces2, status2 = solve(rec, h)
    np.testing.assert_array_equal(forces.view(np.uint32), forces2.view(np.uint32))
    np.testing.assert_array_equal(status, status2)
    # instances are independent: a permuted batch gives the permuted result, bit for bit
    perm = np.random.default_rng(0).permutation(nb)
    forces_p, _ = solve(rec[perm], h)
    np.testing.assert_array_equal(forces_p.view(np.uint32), forces[perm].view(np.uint32))
    # swing-leg variables are exactly zero, stance Fz within [0, f_max/2] (row 7: 0 <= 2 Fz <= f_max)
    g = np.asarray(f["gait"]).reshape(nb, h, 2)
    F = forces.reshape(nb, h, 12)
    for leg in range(2):
        sw = g[:, :, leg] == 0
        cols = [3 * leg, 3 * leg + 1, 3 * leg + 2, 6 + 3 * leg, 7 + 3 * leg, 8 + 3 * leg]
        assert (F[sw][:, cols] == 0.0).all()
        fz = F[:, :, 3 * leg + 2][~sw]
        assert (fz >= -1e-4).all() and (fz <= synthetic.F_MAX / 2 + 1e-3).all()
    # cold start (the reference's strategy) reaches the same optimum as the block warm start
    sub = np.arange(0, nb, max(1, nb // 512))
    cold, st_c = solve(rec[sub], h, warm=False)
    assert (interface.status_code(st_c) == 0).all()
    scale = np.maximum(1.0, np.abs(cold).max(axis=1))
    assert (np.abs(cold - forces[sub]).max(axis=1) / scale).max() < 1e-5
    assert interface.status_iters(st_c).mean() > interface.status_iters(status[sub]).mean()
    # oracle on a sample: feasibility of A x and parity with qpOASES
    samp = sub[:: max(1, len(sub) // 24)]
    ref = oracle.solve_records(rec[samp], h, synthetic.DT_MPC, synthetic.F_MAX)
    assert ref["n_bad"] == 0
    q = ref["q_soln"]
    err = np.abs(forces[samp] - q).max(axis=1) / np.maximum(1.0, np.abs(q).max(axis=1))
    assert err.max() < 1e-4
    for k in samp[:6]:
        a = oracle.assemble_record(rec[k], h, synthetic.DT_MPC, synthetic.F_MAX)
        x = forces[k].astype(np.float64)[a["var_ind"]]
        ax = a["A_red"] @ x
        assert (ax >= a["lb_red"] - 1e-3).all() and (ax <= a["ub_red"] + 1e-3).all()


def test_duplicates_and_ragged_batches():
    f = synthetic.make_batch(5, 10, "mixed", seed=9, phase="random")
    rec = records.pack_records(f, 10)
    big = np.concatenate([rec, rec[::-1], rec[:1]], axis=0)  # 11 instances: duplicates, odd batch size
    forces, status = solve(big, 10)
    assert (interface.status_code(status) == 0).all()
    np.testing.assert_array_equal(forces[:5].view(np.uint32), forces[5:10][::-1].view(np.uint32))
    np.testing.assert_array_equal(forces[0].view(np.uint32), forces[10].view(np.uint32))
    one, _ = solve(rec[2:3], 10)  # batch of one
    np.testing.assert_array_equal(one[0].view(np.uint32), forces[2].view(np.uint32))
    mpc = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, 4)
    mpc.upload(rec[:0])  # empty batch is a no-op
    mpc.solve()
    fe, se = mpc.download()
    assert fe.shape == (0, 120) and se.shape == (0,)
    with pytest.raises(interface.HmpcError):
        mpc.upload(rec)  # 5 > max_batch 4
    mpc.close()


def test_all_swing_and_saturated_instances():
    """Edge cases: no stance leg at all (n = 0) -> zeros; f_max tiny -> Fz caps active everywhere."""
    f = synthetic.make_batch(3, 10, "walking", seed=4, phase="random")
    f["gait"][:] = 0
    forces, status = solve(records.pack_records(f, 10), 10)
    assert (interface.status_code(status) == 0).all() and (forces == 0).all()
    f = synthetic.make_batch(16, 10, "standing", seed=4)
    rec = records.pack_records(f, 10)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, 10, 20.0, 16)  # f_max = 20 N: 2*Fz <= 20 binds (weight ~ 88 N)
    mpc.upload(rec)
    mpc.solve()
    forces, status = mpc.download()
    mpc.close()
    assert (interface.status_code(status) == 0).all()
    fz = forces.reshape(16, 10, 12)[:, :, [2, 5]]
    assert (fz <= 10.0 + 1e-4).all() and (fz > 9.0).any()


def test_stream_ordered_upload_download_matches_blocking_path():
    """hmpc_upload_records_async / hmpc_download_async on a side stream with pinned buffers == the blocking calls."""
    import torch

    nb, h = 96, 10
    rec = records.pack_records(synthetic.make_batch(nb, h, "mixed", seed=33, phase="random"), h)
    ref = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    ref.upload(rec)
    ref.solve()
    f0, s0 = ref.download()
    ref.close()
    stream = torch.cuda.Stream()
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    h_rec = torch.from_numpy(rec.copy()).pin_memory()
    h_f = torch.zeros((nb, 12 * h), dtype=torch.float32).pin_memory()
    h_s = torch.zeros((nb,), dtype=torch.int32).pin_memory()
    for _ in range(2):  # twice: the second round reuses every buffer
        mpc.upload_async(h_rec.data_ptr(), nb, stream.cuda_stream)
        mpc.solve(stream.cuda_stream)
        mpc.download_async(h_f.data_ptr(), h_s.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        np.testing.assert_array_equal(h_f.numpy(), f0)
        np.testing.assert_array_equal(h_s.numpy().astype(np.uint32), s0)
    mpc.close()


def test_bench_line_contract():
    """bench.py prints one JSON line with the keys the driver and the judge read (short run, no CPU baseline)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "1024",
                          "--no-cpu-baseline", "--no-side-configs", "--check", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["parity"]["max_rel_force_err_vs_qpoases"] < 1e-4 and d["solver"]["failed"] == 0
    assert abs(d["value"] - 1024 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6
