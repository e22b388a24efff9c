"""Rows f1+f2 -> a1..a16 -> f3 of SURVEY.md section 8 as ONE device-resident entry, ``hmpc_tick_solve_device``:
tick structs in HBM in, forces / body-frame wrench / joint torques in HBM out, three or four launches on one stream and no
host synchronisation in between (ConvexMPCLocomotion.cpp:283-440, GaitGenerator.cpp:85-103, LegController.cpp:57-61,
108-167).  Every instance is routed ON THE DEVICE to the smallest kernel variant that holds its reduced QP (the size class
the record builder derives from the gait table it has just generated): a walking sweep built on the device runs on the
60-variable variant without any host hint.

Checked here: the entry equals the separate calls bit for bit (per size class), mixed batches solve and match qpOASES, the
routing really is in effect (walking ticks built on the device cost what the hinted 60-variable launch costs), and a
multi-tick closed loop whose tick state never leaves HBM agrees tick by tick with the reference's own caller code
(oracle/_ref/libcaller_ref.so) and the reference QP solver."""
import numpy as np
import pytest

from hector_simulation_amd import interface, synthetic

pytestmark = pytest.mark.gpu
H = 10
LEG_OFFSET = np.tile([0.0, 0.0, 0.3 * 3.14159, -0.6 * 3.14159, 0.3 * 3.14159], 2)  # LegController.cpp:111-113


def _torch():
    import torch

    return torch


def _to_device(t):
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(t).view(np.uint8).reshape(len(t), -1).copy()).cuda()


def _mixed_ticks(nb, seed, motor=True):
    """walking ticks with every second one turned into a standing tick (both feet in stance for the whole horizon)"""
    t = synthetic.make_ticks(nb, H, "walking", seed=seed)
    t["gait_offsets"][1::2] = (0, 0)
    t["gait_durations"][1::2] = (H, H)
    if motor:
        t["leg_q"] = t["leg_q"] - LEG_OFFSET  # raw motor angles
        t["flags"] = 1                        # HMPC_TICK_LEG_Q_MOTOR
    return t


def _pipeline(t, device_repair=False):
    torch = _torch()
    nb = len(t)
    d_t = _to_device(t)
    d_tau = torch.zeros((nb, 10), dtype=torch.float64, device="cuda")
    d_ff = torch.zeros((nb, 12), dtype=torch.float64, device="cuda")
    d_wpd = torch.zeros((nb, 2), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    mpc = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb)
    if device_repair:
        mpc.set_device_repair(True)
    mpc.tick_solve_device(d_t.data_ptr(), nb, synthetic.DT_MPC, d_tau.data_ptr(), d_ff.data_ptr(), d_wpd.data_ptr(), 0)
    forces, status = mpc.download()
    rec = mpc.download_records()
    out = dict(forces=forces, status=status, records=rec, tau=d_tau.cpu().numpy(), f_ff=d_ff.cpu().numpy(),
               wpd=d_wpd.cpu().numpy())
    mpc.close()
    return out


def _separate(t, n_reduced):
    """the same work through the separate entry points, with the host's size hint (one variant for the whole batch)"""
    nb = len(t)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb)
    wpd = mpc.build_records(t, synthetic.DT_MPC)
    interface._check(mpc.L.hmpc_set_max_reduced_vars(mpc.h, n_reduced), "hint")
    mpc.solve()
    forces, status = mpc.download()
    q_motor = t["leg_q"] if int(t["flags"][0]) & 1 else t["leg_q"] - LEG_OFFSET
    f_ff, tau = mpc.leg_torques(t["rBody"], q_motor)
    rec = mpc.download_records()
    mpc.close()
    return dict(forces=forces, status=status, records=rec, tau=tau.reshape(nb, 10), f_ff=f_ff.reshape(nb, 12), wpd=wpd)


@pytest.mark.parametrize("gait,n_reduced", [("walking", 60), ("standing", 120)])
def test_pipeline_equals_the_separate_calls_bit_for_bit(gait, n_reduced):
    nb = 200
    t = synthetic.make_ticks(nb, H, gait, seed=71)
    t["leg_q"] = t["leg_q"] - LEG_OFFSET
    t["flags"] = 1
    a, b = _pipeline(t), _separate(t, n_reduced)
    assert (interface.status_code(a["status"]) == 0).all()
    for k in ("records", "status"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    np.testing.assert_array_equal(a["forces"].view(np.uint32), b["forces"].view(np.uint32))
    for k in ("tau", "f_ff", "wpd"):
        np.testing.assert_array_equal(a[k].view(np.uint64), b[k].view(np.uint64), err_msg=k)


def test_leg_q_flag_semantics_agree():
    """flags = 0 ticks carry data[leg].q AFTER the LegController's in-place offset, flags = HMPC_TICK_LEG_Q_MOTOR ticks the
    motor angle: the same robot state either way -> the same records, forces and torques, bit for bit."""
    nb = 64
    tm = _mixed_ticks(nb, seed=72, motor=True)
    t0 = tm.copy()
    t0["leg_q"] = tm["leg_q"] + LEG_OFFSET  # what LegController.cpp:111-113 leaves in data[leg].q (same double additions)
    t0["flags"] = 0
    a, b = _pipeline(tm), _pipeline(t0)
    np.testing.assert_array_equal(a["records"], b["records"])
    np.testing.assert_array_equal(a["forces"].view(np.uint32), b["forces"].view(np.uint32))
    np.testing.assert_array_equal(a["tau"].view(np.uint64), b["tau"].view(np.uint64))


def test_mixed_batch_is_routed_per_instance(oracle):
    """walking and standing ticks interleaved in one batch built on the device: every instance is solved by the variant of
    its own size class -- bit for bit what a pure batch of its kind gives under the host's hint -- and matches qpOASES."""
    nb = 256
    t = _mixed_ticks(nb, seed=73)
    a = _pipeline(t)
    assert (interface.status_code(a["status"]) == 0).all(), np.bincount(interface.status_code(a["status"]))
    walk, stand = _separate(t[0::2], 60), _separate(t[1::2], 120)
    np.testing.assert_array_equal(a["forces"][0::2].view(np.uint32), walk["forces"].view(np.uint32))
    np.testing.assert_array_equal(a["forces"][1::2].view(np.uint32), stand["forces"].view(np.uint32))
    np.testing.assert_array_equal(a["tau"][0::2].view(np.uint64), walk["tau"].view(np.uint64))
    np.testing.assert_array_equal(a["tau"][1::2].view(np.uint64), stand["tau"].view(np.uint64))
    want, wpd = oracle.build_records(t, H, synthetic.DT_MPC)
    np.testing.assert_array_equal(a["records"], want)
    np.testing.assert_array_equal(a["wpd"].view(np.uint64), wpd.view(np.uint64))
    ref = oracle.solve_records(want, H, synthetic.DT_MPC, synthetic.F_MAX)
    assert ref["n_bad"] == 0
    err = np.abs(a["forces"] - ref["q_soln"]).max(axis=1) / np.maximum(1.0, np.abs(ref["q_soln"]).max(axis=1))
    assert err.max() < 1e-4, err.max()
    f_ff = oracle.body_wrench(a["forces"].astype(np.float64), t["rBody"])
    np.testing.assert_array_equal(a["f_ff"].view(np.uint64), f_ff.reshape(nb, 12).view(np.uint64))
    tau = oracle.leg_torques(f_ff, t["leg_q"]).reshape(nb, 10)
    assert np.abs(a["tau"] - tau).max() <= 2.5e-13


def test_device_resident_records_without_a_hint_are_classified_on_the_device(oracle):
    """hmpc_set_device_records with no hint (max_reduced_vars = -1): the size classes are counted from the records' gait
    bytes at the head of the solve; results equal the hinted launches per class."""
    torch = _torch()
    from hector_simulation_amd import records

    nb = 192
    fw = synthetic.make_batch(nb // 2, H, "walking", seed=74, phase="random")
    fs = synthetic.make_batch(nb // 2, H, "standing", seed=75, phase="random")
    rw, rs = records.pack_records(fw, H), records.pack_records(fs, H)
    rec = np.empty((nb, rw.shape[1]), dtype=np.uint8)
    rec[0::2], rec[1::2] = rw, rs
    d_rec = torch.from_numpy(rec).cuda()
    torch.cuda.synchronize()
    mpc = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb)
    mpc.set_device_records(d_rec.data_ptr(), nb, max_reduced_vars=-1, keepalive=d_rec)
    mpc.solve()
    forces, status = mpc.download()
    mpc.close()
    assert (interface.status_code(status) == 0).all()
    for part, r in ((slice(0, None, 2), rw), (slice(1, None, 2), rs)):
        one = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb // 2)
        one.upload(r)  # host upload: the hint comes from the gait tables
        one.solve()
        f1, s1 = one.download()
        one.close()
        np.testing.assert_array_equal(forces[part].view(np.uint32), f1.view(np.uint32))
        np.testing.assert_array_equal(status[part], s1)


def test_walking_ticks_built_on_the_device_run_on_the_small_variant():
    """The point of the routing: a walking sweep built on the device must cost about what the hinted 60-variable launch
    costs (+ two launches whose workgroups leave at once), not what the 120-variable one does."""
    nb = 8192
    t = synthetic.make_ticks(nb, H, "walking", seed=76)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb)
    mpc.build_records(t, synthetic.DT_MPC)
    mpc.solve()
    mpc.download()
    ms_routed = mpc.time_solve(5)
    interface._check(mpc.L.hmpc_set_max_reduced_vars(mpc.h, 60), "hint")
    ms_small = mpc.time_solve(5)
    interface._check(mpc.L.hmpc_set_max_reduced_vars(mpc.h, 120), "hint")
    ms_big = mpc.time_solve(5)
    mpc.close()
    print(f"walking b{nb} built on the device: routed {ms_routed:.3f} ms, hinted 60-variable {ms_small:.3f} ms, "
          f"120-variable {ms_big:.3f} ms")
    assert ms_routed < 1.15 * ms_small and ms_routed < 0.75 * ms_big


# ------------------------------------------------------------------------------------------------ closed loop in HBM
# offsets (in doubles) of struct hmpc_tick_inputs, include/hector_mpc.h
O_POS, O_V, O_W, O_Q, O_RPY, O_RB, O_LEGQ, O_PFOOT, O_VDES, O_WPD, O_INTS = 0, 3, 6, 9, 13, 16, 25, 35, 41, 46, 48
MASS, TICK, TICKS_PER_STEP = 9.0, 0.005, 8


def _plant_step(torch, T, forces, IB):
    """single-rigid-body integrator (the model the MPC predicts with, SolverMPC.cpp:312-331) on the DEVICE: advances the
    tick structs in place from the step-0 wrench.  Test harness standing in for Gazebo; torch is only the array language."""
    nb = T.shape[0]
    p, v, w, rpy = T[:, O_POS:O_POS + 3], T[:, O_V:O_V + 3], T[:, O_W:O_W + 3], T[:, O_RPY:O_RPY + 3]
    u0 = forces[:, :12].double()
    F, M = u0[:, 0:6].reshape(nb, 2, 3), u0[:, 6:12].reshape(nb, 2, 3)
    Rwb = T[:, O_RB:O_RB + 9].reshape(nb, 3, 3)                 # world -> body
    pf = T[:, O_PFOOT:O_PFOOT + 6].reshape(nb, 2, 3)
    tau = (torch.cross(pf - p[:, None, :], F, dim=-1) + M).sum(dim=1)
    tb = (Rwb * tau[:, None, :]).sum(-1) / IB                   # I_w^-1 tau = R diag(1/I) R' tau
    dw = (Rwb.transpose(1, 2) * tb[:, None, :]).sum(-1)
    acc = F.sum(dim=1) / MASS
    acc[:, 2] -= 9.81
    w += TICK * dw
    v += TICK * acc
    p += TICK * v
    rpy += TICK * w
    # orientation and rBody from the new angles (synthetic.quat_from_rpy / rotation_world_to_body, on the device)
    hr, hp, hy = 0.5 * rpy[:, 0], 0.5 * rpy[:, 1], 0.5 * rpy[:, 2]
    cr, sr, cp, sp, cy, sy = hr.cos(), hr.sin(), hp.cos(), hp.sin(), hy.cos(), hy.sin()
    q = torch.stack([cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy], -1)
    T[:, O_Q:O_Q + 4] = q
    e0, e1, e2, e3 = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2),
                     2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3), 2 * (e2 * e3 - e0 * e1),
                     2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)], -1).reshape(nb, 3, 3)
    T[:, O_RB:O_RB + 9] = R.transpose(1, 2).reshape(nb, 9)


@pytest.mark.parametrize("gait", ["standing", "walking"])
def test_closed_loop_with_the_tick_state_resident_in_hbm(oracle, gait):
    """40 MPC ticks of 48 robots: hmpc_tick_solve_device is the only product call per tick; the plant advances the tick
    structs on the device, the clamped world_position_desired is fed back on the device, the gait iteration advances on the
    device.  After every tick the checker copies the state out and compares it with the reference's own caller code run on
    the same tick (records captured from its updateMPCIfNeeded: bit for bit), the reference QP solver (forces <= 1e-4) and
    its LegController (wrench bit for bit, torques <= 2.5e-13)."""
    torch = _torch()
    from oracle import caller_py

    have_caller = caller_py.available()
    nb, nticks = 48, 40
    t = synthetic.make_ticks(nb, H, gait, seed=91)
    for k in ("v_des_robot", "yaw_rate_des", "roll_des", "pitch_des"):
        t[k] = 0.0
    t["world_position_desired"] = t["position"][:, :2]
    t["gait_iteration"] = 0
    z0 = t["position"][:, 2].copy()
    d_t = _to_device(t)
    T = d_t.view(torch.float64)            # [nb, 51] doubles over the same bytes
    Ti = d_t.view(torch.int32)             # the integer tail: gait_offsets[2], gait_durations[2], gait_iteration, flags
    i_iter = 2 * O_INTS + 4
    d_tau = torch.zeros((nb, 10), dtype=torch.float64, device="cuda")
    d_ff = torch.zeros((nb, 12), dtype=torch.float64, device="cuda")
    d_wpd = torch.zeros((nb, 2), dtype=torch.float64, device="cuda")
    d_forces = torch.zeros((nb, 12 * H), dtype=torch.float32, device="cuda")
    d_status = torch.zeros((nb,), dtype=torch.int32, device="cuda")
    IB = torch.tensor([0.5413, 0.5200, 0.0691], dtype=torch.float64, device="cuda")
    mpc = interface.BatchedMPC(synthetic.DT_MPC, H, synthetic.F_MAX, nb)
    mpc.set_device_outputs(d_forces.data_ptr(), d_status.data_ptr(), keepalive=(d_forces, d_status))
    mpc.set_device_repair(True)            # flagged instances (none expected) would be repaired on the device too
    stream = torch.cuda.current_stream().cuda_stream
    caller = caller_py.Caller() if have_caller else None
    gait_number = 1 if gait == "standing" else 2
    for k in range(nticks):
        Ti[:, i_iter] = (k // TICKS_PER_STEP) % H                                   # device op: the gait table advances
        tick_in = d_t.cpu().numpy().view(interface.TICK_DTYPE).reshape(nb).copy()   # checker's copy of the tick about to run
        # ---- the product path of one tick: a single call, everything stays in HBM
        mpc.tick_solve_device(d_t.data_ptr(), nb, synthetic.DT_MPC, d_tau.data_ptr(), d_ff.data_ptr(), d_wpd.data_ptr(), stream)
        T[:, O_WPD:O_WPD + 2] = d_wpd                                               # device op: clamp fed back
        # ---- checker (host copies; the loop above does not depend on any of it)
        forces, status = d_forces.cpu().numpy(), d_status.cpu().numpy().astype(np.uint32)
        rec = mpc.download_records()
        want, wpd_want = oracle.build_records(tick_in, H, synthetic.DT_MPC)
        np.testing.assert_array_equal(rec, want, err_msg=f"records, tick {k}")
        np.testing.assert_array_equal(d_wpd.cpu().numpy().view(np.uint64), wpd_want.view(np.uint64))
        assert (interface.status_code(status) == 0).all(), (k, np.bincount(interface.status_code(status)))
        ref = oracle.solve_records(want, H, synthetic.DT_MPC, synthetic.F_MAX)
        assert ref["n_bad"] == 0
        err = np.abs(forces - ref["q_soln"]).max(axis=1) / np.maximum(1.0, np.abs(ref["q_soln"]).max(axis=1))
        assert err.max() < 1e-4, (k, err.max())
        f_ff = oracle.body_wrench(forces.astype(np.float64), tick_in["rBody"]).reshape(nb, 12)
        np.testing.assert_array_equal(d_ff.cpu().numpy().view(np.uint64), f_ff.view(np.uint64))
        tau_want = oracle.leg_torques(f_ff, tick_in["leg_q"] - LEG_OFFSET).reshape(nb, 10)
        assert np.abs(d_tau.cpu().numpy() - tau_want).max() <= 2e-11  # (leg_q - offset + offset: one rounding in the argument)
        if have_caller and k % 4 == 0:
            # the reference's OWN updateMPCIfNeeded on the same tick: captured update_problem_data arguments == device records
            from tests.test_caller_reference import _assert_records_equal, _record_from_capture

            for j in range(0, nb, 8):
                caller.set_solution(forces[j].astype(np.float64))
                cap = caller_py.tick_through_reference(caller, tick_in[j], gait_number)
                _assert_records_equal(_record_from_capture(cap), rec[j], f"tick {k} robot {j}")
                np.testing.assert_array_equal(cap["world_position_desired"][:2].view(np.uint64), wpd_want[j].view(np.uint64))
                # (as values: a swing leg's exact-zero force gives -0.0 as a bare product and +0.0 under the Eigen stand-in,
                #  whose products start from +0 -- the documented sign-of-zero exception of DESIGN.md section 2)
                np.testing.assert_array_equal(cap["f_ff"].reshape(12), f_ff[j])
        # ---- plant, on the device
        _plant_step(torch, T, d_forces, IB)
    torch.cuda.synchronize()
    if caller:
        caller.close()
    mpc.close()
    end = d_t.cpu().numpy().view(interface.TICK_DTYPE).reshape(nb)
    if gait == "standing":
        p, rpy, w, v = end["position"], end["rpy"], end["omegaWorld"], end["vWorld"]
        assert np.abs(p[:, 2] - synthetic.NOMINAL_HEIGHT).max() < np.abs(z0 - synthetic.NOMINAL_HEIGHT).max() + 0.01
        assert np.abs(p[:, 2] - synthetic.NOMINAL_HEIGHT).max() < 0.04
        assert np.abs(rpy[:, :2]).max() < 0.15 and np.abs(w).max() < 1.0 and np.abs(v).max() < 0.5
