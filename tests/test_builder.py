"""device-side record builder (ConvexMPCLocomotion.cpp:283-406 +
GaitGenerator.cpp:85-103), body-frame wrench consumer (ConvexMPCLocomotion.cpp:419-440).  Bit-exact bar: binary64
arithmetic in the reference's order, narrowed once to the record's float32/u8."""
import math

import numpy as np
import pytest

from hector_simulation_amd import interface, records, synthetic


def python_build_record(t, h, dt):
    """Independent scalar-Python restatement (IEEE doubles, same operation order) used to pin the C oracle."""
    PI = 3.14159265359
    q = [float(x) for x in t["leg_q"]]
    for leg in (0, 5):
        q[leg + 2] += 0.3 * PI
        q[leg + 3] -= 0.6 * PI
        q[leg + 4] += 0.3 * PI
    q = [math.fmod(x, 2 * PI) for x in q]
    p = [float(x) for x in t["position"]]
    r = [float(t["pFoot"][3 * (i % 2) + i // 2]) - p[i // 2] for i in range(6)]
    rb = [float(x) for x in t["rBody"]]
    vr = [float(x) for x in t["v_des_robot"]]
    vdw = [(rb[0 * 3 + i] * vr[0] + rb[1 * 3 + i] * vr[1]) + rb[2 * 3 + i] * 0.0 for i in range(3)]
    xs, ys = float(t["world_position_desired"][0]), float(t["world_position_desired"][1])
    if xs - p[0] > .05: xs = p[0] + .05
    if p[0] - xs > .05: xs = p[0] - .05
    if ys - p[1] > .05: ys = p[1] + .05
    if p[1] - ys > .05: ys = p[1] - .05
    yr = float(t["yaw_rate_des"])
    init = [float(t["roll_des"]), float(t["pitch_des"]), 0.0, xs, ys, 0.55, 0.0, 0.0, yr, vdw[0], vdw[1], 0.0]
    traj = []
    rpy = [float(x) for x in t["rpy"]]
    for i in range(h):
        row = list(init)
        if i == 0:
            row[0:3] = rpy
            row[3:6] = p
        else:
            row[3] = (init[3] if vdw[0] == 0 else p[0]) + i * dt * vdw[0]
            row[4] = (init[4] if vdw[1] == 0 else p[1]) + i * dt * vdw[1]
            row[2] = init[2] if yr == 0 else rpy[2] + i * dt * yr
        traj += row
    gait = synthetic.mpc_gait(h, t["gait_offsets"], t["gait_durations"], int(t["gait_iteration"]))
    f = dict(p=[p], v=[t["vWorld"]], q=[t["orientation"]], w=[t["omegaWorld"]], r=[r], joint_angles=[q], yaw=[rpy[2]],
             weights=[synthetic.Q_WEIGHTS], Alpha_K=[synthetic.ALPHA], traj=[traj], gait=[gait])
    return records.pack_records(f, h)[0], (xs, ys)


@pytest.mark.parametrize("gait,h", [("walking", 10), ("standing", 10), ("walking", 20)])
def test_oracle_builder_matches_python(oracle, gait, h):
    t = synthetic.make_ticks(12, h, gait, seed=31)
    rec, wpd = oracle.build_records(t, h, synthetic.DT_MPC)
    for k in range(12):
        want, (xs, ys) = python_build_record(t[k], h, synthetic.DT_MPC)
        np.testing.assert_array_equal(rec[k], want)
        assert wpd[k, 0] == xs and wpd[k, 1] == ys
    # the clamp is exercised: some desired positions were more than 5 cm away
    assert (np.abs(wpd - t["position"][:, :2]).max(axis=1) <= 0.05 + 1e-15).all()
    assert (np.abs(t["world_position_desired"] - t["position"][:, :2]).max(axis=1) > 0.05).any()


def test_oracle_gait_matches_reference_semantics(oracle):
    for h in (10, 20):
        for it in range(h):
            np.testing.assert_array_equal(oracle.mpc_gait(h, (0, h // 2), (h // 2, h - h // 2), it),
                                          synthetic.mpc_gait(h, (0, h // 2), (h // 2, h - h // 2), it))
    assert oracle.mpc_gait(10, (0, 0), (10, 10), 3).all()


def test_oracle_body_wrench(oracle):
    rng = np.random.default_rng(5)
    q = rng.normal(size=(4, 120)) * 30
    quat = synthetic.quat_from_rpy(rng.uniform(-.2, .2, 4), rng.uniform(-.2, .2, 4), rng.uniform(-1, 1, 4))
    rb = synthetic.rotation_world_to_body(quat)
    f = oracle.body_wrench(q, rb)
    for k in range(4):
        for leg in range(2):
            np.testing.assert_allclose(f[k, leg, :3], -rb[k] @ q[k, 3 * leg:3 * leg + 3], rtol=1e-14, atol=1e-13)
            np.testing.assert_allclose(f[k, leg, 3:], -rb[k] @ q[k, 6 + 3 * leg:9 + 3 * leg], rtol=1e-14, atol=1e-13)


def test_oracle_jacobian_structure(oracle):
    rng = np.random.default_rng(3)
    for leg in (0, 1):
        q = rng.uniform(-0.3, 0.3, 5)
        J = oracle.leg_jacobian(q, leg)
        s0, c0, s1, c1 = math.sin(q[0]), math.cos(q[0]), math.sin(q[1]), math.cos(q[1])
        # angular rows: joint axes of a Rz(q0) Rx(q1) Ry(...) chain (LegController.cpp:134-165 rows 3-5)
        np.testing.assert_allclose(J[3:, 0], [0, 0, 1], atol=1e-15)
        np.testing.assert_allclose(J[3:, 1], [c0, s0, 0], atol=1e-15)
        for c in (2, 3, 4):
            np.testing.assert_allclose(J[3:, c], [-c1 * s0, c0 * c1, s1], atol=1e-15)
        # translational columns 2..4 shrink with the lever arm: |J_v(:,4)| = 0.04 (the 4 cm foot link)
        assert abs(np.linalg.norm(J[:3, 4]) - 0.04) < 1e-12
        # column 3 = column 4 + 0.22 m shank contribution, column 2 adds the 0.22 m thigh: norms are ordered
        assert np.linalg.norm(J[:3, 2]) > np.linalg.norm(J[:3, 4])
    # integrability: the translational columns are the gradient of ONE foot position, so d J(:,j)/d q_k == d J(:,k)/d q_j
    # (a transcription slip in any entry breaks this); central differences on the restated Jacobian itself
    for leg in (0, 1):
        q = rng.uniform(-0.4, 0.4, 5)
        hstep = 1e-5
        D = np.zeros((5, 5, 3))
        for k in range(5):
            dq = np.zeros(5)
            dq[k] = hstep
            D[:, k, :] = ((oracle.leg_jacobian(q + dq, leg)[:3] - oracle.leg_jacobian(q - dq, leg)[:3]) / (2 * hstep)).T
        for j in range(5):
            for k in range(5):
                np.testing.assert_allclose(D[j, k], D[k, j], atol=2e-8)
    f = rng.normal(size=(3, 12)) * 20
    lq = rng.uniform(-0.3, 0.3, (3, 10))
    tau = oracle.leg_torques(f, lq)
    for k in range(3):
        for leg in (0, 1):
            J = oracle.leg_jacobian(lq[k, 5 * leg:5 * leg + 5], leg)
            np.testing.assert_allclose(tau[k, leg], J.T @ f[k, 6 * leg:6 * leg + 6], rtol=1e-13, atol=1e-13)


def _legcontroller_mutation(q_motor):
    """common/LegController.cpp:48-52, 108-113: updateData() passes data[leg].q BY REFERENCE to
    computeLegJacobianAndPosition, which adds its offsets (pi ~ 3.14159) in place -- what the MPC reads afterwards."""
    q = np.array(q_motor, dtype=np.float64, copy=True)
    for leg in (0, 1):
        q[..., 5 * leg + 2] = q[..., 5 * leg + 2] + 0.3 * 3.14159
        q[..., 5 * leg + 3] = q[..., 5 * leg + 3] - 0.6 * 3.14159
        q[..., 5 * leg + 4] = q[..., 5 * leg + 4] + 0.3 * 3.14159
    return q


def test_motor_angle_flag_models_the_legcontroller_mutation(oracle):
    """The live reference feeds the MPC data[leg].q AFTER the LegController's in-place offset (SURVEY A.9 "triple
    offset").  A tick carrying raw motor angles with HMPC_TICK_LEG_Q_MOTOR builds the same record, bit for bit, as a tick
    carrying the mutated angles with flags = 0; the record's joint field is motor + 0.3*3.14159 + 0.3*PI (fmod 2 PI)."""
    h = 10
    t = synthetic.make_ticks(24, h, "walking", seed=5)
    q_motor = np.array(t["leg_q"], copy=True)          # treat the synthetic angles as what the motors report
    t_mut = t.copy()
    t_mut["leg_q"] = _legcontroller_mutation(q_motor)  # what updateMPCIfNeeded reads on the live path
    t_mot = t.copy()
    t_mot["flags"] = 1
    rec_mut, _ = oracle.build_records(t_mut, h, synthetic.DT_MPC)
    rec_mot, _ = oracle.build_records(t_mot, h, synthetic.DT_MPC)
    np.testing.assert_array_equal(rec_mot, rec_mut)
    rec_plain, _ = oracle.build_records(t, h, synthetic.DT_MPC)  # same angles WITHOUT the flag: a different record
    ja = lambda r: records.unpack_records(r, h)["joint_angles"].astype(np.float64)
    PI = 3.14159265359
    want = q_motor.copy()
    for leg in (0, 1):
        for j, c in ((2, 0.3), (3, -0.6), (4, 0.3)):
            want[:, 5 * leg + j] = np.fmod((want[:, 5 * leg + j] + c * 3.14159) + c * PI, 2 * PI)
    np.testing.assert_array_equal(ja(rec_mot), want.astype(np.float32).astype(np.float64))
    assert np.abs(ja(rec_plain)[:, 2] - ja(rec_mot)[:, 2]).min() > 0.9  # ~0.3*pi: the mistake the flag exists to prevent
    # the Jacobian side (hmpc_leg_torques) takes the MOTOR angle: it applies LegController.cpp:111-113 itself
    J = oracle.leg_jacobian(q_motor[0, :5], 0)
    assert np.isfinite(J).all()


@pytest.mark.gpu
def test_device_builder_motor_angle_flag_bitwise(oracle):
    h, nb = 10, 40
    t = synthetic.make_ticks(nb, h, "walking", seed=9)
    t["flags"] = 1
    want, _ = oracle.build_records(t, h, synthetic.DT_MPC)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    mpc.build_records(t, synthetic.DT_MPC)
    np.testing.assert_array_equal(mpc.download_records(), want)
    t2 = t.copy()
    t2["flags"] = 0
    t2["leg_q"] = _legcontroller_mutation(t["leg_q"])
    mpc.build_records(t2, synthetic.DT_MPC)
    np.testing.assert_array_equal(mpc.download_records(), want)
    mpc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gait,h,nb", [("walking", 10, 64), ("standing", 10, 33), ("walking", 20, 16)])
def test_device_builder_bitwise(oracle, gait, h, nb):
    t = synthetic.make_ticks(nb, h, gait, seed=77)
    want, wpd_want = oracle.build_records(t, h, synthetic.DT_MPC)
    mpc = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    wpd = mpc.build_records(t, synthetic.DT_MPC)
    got = mpc.download_records()
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(wpd.view(np.uint64), wpd_want.view(np.uint64))
    # end to end on the device: build -> solve -> body-frame wrench, against the CPU reference path on the same ticks
    mpc.solve()
    forces, status = mpc.download()
    assert (interface.status_code(status) == 0).all()
    ref = oracle.solve_records(want, h, synthetic.DT_MPC, synthetic.F_MAX)
    assert ref["n_bad"] == 0
    err = np.abs(forces - ref["q_soln"]).max(axis=1) / np.maximum(1.0, np.abs(ref["q_soln"]).max(axis=1))
    assert err.max() < 1e-4
    fff = mpc.body_wrench(t["rBody"])
    want_f = oracle.body_wrench(forces.astype(np.float64), t["rBody"])
    np.testing.assert_array_equal(fff.view(np.uint64), want_f.view(np.uint64))
    fff2, tau = mpc.leg_torques(t["rBody"], t["leg_q"])
    np.testing.assert_array_equal(fff2.view(np.uint64), want_f.view(np.uint64))
    want_tau = oracle.leg_torques(want_f, t["leg_q"])
    np.testing.assert_array_equal(tau.view(np.uint64), want_tau.view(np.uint64))
    mpc.close()
