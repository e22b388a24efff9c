"""Synthetic MPC-tick inputs for the BASELINE.json configs (distribution: SURVEY.md section 8d).

Everything here restates what the reference's caller builds before it crosses the C boundary
(ConvexMPC/ConvexMPCLocomotion.cpp:283-406: Q/alpha at :321-322, the 10-step reference trajectory at :351-406,
foot vectors r[i] = pFoot[i%2][i/2] - position[i/2] at :313-316) and the gait tables of
ConvexMPC/GaitGenerator.cpp:85-103 (``Gait(10,(0,5),(5,5))`` walking, ``Gait(10,(0,0),(10,10))`` standing,
ConvexMPCLocomotion.cpp:16-17).  RNG = numpy.random.default_rng(seed).
"""
from __future__ import annotations

import numpy as np

Q_WEIGHTS = np.array([100, 100, 250, 200, 200, 300, 1, 1, 1, 1, 1, 1], dtype=np.float64)
ALPHA = np.array([1e-4, 1e-4, 5e-4, 1e-4, 1e-4, 5e-4, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2], dtype=np.float64)
DT_MPC = 0.001 * 40  # FSMState_Walking.cpp:5
F_MAX = 500.0  # ConvexMPCLocomotion.cpp:410
NOMINAL_HEIGHT = 0.55


def mpc_gait(horizon: int, offsets, durations, iteration: int, n_iter: int | None = None) -> np.ndarray:
    """GaitGenerator.cpp:85-103 for one phase ``iteration``; returns int table [2*horizon], [2*i+leg]."""
    n_iter = horizon if n_iter is None else n_iter
    t = np.zeros(2 * horizon, dtype=np.int32)
    for i in range(horizon):
        it = (i + iteration) % n_iter
        for j in range(2):
            prog = it - offsets[j]
            if prog < 0:
                prog += n_iter
            t[2 * i + j] = 1 if prog < durations[j] else 0
    return t


def quat_from_rpy(roll, pitch, yaw):
    cr, sr = np.cos(roll / 2), np.sin(roll / 2)
    cp, sp = np.cos(pitch / 2), np.sin(pitch / 2)
    cy, sy = np.cos(yaw / 2), np.sin(yaw / 2)
    return np.stack(
        [cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy],
        axis=-1,
    )


def make_batch(batch: int, horizon: int = 10, gait: str = "walking", seed: int = 0, randomize: bool = True,
               phase: int | str = 0, yaw_rate_cmd: bool = False) -> dict:
    """Returns the field dict ``records.pack_records`` accepts (float64 / int32 arrays, leading dim = batch).

    gait: "standing" (both feet in stance every step), "walking" (reference walking gait, single support),
          "single" (one leg in stance for the whole horizon; the h=20 stress config),
          "mixed" (walking offsets with 70 % duty -> double-support phases).
    phase: int (fixed gait iteration) or "random".
    """
    rng = np.random.default_rng(seed)
    b, h = batch, horizon
    z = lambda *s: np.zeros((b,) + s)
    u = (lambda lo, hi, *s: rng.uniform(lo, hi, (b,) + s)) if randomize else (lambda lo, hi, *s: np.zeros((b,) + s))
    rpy = np.stack([u(-0.1, 0.1), u(-0.1, 0.1), u(-0.1, 0.1)], -1)
    p = np.stack([u(-0.02, 0.02), u(-0.02, 0.02), NOMINAL_HEIGHT + u(-0.03, 0.03)], -1)
    v = u(-0.3, 0.3, 3)
    w = u(-0.5, 0.5, 3)
    q = quat_from_rpy(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    foot = z(2, 3)  # [leg][axis], world frame relative to body
    foot[:, 0, 1], foot[:, 1, 1] = 0.06, -0.06
    foot[:, :, 0] += u(-0.1, 0.1, 2)
    foot[:, :, 1] += u(-0.03, 0.03, 2)
    foot[:, :, 2] = -p[:, 2:3]
    r = np.stack([foot[:, i % 2, i // 2] for i in range(6)], -1)  # r[2*axis+leg]
    joints = u(-0.15, 0.15, 10)
    vx_cmd = u(-0.5, 0.5)
    yaw_rate = u(-0.3, 0.3) if yaw_rate_cmd else z()
    # reference trajectory (ConvexMPCLocomotion.cpp:351-406); xStart/yStart = current xy (desired within clamp)
    traj = z(h, 12)
    traj[:, :, 3], traj[:, :, 4], traj[:, :, 5] = p[:, 0:1], p[:, 1:2], NOMINAL_HEIGHT
    traj[:, :, 8] = yaw_rate[:, None]
    traj[:, :, 9] = vx_cmd[:, None]
    traj[:, 0, 0:3] = rpy
    traj[:, 0, 3:6] = p
    for i in range(1, h):
        traj[:, i, 3] = p[:, 0] + i * DT_MPC * vx_cmd
        if yaw_rate_cmd:
            traj[:, i, 2] = rpy[:, 2] + i * DT_MPC * yaw_rate
    # gait tables
    if isinstance(phase, str):
        ph = rng.integers(0, h, size=b)
    else:
        ph = np.full(b, int(phase))
    half = h // 2
    tables = {}
    g = np.zeros((b, 2 * h), dtype=np.int32)
    for k in range(b):
        key = int(ph[k])
        if key not in tables:
            if gait == "standing":
                tables[key] = mpc_gait(h, (0, 0), (h, h), key)
            elif gait == "walking":
                tables[key] = mpc_gait(h, (0, half), (half, h - half), key)
            elif gait == "single":
                tables[key] = np.tile(np.array([1, 0] if key % 2 == 0 else [0, 1], dtype=np.int32), h)
            elif gait == "mixed":
                dur = (7 * h + 9) // 10
                tables[key] = mpc_gait(h, (0, half), (dur, dur), key)
            else:
                raise ValueError(gait)
        g[k] = tables[key]
    return dict(p=p, v=v, q=q, w=w, r=r, joint_angles=joints, yaw=rpy[:, 2], weights=np.tile(Q_WEIGHTS, (b, 1)),
                Alpha_K=np.tile(ALPHA, (b, 1)), traj=traj.reshape(b, 12 * h), gait=g)


def rpy_from_quat(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([np.arctan2(2 * (w * x + y * z), 1 - 2 * (x * x + y * y)),
                     np.arcsin(np.clip(2 * (w * y - x * z), -1, 1)),
                     np.arctan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))], -1)


def advance_tick(fields: dict, horizon: int, seed: int = 0, dt_tick: float = 0.005, noise: float = 1.0) -> dict:
    """The same instances one MPC tick later (the reference re-solves every 5 ms, ConvexMPCLocomotion.cpp:277, while
    a horizon step is 40 ms): the body moved by v dt and rotated by w dt, velocities drifted, the stance feet stayed
    where they were in the world (so r shrinks by the body motion), the joints crept, and the reference trajectory
    is re-anchored at the new position exactly as ``make_batch`` builds it.  Gait table unchanged (replace
    ``out["gait"]`` to model a phase advance).  Used to exercise the warm start across ticks."""
    rng = np.random.default_rng(seed + 104729)
    b, h = np.asarray(fields["p"]).shape[0], horizon
    nz = lambda sc, *sh: noise * sc * rng.standard_normal((b,) + sh)
    out = {k: np.array(v, copy=True) for k, v in fields.items()}
    dp = fields["v"] * dt_tick
    out["p"] = fields["p"] + dp
    out["v"] = fields["v"] + nz(0.02, 3)
    out["w"] = fields["w"] + nz(0.03, 3)
    rpy = rpy_from_quat(np.asarray(fields["q"], dtype=np.float64)) + fields["w"] * dt_tick + nz(5e-4, 3)
    out["q"] = quat_from_rpy(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    out["yaw"] = rpy[:, 2]
    nc = np.asarray(fields["r"]).shape[1] // 3
    out["r"] = (np.asarray(fields["r"]).reshape(b, 3, nc) - dp[:, :, None]).reshape(b, 3 * nc)
    out["joint_angles"] = fields["joint_angles"] + nz(2e-3, 10)
    old = np.asarray(fields["traj"]).reshape(b, h, 12)
    traj = old.copy()
    vx = old[:, 0, 9]
    traj[:, 0, 0:3] = rpy
    traj[:, 0, 3:6] = out["p"]
    traj[:, :, 3] = out["p"][:, 0:1] + np.arange(h)[None, :] * DT_MPC * vx[:, None]
    traj[:, 1:, 4] = out["p"][:, 1:2]
    if np.any(old[:, :, 8] != 0):
        traj[:, 1:, 2] = rpy[:, 2:3] + np.arange(1, h)[None, :] * DT_MPC * old[:, 1:, 8]
    out["traj"] = traj.reshape(b, 12 * h)
    return out


def hard_batch(batch: int, horizon: int = 10, gait: str = "standing", seed: int = 17, scale: float = 1.0) -> dict:
    """``make_batch`` (random phase, yaw-rate command) with the attitude, velocity, angular-velocity, joint and commanded-
    velocity ranges of SURVEY.md section 8d multiplied by ``scale`` -- the off-nominal stress rows (scripts/stress.py,
    tests/test_gpu_robustness.py, bench.py's ``range_scale_*`` side configs): at 3x / 6x / 10x the final working sets of the
    2-contact QP grow from ~50 to 72 / 91 / 95 rows and a cold qpOASES run takes up to 104 / 193 / 242 iterations."""
    f = make_batch(batch, horizon, gait, seed=seed, phase="random", yaw_rate_cmd=True)
    rng = np.random.default_rng(seed + 1)
    rpy = rng.uniform(-0.1 * scale, 0.1 * scale, (batch, 3))
    f["q"] = quat_from_rpy(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    f["v"] = rng.uniform(-0.3 * scale, 0.3 * scale, (batch, 3))
    f["w"] = rng.uniform(-0.5 * scale, 0.5 * scale, (batch, 3))
    f["joint_angles"] = rng.uniform(-0.15 * scale, 0.15 * scale, (batch, 10))
    tr = f["traj"].reshape(batch, horizon, 12)
    tr[:, :, 9] *= scale
    f["traj"] = tr.reshape(batch, -1)
    return f


ALPHA3 = np.array([1e-4, 1e-4, 5e-4] * 3 + [1e-2] * 9, dtype=np.float64)
F_MAX_HAND = 150.0


def rot_from_rpy(roll, pitch, yaw):
    """Rz(yaw) Ry(pitch) Rx(roll), batched -> [..., 3, 3]."""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    R = np.stack([cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
                  sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
                  -sp, cp * sr, cp * cr], axis=-1)
    return R.reshape(np.shape(roll) + (3, 3))


def make_batch3(batch: int, horizon: int = 10, gait: str = "standing", seed: int = 0, randomize: bool = True,
                phase: int | str = 0, hand: str = "contact") -> dict:
    """Three-contact (two feet + one hand) instances of the extension formulation -- BASELINE config 5's shape
    (180 variables x 240 rows at h = 10 when every contact is in stance).  The feet and the body come from
    ``make_batch`` (same seed -> same values); the hand rests on a surface in front of the body.

    hand: "contact" (in stance every step), "window" (in stance for a random window of the horizon), "off" (never)."""
    f = make_batch(batch, horizon, gait, seed, randomize, phase)
    rng = np.random.default_rng(seed + 7919)
    b, h = batch, horizon
    u = (lambda lo, hi, *s: rng.uniform(lo, hi, (b,) + s)) if randomize else (lambda lo, hi, *s: np.zeros((b,) + s))
    rh = np.stack([0.25 + u(-0.05, 0.05), -0.15 + u(-0.05, 0.05), 0.10 + u(-0.05, 0.05)], -1)  # hand relative to the body
    r2 = f["r"].reshape(b, 3, 2)
    r3 = np.concatenate([r2, rh[:, :, None]], axis=2).reshape(b, 9)  # r[3*axis + contact]
    Rh = rot_from_rpy(u(-0.2, 0.2), u(-0.2, 0.2), u(-0.4, 0.4)).reshape(b, 9)
    g2 = f["gait"].reshape(b, h, 2)
    if hand == "contact":
        gh = np.ones((b, h), dtype=np.int32)
    elif hand == "off":
        gh = np.zeros((b, h), dtype=np.int32)
    elif hand == "window":
        a = rng.integers(0, h, size=b)
        ln = rng.integers(1, h + 1, size=b)
        idx = np.arange(h)[None, :]
        gh = ((idx >= a[:, None]) & (idx < (a + ln)[:, None])).astype(np.int32)
    else:
        raise ValueError(hand)
    g3 = np.concatenate([g2, gh[:, :, None]], axis=2).reshape(b, 3 * h)
    out = dict(f)
    out.update(r=r3, Alpha_K=np.tile(ALPHA3, (b, 1)), gait=g3, Rhand=Rh, f_max_hand=np.full(b, F_MAX_HAND))
    return out


CONFIGS = {
    # BASELINE.json configs -> generator arguments (seed = config index)
    "cfg1_stand_single": dict(batch=1, horizon=10, gait="standing", seed=1, randomize=False),
    "cfg2_walk_1024": dict(batch=1024, horizon=10, gait="walking", seed=2, phase=0),
    "cfg3_walk_sweep_65536": dict(batch=65536, horizon=10, gait="walking", seed=3, phase="random"),
    "cfg4_h20_single_4096": dict(batch=4096, horizon=20, gait="single", seed=4, phase="random"),
    "metric_2contact_1024": dict(batch=1024, horizon=10, gait="standing", seed=6),
}
# BASELINE config 5 (extension: make_batch3 arguments); 8192 instances = 2048 per GPU on 4 GPUs
CONFIG5 = dict(batch=8192, horizon=10, gait="standing", seed=5, hand="contact")


def rotation_world_to_body(q):
    """orientation_tools.h:182-200 quaternionToRotationMatrix (returns the transposed matrix = world -> body)."""
    e0, e1, e2, e3 = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.stack([1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2),
                  2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3), 2 * (e2 * e3 - e0 * e1),
                  2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)], axis=-1)
    R = R.reshape(q.shape[:-1] + (3, 3))
    return np.swapaxes(R, -1, -2)


def make_ticks(batch: int, horizon: int = 10, gait: str = "walking", seed: int = 0, randomize: bool = True):
    """What ConvexMPCLocomotion::updateMPCIfNeeded reads for one tick (ConvexMPCLocomotion.cpp:283-406), as a
    structured array laid out like ``struct hmpc_tick_inputs`` -- the input of the device-side record builder."""
    from .interface import TICK_DTYPE

    rng = np.random.default_rng(seed)
    b, h = batch, horizon
    u = (lambda lo, hi, *s: rng.uniform(lo, hi, (b,) + s)) if randomize else (lambda lo, hi, *s: np.zeros((b,) + s))
    t = np.zeros(b, dtype=TICK_DTYPE)
    rpy = np.stack([u(-0.1, 0.1), u(-0.1, 0.1), u(-0.5, 0.5)], -1)
    t["rpy"] = rpy
    t["orientation"] = quat_from_rpy(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    t["rBody"] = rotation_world_to_body(t["orientation"]).reshape(b, 9)
    t["position"] = np.stack([u(-2, 2), u(-2, 2), NOMINAL_HEIGHT + u(-0.03, 0.03)], -1)
    t["vWorld"] = u(-0.3, 0.3, 3)
    t["omegaWorld"] = u(-0.5, 0.5, 3)
    foot = np.zeros((b, 2, 3))
    foot[:, 0, 1], foot[:, 1, 1] = 0.06, -0.06
    foot[:, :, 0] += u(-0.1, 0.1, 2)
    foot[:, :, 1] += u(-0.03, 0.03, 2)
    foot[:, :, 2] = -t["position"][:, 2:3]
    t["pFoot"] = (foot + t["position"][:, None, :]).reshape(b, 6)
    t["leg_q"] = u(-0.15, 0.15, 10) + np.tile([0, 0, -0.3 * 3.14159265359, 0.6 * 3.14159265359, -0.3 * 3.14159265359], 2)
    t["v_des_robot"] = u(-0.5, 0.5, 2) * (rng.random((b, 2)) > 0.25)     # some commands are exactly zero
    t["yaw_rate_des"] = u(-0.3, 0.3) * (rng.random(b) > 0.5)
    t["roll_des"], t["pitch_des"] = u(-0.02, 0.02), u(-0.02, 0.02)
    t["world_position_desired"] = t["position"][:, :2] + u(-0.12, 0.12, 2)  # beyond the 5 cm clamp on some
    half = h // 2
    if gait == "standing":
        t["gait_offsets"], t["gait_durations"] = (0, 0), (h, h)
    else:
        t["gait_offsets"], t["gait_durations"] = (0, half), (half, h - half)
    t["gait_iteration"] = rng.integers(0, h, size=b)
    return t
