"""Independent float64 numpy restatement of the reference QP, used only to cross-check
the C oracle (oracle/hmpc_oracle.c) -- a second implementation written from the mathematical specification:
ConvexMPC/SolverMPC.cpp:302-342 (model), :133-193 (discretise/stack), :450-570 (cost, bounds, constraint block),
:589-697 (swing elimination).  Dense matrices throughout, no structure exploited, libm trig."""
from __future__ import annotations

import numpy as np

BIG = float(np.float32(5e10))


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1.0, 0, 0], [0, c, -s], [0, s, c]])


def _ry(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1.0, 0], [-s, 0, c]])


def _cross(r):
    return np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0.0]])


def assemble(f: dict, h: int, dt: float, f_max: float, nc: int = 2) -> dict:
    """f: one instance's fields (float32-narrowed values, as the C boundary sees them).

    nc = 3: the hand-contact extension written from its specification ("cfg-5 default extension"):
    u_k = [F_L F_R F_H M_L M_R M_H]; B_ct gains the hand's columns; the hand gets the left foot's 8-row block in the contact
    frame ``Rhand`` with its own force cap ``f_max_hand`` and stance flag gait[3i+2]."""
    U, C8 = 6 * nc, 8 * nc
    p, v, q, w, r = (np.asarray(f[k], dtype=np.float64) for k in ("p", "v", "q", "w", "r"))
    ja = np.asarray(f["joint_angles"], dtype=np.float64).copy()
    PI = 3.14159265359
    for leg in range(2):
        ja[5 * leg + 2] += 0.3 * PI
        ja[5 * leg + 3] -= 0.6 * PI
        ja[5 * leg + 4] += 0.3 * PI
    ja = np.fmod(ja, 2 * PI)
    qw, qx, qy, qz = q
    R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy)],
                  [2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx)],
                  [2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)]])
    roll = np.arctan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy))
    pitch = np.arcsin(min(2 * (qw * qy - qx * qz), 0.99999))
    yaw = np.arctan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz))
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Rb = np.array([[cy * cp, -sy, 0], [sy * cp, cy, 0], [-sp, 0, 1.0]])
    x0 = np.concatenate([[roll, pitch, yaw], p, w, v, [9.81]])
    Iw = R @ np.diag([0.5413, 0.5200, 0.0691]) @ R.T
    Iinv = np.linalg.inv(Iw)
    A = np.zeros((13, 13))
    A[0:3, 6:9] = np.linalg.inv(Rb)
    A[3:6, 9:12] = np.eye(3)
    A[11, 12] = -1.0
    B = np.zeros((13, U))
    for leg in range(nc):
        rl = np.array([r[0 * nc + leg], r[1 * nc + leg], r[2 * nc + leg]])
        B[6:9, 3 * leg:3 * leg + 3] = Iinv @ _cross(rl)
        B[6:9, 3 * nc + 3 * leg:3 * nc + 3 * leg + 3] = Iinv
        B[9:12, 3 * leg:3 * leg + 3] = np.eye(3) / 9.0
    Acd = np.eye(13) + dt * A
    Bcd = dt * B
    Aqp = np.zeros((13 * h, 13))
    Bqp = np.zeros((13 * h, U * h))
    pw = [np.eye(13)]
    for _ in range(h):
        pw.append(pw[-1] @ Acd)
    for i in range(h):
        Aqp[13 * i:13 * i + 13] = pw[i + 1]
        for j in range(i + 1):
            Bqp[13 * i:13 * i + 13, U * j:U * j + U] = pw[i - j] @ Bcd
    S = np.diag(np.tile(np.concatenate([np.asarray(f["weights"], dtype=np.float64), [0.0]]), h))
    Xd = np.zeros(13 * h)
    traj = np.asarray(f["traj"], dtype=np.float64)
    for i in range(h):
        Xd[13 * i:13 * i + 12] = traj[12 * i:12 * i + 12]
    alpha = np.tile(np.asarray(f["Alpha_K"], dtype=np.float64), h)
    H = 2 * (Bqp.T @ S @ Bqp + np.diag(alpha))
    g = 2 * Bqp.T @ S @ (Aqp @ x0 - Xd)
    mu, lt, lh = 2.0, float(np.float32(0.09)), float(np.float32(0.06))
    Fc = np.zeros((C8, U))
    for leg in range(nc):
        if leg < 2:
            a = ja[5 * leg:5 * leg + 5]
            Rf = _rz(a[0]) @ _rx(a[1]) @ _ry(a[2] + a[3] + a[4])
        else:
            Rf = np.asarray(f["Rhand"], dtype=np.float64).reshape(3, 3)
        T = Rf.T @ R.T  # rows: contact-frame x, y, z axes in world
        rows = Fc[8 * leg:8 * leg + 8]
        cf, cm = slice(3 * leg, 3 * leg + 3), slice(3 * nc + 3 * leg, 3 * nc + 3 * leg + 3)
        rows[0, cf] = [-mu, 0, 1]
        rows[1, cf] = [mu, 0, 1]
        rows[2, cf] = [0, -mu, 1]
        rows[3, cf] = [0, mu, 1]
        rows[4, cm] = T[0]
        rows[5, cf] = -lt * T[2]
        rows[5, cm] = T[1]
        rows[6, cf] = -lh * T[2]
        rows[6, cm] = T[1] if leg == 1 else -T[1]
        rows[7, cf] = [0, 0, 2]
    gait = np.asarray(f["gait"]).astype(np.int64)
    lb = np.zeros(C8 * h)
    ub = np.zeros(C8 * h)
    caps = [float(np.float32(f_max))] * 2 + ([float(np.float32(np.asarray(f["f_max_hand"]).reshape(-1)[0]))] if nc == 3 else [])
    for i in range(h):
        for leg in range(nc):
            o = C8 * i + 8 * leg
            lb[o:o + 4], ub[o:o + 4] = 0, BIG
            lb[o + 4], ub[o + 4] = 0, float(np.float32(0.01))
            lb[o + 5:o + 7], ub[o + 5:o + 7] = -BIG, 0
            lb[o + 7], ub[o + 7] = 0, float(np.float32(np.float32(caps[leg]) * np.float32(gait[nc * i + leg])))
    # elimination
    keep_v, keep_c = [], []
    for i in range(h):
        st = [abs(ub[C8 * i + 8 * leg + 7]) >= 1e-4 for leg in range(nc)]
        for c in range(U):
            if st[(c // 3) % nc]:
                keep_v.append(U * i + c)
        for c in range(C8):
            if st[c // 8]:
                keep_c.append(C8 * i + c)
    keep_v, keep_c = np.array(keep_v, dtype=int), np.array(keep_c, dtype=int)
    Afull = np.zeros((C8 * h, U * h))
    for i in range(h):
        Afull[C8 * i:C8 * i + C8, U * i:U * i + U] = Fc
    return dict(R=R, rpy=np.array([roll, pitch, yaw]), x0=x0, Acd=Acd, Bcd=Bcd, H=H, g=g, Fc=Fc, lb=lb, ub=ub,
                var_ind=keep_v, con_ind=keep_c, H_red=H[np.ix_(keep_v, keep_v)], g_red=g[keep_v],
                A_red=Afull[np.ix_(keep_c, keep_v)], lb_red=lb[keep_c], ub_red=ub[keep_c], qj=ja)
