"""Packed per-instance input record of the batched MPC solver (host and device share this layout).

One record = everything ``update_problem_data`` receives for one MPC tick
(reference: ConvexMPC/convexMPC_interface.h:19-37 ``update_data_t``; extents SURVEY.md section 8b), narrowed to
float32 exactly as ``update_problem_data`` does (convexMPC_interface.cpp:83-103), laid out contiguously so that one
workgroup fetches its instance with a single coalesced burst:

    float32 index   field
    0..2            p            body position (world)
    3..5            v            body velocity (world)
    6..9            q            orientation quaternion (w, x, y, z)
    10..12          w            angular velocity (world)
    13..18          r            foot positions relative to the body, r[2*axis + leg]
    19..28          joint_angles left leg q0..q4, right leg q0..q4
    29              yaw          (carried for interface fidelity; unused by solve_mpc)
    30..41          weights      Q diagonal
    42..53          Alpha_K      R diagonal
    54..54+12h-1    traj         reference trajectory, 12 floats per horizon step
    then 2h bytes   gait         gait[2*step + leg], 1 = stance

Record stride = (54 + 12h)*4 + 2h rounded up to 16 bytes (720 B at h = 10; 716 B of payload).

Extension record (``contacts=3``: two feet + one hand, BASELINE config 5 -- no reference code, include/hector_mpc.h):

    0..12 as above | 13..21 r[3*axis + contact] | 22..31 joint_angles | 32 yaw | 33..44 weights |
    45..62 Alpha_K [F of each contact (9), M of each contact (9)] | 63..71 Rhand (hand contact frame, body frame,
    row-major) | 72 f_max_hand | 73.. traj (12h) | then 3h bytes gait[3*step + contact]      (816 B at h = 10)
"""
from __future__ import annotations

import numpy as np

N_FIXED = 54
OFF = dict(p=0, v=3, q=6, w=10, r=13, joint_angles=19, yaw=29, weights=30, Alpha_K=42, traj=54)
LEN = dict(p=3, v=3, q=4, w=3, r=6, joint_angles=10, yaw=1, weights=12, Alpha_K=12)
N_FIXED3 = 73
OFF3 = dict(p=0, v=3, q=6, w=10, r=13, joint_angles=22, yaw=32, weights=33, Alpha_K=45, Rhand=63, f_max_hand=72, traj=73)
LEN3 = dict(p=3, v=3, q=4, w=3, r=9, joint_angles=10, yaw=1, weights=12, Alpha_K=18, Rhand=9, f_max_hand=1)


def _layout(contacts: int):
    if contacts == 2:
        return N_FIXED, OFF, LEN
    if contacts == 3:
        return N_FIXED3, OFF3, LEN3
    raise ValueError("contacts must be 2 or 3")


def payload_bytes(horizon: int, contacts: int = 2) -> int:
    return (_layout(contacts)[0] + 12 * horizon) * 4 + contacts * horizon


def record_stride(horizon: int, contacts: int = 2) -> int:
    return (payload_bytes(horizon, contacts) + 15) // 16 * 16


def pack_records(fields: dict, horizon: int, contacts: int = 2) -> np.ndarray:
    """fields: dict of arrays with leading batch dim (p,v,q,w,r,joint_angles,yaw,weights,Alpha_K,traj,gait).

    Values are narrowed double->float32 / int->uint8 here, which is the narrowing the reference performs at its
    C boundary.  Returns a uint8 array [batch, stride].
    """
    NF, O, L = _layout(contacts)
    b = int(np.asarray(fields["p"]).shape[0])
    stride = record_stride(horizon, contacts)
    rec = np.zeros((b, stride), dtype=np.uint8)
    f32 = rec[:, : (NF + 12 * horizon) * 4].view(np.float32)
    for k, n in L.items():
        a = np.asarray(fields[k], dtype=np.float64).reshape(b, n)
        f32[:, O[k] : O[k] + n] = a.astype(np.float32)
    tr = np.asarray(fields["traj"], dtype=np.float64).reshape(b, -1)[:, : 12 * horizon]
    f32[:, O["traj"] : O["traj"] + 12 * horizon] = tr.astype(np.float32)
    g = np.asarray(fields["gait"]).reshape(b, -1)[:, : contacts * horizon]
    goff = (NF + 12 * horizon) * 4
    rec[:, goff : goff + contacts * horizon] = g.astype(np.uint8)
    return rec


def unpack_records(rec: np.ndarray, horizon: int, contacts: int = 2) -> dict:
    NF, O, L = _layout(contacts)
    rec = np.ascontiguousarray(rec)
    f32 = rec[:, : (NF + 12 * horizon) * 4].view(np.float32)
    out = {k: f32[:, O[k] : O[k] + n].copy() for k, n in L.items()}
    out["traj"] = f32[:, O["traj"] : O["traj"] + 12 * horizon].copy()
    goff = (NF + 12 * horizon) * 4
    out["gait"] = rec[:, goff : goff + contacts * horizon].copy()
    return out
