"""ctypes binding of the CPU oracle (oracle/libhmpc_oracle.so).  TEST INFRASTRUCTURE ONLY.

Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package.
See oracle/hmpc_oracle.h for what the oracle restates and its parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libhmpc_oracle.so")
MAXH = 36


def build(force: bool = False) -> None:
    """Builds the oracle with oracle/Makefile (gcc; _ref only when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or not os.path.exists(os.path.join(HERE, "_ref", "libqpoases_ref.so")):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    elif os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "hmpc_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"])


class Update(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("v", C.c_float * 3), ("q", C.c_float * 4), ("w", C.c_float * 3),
                ("r", C.c_float * 9), ("joint_angles", C.c_float * 10), ("yaw", C.c_float),
                ("weights", C.c_float * 12), ("traj", C.c_float * (12 * MAXH)), ("Alpha_K", C.c_float * 18),
                ("gait", C.c_ubyte * (3 * MAXH)), ("nc", C.c_int), ("Rhand", C.c_float * 9),
                ("f_max_hand", C.c_float)]


class Setup(C.Structure):
    _fields_ = [("dt", C.c_float), ("mu", C.c_float), ("f_max", C.c_float), ("horizon", C.c_int)]


class QP(C.Structure):
    _fields_ = [("horizon", C.c_int), ("qj", C.c_float * 10), ("R", C.c_float * 9), ("rpy", C.c_float * 3),
                ("x0", C.c_float * 13), ("Acd", C.c_float * 169), ("nc", C.c_int), ("Bcd", C.c_float * (13 * 18)),
                ("Rfoot", (C.c_float * 9) * 3), ("Fc", C.c_float * (24 * 18)),
                ("Phi", C.POINTER(C.c_float)), ("Apow", C.POINTER(C.c_float)), ("H", C.POINTER(C.c_float)),
                ("g", C.POINTER(C.c_float)), ("lb", C.POINTER(C.c_float)), ("ub", C.POINTER(C.c_float))]


class Red(C.Structure):
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("var_ind", C.POINTER(C.c_int)), ("con_ind", C.POINTER(C.c_int)),
                ("H", C.POINTER(C.c_double)), ("g", C.POINTER(C.c_double)), ("A", C.POINTER(C.c_double)),
                ("lb", C.POINTER(C.c_double)), ("ub", C.POINTER(C.c_double))]


class Params(C.Structure):
    """orc_params_t == include/hector_mpc.h struct hmpc_params"""
    _fields_ = [("mass", C.c_float), ("inertia", C.c_float * 3), ("mu", C.c_float), ("lt", C.c_float), ("lh", C.c_float),
                ("gravity", C.c_float)]


class Tick(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("vWorld", C.c_double * 3), ("omegaWorld", C.c_double * 3),
                ("orientation", C.c_double * 4), ("rpy", C.c_double * 3), ("rBody", C.c_double * 9),
                ("leg_q", C.c_double * 10), ("pFoot", C.c_double * 6), ("v_des_robot", C.c_double * 2),
                ("yaw_rate_des", C.c_double), ("roll_des", C.c_double), ("pitch_des", C.c_double),
                ("world_position_desired", C.c_double * 2), ("gait_offsets", C.c_int * 2),
                ("gait_durations", C.c_int * 2), ("gait_iteration", C.c_int), ("flags", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_qp_alloc.restype = C.POINTER(QP)
        L.orc_qp_alloc.argtypes = [C.c_int]
        L.orc_qp_free.argtypes = [C.POINTER(QP)]
        L.orc_red_alloc.restype = C.POINTER(Red)
        L.orc_red_alloc.argtypes = [C.c_int]
        L.orc_red_free.argtypes = [C.POINTER(Red)]
        L.orc_assemble.argtypes = [C.POINTER(Update), C.POINTER(Setup), C.POINTER(QP)]
        L.orc_reduce.argtypes = [C.POINTER(QP), C.POINTER(Red)]
        L.orc_solve_reduced.argtypes = [C.POINTER(Red), C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_atan2.restype = C.c_double
        L.orc_atan2.argtypes = [C.c_double, C.c_double]
        L.orc_asin.restype = C.c_double
        L.orc_asin.argtypes = [C.c_double]
        L.orc_set_dense_chain.argtypes = [C.c_int]
        L.orc_set_params.argtypes = [C.POINTER(Params)]
        L.orc_get_params.argtypes = [C.POINTER(Params)]
        L.orc_setup_problem.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double]
        L.orc_update_problem_data.argtypes = [C.c_void_p] * 6 + [C.c_double] + [C.c_void_p] * 4
        L.orc_get_solution.restype = C.c_double
        L.orc_get_solution.argtypes = [C.c_int]
        L.orc_solve_records.restype = C.c_int
        L.orc_solve_records_ex.restype = C.c_int
        L.orc_solve_records_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_solve_records.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_unpack_record.argtypes = [C.c_void_p, C.c_int, C.POINTER(Update)]
        L.orc_unpack_record3.argtypes = [C.c_void_p, C.c_int, C.POINTER(Update)]
        L.orc_set_records_nc.argtypes = [C.c_int]
        L.orc_set_unfused_chain.argtypes = [C.c_int]
        L.orc_set_libm_trig.argtypes = [C.c_int]
        L.orc_mpc_gait.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_build_record.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.orc_body_wrench.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_leg_torques.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_leg_jacobian.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_qpoases_solve.restype = C.c_int
        L.ref_qpoases_solve.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 4
        _lib = L
    return _lib


def _np(ptr, shape, dtype):
    n = int(np.prod(shape))
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype).reshape(shape).copy()


def set_params(mass=None, inertia=None, mu=None, lt=None, lh=None, gravity=None) -> None:
    """Robot / contact constants of the oracle (process-global); no arguments = the reference's literals."""
    L = lib()
    if all(v is None for v in (mass, inertia, mu, lt, lh, gravity)):
        L.orc_set_params(None)
        return
    p = Params()
    L.orc_set_params(None)
    L.orc_get_params(C.byref(p))
    if mass is not None:
        p.mass = np.float32(mass)
    if inertia is not None:
        p.inertia[:] = [np.float32(v) for v in inertia]
    if mu is not None:
        p.mu = np.float32(mu)
    if lt is not None:
        p.lt = np.float32(lt)
    if lh is not None:
        p.lh = np.float32(lh)
    if gravity is not None:
        p.gravity = np.float32(gravity)
    L.orc_set_params(C.byref(p))


def update_from_record(rec_row: np.ndarray, horizon: int, nc: int = 2) -> Update:
    u = Update()
    row = np.ascontiguousarray(rec_row)
    (lib().orc_unpack_record3 if nc == 3 else lib().orc_unpack_record)(row.ctypes.data, horizon, C.byref(u))
    return u


def assemble_record(rec_row: np.ndarray, horizon: int, dt: float, f_max: float, reduce: bool = True,
                    dense_chain: bool = False, nc: int = 2) -> dict:
    """Runs orc_assemble (+ orc_reduce) on one packed record and returns every intermediate as numpy arrays.

    nc = 3 reads the extension record layout (hand contact, BASELINE config 5)."""
    L = lib()
    u = update_from_record(rec_row, horizon, nc)
    s = Setup(np.float32(dt), np.float32(0.25), np.float32(f_max), horizon)
    qp = L.orc_qp_alloc(horizon)
    L.orc_set_dense_chain(1 if dense_chain else 0)
    L.orc_assemble(C.byref(u), C.byref(s), qp)
    L.orc_set_dense_chain(0)
    q = qp.contents
    U, C8 = 6 * nc, 8 * nc
    h, N, M = horizon, U * horizon, C8 * horizon
    out = dict(
        qj=np.array(q.qj, dtype=np.float32), R=np.array(q.R, dtype=np.float32).reshape(3, 3),
        rpy=np.array(q.rpy, dtype=np.float32), x0=np.array(q.x0, dtype=np.float32),
        Acd=np.array(q.Acd, dtype=np.float32).reshape(13, 13), Bcd=np.array(q.Bcd, dtype=np.float32)[:13 * U].reshape(13, U),
        Rfoot=np.array(q.Rfoot, dtype=np.float32).reshape(3, 3, 3)[:nc],
        Fc=np.array(q.Fc, dtype=np.float32)[:C8 * U].reshape(C8, U),
        Phi=_np(q.Phi, (h, 13, U), np.float32), Apow=_np(q.Apow, (h + 1, 13, 13), np.float32),
        H=_np(q.H, (N, N), np.float32), g=_np(q.g, (N,), np.float32), lb=_np(q.lb, (M,), np.float32),
        ub=_np(q.ub, (M,), np.float32),
    )
    if reduce:
        red = L.orc_red_alloc(horizon)
        L.orc_reduce(qp, red)
        r = red.contents
        n, m = r.n, r.m
        out.update(n=n, m=m, var_ind=_np(r.var_ind, (n,), np.int32), con_ind=_np(r.con_ind, (m,), np.int32),
                   H_red=_np(r.H, (n, n), np.float64), g_red=_np(r.g, (n,), np.float64),
                   A_red=_np(r.A, (m, n), np.float64), lb_red=_np(r.lb, (m,), np.float64),
                   ub_red=_np(r.ub, (m,), np.float64))
        L.orc_red_free(red)
    L.orc_qp_free(qp)
    return out


def qpoases_solve(H, g, A, lb, ub, nwsr_max: int = 500):
    """The reference's vendored qpOASES on an explicit dense QP (float64, row-major).  Returns x, y, obj, nWSR, status."""
    H = np.ascontiguousarray(H, dtype=np.float64)
    g = np.ascontiguousarray(g, dtype=np.float64)
    A = np.ascontiguousarray(A, dtype=np.float64)
    lb = np.ascontiguousarray(lb, dtype=np.float64)
    ub = np.ascontiguousarray(ub, dtype=np.float64)
    n, m = g.shape[0], lb.shape[0]
    x = np.zeros(n)
    y = np.zeros(n + m)
    obj = C.c_double(0)
    nwsr = C.c_int(0)
    st = lib().ref_qpoases_solve(n, m, H.ctypes.data, g.ctypes.data, A.ctypes.data, lb.ctypes.data, ub.ctypes.data,
                                 nwsr_max, x.ctypes.data, y.ctypes.data, C.byref(obj), C.byref(nwsr))
    return x, y, obj.value, nwsr.value, st


def solve_records(records: np.ndarray, horizon: int, dt: float, f_max: float, first: int = 0, count: int | None = None,
                  nc: int = 2):
    """Full reference path (assembly + elimination + qpOASES + scatter) on packed records.

    Returns dict(q_soln [count,6 nc h] float64, nwsr, obj, n_bad, bad [count] bool: qpOASES did not solve this instance,
    t_assemble, t_solve)."""
    records = np.ascontiguousarray(records)
    count = records.shape[0] - first if count is None else count
    q = np.zeros((count, 6 * nc * horizon))
    lib().orc_set_records_nc(nc)
    nwsr = np.zeros(count, dtype=np.int32)
    obj = np.zeros(count)
    ta = C.c_double(0)
    ts = C.c_double(0)
    rv = np.zeros(count, dtype=np.int32)
    bad = lib().orc_solve_records_ex(records.ctypes.data, records.shape[1], first, count, horizon, np.float32(dt),
                                     np.float32(f_max), q.ctypes.data, nwsr.ctypes.data, obj.ctypes.data,
                                     C.addressof(ta), C.addressof(ts), rv.ctypes.data)
    lib().orc_set_records_nc(2)
    return dict(q_soln=q, nwsr=nwsr, obj=obj, n_bad=bad, bad=rv != 0, t_assemble=ta.value, t_solve=ts.value)


def legacy_tick(fields_row: dict, horizon: int, dt: float, mu: float, f_max: float) -> np.ndarray:
    """setup_problem / update_problem_data / get_solution sequence of ConvexMPCLocomotion.cpp:410-429 on the oracle."""
    L = lib()
    L.orc_setup_problem(dt, horizon, mu, f_max)
    arrs = [np.ascontiguousarray(fields_row[k], dtype=np.float64) for k in ("p", "v", "q", "w", "r", "joint_angles")]
    tail = [np.ascontiguousarray(fields_row[k], dtype=np.float64) for k in ("weights", "traj", "Alpha_K")]
    gait = np.ascontiguousarray(fields_row["gait"], dtype=np.int32)
    L.orc_update_problem_data(*[a.ctypes.data for a in arrs], float(np.asarray(fields_row["yaw"]).reshape(-1)[0]),
                              *[a.ctypes.data for a in tail], gait.ctypes.data)
    return np.array([L.orc_get_solution(i) for i in range(12 * horizon)])


# ---- input builder (f1), gait table (f2), body-frame wrench (f3)
def mpc_gait(n_segments: int, offsets, durations, iteration: int) -> np.ndarray:
    off = np.ascontiguousarray(offsets, dtype=np.int32)
    dur = np.ascontiguousarray(durations, dtype=np.int32)
    out = np.zeros(2 * n_segments, dtype=np.int32)
    lib().orc_mpc_gait(n_segments, off.ctypes.data, dur.ctypes.data, int(iteration), out.ctypes.data)
    return out


def build_records(ticks: np.ndarray, horizon: int, dt_mpc: float):
    """ticks: structured array laid out as struct orc_tick_t (= hmpc_tick_inputs).  Returns (records, wpd)."""
    ticks = np.ascontiguousarray(ticks)
    assert ticks.dtype.itemsize == C.sizeof(Tick)
    stride = ((54 + 12 * horizon) * 4 + 2 * horizon + 15) // 16 * 16
    rec = np.zeros((ticks.shape[0], stride), dtype=np.uint8)
    wpd = np.zeros((ticks.shape[0], 2), dtype=np.float64)
    L = lib()
    for k in range(ticks.shape[0]):
        L.orc_build_record(ticks[k:k + 1].ctypes.data, horizon, float(dt_mpc), rec[k].ctypes.data, wpd[k].ctypes.data)
    return rec, wpd


def body_wrench(q_soln: np.ndarray, rBody: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q_soln, dtype=np.float64)
    rb = np.ascontiguousarray(rBody, dtype=np.float64).reshape(q.shape[0], 9)
    out = np.zeros((q.shape[0], 2, 6), dtype=np.float64)
    L = lib()
    for k in range(q.shape[0]):
        L.orc_body_wrench(q[k].ctypes.data, rb[k].ctypes.data, out[k].ctypes.data)
    return out


def leg_torques(f_ff: np.ndarray, leg_q: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(f_ff, dtype=np.float64).reshape(-1, 12)
    q = np.ascontiguousarray(leg_q, dtype=np.float64).reshape(-1, 10)
    out = np.zeros((f.shape[0], 2, 5), dtype=np.float64)
    L = lib()
    for k in range(f.shape[0]):
        L.orc_leg_torques(f[k].ctypes.data, q[k].ctypes.data, out[k].ctypes.data)
    return out


def leg_jacobian(q5: np.ndarray, leg: int) -> np.ndarray:
    q = np.ascontiguousarray(q5, dtype=np.float64)
    J = np.zeros(30, dtype=np.float64)
    lib().orc_leg_jacobian(q.ctypes.data, int(leg), J.ctypes.data)
    return J.reshape(6, 5)
