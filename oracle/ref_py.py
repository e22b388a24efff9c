"""ctypes binding of oracle/_ref/libsolvempc_ref.so.  TEST INFRASTRUCTURE ONLY.

That library is the reference's OWN ``ConvexMPC/SolverMPC.cpp``, ``RobotState.cpp`` and ``convexMPC_interface.cpp``,
compiled unmodified from /root/reference against the Eigen stand-in ``oracle/mini_eigen`` and linked with the
reference's vendored qpOASES (recipe: ``oracle/Makefile``).  It is what pins the assembly half of the oracle
(``hmpc_oracle.c``) to an execution of the reference's text; see ``mini_eigen/eigen3/Eigen/Dense`` for exactly what a
build against the stand-in does and does not pin.  Used by tests/ and by bench.py's ``cpu_baseline`` leg only.

The reference prints three lines per solve (``SolverMPC.cpp:639-640, 717``); ``quiet()`` sends file descriptor 1 to
/dev/null around the calls.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libsolvempc_ref.so")
K_MAX_GAIT_SEGMENTS = 36  # convexMPC_interface.h:3


class UpdateData(C.Structure):  # update_data_t, convexMPC_interface.h:19-37
    _fields_ = [("p", C.c_float * 3), ("v", C.c_float * 3), ("q", C.c_float * 4), ("w", C.c_float * 3),
                ("r", C.c_float * 6), ("joint_angles", C.c_float * 10), ("yaw", C.c_float),
                ("weights", C.c_float * 12), ("traj", C.c_float * (12 * K_MAX_GAIT_SEGMENTS)),
                ("Alpha_K", C.c_float * 12), ("gait", C.c_ubyte * K_MAX_GAIT_SEGMENTS),
                ("hack_pad", C.c_ubyte * 1000), ("max_iterations", C.c_int), ("rho", C.c_double),
                ("sigma", C.c_double), ("solver_alpha", C.c_double), ("terminate", C.c_double)]


def available() -> bool:
    return os.path.exists(LIB_PATH)


def build() -> None:
    """(Re)builds through oracle/Makefile; a no-op where /root/reference is absent and the prebuilt file exists."""
    subprocess.check_call(["make", "-C", HERE, "-s", os.path.join(HERE, "_ref", "libsolvempc_ref.so")])


_lib = None
_libc = C.CDLL(None)
ASSOC_LIB_PATH = os.path.join(HERE, "_ref", "libsolvempc_ref_assoc.so")


def use_variant(name: str | None) -> None:
    """Switches this module to another build of the same reference sources: ``"assoc"`` = the sensitivity build whose
    stand-in associates three-term sums as d0 + (d1 + d2) (oracle/Makefile); ``None`` = the default build."""
    global _lib, LIB_PATH
    LIB_PATH = ASSOC_LIB_PATH if name == "assoc" else os.path.join(HERE, "_ref", "libsolvempc_ref.so")
    _lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", HERE, "-s", LIB_PATH])
        L = C.CDLL(LIB_PATH)
        L.setup_problem.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double]
        L.update_problem_data.argtypes = [C.c_void_p] * 6 + [C.c_double] + [C.c_void_p] * 4
        L.get_solution.restype = C.c_double
        L.get_solution.argtypes = [C.c_int]
        L.ref_get_matrix.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ref_get_array.restype = C.POINTER(C.c_double)
        L.ref_get_array.argtypes = [C.c_char_p]
        L.ref_var_elim.restype = C.POINTER(C.c_char)
        L.ref_con_elim.restype = C.POINTER(C.c_char)
        L.ref_update.restype = C.POINTER(UpdateData)
        assert L.ref_sizeof_update() == C.sizeof(UpdateData)
        _lib = L
    return _lib


@contextlib.contextmanager
def quiet():
    """fd 1 -> /dev/null for the duration (the reference's printf/cout lines), C stdio flushed on both edges."""
    _libc.fflush(None)
    saved = os.dup(1)
    null = os.open(os.devnull, os.O_WRONLY)
    os.dup2(null, 1)
    os.close(null)
    try:
        yield
    finally:
        _libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def silence_forever(path: str | None = None) -> None:
    """For worker processes that only ever time the reference: stdout of this process goes to /dev/null (or to `path`,
    so that the reference's "failed to solve!" lines, SolverMPC.cpp:714-715, can be counted afterwards)."""
    _libc.fflush(None)
    null = os.open(path or os.devnull, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
    os.dup2(null, 1)
    os.close(null)


def flush_stdio() -> None:
    _libc.fflush(None)


def setup_problem(dt: float, horizon: int, mu: float, f_max: float) -> None:
    lib().setup_problem(float(dt), int(horizon), float(mu), float(f_max))


def update_problem_data(fields_row: dict) -> None:
    """One blocking tick through the reference's own C entry (convexMPC_interface.cpp:83-103): narrows and solves."""
    L = lib()
    arrs = [np.ascontiguousarray(fields_row[k], dtype=np.float64) for k in ("p", "v", "q", "w", "r", "joint_angles")]
    tail = [np.ascontiguousarray(fields_row[k], dtype=np.float64) for k in ("weights", "traj", "Alpha_K")]
    gait = np.ascontiguousarray(fields_row["gait"], dtype=np.int32)
    L.update_problem_data(*[a.ctypes.data for a in arrs], float(np.asarray(fields_row["yaw"]).reshape(-1)[0]),
                          *[a.ctypes.data for a in tail], gait.ctypes.data)


def matrix(name: str) -> np.ndarray:
    """One of the reference's float Eigen globals (SolverMPC.cpp:19-45, 365-368), as a float32 [rows, cols] array."""
    cap = 1 << 20
    buf = np.zeros(cap, dtype=np.float32)
    r, c = C.c_int(0), C.c_int(0)
    rc = lib().ref_get_matrix(name.encode(), buf.ctypes.data, cap, C.byref(r), C.byref(c))
    if rc != 0:
        raise KeyError(f"{name}: {rc}")
    return buf[: r.value * c.value].reshape(r.value, c.value).copy()


def array(name: str, n: int) -> np.ndarray:
    p = lib().ref_get_array(name.encode())
    if not p:
        raise KeyError(name)
    return np.ctypeslib.as_array(p, shape=(n,)).copy()


def tick(fields_row: dict, horizon: int, dt: float, mu: float, f_max: float, setup: bool = True) -> dict:
    """setup_problem + update_problem_data on the reference's own code, then everything it left behind.

    Returns the unreduced float data (qH, qg, fmat, U_b, L_b, A_qp, B_qp, x_0, A_ct, B_ct_r, R, I_world), the elimination
    marks (var_elim, con_elim -> var_ind, con_ind, n, m), the binary64 reduced QP handed to qpOASES (H_red, g_red, A_red,
    lb_red, ub_red) and q_soln[12 h]."""
    L = lib()
    h = int(horizon)
    with quiet():
        if setup:
            setup_problem(dt, h, mu, f_max)
        update_problem_data(fields_row)
    N, M = 12 * h, 16 * h
    out = {k: matrix(k) for k in ("qH", "qg", "fmat", "U_b", "L_b", "A_qp", "B_qp", "x_0", "A_ct", "B_ct_r", "R",
                                  "I_world", "X_d")}
    ve = np.frombuffer(C.string_at(L.ref_var_elim(), N), dtype=np.int8).copy()
    ce = np.frombuffer(C.string_at(L.ref_con_elim(), M), dtype=np.int8).copy()
    var_ind = np.nonzero(ve == 0)[0].astype(np.int32)
    con_ind = np.nonzero(ce == 0)[0].astype(np.int32)
    n, m = len(var_ind), len(con_ind)
    out.update(var_elim=ve, con_elim=ce, var_ind=var_ind, con_ind=con_ind, n=n, m=m,
               H_red=array("H_red", n * n).reshape(n, n), g_red=array("g_red", n),
               A_red=array("A_red", m * n).reshape(m, n), lb_red=array("lb_red", m), ub_red=array("ub_red", m),
               q_soln=array("q_soln", N),
               get_solution=np.array([L.get_solution(i) for i in range(N)]))
    return out


def solve_fields(fields: dict, horizon: int, dt: float, mu: float, f_max: float, first: int = 0,
                 count: int | None = None) -> np.ndarray:
    """The reference's own path end to end over rows [first, first+count) of a field dict -> q_soln [count, 12 h]."""
    b = np.asarray(fields["p"]).shape[0]
    count = b - first if count is None else count
    h = int(horizon)
    out = np.zeros((count, 12 * h))
    L = lib()
    keys = ("p", "v", "q", "w", "r", "joint_angles", "weights", "traj", "Alpha_K")
    f64 = {k: np.ascontiguousarray(fields[k], dtype=np.float64) for k in keys}
    gait = np.ascontiguousarray(fields["gait"], dtype=np.int32)
    yaw = np.asarray(fields["yaw"], dtype=np.float64).reshape(-1)
    with quiet():
        setup_problem(dt, h, mu, f_max)  # ConvexMPCLocomotion.cpp:410 calls it every tick; once is equivalent
        for k in range(first, first + count):
            L.update_problem_data(*[f64[key][k].ctypes.data for key in ("p", "v", "q", "w", "r", "joint_angles")],
                                  float(yaw[k]),
                                  *[f64[key][k].ctypes.data for key in ("weights", "traj", "Alpha_K")],
                                  gait[k].ctypes.data)
            out[k - first] = np.ctypeslib.as_array(L.ref_get_array(b"q_soln"), shape=(12 * h,))
    return out
