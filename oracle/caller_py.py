"""ctypes binding of oracle/_ref/libcaller_ref.so.  TEST INFRASTRUCTURE ONLY.

That library is the reference's OWN caller-side code for SURVEY.md section 8(f) rows f1-f3 --
``ConvexMPC/GaitGenerator.cpp``, ``ConvexMPC/ConvexMPCLocomotion.cpp``, ``src/common/LegController.cpp`` (+
``FootSwingTrajectory.cpp``, ``DesiredCommand.cpp``) -- compiled unmodified from /root/reference against the Eigen
stand-in ``oracle/mini_eigen`` (recipe: ``oracle/Makefile``); ``oracle/caller_ref_shim.cpp`` says what the shim supplies.
Used by tests/ (and tests/golden/make_caller_golden.py) only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

from . import ref_py

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libcaller_ref.so")
KMAX = ref_py.K_MAX_GAIT_SEGMENTS
DT_CONTROL, ITERATIONS_BETWEEN_MPC = 0.001, 40  # FSMState_Walking.cpp:5  Cmpc(0.001, 40)


class Capture(C.Structure):  # struct refc_capture, caller_ref_shim.cpp
    _fields_ = [("n_setup", C.c_int), ("n_update", C.c_int), ("horizon", C.c_int), ("pad", C.c_int),
                ("dt", C.c_double), ("mu", C.c_double), ("f_max", C.c_double),
                ("p", C.c_double * 3), ("v", C.c_double * 3), ("q", C.c_double * 4), ("w", C.c_double * 3),
                ("r", C.c_double * 6), ("joint_angles", C.c_double * 10), ("yaw", C.c_double),
                ("weights", C.c_double * 12), ("Alpha_K", C.c_double * 12), ("traj", C.c_double * (12 * KMAX)),
                ("gait", C.c_int * (2 * KMAX))]


def available() -> bool:
    return os.path.exists(LIB_PATH)


def build() -> None:
    subprocess.check_call(["make", "-C", HERE, "-s", LIB_PATH])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.refc_create.restype = vp
        L.refc_create.argtypes = [C.c_double, C.c_int, C.c_char_p]
        L.refc_destroy.argtypes = [vp]
        L.refc_set_backend.argtypes = [vp, vp, vp]
        L.refc_set_solution.argtypes = [vp, C.c_int]
        L.refc_set_state.argtypes = [vp] * 7
        L.refc_set_command.argtypes = [vp, vp]
        L.refc_update_leg_data_from_motors.argtypes = [vp, vp, vp]
        L.refc_set_leg_q.argtypes = [vp, vp]
        L.refc_poke_leg_q.argtypes = [vp, vp]
        L.refc_get_leg.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.refc_set_members.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
        L.refc_get_members.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int)]
        L.refc_update_mpc.argtypes = [vp]
        L.refc_run.argtypes = [vp, C.c_int]
        L.refc_get_capture.argtypes = [C.POINTER(Capture)]
        L.refc_update_command.argtypes = [vp, vp, vp]
        L.refc_leg_tau_f64.argtypes = [vp, vp, vp]
        L.refc_gait.argtypes = [C.c_int] * 7 + [vp] * 4
        assert L.refc_sizeof_capture() == C.sizeof(Capture)
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data


def gait(n_segments: int, offsets, durations, iterations_per_mpc: int, current_iteration: int) -> dict:
    """Gait(n, offsets, durations) -> setIterations(iterations_per_mpc, current_iteration) -> mpc_gait() and the two
    sub-phase functions, on the reference's own GaitGenerator.cpp."""
    table = np.zeros(2 * n_segments, dtype=np.int32)
    contact, swing = np.zeros(2), np.zeros(2)
    ss = np.zeros(2, dtype=np.int32)
    lib().refc_gait(int(n_segments), int(offsets[0]), int(offsets[1]), int(durations[0]), int(durations[1]),
                    int(iterations_per_mpc), int(current_iteration), _p(table), _p(contact), _p(swing), _p(ss))
    return dict(table=table, contact=contact, swing=swing, stance=int(ss[0]), swing_segments=int(ss[1]))


class Caller:
    """One ConvexMPCLocomotion of the reference with the object graph main.cpp builds around it."""

    def __init__(self, backend: str | None = None):
        L = lib()
        self._scratch = tempfile.TemporaryDirectory(prefix="refcaller_")  # the constructor opens ./foot_pos.txt
        self.h = L.refc_create(DT_CONTROL, ITERATIONS_BETWEEN_MPC, self._scratch.name.encode())
        assert self.h
        self.set_backend(backend)

    def set_backend(self, backend: str | None) -> None:
        """None: get_solution answers from set_solution(); "reference": the captured calls are forwarded to the
        reference's own solver (oracle/_ref/libsolvempc_ref.so); a ctypes library object: forwarded to that library's
        setup_problem / update_problem_data / get_solution."""
        L = lib()
        cast = lambda f: C.cast(f, C.c_void_p)
        if backend == "reference":
            R = ref_py.lib()
            L.refc_set_backend(cast(R.setup_problem), cast(R.update_problem_data), cast(R.get_solution))
        elif backend is not None and not isinstance(backend, str):
            # any library that exports the reference's C interface (convexMPC_interface.h:39-43) -- the tests pass the
            # PRODUCT library here: the reference's own controller code then drives the HIP solver, unchanged
            L.refc_set_backend(cast(backend.setup_problem), cast(backend.update_problem_data), cast(backend.get_solution))
            self._backend_keepalive = backend
        else:
            L.refc_set_backend(None, None, None)

    def close(self):
        if self.h:
            lib().refc_destroy(self.h)
            self.h = None
            self._scratch.cleanup()

    def set_solution(self, sol):
        s = np.ascontiguousarray(sol, dtype=np.float64)
        lib().refc_set_solution(_p(s), s.size)

    def set_state(self, position, vWorld, omegaWorld, orientation, rpy, rBody):
        a = [np.ascontiguousarray(x, dtype=np.float64).reshape(-1) for x in (position, vWorld, omegaWorld, orientation, rpy, rBody)]
        lib().refc_set_state(self.h, *[_p(x) for x in a])

    def set_command(self, roll_des=0.0, pitch_des=0.0, vx=0.0, vy=0.0, yaw_rate=0.0):
        s = np.zeros(12)
        s[3], s[4], s[6], s[7], s[11] = roll_des, pitch_des, vx, vy, yaw_rate  # DesiredCommand.cpp:16-44
        lib().refc_set_command(self.h, _p(s))

    def update_leg_data_from_motors(self, q10):
        q = np.ascontiguousarray(q10, dtype=np.float32)
        lib().refc_update_leg_data_from_motors(self.h, _p(q), None)

    def set_leg_q(self, q10):
        q = np.ascontiguousarray(q10, dtype=np.float64)
        lib().refc_set_leg_q(self.h, _p(q))

    def poke_leg_q(self, q10):
        q = np.ascontiguousarray(q10, dtype=np.float64)
        lib().refc_poke_leg_q(self.h, _p(q))

    def leg(self, leg: int) -> dict:
        q, J, Jf, p = np.zeros(5), np.zeros((6, 5)), np.zeros((3, 5)), np.zeros(3)
        lib().refc_get_leg(self.h, leg, _p(q), _p(J), _p(Jf), _p(p))
        return dict(q=q, J_force_moment=J, J_force=Jf, p=p)

    def set_members(self, world_position_desired, pFoot, iteration_counter: int, gait_number: int, first_run: bool = False):
        w = np.zeros(3)
        w[:len(world_position_desired)] = world_position_desired
        f = np.ascontiguousarray(pFoot, dtype=np.float64).reshape(-1)
        lib().refc_set_members(self.h, _p(w), _p(f), int(iteration_counter), int(gait_number), int(first_run))

    def members(self) -> dict:
        w, f, ff = np.zeros(3), np.zeros(6), np.zeros(12)
        it = C.c_int(0)
        lib().refc_get_members(self.h, _p(w), _p(f), _p(ff), C.byref(it))
        return dict(world_position_desired=w, pFoot=f.reshape(2, 3), f_ff=ff.reshape(2, 6), iterationCounter=it.value)

    def update_mpc(self):
        with ref_py.quiet():
            lib().refc_update_mpc(self.h)

    def run(self, gait_number: int):
        with ref_py.quiet():
            lib().refc_run(self.h, int(gait_number))

    @staticmethod
    def capture() -> dict:
        c = Capture()
        lib().refc_get_capture(C.byref(c))
        h = c.horizon
        out = {k: np.array(getattr(c, k)) for k in ("p", "v", "q", "w", "r", "joint_angles", "weights", "Alpha_K")}
        out.update(yaw=c.yaw, traj=np.array(c.traj)[:12 * h], gait=np.array(c.gait)[:2 * h], horizon=h, dt=c.dt, mu=c.mu,
                   f_max=c.f_max, n_setup=c.n_setup, n_update=c.n_update)
        return out

    def update_command(self, f_ff):
        f = np.ascontiguousarray(f_ff, dtype=np.float64).reshape(-1)
        tau = np.zeros(10, dtype=np.float32)
        lib().refc_update_command(self.h, _p(f), _p(tau))
        return tau

    def leg_tau_f64(self, f_ff):
        f = np.ascontiguousarray(f_ff, dtype=np.float64).reshape(-1)
        tau = np.zeros(10)
        lib().refc_leg_tau_f64(self.h, _p(f), _p(tau))
        return tau


def tick_through_reference(caller: Caller, t, gait_number: int) -> dict:
    """One ``hmpc_tick_inputs`` row through the reference's updateMPCIfNeeded (members injected, see the shim):
    returns the captured update_problem_data arguments, the clamped world_position_desired and f_ff."""
    caller.set_state(t["position"], t["vWorld"], t["omegaWorld"], t["orientation"], t["rpy"], t["rBody"])
    caller.set_command(t["roll_des"], t["pitch_des"], t["v_des_robot"][0], t["v_des_robot"][1], t["yaw_rate_des"])
    if int(t["flags"]) & 1:
        caller.set_leg_q(t["leg_q"])   # raw motor angles: the reference's Jacobian call mutates data[leg].q
    else:
        caller.poke_leg_q(t["leg_q"])  # data[leg].q as updateMPCIfNeeded reads it
    caller.set_members(t["world_position_desired"], t["pFoot"], ITERATIONS_BETWEEN_MPC * int(t["gait_iteration"]), gait_number)
    caller.update_mpc()
    cap = caller.capture()
    cap.update(caller.members())
    return cap
