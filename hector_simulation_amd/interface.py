"""Python mirror of the MPC boundary -- same names, argument meaning and error behaviour as the reference's
``ConvexMPC/convexMPC_interface.h:39-43`` (``setup_problem`` / ``update_problem_data`` / ``get_solution`` /
``update_solver_settings``) plus the batched handle API of ``include/hector_mpc.h``.

Everything here is a thin ctypes call into ``libhector_mpc_hip.so``; there is no Python/NumPy/PyTorch compute path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, records

TICK_DTYPE = np.dtype([("position", "f8", 3), ("vWorld", "f8", 3), ("omegaWorld", "f8", 3), ("orientation", "f8", 4),
                       ("rpy", "f8", 3), ("rBody", "f8", 9), ("leg_q", "f8", 10), ("pFoot", "f8", 6),
                       ("v_des_robot", "f8", 2), ("yaw_rate_des", "f8"), ("roll_des", "f8"), ("pitch_des", "f8"),
                       ("world_position_desired", "f8", 2), ("gait_offsets", "i4", 2), ("gait_durations", "i4", 2),
                       ("gait_iteration", "i4"), ("flags", "i4")], align=True)

STATUS_NAMES = {0: "ok", 1: "max_iter", 2: "infeasible", 3: "too_large", 4: "kkt", 5: "working_set_full", 6: "ok_relaxed", 7: "sweep_mismatch", 8: "hessian_not_positive_definite", 9: "regularisation_step"}


class HmpcError(RuntimeError):
    pass


def _check(rc: int, what: str):
    if rc != 0:
        raise HmpcError(f"{what} failed with {rc}: {_lib.load().hmpc_last_hip_error().decode()}")


# ---------------------------------------------------------------- reference (legacy) interface, process-global
def setup_problem(dt: float, horizon: int, mu: float, f_max: float) -> None:
    _lib.load().setup_problem(float(dt), int(horizon), float(mu), float(f_max))


def update_problem_data(p, v, q, w, r, joint_angles, yaw, weights, state_trajectory, Alpha_K, gait) -> None:
    """Blocking: narrows to float, solves on the GPU, leaves the solution for ``get_solution`` (as the reference does)."""
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (p, v, q, w, r, joint_angles)]
    b = [np.ascontiguousarray(x, dtype=np.float64) for x in (weights, state_trajectory, Alpha_K)]
    g = np.ascontiguousarray(gait, dtype=np.int32)
    _lib.load().update_problem_data(*[x.ctypes.data for x in a], float(yaw), *[x.ctypes.data for x in b], g.ctypes.data)


def get_solution(index: int) -> float:
    return float(_lib.load().get_solution(int(index)))


def update_solver_settings(max_iter, rho, sigma, solver_alpha, terminate, use_jcqp) -> None:
    """Stored and read by nothing, exactly as in the reference (convexMPC_interface.cpp:112-118)."""
    _lib.load().update_solver_settings(int(max_iter), float(rho), float(sigma), float(solver_alpha), float(terminate),
                                       float(use_jcqp))


def legacy_set_max_iterations(max_iter: int) -> None:
    """Explicit opt-in: cap on the active-set iterations of the process-global (legacy) solver; 0 = none."""
    _check(_lib.load().hmpc_legacy_set_max_iterations(int(max_iter)), "hmpc_legacy_set_max_iterations")


def last_status() -> int:
    return int(_lib.load().hmpc_last_status())


# ---------------------------------------------------------------- batched handle API
class BatchedMPC:
    """One handle = one GPU, one (dt, f_max, horizon) problem shape, up to ``max_batch`` independent MPC instances.

    ``contacts=3`` selects the hand-contact extension (BASELINE config 5; include/hector_mpc.h ``hmpc_create_ex``)."""

    def __init__(self, dt: float, horizon: int, f_max: float, max_batch: int, mu: float = 0.25, device: int = 0,
                 contacts: int = 2):
        self.L = _lib.load()
        self.horizon, self.max_batch, self.device = int(horizon), int(max_batch), int(device)
        self.contacts = int(contacts)
        self.nvar = 6 * self.contacts * self.horizon  # forces per instance: [step][F of each contact, M of each contact]
        self.setup = _lib.ProblemSetup(np.float32(dt), np.float32(mu), np.float32(f_max), int(horizon))
        self.h = C.c_void_p()
        _check(self.L.hmpc_create_ex(C.byref(self.h), C.byref(self.setup), self.max_batch, self.device, self.contacts),
               "hmpc_create_ex")
        self.stride = int(self.L.hmpc_record_stride_ex(self.horizon, self.contacts))
        self._keep = None
        self._keep_out = None

    def close(self):
        if getattr(self, "h", None):
            self.L.hmpc_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def batch(self) -> int:
        return int(self.L.hmpc_batch(self.h))

    def upload(self, recs: np.ndarray) -> None:
        recs = np.ascontiguousarray(recs, dtype=np.uint8)
        assert recs.ndim == 2 and recs.shape[1] == self.stride, (recs.shape, self.stride)
        _check(self.L.hmpc_upload_records(self.h, recs.ctypes.data, recs.shape[0]), "hmpc_upload_records")

    def upload_async(self, host_ptr: int, batch: int, stream: int = 0) -> None:
        """Stream-ordered upload from a (pinned) host buffer of ``batch`` packed records."""
        _check(self.L.hmpc_upload_records_async(self.h, C.c_void_p(host_ptr), int(batch), C.c_void_p(stream)),
               "hmpc_upload_records_async")

    def download_async(self, forces_ptr: int, status_ptr: int, stream: int = 0) -> None:
        """Stream-ordered download into (pinned) host buffers; no safe pass (see include/hector_mpc.h)."""
        _check(self.L.hmpc_download_async(self.h, C.c_void_p(forces_ptr), C.c_void_p(status_ptr), C.c_void_p(stream)),
               "hmpc_download_async")

    def upload_fields(self, fields: dict) -> None:
        self.upload(records.pack_records(fields, self.horizon, self.contacts))

    def set_device_records(self, device_ptr: int, batch: int, max_reduced_vars: int = -1, keepalive=None) -> None:
        self._keep = keepalive
        _check(self.L.hmpc_set_device_records(self.h, C.c_void_p(device_ptr), int(batch)), "hmpc_set_device_records")
        _check(self.L.hmpc_set_max_reduced_vars(self.h, int(max_reduced_vars)), "hmpc_set_max_reduced_vars")

    def set_device_outputs(self, forces_ptr: int, status_ptr: int, keepalive=None) -> None:
        self._keep_out = keepalive
        _check(self.L.hmpc_set_device_outputs(self.h, C.c_void_p(forces_ptr), C.c_void_p(status_ptr)),
               "hmpc_set_device_outputs")

    def set_warm_start(self, on: bool) -> None:
        _check(self.L.hmpc_set_warm_start(self.h, 1 if on else 0), "hmpc_set_warm_start")

    def set_tick_warm_start(self, on: bool, horizon_shift: int = 0) -> None:
        """Carry each instance's final working set to the next solve of this handle (off by default: the reference
        cold-starts every tick).  ``horizon_shift`` = steps the gait table advanced since the previous solve."""
        _check(self.L.hmpc_set_tick_warm_start(self.h, 1 if on else 0, int(horizon_shift)), "hmpc_set_tick_warm_start")

    def reset_tick_warm_start(self) -> None:
        _check(self.L.hmpc_reset_tick_warm_start(self.h), "hmpc_reset_tick_warm_start")

    def set_auto_resolve(self, on: bool) -> None:
        _check(self.L.hmpc_set_auto_resolve(self.h, 1 if on else 0), "hmpc_set_auto_resolve")

    def set_params(self, mass=None, inertia=None, mu=None, lt=None, lh=None, gravity=None) -> None:
        """Robot / contact constants (include/hector_mpc.h struct hmpc_params); arguments left out keep the reference's literals
        (mass 9.0, inertia diag (0.5413, 0.5200, 0.0691), mu 2.0, lt 0.09, lh 0.06, gravity 9.81); no arguments = all defaults."""
        p = _lib.Params()
        self.L.hmpc_default_params(C.byref(p))
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
        _check(self.L.hmpc_set_params(self.h, C.byref(p)), "hmpc_set_params")

    def set_instance_mu(self, device_ptr: int, keepalive=None) -> None:
        """Per-instance friction parameter (float32[batch] in HBM; 0 / None = off): terrain sweeps, also inside a command-sweep group
        (include/hector_mpc.h hmpc_set_instance_mu)."""
        self._keep_mu = keepalive
        _check(self.L.hmpc_set_instance_mu(self.h, C.c_void_p(int(device_ptr or 0))), "hmpc_set_instance_mu")

    def get_params(self) -> dict:
        p = _lib.Params()
        _check(self.L.hmpc_get_params(self.h, C.byref(p)), "hmpc_get_params")
        return dict(mass=float(p.mass), inertia=[float(v) for v in p.inertia], mu=float(p.mu), lt=float(p.lt), lh=float(p.lh),
                    gravity=float(p.gravity))

    def set_handover(self, on: bool) -> None:
        """Continue (default) or re-solve cold the instances whose working set outgrew the fast variant (hmpc_set_handover)."""
        _check(self.L.hmpc_set_handover(self.h, 1 if on else 0), "hmpc_set_handover")

    def set_device_repair(self, on) -> None:
        """Device-side safe pass inside every solve (include/hector_mpc.h hmpc_set_device_repair); 2 = continuation pass only."""
        _check(self.L.hmpc_set_device_repair(self.h, 2 if on == 2 else (1 if on else 0)), "hmpc_set_device_repair")

    def set_max_iterations(self, max_iter: int) -> None:
        """Cap on the active-set iterations (0 = none); the nWSR analogue (include/hector_mpc.h hmpc_set_max_iterations)."""
        _check(self.L.hmpc_set_max_iterations(self.h, int(max_iter)), "hmpc_set_max_iterations")

    def set_dispatch_order(self, mode) -> None:
        """0 / False: natural order; 1 / True (default): longest previous solve first, a cold handle by the cost predicted
        from the records; 2: always by the predictor (include/hector_mpc.h hmpc_set_dispatch_order).  Results do not depend
        on it."""
        _check(self.L.hmpc_set_dispatch_order(self.h, int(mode)), "hmpc_set_dispatch_order")

    def tick_solve_device(self, ticks_ptr: int, batch: int, dt_mpc: float, tau_ptr: int, f_ff_ptr: int = 0, wpd_ptr: int = 0,
                          stream: int = 0) -> None:
        """f1+f2 -> solve -> f3 on one stream, everything device-resident (include/hector_mpc.h hmpc_tick_solve_device)."""
        _check(self.L.hmpc_tick_solve_device(self.h, C.c_void_p(ticks_ptr), int(batch), float(dt_mpc), C.c_void_p(wpd_ptr),
                                             C.c_void_p(f_ff_ptr), C.c_void_p(tau_ptr), C.c_void_p(stream)),
               "hmpc_tick_solve_device")

    def resolve_failed(self) -> int:
        n = C.c_int(0)
        _check(self.L.hmpc_resolve_failed(self.h, C.byref(n)), "hmpc_resolve_failed")
        return int(n.value)

    def solve(self, stream: int = 0) -> None:
        _check(self.L.hmpc_solve(self.h, C.c_void_p(stream)), "hmpc_solve")

    def solve_command_sweep(self, group_size: int, stream: int = 0) -> None:
        """The current batch as groups of ``group_size`` consecutive records that differ in the reference trajectory only: H is
        assembled and inverted once per group, every instance solves with its group's inverse -- bit-identical results
        (include/hector_mpc.h hmpc_solve_command_sweep)."""
        _check(self.L.hmpc_solve_command_sweep(self.h, int(group_size), C.c_void_p(stream)), "hmpc_solve_command_sweep")

    def download(self):
        b = self.batch
        forces = np.zeros((b, self.nvar), dtype=np.float32)
        status = np.zeros(b, dtype=np.uint32)
        _check(self.L.hmpc_download(self.h, forces.ctypes.data, status.ctypes.data), "hmpc_download")
        return forces, status

    def download_f64(self):
        b = self.batch
        x = np.zeros((b, self.nvar), dtype=np.float64)
        obj = np.zeros(b, dtype=np.float64)
        _check(self.L.hmpc_download_f64(self.h, x.ctypes.data, obj.ctypes.data), "hmpc_download_f64")
        return x, obj

    def solve_external_qp(self, H: np.ndarray, g: np.ndarray, Fc: np.ndarray) -> None:
        """Parity hook (hmpc_debug_solve_external_qp): solver stages on caller-supplied QP data; H [batch, ld, ld] and
        g [batch, ld] in the reference's reduced order, Fc [batch, 8 nc, 6 nc]."""
        H = np.ascontiguousarray(H, dtype=np.float32)
        g = np.ascontiguousarray(g, dtype=np.float32)
        Fc = np.ascontiguousarray(Fc, dtype=np.float32)
        if not (H.ndim == 3 and H.shape[0] == self.batch and H.shape[1] == H.shape[2] == g.shape[1] and g.shape[0] == self.batch):
            raise ValueError(f"H {H.shape} / g {g.shape}: expected [batch={self.batch}, ld, ld] and [batch, ld]")
        if Fc.shape != (self.batch, 8 * self.contacts, 6 * self.contacts):
            raise ValueError(f"Fc {Fc.shape}: expected {(self.batch, 8 * self.contacts, 6 * self.contacts)}")
        _check(self.L.hmpc_debug_solve_external_qp(self.h, H.ctypes.data, g.ctypes.data, Fc.ctypes.data, H.shape[1]),
               "hmpc_debug_solve_external_qp")

    # ---- rows either side of the solve
    def build_records(self, ticks: np.ndarray, dt_mpc: float):
        """f1+f2 on the device: ticks = structured array with dtype ``TICK_DTYPE``; returns the clamped
        world_position_desired [batch, 2].  The built records become the current batch."""
        ticks = np.ascontiguousarray(ticks, dtype=TICK_DTYPE)
        wpd = np.zeros((ticks.shape[0], 2), dtype=np.float64)
        _check(self.L.hmpc_build_records(self.h, ticks.ctypes.data, ticks.shape[0], float(dt_mpc), wpd.ctypes.data),
               "hmpc_build_records")
        return wpd

    def download_records(self) -> np.ndarray:
        rec = np.zeros((self.batch, self.stride), dtype=np.uint8)
        _check(self.L.hmpc_download_records(self.h, rec.ctypes.data), "hmpc_download_records")
        return rec

    def body_wrench(self, rBody: np.ndarray) -> np.ndarray:
        """f3: f_ff[batch, 2, 6] = -rBody [GRF; GRM] from the last solve's forces."""
        rb = np.ascontiguousarray(rBody, dtype=np.float64).reshape(self.batch, 9)
        out = np.zeros((self.batch, 2, 6), dtype=np.float64)
        _check(self.L.hmpc_body_wrench(self.h, rb.ctypes.data, out.ctypes.data), "hmpc_body_wrench")
        return out

    def leg_torques(self, rBody: np.ndarray, leg_q: np.ndarray):
        """f3 with torques: returns (f_ff[batch,2,6], tau[batch,2,5]) from the last solve's forces."""
        rb = np.ascontiguousarray(rBody, dtype=np.float64).reshape(self.batch, 9)
        lq = np.ascontiguousarray(leg_q, dtype=np.float64).reshape(self.batch, 10)
        fff = np.zeros((self.batch, 2, 6), dtype=np.float64)
        tau = np.zeros((self.batch, 2, 5), dtype=np.float64)
        _check(self.L.hmpc_leg_torques(self.h, rb.ctypes.data, lq.ctypes.data, fff.ctypes.data, tau.ctypes.data),
               "hmpc_leg_torques")
        return fff, tau

    def time_solve(self, reps: int, stream: int = 0) -> float:
        ms = C.c_float(0)
        _check(self.L.hmpc_time_solve(self.h, C.c_void_p(stream), int(reps), C.byref(ms)), "hmpc_time_solve")
        return float(ms.value)

    def debug_assemble(self, index: int) -> dict:
        """Assembly stage only (same device code the solve kernel runs) -> the reduced QP as the solver sees it."""
        h, nc = self.horizon, self.contacts
        nmax = 180 if nc == 3 else 240  # HMPC_MAX_VARS_3C / HMPC_MAX_VARS_WIDE
        n, m = C.c_int(0), C.c_int(0)
        var_ind = np.zeros(nmax, dtype=np.int32)
        H = np.zeros(nmax * nmax, dtype=np.float32)
        g = np.zeros(nmax, dtype=np.float32)
        Fc = np.zeros(48 * nc * nc, dtype=np.float32)
        lb = np.zeros(8 * nc * h, dtype=np.float32)
        ub = np.zeros(8 * nc * h, dtype=np.float32)
        x0 = np.zeros(13, dtype=np.float32)
        Acd = np.zeros(169, dtype=np.float32)
        Bcd = np.zeros(78 * nc, dtype=np.float32)
        _check(self.L.hmpc_debug_assemble(self.h, int(index), C.byref(n), C.byref(m), var_ind.ctypes.data,
                                          H.ctypes.data, g.ctypes.data, Fc.ctypes.data, lb.ctypes.data,
                                          ub.ctypes.data, x0.ctypes.data, Acd.ctypes.data, Bcd.ctypes.data),
               "hmpc_debug_assemble")
        nn = n.value
        if nn > nmax:
            return dict(n=nn, m=m.value)
        return dict(n=nn, m=m.value, var_ind=var_ind[:nn].copy(), H=H[: nn * nn].reshape(nn, nn).copy(), g=g[:nn].copy(),
                    Fc=Fc.reshape(8 * nc, 6 * nc), lb=lb, ub=ub, x0=x0, Acd=Acd.reshape(13, 13),
                    Bcd=Bcd.reshape(13, 6 * nc))


class DeviceGroup:
    """``hmpc_group_*`` (include/hector_mpc.h, csrc/hmpc_group.hip): one process, several GPUs, contiguous slices, the
    step-0 wrench + status gathered on every member and on the host.  ``devices`` may repeat an index with
    ``transport="p2p"`` (one-GPU exercise of the multi-member path)."""

    TRANSPORT = {"auto": 0, "rccl": 1, "p2p": 2}

    def __init__(self, dt: float, horizon: int, f_max: float, max_batch: int, devices, transport: str = "auto",
                 mu: float = 0.25, contacts: int = 2):
        self.L = _lib.load()
        self.horizon, self.max_batch, self.contacts = int(horizon), int(max_batch), int(contacts)
        self.setup = _lib.ProblemSetup(np.float32(dt), np.float32(mu), np.float32(f_max), int(horizon))
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self.g = C.c_void_p()
        rc = self.L.hmpc_group_create_ex(C.byref(self.g), C.byref(self.setup), devs.ctypes.data, len(devs), self.max_batch,
                                         self.TRANSPORT[transport], self.contacts)
        if rc != 0:
            raise HmpcError(f"hmpc_group_create_ex failed with {rc}: {self.L.hmpc_group_last_error().decode()}")
        self.size = int(self.L.hmpc_group_size(self.g))
        self.stride = int(self.L.hmpc_record_stride_ex(self.horizon, self.contacts))
        self.wrench_width = 6 * self.contacts

    def _check(self, rc, what):
        if rc != 0:
            raise HmpcError(f"{what} failed with {rc}: {self.L.hmpc_group_last_error().decode()}")

    def close(self):
        if getattr(self, "g", None):
            self.L.hmpc_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def transport(self) -> str:
        return {1: "rccl", 2: "p2p"}[int(self.L.hmpc_group_transport(self.g))]

    @property
    def batch(self) -> int:
        return int(self.L.hmpc_group_batch(self.g))

    def member(self, i: int):
        """(handle pointer, device, lo, n, solve stream) of member i."""
        h, st = C.c_void_p(), C.c_void_p()
        dev, lo, n = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self.L.hmpc_group_member(self.g, int(i), C.byref(h), C.byref(dev), C.byref(lo), C.byref(n), C.byref(st)),
                    "hmpc_group_member")
        return h, dev.value, lo.value, n.value, st.value

    def upload(self, recs: np.ndarray) -> None:
        recs = np.ascontiguousarray(recs, dtype=np.uint8)
        assert recs.ndim == 2 and recs.shape[1] == self.stride
        self._check(self.L.hmpc_group_upload_records(self.g, recs.ctypes.data, recs.shape[0]), "hmpc_group_upload_records")

    def set_device_records(self, member_ptrs, batch: int, max_reduced_vars: int = -1, keepalive=None) -> None:
        """Records already resident on each member's device: member_ptrs[i] = device pointer of member i's first record
        (slice sizes follow ``shard_bounds``)."""
        self._keep = keepalive
        arr = (C.c_void_p * self.size)(*[C.c_void_p(int(p)) for p in member_ptrs])
        self._check(self.L.hmpc_group_set_device_records(self.g, arr, int(batch), int(max_reduced_vars)),
                    "hmpc_group_set_device_records")

    def solve(self) -> None:
        self._check(self.L.hmpc_group_solve(self.g), "hmpc_group_solve")

    def solve_command_sweep(self, group_size: int) -> None:
        """hmpc_solve_command_sweep on every member (slices must consist of whole groups)."""
        self._check(self.L.hmpc_group_solve_command_sweep(self.g, int(group_size)), "hmpc_group_solve_command_sweep")

    def set_deal(self, striped: bool) -> None:
        """Contiguous slices (default) or round-robin: member i holds instances i, i + G, ... (hmpc_group_set_deal)."""
        self._check(self.L.hmpc_group_set_deal(self.g, 1 if striped else 0), "hmpc_group_set_deal")

    def member_step(self, i: int) -> int:
        return int(self.L.hmpc_group_member_step(self.g, int(i)))

    def set_exchange_repair(self, on: bool) -> None:
        self._check(self.L.hmpc_group_set_exchange_repair(self.g, 1 if on else 0), "hmpc_group_set_exchange_repair")

    def post_gather(self) -> None:
        self._check(self.L.hmpc_group_post_gather(self.g), "hmpc_group_post_gather")

    def wait_gather(self) -> None:
        self._check(self.L.hmpc_group_wait_gather(self.g), "hmpc_group_wait_gather")

    def gather_wrench(self):
        b = self.batch
        wrench = np.zeros((b, self.wrench_width), dtype=np.float32)
        status = np.zeros(b, dtype=np.uint32)
        self._check(self.L.hmpc_group_gather_wrench(self.g, wrench.ctypes.data, status.ctypes.data), "hmpc_group_gather_wrench")
        return wrench, status

    def device_gathered(self, member: int):
        """(device pointer of member's gathered block, rows per slot)."""
        p, rows = C.c_void_p(), C.c_int(0)
        self._check(self.L.hmpc_group_device_gathered(self.g, int(member), C.byref(p), C.byref(rows)),
                    "hmpc_group_device_gathered")
        return p.value, rows.value

    def download(self):
        b = self.batch
        forces = np.zeros((b, self.wrench_width * self.horizon), dtype=np.float32)
        status = np.zeros(b, dtype=np.uint32)
        self._check(self.L.hmpc_group_download(self.g, forces.ctypes.data, status.ctypes.data), "hmpc_group_download")
        return forces, status

    def synchronize(self) -> None:
        self._check(self.L.hmpc_group_synchronize(self.g), "hmpc_group_synchronize")


def shard_bounds(global_batch: int, n_shards: int, index: int):
    """``hmpc_shard_bounds`` of the C ABI (pure host arithmetic; works without a GPU)."""
    lo, hi = C.c_int(0), C.c_int(0)
    rc = _lib.load().hmpc_shard_bounds(int(global_batch), int(n_shards), int(index), C.byref(lo), C.byref(hi))
    if rc != 0:
        raise ValueError((global_batch, n_shards, index))
    return lo.value, hi.value


def status_code(status: np.ndarray) -> np.ndarray:
    return (np.asarray(status) & 0xFF).astype(np.int32)


def status_iters(status: np.ndarray) -> np.ndarray:
    return ((np.asarray(status) >> 8) & 0xFFF).astype(np.int32)


def status_nactive(status: np.ndarray) -> np.ndarray:
    return ((np.asarray(status) >> 20) & 0xFFF).astype(np.int32)
