"""ctypes loader of libhector_mpc_hip.so.  Fails loudly: there is no CPU / PyTorch fallback for the solve path."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_lib = None

# every symbol include/hector_mpc.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "setup_problem", "update_problem_data", "get_solution", "update_solver_settings", "hmpc_solve_mpc", "solveDenseMPC",
    "hmpc_get_q_soln", "hmpc_last_status", "hmpc_record_stride", "hmpc_pack_record", "hmpc_create", "hmpc_destroy",
    "hmpc_upload_records", "hmpc_set_device_records", "hmpc_set_max_reduced_vars", "hmpc_set_warm_start", "hmpc_resolve_failed", "hmpc_set_auto_resolve", "hmpc_set_device_outputs",
    "hmpc_solve", "hmpc_download", "hmpc_get_device_outputs", "hmpc_batch", "hmpc_horizon", "hmpc_time_solve",
    "hmpc_build_records", "hmpc_build_records_device", "hmpc_body_wrench", "hmpc_body_wrench_device", "hmpc_leg_torques", "hmpc_leg_torques_device",
    "hmpc_download_records", "hmpc_debug_assemble", "hmpc_debug_phase_cycles", "hmpc_download_f64", "hmpc_last_hip_error", "hmpc_version",
    "hmpc_upload_records_async", "hmpc_download_async", "hmpc_set_tick_warm_start", "hmpc_reset_tick_warm_start", "hmpc_create_ex", "hmpc_contacts", "hmpc_record_stride_ex", "hmpc_pack_record_ex",
    "hmpc_enable_f64_output", "hmpc_debug_solve_external_qp", "hmpc_set_device_repair", "hmpc_group_set_exchange_repair",
    "hmpc_shard_bounds", "hmpc_group_create", "hmpc_group_destroy", "hmpc_group_size", "hmpc_group_transport",
    "hmpc_group_batch", "hmpc_group_member", "hmpc_group_upload_records", "hmpc_group_set_device_records",
    "hmpc_group_solve", "hmpc_group_post_gather", "hmpc_group_wait_gather", "hmpc_group_device_gathered",
    "hmpc_group_gather_wrench", "hmpc_group_download", "hmpc_group_synchronize", "hmpc_group_last_error",
    "hmpc_group_create_ex", "hmpc_group_contacts", "hmpc_group_set_deal", "hmpc_group_deal", "hmpc_group_member_step",
    "hmpc_upload_records_strided_async", "hmpc_set_max_iterations", "hmpc_legacy_set_max_iterations", "hmpc_tick_solve_device", "hmpc_set_dispatch_order",
    "hmpc_set_handover", "hmpc_default_params", "hmpc_set_params", "hmpc_get_params", "hmpc_legacy_set_params", "hmpc_group_set_params",
    "hmpc_solve_command_sweep", "hmpc_set_instance_mu", "hmpc_group_solve_command_sweep",
]


class ProblemSetup(C.Structure):
    _fields_ = [("dt", C.c_float), ("mu", C.c_float), ("f_max", C.c_float), ("horizon", C.c_int)]


class Params(C.Structure):
    """include/hector_mpc.h struct hmpc_params (robot / contact constants; defaults = the reference's literals)."""
    _fields_ = [("mass", C.c_float), ("inertia", C.c_float * 3), ("mu", C.c_float), ("lt", C.c_float), ("lh", C.c_float),
                ("gravity", C.c_float)]


class TickInputs(C.Structure):
    """include/hector_mpc.h struct hmpc_tick_inputs (what updateMPCIfNeeded reads for one tick)."""
    _fields_ = [("position", C.c_double * 3), ("vWorld", C.c_double * 3), ("omegaWorld", C.c_double * 3),
                ("orientation", C.c_double * 4), ("rpy", C.c_double * 3), ("rBody", C.c_double * 9),
                ("leg_q", C.c_double * 10), ("pFoot", C.c_double * 6), ("v_des_robot", C.c_double * 2),
                ("yaw_rate_des", C.c_double), ("roll_des", C.c_double), ("pitch_des", C.c_double),
                ("world_position_desired", C.c_double * 2), ("gait_offsets", C.c_int * 2),
                ("gait_durations", C.c_int * 2), ("gait_iteration", C.c_int), ("flags", C.c_int)]


class UpdateData(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("v", C.c_float * 3), ("q", C.c_float * 4), ("w", C.c_float * 3),
                ("r", C.c_float * 6), ("joint_angles", C.c_float * 10), ("yaw", C.c_float),
                ("weights", C.c_float * 12), ("traj", C.c_float * (12 * 36)), ("Alpha_K", C.c_float * 12),
                ("gait", C.c_ubyte * 36), ("hack_pad", C.c_ubyte * 1000), ("max_iterations", C.c_int),
                ("rho", C.c_double), ("sigma", C.c_double), ("solver_alpha", C.c_double), ("terminate", C.c_double)]


def lib_path() -> str:
    return _build.LIB


def load():
    """Loads (building first if the sources are newer) the native library; raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build()
    if not os.path.exists(path):
        raise RuntimeError("libhector_mpc_hip.so is missing and could not be built; the solver has no fallback path")
    # ONE HIP runtime per process: PyTorch-ROCm bundles a libamdhip64.so.7 and this library's DT_NEEDED names the same SONAME, so the
    # copy the loader maps first serves both.  torch first = torch's bundled runtime for both (works; bench.py's order); this library
    # first = the system runtime for both, on which torch has been seen to report "no ROCm-capable device".  So if torch is installed,
    # let it come up first.  (A C++ host such as the reference controller has no torch in the process: the system runtime, the one the
    # library was built against, serves it.  tests/test_gpu_runtime.py asserts the single mapping.)
    if os.environ.get("HMPC_NO_TORCH_PRELOAD") != "1":
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass
    L = C.CDLL(path)
    vp, ci, cd, cf = C.c_void_p, C.c_int, C.c_double, C.c_float
    L.setup_problem.argtypes = [cd, ci, cd, cd]
    L.setup_problem.restype = None
    L.update_problem_data.argtypes = [vp] * 6 + [cd] + [vp] * 4
    L.update_problem_data.restype = None
    L.get_solution.argtypes = [ci]
    L.get_solution.restype = cd
    L.update_solver_settings.argtypes = [ci, cd, cd, cd, cd, cd]
    L.update_solver_settings.restype = None
    L.hmpc_solve_mpc.argtypes = [C.POINTER(UpdateData), C.POINTER(ProblemSetup)]
    L.hmpc_solve_mpc.restype = None
    L.solveDenseMPC.argtypes = [C.POINTER(UpdateData), C.POINTER(ProblemSetup)]
    L.solveDenseMPC.restype = None
    L.hmpc_get_q_soln.restype = C.POINTER(cd)
    L.hmpc_last_status.restype = C.c_uint32
    L.hmpc_record_stride.argtypes = [ci]
    L.hmpc_record_stride.restype = C.c_size_t
    L.hmpc_pack_record.argtypes = [vp, ci] + [vp] * 6 + [cd] + [vp] * 4
    L.hmpc_create.argtypes = [C.POINTER(vp), C.POINTER(ProblemSetup), ci, ci]
    L.hmpc_create_ex.argtypes = [C.POINTER(vp), C.POINTER(ProblemSetup), ci, ci, ci]
    L.hmpc_contacts.argtypes = [vp]
    L.hmpc_record_stride_ex.argtypes = [ci, ci]
    L.hmpc_record_stride_ex.restype = C.c_size_t
    L.hmpc_pack_record_ex.argtypes = [vp, ci, ci] + [vp] * 6 + [cd] + [vp] * 5 + [cd]
    L.hmpc_destroy.argtypes = [vp]
    L.hmpc_upload_records.argtypes = [vp, vp, ci]
    L.hmpc_set_device_records.argtypes = [vp, vp, ci]
    L.hmpc_set_max_reduced_vars.argtypes = [vp, ci]
    L.hmpc_set_warm_start.argtypes = [vp, ci]
    L.hmpc_upload_records_async.argtypes = [vp, vp, ci, vp]
    L.hmpc_download_async.argtypes = [vp, vp, vp, vp]
    L.hmpc_set_tick_warm_start.argtypes = [vp, ci, ci]
    L.hmpc_reset_tick_warm_start.argtypes = [vp]
    L.hmpc_resolve_failed.argtypes = [vp, C.POINTER(ci)]
    L.hmpc_set_auto_resolve.argtypes = [vp, ci]
    L.hmpc_set_handover.argtypes = [vp, ci]
    L.hmpc_default_params.argtypes = [C.POINTER(Params)]
    L.hmpc_default_params.restype = None
    L.hmpc_set_params.argtypes = [vp, C.POINTER(Params)]
    L.hmpc_get_params.argtypes = [vp, C.POINTER(Params)]
    L.hmpc_legacy_set_params.argtypes = [C.POINTER(Params)]
    L.hmpc_group_set_params.argtypes = [vp, C.POINTER(Params)]
    L.hmpc_set_device_repair.argtypes = [vp, ci]
    L.hmpc_group_set_exchange_repair.argtypes = [vp, ci]
    L.hmpc_set_device_outputs.argtypes = [vp, vp, vp]
    L.hmpc_solve.argtypes = [vp, vp]
    L.hmpc_solve_command_sweep.argtypes = [vp, ci, vp]
    L.hmpc_set_instance_mu.argtypes = [vp, vp]
    L.hmpc_group_solve_command_sweep.argtypes = [vp, ci]
    L.hmpc_download.argtypes = [vp, vp, vp]
    L.hmpc_get_device_outputs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.hmpc_batch.argtypes = [vp]
    L.hmpc_horizon.argtypes = [vp]
    L.hmpc_time_solve.argtypes = [vp, vp, ci, C.POINTER(cf)]
    L.hmpc_debug_assemble.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)] + [vp] * 9
    L.hmpc_download_f64.argtypes = [vp, vp, vp]
    L.hmpc_build_records.argtypes = [vp, vp, ci, cd, vp]
    L.hmpc_build_records_device.argtypes = [vp, vp, ci, cd, vp, vp]
    L.hmpc_body_wrench.argtypes = [vp, vp, vp]
    L.hmpc_body_wrench_device.argtypes = [vp, vp, vp, vp]
    L.hmpc_download_records.argtypes = [vp, vp]
    L.hmpc_leg_torques.argtypes = [vp, vp, vp, vp, vp]
    L.hmpc_leg_torques_device.argtypes = [vp, vp, vp, vp, vp, vp]
    L.hmpc_enable_f64_output.argtypes = [vp]
    L.hmpc_debug_solve_external_qp.argtypes = [vp, vp, vp, vp, ci]
    L.hmpc_shard_bounds.argtypes = [ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
    L.hmpc_group_create.argtypes = [C.POINTER(vp), C.POINTER(ProblemSetup), vp, ci, ci, ci]
    L.hmpc_group_create_ex.argtypes = [C.POINTER(vp), C.POINTER(ProblemSetup), vp, ci, ci, ci, ci]
    L.hmpc_group_contacts.argtypes = [vp]
    L.hmpc_group_set_deal.argtypes = [vp, ci]
    L.hmpc_group_deal.argtypes = [vp]
    L.hmpc_group_member_step.argtypes = [vp, ci]
    L.hmpc_upload_records_strided_async.argtypes = [vp, vp, ci, C.c_size_t, vp]
    L.hmpc_set_max_iterations.argtypes = [vp, ci]
    L.hmpc_legacy_set_max_iterations.argtypes = [ci]
    L.hmpc_legacy_set_max_iterations.restype = ci
    L.hmpc_set_dispatch_order.argtypes = [vp, ci]
    L.hmpc_tick_solve_device.argtypes = [vp, vp, ci, cd, vp, vp, vp, vp]
    for name in ("hmpc_group_destroy", "hmpc_group_size", "hmpc_group_transport", "hmpc_group_batch", "hmpc_group_solve",
                 "hmpc_group_post_gather", "hmpc_group_wait_gather", "hmpc_group_synchronize"):
        getattr(L, name).argtypes = [vp]
    L.hmpc_group_member.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), C.POINTER(vp)]
    L.hmpc_group_upload_records.argtypes = [vp, vp, ci]
    L.hmpc_group_set_device_records.argtypes = [vp, vp, ci, ci]
    L.hmpc_group_device_gathered.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(ci)]
    L.hmpc_group_gather_wrench.argtypes = [vp, vp, vp]
    L.hmpc_group_download.argtypes = [vp, vp, vp]
    L.hmpc_group_last_error.restype = C.c_char_p
    L.hmpc_debug_phase_cycles.argtypes = [vp, vp]
    L.hmpc_last_hip_error.restype = C.c_char_p
    L.hmpc_version.restype = C.c_char_p
    _lib = L
    return L
