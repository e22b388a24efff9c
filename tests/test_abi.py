"""The C-ABI library loads on a CPU-only box, exports every symbol include/hector_mpc.h declares, its host-side
helpers agree with the Python mirror, and the solve path FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from hector_simulation_amd import _lib, interface, records, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "hector_mpc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\([^;{]*\)\s*;", src, flags=re.M)
    return sorted(set(names))


def test_exports_every_declared_symbol():
    L = _lib.load()
    decl = declared_functions()
    assert len(decl) >= 25, decl
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/hector_mpc.h but not exported"
    assert set(decl) == set(_lib.EXPORTS)
    # the reference's C++-linkage names are exported as well (SolverMPC.h:56,63)
    syms = os.popen(f"nm -D --defined-only {_lib.lib_path()}").read()
    assert "_Z9solve_mpcP13update_data_tP13problem_setup" in syms and "_Z10get_q_solnv" in syms


def test_pod_layout_matches_reference():
    # convexMPC_interface.h:11-37 (float/int/uchar/double fields, natural alignment)
    assert C.sizeof(_lib.ProblemSetup) == 16
    assert _lib.UpdateData.traj.offset == (3 + 3 + 4 + 3 + 6 + 10 + 1 + 12) * 4
    assert _lib.UpdateData.gait.offset == _lib.UpdateData.Alpha_K.offset + 48
    assert _lib.UpdateData.max_iterations.offset % 4 == 0 and _lib.UpdateData.rho.offset % 8 == 0


def test_record_helpers_agree_with_python():
    L = _lib.load()
    for h in (1, 5, 10, 19, 20):
        assert L.hmpc_record_stride(h) == records.record_stride(h)
    f = synthetic.make_batch(3, 10, "walking", seed=8, phase="random")
    rec = records.pack_records(f, 10)
    for k in range(3):
        buf = np.zeros(720, dtype=np.uint8)
        arrs = [np.ascontiguousarray(np.asarray(f[key])[k], dtype=np.float64)
                for key in ("p", "v", "q", "w", "r", "joint_angles")]
        tail = [np.ascontiguousarray(np.asarray(f[key])[k], dtype=np.float64) for key in ("weights", "traj", "Alpha_K")]
        gait = np.ascontiguousarray(np.asarray(f["gait"])[k], dtype=np.int32)
        rc = L.hmpc_pack_record(buf.ctypes.data, 10, *[a.ctypes.data for a in arrs], float(f["yaw"][k]),
                                *[a.ctypes.data for a in tail], gait.ctypes.data)
        assert rc == 0
        np.testing.assert_array_equal(buf, rec[k])
    assert L.hmpc_pack_record(None, 10, *([None] * 6), 0.0, *([None] * 4)) == -1


def test_legacy_entry_points_do_not_throw():
    L = _lib.load()
    assert L.get_solution(0) == 0.0  # convexMPC_interface.cpp:107: 0 before the first solve
    L.update_solver_settings(100, 1e-6, 1e-6, 1.5, 1e-9, 0.0)  # stored, read by nothing (as in the reference)
    L.setup_problem(0.04, 25, 0.25, 500.0)  # the reference throws for horizon > 19; we never throw across C
    assert L.get_solution(0) == 0.0


def test_fails_loudly_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    with pytest.raises(interface.HmpcError):
        interface.BatchedMPC(0.04, 10, 500.0, 4)
    h = C.c_void_p()
    s = _lib.ProblemSetup(0.04, 0.25, 500.0, 10)
    assert _lib.load().hmpc_create(C.byref(h), C.byref(s), 4, 0) == -5  # HMPC_E_NO_DEVICE
    s.horizon = 21
    assert _lib.load().hmpc_create(C.byref(h), C.byref(s), 4, 0) == -2  # HMPC_E_HORIZON


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "hector_simulation_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), (dp, fn)
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*oracle", txt), (dp, fn)
                assert "libhmpc_oracle" not in txt and "libqpoases_ref" not in txt and "orc_" not in txt, (dp, fn)
    assert "hmpc_oracle" not in os.popen(f"ldd {_lib.lib_path()}").read()


def test_shard_bounds_in_c_matches_the_python_sharding():
    """hmpc_shard_bounds (C ABI, pure host arithmetic) == hector_simulation_amd.sharding.shard_bounds: contiguous cover,
    the first batch % n shards one longer."""
    from hector_simulation_amd import interface, sharding

    for batch in (0, 1, 7, 64, 100, 1000, 8192, 65536):
        for n in (1, 2, 3, 4, 8):
            prev = 0
            for i in range(n):
                lo, hi = interface.shard_bounds(batch, n, i)
                assert (lo, hi) == sharding.shard_bounds(batch, n, i)
                assert lo == prev
                prev = hi
            assert prev == batch
    with pytest.raises(ValueError):
        interface.shard_bounds(10, 2, 2)


def test_developer_switches_cannot_reach_the_product_library():
    """-DHMPC_DEBUG_STATS (debug counters written where the objective value goes) and -DHMPC_PROFILE are developer builds: build.py
    refuses them for the library the package loads -- also when a library built with them is already there -- unless
    HMPC_ALLOW_DEV_BUILD=1.  The timing switches of rounds 4-5 that left stages of the kernel out (HMPC_MFS_NO_*) are gone."""
    from hector_simulation_amd import build

    saved, saved_env = build.EXTRA[:], os.environ.pop("HMPC_ALLOW_DEV_BUILD", None)
    try:
        for flag in ("-DHMPC_DEBUG_STATS", "-DHMPC_PROFILE"):
            build.EXTRA[:] = [flag]
            with pytest.raises(RuntimeError):
                build._check_flags()
            with pytest.raises(RuntimeError):
                build.build()  # (checked before the "library is up to date" shortcut)
        build.EXTRA[:] = ["-DHMPC_CONT_ROUNDS=2"]  # a same-results tuning switch is fine
        build._check_flags()
        os.environ["HMPC_ALLOW_DEV_BUILD"] = "1"
        build.EXTRA[:] = ["-DHMPC_DEBUG_STATS"]
        build._check_flags()
    finally:
        build.EXTRA[:] = saved
        os.environ.pop("HMPC_ALLOW_DEV_BUILD", None)
        if saved_env is not None:
            os.environ["HMPC_ALLOW_DEV_BUILD"] = saved_env
    ksrc = open(os.path.join(ROOT, "hector_simulation_amd", "csrc", "hmpc_kernel.h")).read()
    for gone in ("HMPC_MFS_NO_", "HMPC_MFS_ONLY_WAVE", "HMPC_DEV_TIMING", "HMPC_MFMA_SWEEP1", "HMPC_FLIP4", "HMPC_PIN_SWEEP", "HMPC_S0_ACTIVE_ROWS"):
        assert gone not in ksrc, gone


def test_every_kernel_variant_is_built_by_exactly_one_translation_unit():
    """csrc/hmpc_variants.h: sixteen variants over HMPC_VARIANT_GROUPS groups, build.py compiles one unit per group."""
    from hector_simulation_amd import build

    hdr = open(os.path.join(ROOT, "hector_simulation_amd", "csrc", "hmpc_variants.h")).read()
    rows = re.findall(r"X\((\d+), (\d+),", hdr)
    assert [int(i) for i, _ in rows] == list(range(16))
    groups = sorted({int(g) for _, g in rows})
    assert groups == list(range(build.VARIANT_GROUPS))
    assert f"HMPC_VARIANT_GROUPS = {build.VARIANT_GROUPS}" in hdr
    units = build.compile_commands("/tmp/x", "hipcc")
    assert len(units) == build.VARIANT_GROUPS + len(build.HOST_SOURCES)
