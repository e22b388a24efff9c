"""examples/: the C ABI used from the reference's own languages (C++ and C), compiled against include/hector_mpc.h and
linked to the in-tree library.  Without a GPU the programs must fail loudly (no CPU fallback); with one they must solve."""
import os
import subprocess

import pytest

from hector_simulation_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hector_simulation_amd")


def _compile(tmp_path, src, cc, std, folder="examples"):
    build.build()
    exe = str(tmp_path / os.path.splitext(src)[0])
    cmd = [cc, std, "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, folder, src), "-L" + PKG,
           "-lhector_mpc_hip", "-lm", "-Wl,-rpath," + PKG, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _has_gpu():
    import torch

    return torch.cuda.is_available()


@pytest.mark.parametrize("src,cc,std", [("legacy_tick.cpp", "g++", "-std=c++17"), ("batched.c", "gcc", "-std=c11"),
                                        ("batched_multi.c", "gcc", "-std=c11"), ("friction_sweep.c", "gcc", "-std=c11")])
def test_examples_compile_and_fail_loudly_without_gpu(tmp_path, src, cc, std):
    exe = _compile(tmp_path, src, cc, std)
    if _has_gpu():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no HIP device" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("src,cc,std", [("legacy_tick.cpp", "g++", "-std=c++17"), ("batched.c", "gcc", "-std=c11"),
                                        ("friction_sweep.c", "gcc", "-std=c11")])
def test_examples_run_on_gpu(tmp_path, src, cc, std):
    exe = _compile(tmp_path, src, cc, std)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    if src == "friction_sweep.c":  # hmpc_set_params sweep + command sweep (round 6)
        assert "bit-identical to the independent solves" in r.stdout and r.stdout.count("friction pyramid") == 8
    if src == "legacy_tick.cpp":
        u0 = [float(x) for x in r.stdout.split("u0 =")[1].split()]
        assert abs(u0[2] - 47.84) < 0.05 and abs(u0[5] - 47.84) < 0.05  # the nominal standing tick (tests/test_gpu_solve.py)


@pytest.mark.gpu
@pytest.mark.parametrize("args", [["1"], ["3", "p2p"], ["4", "p2p", "3"], ["1", "auto", "3"], ["3", "p2p", "2", "striped"]])
def test_multi_device_example_runs_on_gpu(tmp_path, args):
    """examples/batched_multi.c: a group of one over RCCL (what a one-GPU box can run of the real transport) and a group of
    three members on device 0 over the P2P transport (ragged 334/333/333 slices); with "3": groups of three-contact handles
    (BASELINE config 5's split: four members over P2P, one over RCCL)."""
    exe = _compile(tmp_path, "batched_multi.c", "gcc", "-std=c11")
    r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of 1000 not ok, 0 gathered rows differ" in r.stdout
    assert f"{args[2] if len(args) > 2 else 2} contacts" in r.stdout
    if "striped" in args:
        assert "334 instances 0, 3, 6, ..." in r.stdout and "333 instances 2, 5, 8, ..." in r.stdout


def test_host_api_sweep_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = _compile(tmp_path, "host_api_sweep.c", "gcc", "-std=c11", folder=os.path.join("tests", "src"))
    if _has_gpu():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_host_api_sweep_on_gpu(tmp_path):
    """tests/src/host_api_sweep.c: every batched entry point from plain C, error paths, scratch reuse, safe pass, f64
    copy-out before/after, tick warm start, device group (also the program scripts/sanitize_host.sh runs under ASan/UBSan)."""
    exe = _compile(tmp_path, "host_api_sweep.c", "gcc", "-std=c11", folder=os.path.join("tests", "src"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, cwd=ROOT)  # (reads tests/golden/indefinite_*.bin relative to the root)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host API sweep ok" in r.stdout
    assert "indefinite Hessians: 2 of 3 regularised as the reference does" in r.stdout
