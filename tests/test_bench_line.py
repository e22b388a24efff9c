"""bench.py keeps its contract: ONE JSON line with the keys the driver and the judge read; without a GPU it fails loudly."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch

    return torch.cuda.is_available()


def test_bench_fails_loudly_without_gpu():
    if _has_gpu():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--batch", "2048",
                        "--no-side-configs", "--check", "16", "--cpu-per-core", "8", "--cpu-seconds", "1.5"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line on stdout (the reference-source baseline's prints go to /dev/null)
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0
    assert abs(d["value"] - 2048 * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6  # value = units / timed seconds
    assert d["parity"]["max_rel_force_err_vs_qpoases"] < 1e-4 and d["solver"]["failed"] == 0
    e2e = d["parity"]["vs_reference_source_end_to_end"]  # measured in this run, not a prose string
    assert e2e["checked"] == 16 and 0 < e2e["max_rel_force_err"] < 7.2e-4 and 0 <= e2e["fraction_above_1e-4"] <= 1
    assert len(cb["process_sweep"]) >= 2 and cb["process_sweep"][0]["processes"] == 1 and cb["host"]["nproc_affinity"] >= 1
    assert cb["saturation_processes"] >= 1 and "scaling_note" in cb
    # quotable: the value is the median of three timed-to-target runs, the spread is in the line
    assert len(cb["runs_solves_per_s"]) == 3 and cb["value_min"] <= cb["value"] <= cb["value_max"]
    assert min(cb["runs_wall_s"]) > 0.4 * 1.5  # (timed to a target from a rate estimate: a loaded box lands below it)
    assert d["parity"]["max_rel_objective_gap"] < 1e-4 and d["parity"]["kkt"]["max_rel_row_violation"] < 1e-6
    assert abs(d["parity"]["kkt"]["max_rel_suboptimality"]) < 1e-6 and d["parity"]["kkt"]["max_rel_stationarity_residual"] < 1e-6
    for k in ("fp64_valu_frac", "iterations_per_solve", "single_stream"):
        assert k in d, k


@pytest.mark.gpu
def test_bench_torchrun_code_path_on_one_gpu():
    """The N>1 code path of bench.py -- process group over RCCL, one posted all_gather of the step-0 wrench + status per
    solve on the communicator's stream, double-buffered against the two launch streams -- launched exactly as the driver
    launches it (torch.distributed.run), with a group of one and the collective forced on, so that it executes on hardware
    in the GPU test run although no multi-GPU box is available to this build."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--steps", "6", "--warmup", "2", "--batch", "1024", "--no-side-configs", "--check", "8",
                        "--no-cpu-baseline", "--force-exchange"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["exchange_selfcheck_ok"] is True
    assert "exchange code path forced on" in d["config"]["parallelism"]
    assert d["solver"]["failed"] == 0 and d["parity"]["max_rel_force_err_vs_qpoases"] < 1e-4
    assert d["parity"]["max_rel_objective_gap"] < 1e-4


def _torchrun(nproc, port, *bench_args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(nproc), *bench_args], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone prints
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra,batch", [
    ("headline_2contact", [], 1024),
    ("cfg3_walking_sweep", ["--gait", "walking"], 2048),
    ("cfg5_three_contact", ["--contacts", "3"], 256),
])
def test_bench_two_ranks_on_one_gpu(name, extra, batch):
    """Rank > 0 code on hardware: bench.py launched exactly as the driver launches N = 2 (torch.distributed.run, one
    process per rank), both ranks on the one GPU of the test box, the exchange over the gloo TEST transport (RCCL refuses
    two ranks on one device).  Everything but the collective's transport is the N > 1 path of the driver's scaling run:
    per-rank seed / shard, one HIP runtime per process (torch's, loaded first) x 2 processes, the lock-protected library load, the posted exchange per solve
    with its double buffering, barriers, all_reduce(MAX) of the elapsed time -- for the headline config, BASELINE config 3
    (walking sweep) and config 5 (three contacts) -- and EVERY rank checks its own shard against the oracle."""
    d = _torchrun(2, 29531, "--steps", "4", "--warmup", "1", "--batch", str(batch), "--check", "24", "--backend", "gloo",
                  *extra)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2 * batch and d["config"]["batch_per_gpu"] == batch
    assert d["config"]["exchange_selfcheck_ok"] is True
    assert "gloo" in d["config"]["exchange_backend"]
    assert d["config"]["contacts"] == (3 if "--contacts" in extra else 2)
    assert abs(d["value"] - 2 * batch * 4 / (d["ms_per_step"] * 4e-3)) / d["value"] < 1e-6  # whole-job units / max-over-ranks time
    p = d["parity"]
    assert p["ranks_checked"] == 2 and p["checked"] == 48 and len(p["per_rank_max_rel_force_err"]) == 2
    assert p["max_rel_force_err_vs_qpoases"] < 1e-4 and p["max_rel_objective_gap"] < 1e-4
    assert p["not_ok_over_all_shards"] == 0 and p["qpoases_failed"] == 0
    assert d["solver"]["failed_over_all_ranks"] == 0 and len(d["solver"]["kernel_ms_per_rank"]) == 2
    if name == "headline_2contact":
        e2e = p["vs_reference_source_end_to_end"]
        assert e2e["checked"] == 48 and e2e["max_rel_force_err"] < 7.2e-4  # cond(H) x binary32 round-off (test_reference_source.py)
    if name == "cfg5_three_contact":
        assert p["vs_reference_source_end_to_end"] is None  # the reference has no code for this shape
        assert "180x240" in d["config"]["workload"]


def test_bench_gpus_n_without_gpu_is_an_error_not_a_one_gpu_run():
    if _has_gpu():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


@pytest.mark.gpu
def test_bench_gpus_n_launches_n_ranks_by_itself():
    """`python bench.py --gpus 2` WITHOUT torchrun: bench.py re-executes itself under
    torch.distributed.run and the line says n_gpus = 2 with one parity block per rank -- on the one-GPU test box over the gloo
    TEST transport (both ranks on device 0)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "4",
                        "--warmup", "1", "--batch", "1024", "--check", "16"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["world"] == 2 and d["config"]["global_batch"] == 2048
    assert "re-executed itself" in d["config"]["launcher"]
    assert len(d["config"]["rank_devices"]) == 2 and len(d["solver"]["kernel_ms_per_rank"]) == 2
    assert d["parity"]["ranks_checked"] == 2 and d["parity"]["max_rel_force_err_vs_qpoases"] < 1e-4
    assert d["config"]["exchange_selfcheck_ok"] is True and d["solver"]["failed_over_all_ranks"] == 0


@pytest.mark.gpu
def test_bench_more_gpus_than_visible_fails_loudly():
    """--gpus 8 on a box with fewer devices: non-zero exit and the reason on stderr, never an n_gpus line from fewer GPUs."""
    import torch

    n = torch.cuda.device_count()
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 7), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "visible" in r.stderr and f"--gpus {n + 7}" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    # ... and --gpus that contradicts the torchrun it was launched under is an error as well
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_self_launch_builds_the_torchrun_command(monkeypatch):
    """bench.py --gpus N without WORLD_SIZE: the re-exec goes through torch.distributed.run with N ranks on 127.0.0.1, passes the
    caller's own arguments on, marks the children (HMPC_BENCH_SELF_LAUNCHED) and refuses when fewer devices are visible -- checked
    here without a GPU by standing in for the device count and the process launch."""
    import argparse
    import importlib.util

    import torch

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    argv = ["--gpus", "4", "--contacts", "3", "--batch", "2048", "--steps", "5"]
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    rc = bench.self_launch(argparse.Namespace(gpus=4, backend="nccl"), argv)
    assert rc == 7  # the ranks' exit code is passed on
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv):] == argv and os.path.basename(cmd[-len(argv) - 1]) == "bench.py"
    assert seen["env"]["HMPC_BENCH_SELF_LAUNCHED"] == "1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # fewer devices than ranks: refused (exit code 2, nothing launched) -- unless the ranks may share devices (gloo test transport)
    seen.clear()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.self_launch(argparse.Namespace(gpus=4, backend="nccl"), argv) == 2 and not seen
    assert bench.self_launch(argparse.Namespace(gpus=4, backend="gloo"), argv) == 7 and seen


def test_rank_bring_up_fails_with_diagnostics_within_its_timeout(tmp_path):
    """a rank that cannot bring its process group up (here: WORLD_SIZE = 2 with only rank 0 ever started, so
    the rendezvous never completes) must not sit in the default 10-minute time-outs: bench.py's bring_up_process_group prints the
    rank's device list, the RCCL version and the HSA_* / NCCL_* / MASTER_* environment to stderr and the PROCESS exits 3 -- within
    the bring-up time-out plus the watchdog's margin.  Runs without a GPU (gloo control plane, the function under test is the same)."""
    import socket
    import time

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    code = (
        "import importlib.util, os, sys\n"
        f"spec = importlib.util.spec_from_file_location('bench_under_test', {os.path.join(ROOT, 'bench.py')!r})\n"
        "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "import torch, torch.distributed as dist\n"
        "b.bring_up_process_group(torch, dist, 'gloo', 0, 0, 4.0)\n"
        "print('UNEXPECTED: bring-up returned')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0",
               NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    took = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    assert took < 60, took
    assert "UNEXPECTED" not in r.stdout
    err = r.stderr
    assert "[bench] rank 0 (LOCAL_RANK 0): ERROR" in err and "process group" in err
    assert "device(s) visible" in err or "device query failed" in err
    assert "NCCL_DEBUG=WARN" in err and "HSA_ENABLE_IPC_MODE_LEGACY=0" in err and f"MASTER_PORT={port}" in err and "WORLD_SIZE=2" in err
    assert "--exchange none" in err  # (the hint at the diagnostic mode)


@pytest.mark.gpu
def test_bench_exchange_none_is_a_diagnosable_line_without_any_collective():
    """`--exchange none`: N ranks, NO data-path collective, control plane on gloo -- what to run when the RCCL
    exchange of the default mode cannot be brought up.  Two ranks on the one GPU of the test box: the line carries both ranks' kernel
    times and parity blocks, and says that nothing was gathered."""
    d = _torchrun(2, 29547, "--steps", "4", "--warmup", "1", "--batch", "1024", "--check", "16", "--backend", "gloo", "--exchange", "none",
                  "--windows", "3")
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2048
    assert "NO exchange" in d["config"]["parallelism"] and d["config"]["exchange_backend"].startswith("none")
    assert "exchange_selfcheck_ok" not in d["config"]
    assert len(d["solver"]["kernel_ms_per_rank"]) == 2 and d["parity"]["ranks_checked"] == 2
    assert d["parity"]["max_rel_force_err_vs_qpoases"] < 1e-4 and d["solver"]["failed_over_all_ranks"] == 0
    assert d["windows"]["n"] == 3 and len(d["windows"]["values"]) == 3
    assert d["windows"]["value_min"] <= d["value"] <= d["windows"]["value_max"]
