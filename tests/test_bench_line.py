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
                        "--no-side-configs", "--check", "16", "--cpu-per-core", "2"], capture_output=True, text=True,
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
