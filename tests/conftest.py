import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """ONE HIP runtime serves the whole Python process: PyTorch-ROCm bundles a libamdhip64.so.7 (ROCm 7.0) and libhector_mpc_hip.so
    names the same SONAME in its DT_NEEDED (it is built against the ROCm 7.2 headers of /opt/rocm), so whichever copy the dynamic
    loader maps FIRST is the one both use.  torch first (the order bench.py uses) = torch's bundled runtime for both, which works;
    our library first = the system runtime for both, and torch on a runtime it was not built with has been observed to fail with
    "no ROCm-capable device".  So on a GPU box the test session brings torch.cuda up first; tests/test_gpu_runtime.py asserts
    that exactly one libamdhip64 is mapped and that it is torch's.  (A C++ host such as the reference controller has no torch in
    the process: the library then runs on the system runtime it was built against.)"""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/): the checker.  Never imported by the product package."""
    from oracle import oracle_py

    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_available():
    import torch

    return torch.cuda.is_available()
