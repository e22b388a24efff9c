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
    """PyTorch-ROCm bundles its own libamdhip64 while libhector_mpc_hip.so links the system one: two HIP runtimes in one
    process.  That works when torch's runtime comes up first (the order bench.py uses) -- initialising it after ours has
    been observed to fail with "no ROCm-capable device".  So on a GPU box the test session brings torch.cuda up first."""
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
