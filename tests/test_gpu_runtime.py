"""Which HIP runtime the library runs on under Python: exactly ONE libamdhip64 is mapped into the process --
torch's bundled one, because torch comes up first (hector_simulation_amd/_lib.py) and the library's DT_NEEDED names the same SONAME --
and the product library is among the mapped objects (no silent fallback path)."""
import os
import re

import pytest

pytestmark = pytest.mark.gpu


def _mapped(pattern: str) -> set:
    paths = set()
    with open("/proc/self/maps") as f:
        for line in f:
            m = re.search(r"(/\S*" + pattern + r"\S*)", line)
            if m:
                paths.add(os.path.realpath(m.group(1)))
    return paths


def test_one_hip_runtime_serves_torch_and_the_library():
    import torch

    from hector_simulation_amd import _lib, interface, records, synthetic

    assert torch.cuda.is_available()
    _lib.load()
    m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, 4)  # (the library has really talked to the runtime)
    m.upload(records.pack_records(synthetic.make_batch(4, 10, "standing", seed=1), 10))
    m.solve()
    _, st = m.download()
    m.close()
    assert (interface.status_code(st) == 0).all()
    hip = _mapped(r"libamdhip64\.so")
    assert len(hip) == 1, hip                                   # one runtime in the process, not two
    assert os.path.dirname(torch.__file__) in next(iter(hip))   # ... torch's bundled copy (it was loaded first)
    assert any(p.endswith("libhector_mpc_hip.so") for p in _mapped(r"libhector_mpc_hip\.so"))
    # the library asks for the runtime by SONAME (its RUNPATH into /opt/rocm only matters when nothing with that SONAME is mapped yet:
    # a C++ host without torch)
    needed = os.popen(f"readelf -d {_lib.lib_path()}").read()
    assert re.search(r"NEEDED.*libamdhip64\.so\.\d+", needed)
