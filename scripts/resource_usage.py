#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy report of the product library's device code (hipcc cross-compiles: no GPU).

    python scripts/resource_usage.py [extra hipcc flags]        e.g.  -DHMPC_QCAP_3C=88
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import concurrent.futures  # noqa: E402

from hector_simulation_amd import build as hip_build  # noqa: E402

# the product's own translation units and flags (hector_simulation_amd/build.py), each with the resource-usage remarks on
extra = [a for a in sys.argv[1:] if a != "--all"]
units = hip_build.compile_commands("/tmp", "/opt/rocm/bin/hipcc")
cmds = [c[:1] + ["-Rpass-analysis=kernel-resource-usage"] + extra + c[1:-1] + ["/dev/null"] for _, c in units]
with concurrent.futures.ThreadPoolExecutor(max_workers=len(cmds)) as ex:
    out = "\n".join(ex.map(lambda c: subprocess.run(c, capture_output=True, text=True).stderr, cmds))
pats = {"vgpr": r" VGPRs: (\d+)", "agpr": r"AGPRs: (\d+)", "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
        "spill": r"VGPRs? Spill: (\d+)", "occ": r"Occupancy \[waves/SIMD\]: (\d+)", "lds": r"LDS Size \[bytes/block\]: (\d+)"}
name, row = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name, row = m.group(1), {}
        continue
    for k, p in pats.items():
        m = re.search(p, line)
        if m:
            row[k] = int(m.group(1))
    if "LDS Size" in line and name:
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"^void hmpc::", "", short).replace("(hmpc::KernelArgs)", "")
        if "true" not in short or "--all" in sys.argv:
            print(f"{short:64s} vgpr {row.get('vgpr'):>3} agpr {row.get('agpr'):>3} spill {row.get('spill'):>3} "
                  f"scratch {row.get('scratch'):>3} B  occupancy {row.get('occ')}")
        name = None
if "error" in out:
    print(out[-3000:])
