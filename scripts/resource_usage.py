#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy report of the product library's device code (hipcc cross-compiles: no GPU).

    python scripts/resource_usage.py [extra hipcc flags]        e.g.  -DHMPC_QCAP_3C=88
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-c",
       "-Wno-unused-value", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:] + \
      [os.path.join(ROOT, "hector_simulation_amd", "csrc", "hmpc_capi.hip"), "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
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
