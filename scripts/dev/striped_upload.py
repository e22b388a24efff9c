#!/usr/bin/env python3
"""Developer tool: host -> device upload time of a device group, contiguous slices against the striped deal (one
hipMemcpy2DAsync per member, width = one record), pageable host memory.    python scripts/dev/striped_upload.py [batch] [members]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rec = records.pack_records(synthetic.make_batch(B, 10, "standing", seed=3, phase="random"), 10)
g = interface.DeviceGroup(synthetic.DT_MPC, 10, synthetic.F_MAX, B, [0] * G, transport="p2p")
for striped in (False, True, False, True):
    g.set_deal(striped)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        g.upload(rec)
        g.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"b{B} x {G} members, {'striped' if striped else 'contiguous'} deal: upload + sync {1e3 * min(ts[1:]):.3f} ms (median {1e3 * sorted(ts[1:])[2]:.3f})", flush=True)
g.close()
