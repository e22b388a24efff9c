#!/usr/bin/env python3
"""Developer tool: the instances the continuation pass leaves for the safe pass at `scale` x the input ranges -- their status words after
the continuation alone (hmpc_set_device_repair 2) and after the whole device chain (1): iterations and |W| of the safe pass's solves.
    python scripts/dev/safe_tail.py [scale] [batch] [gait] [h]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
gait = sys.argv[3] if len(sys.argv) > 3 else "standing"
h = int(sys.argv[4]) if len(sys.argv) > 4 else 10
rec = records.pack_records(synthetic.hard_batch(nb, h, gait, 17, scale), h)
st = {}
for mode in (2, 1):
    m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
    m.set_auto_resolve(False)
    m.set_device_repair(mode)
    m.upload(rec)
    m.solve()
    _, s = m.download()
    ms = min(m.time_solve(1) for _ in range(3))
    st[mode] = (s.copy(), ms)
    m.close()
left = interface.status_code(st[2][0]) != 0
it2, it1 = interface.status_iters(st[2][0])[left], interface.status_iters(st[1][0])[left]
print(f"{gait} h={h} x{scale} b{nb}: continuation only {st[2][1]:.3f} ms, whole chain {st[1][1]:.3f} ms; left for the safe pass {int(left.sum())}: codes {dict(zip(*np.unique(interface.status_code(st[2][0])[left], return_counts=True)))}")
print("  iterations when the continuation gave up:", np.sort(it2)[::-1][:40])
print("  iterations of the safe pass's solve      :", np.sort(it1)[::-1][:40], "codes", dict(zip(*np.unique(interface.status_code(st[1][0])[left], return_counts=True))))
print("  |W| after the safe pass:", np.sort(interface.status_nactive(st[1][0])[left])[::-1][:40])
idx = np.nonzero(left)[0]
order = np.argsort(-interface.status_iters(st[1][0])[idx])
print("  per instance (sorted by the safe pass's iterations): instance, continuation code / iterations / |W|, safe pass iterations / |W|")
for i in idx[order][:40]:
    print(f"   {i:5d}  cont code {interface.status_code(st[2][0])[i]} it {interface.status_iters(st[2][0])[i]:4d} |W| {interface.status_nactive(st[2][0])[i]:3d}   safe it {interface.status_iters(st[1][0])[i]:4d} |W| {interface.status_nactive(st[1][0])[i]:3d}")
