#!/usr/bin/env python3
"""Developer tool: what an off-nominal batch costs.  2-contact standing h=10 (or argv gait/h), b8192, input ranges at 1x/3x/6x/10x
(synthetic.hard_batch): kernel time of the fast pass alone, of the whole device-side pipeline (hmpc_set_device_repair) with the
hand-over of full working sets on and off, flagged fractions, statuses after the device passes.
    python scripts/dev/range_scale.py [batch] [gait] [h]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
gait = sys.argv[2] if len(sys.argv) > 2 else "standing"
h = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cnt = lambda c: {int(k): int(v) for k, v in zip(*np.unique(c, return_counts=True))}
for scale in (1, 3, 6, 10):
    fs = synthetic.hard_batch(B, h, gait, 17, scale)
    rec_a = records.pack_records(fs, h)
    rec_b = records.pack_records(synthetic.advance_tick(fs, h, seed=9), h)
    row = {}
    for mode in ("fast_only", "repair_handover", "repair_cold"):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B)
        m.set_auto_resolve(False)
        m.set_device_repair(mode != "fast_only")
        m.set_handover(mode != "repair_cold")
        ts = []
        for _ in range(3):
            m.upload(rec_a)
            m.solve()
            torch.cuda.synchronize()
            m.upload(rec_b)
            ts.append(m.time_solve(1))
        _, st = m.download()
        m.close()
        row[mode] = (min(ts), cnt(interface.status_code(st)), float(interface.status_iters(st).mean()), int(interface.status_nactive(st).max()))
    t1 = row["fast_only"][0]
    print(f"{gait} h={h} b{B} x{scale:<2d} fast-only {t1:.3f} ms {row['fast_only'][1]} | device repair, hand-over: {row['repair_handover'][0]:.3f} ms "
          f"= {B / row['repair_handover'][0] / 1e3:.2f} M/s {row['repair_handover'][1]} iters {row['repair_handover'][2]:.1f} |W|max {row['repair_handover'][3]} | "
          f"cold re-solve: {row['repair_cold'][0]:.3f} ms = {B / row['repair_cold'][0] / 1e3:.2f} M/s {row['repair_cold'][1]} iters {row['repair_cold'][2]:.1f}", flush=True)
