#!/usr/bin/env python3
"""Developer tool: what the continuation pass alone does (HMPC_DEBUG_CONT_ONLY=1: the safe pass behind it is skipped).
    python scripts/dev/cont_probe.py [scale] [batch]"""
import os
import sys

import numpy as np

os.environ["HMPC_DEBUG_CONT_ONLY"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
h = 10
cnt = lambda c: {int(k): int(v) for k, v in zip(*np.unique(c, return_counts=True))}
rec = records.pack_records(synthetic.hard_batch(B, h, "standing", 17, scale), h)
m0 = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B)
m0.set_auto_resolve(False)
m0.upload(rec)
m0.solve()
_, st0 = m0.download()
t0 = m0.time_solve(1)
m0.close()
m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, B)
m.set_auto_resolve(False)
m.set_device_repair(True)
m.upload(rec)
m.solve()
_, st1 = m.download()
t1 = m.time_solve(1)
dbg = None
if "HMPC_DEBUG_STATS" in os.environ.get("HMPC_EXTRA_FLAGS", ""):
    interface._check(m.L.hmpc_enable_f64_output(m.h), "f64")
    m.solve()
    torch.cuda.synchronize()
    _, dbg = m.download_f64()
m.close()
full = interface.status_code(st0) == 5
print(f"x{scale:g} b{B}: fast pass {t0:.3f} ms {cnt(interface.status_code(st0))}; fast + continuation {t1:.3f} ms {cnt(interface.status_code(st1))}")
it0, it1 = interface.status_iters(st0)[full], interface.status_iters(st1)[full]
print(f"  handed over: {full.sum()}; iterations at hand-over mean {it0.mean():.1f}; after continuation mean {it1.mean():.1f} (added {np.mean(it1 - it0):.1f}, p50 {np.median(it1 - it0):.0f}, p90 {np.percentile(it1 - it0, 90):.0f}, max {(it1 - it0).max()})")
print(f"  |W| after continuation: mean {interface.status_nactive(st1)[full].mean():.1f} max {interface.status_nactive(st1)[full].max()}; codes of the handed-over {cnt(interface.status_code(st1)[full])}")
if dbg is not None:
    hit = dbg[full] < 0
    if hit.any():
        v = -dbg[full][hit]
        print(f"  budget ran out in {hit.sum()} resumed solves; most violated row (unit scale) at that point: min {v.min():.1e} p25 {np.percentile(v, 25):.1e} "
              f"median {np.median(v):.1e} p75 {np.percentile(v, 75):.1e} max {v.max():.1e}; their final codes {cnt(interface.status_code(st1)[full][hit])}")
    d = np.where(dbg[full] < 0, 0, dbg[full]).astype(np.int64)
    bad, rounds, nocap, nofew, dep = d % 10, (d // 10) % 100, (d // 1000) % 10, (d // 10000) % 10, d // 100000
    added = it1 - it0
    for name, sel in (("all handed over", np.ones_like(added, dtype=bool)), ("added > 60", added > 60), ("not ok", interface.status_code(st1)[full] != 0)):
        if sel.any():
            print(f"  [{name}: {sel.sum()}] bad starts {bad[sel].mean():.2f} (any: {(bad[sel] > 0).mean():.2f}) rounds run {rounds[sel].mean():.2f} refused(capacity) {nocap[sel].mean():.2f} "
                  f"refused(few new) {nofew[sel].mean():.2f} rows set aside as dependent {dep[sel].mean():.2f}; added p50 {np.median(added[sel]):.0f}")
