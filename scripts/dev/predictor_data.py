#!/usr/bin/env python3
"""Developer tool: (record fields, iteration counts, working-set sizes) of solved batches -> gpurun_out/pred/*.npz, the data the
cold-handle dispatch-order predictor (cost_hint_kernel, hmpc_builder.h) was fitted on (scripts/dev/fit_predictor.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "pred")
os.makedirs(OUT, exist_ok=True)


hard_batch = synthetic.hard_batch  # (nb, h, gait, seed, scale): the off-nominal stress rows


def run(name, f, h, nc):
    rec = records.pack_records(f, h, nc)
    nb = rec.shape[0]
    m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb, contacts=nc)
    m.set_dispatch_order(False)
    m.upload(rec)
    m.solve()
    forces, st = m.download()
    ms = min(m.time_solve(5) for _ in range(2))
    m.close()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), rec=rec, iters=interface.status_iters(st), nact=interface.status_nactive(st),
                        code=interface.status_code(st), h=h, nc=nc, ms_natural=ms)
    it = interface.status_iters(st)
    print(f"{name:28s} nb {nb} ms {ms:.4f} iters mean {it.mean():.2f} p50 {np.median(it):.0f} p90 {np.percentile(it, 90):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()}", flush=True)


for seed in (6, 1006, 2):
    f = synthetic.make_batch(8192, 10, "standing", seed=seed, phase="random")
    run(f"standing_h10_s{seed}", f, 10, 2)
run("standing_h10_s6_next", synthetic.advance_tick(synthetic.make_batch(8192, 10, "standing", seed=6, phase="random"), 10, seed=7), 10, 2)
run("standing_h10_s2_b1024", synthetic.make_batch(1024, 10, "standing", seed=2, phase="random"), 10, 2)
for seed in (5, 6):
    run(f"3contact_s{seed}", synthetic.make_batch3(8192, 10, "standing", seed=seed, phase="random", hand="contact"), 10, 3)
run("3contact_s5_b2048", synthetic.make_batch3(2048, 10, "standing", seed=5, hand="contact"), 10, 3)
run("single_h20_s2", synthetic.make_batch(4096, 20, "single", seed=2, phase="random"), 20, 2)
run("standing_h20_s2", synthetic.make_batch(2048, 20, "standing", seed=2, phase="random"), 20, 2)
run("walking_h10_s6", synthetic.make_batch(8192, 10, "walking", seed=6, phase="random"), 10, 2)
run("mixed_h10_s3", synthetic.make_batch(4096, 10, "mixed", seed=3, phase="random"), 10, 2)
run("standing_h10_x3", hard_batch(2048, 10, "standing", 17, 3), 10, 2)
