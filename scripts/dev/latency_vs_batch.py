#!/usr/bin/env python3
"""Developer tool: kernel time vs batch size for the metric's 2-contact case -- the per-instance dependency chain that bounds
small batches (one round of workgroups up to 768 instances) -- in natural dispatch order, ordered by the record-only cost
predictor (a cold handle), and ordered by the previous solve of the same batch.  20 back-to-back launches, best of 3."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

print(f"{'batch':>6s} {'natural':>22s} {'predicted order':>24s} {'previous-solve order':>24s}   active-set iterations of the batch")
one_round = {}  # batch -> (natural-order ms, largest iteration count), batches that are one round of resident workgroups (<= 768)
for nb in (64, 256, 512, 768, 1024, 1536, 2304):
    f = synthetic.make_batch(nb, 10, "standing", seed=2, phase="random")
    rec = records.pack_records(f, 10)
    row = []
    for mode in (0, 2, 1):
        m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb)
        m.set_dispatch_order(mode)
        m.upload(rec)
        m.solve()
        _, st = m.download()
        itmax, itmean = int(interface.status_iters(st).max()), float(interface.status_iters(st).mean())
        row.append(min(m.time_solve(20) for _ in range(3)))
        m.close()
    # (up to 768 instances are ONE round of workgroups: the launch lasts as long as its slowest instance's dependency chain, i.e. it
    #  follows the batch's LARGEST iteration count -- every batch size is its own random draw -- not the batch size)
    print(f"b{nb:5d} " + "   ".join(f"{ms:.4f} ms {nb / ms / 1e3:6.3f} M/s" for ms in row) + f"   iterations mean {itmean:.2f} max {itmax}", flush=True)
    if nb <= 768:
        one_round[nb] = (row[0], itmax)
# Why a smaller batch can be slower than a larger one (b256 vs b512), from this run's own numbers
(t64, i64), (t512, i512), (t256, i256) = one_round[64], one_round[512], one_round[256]
slope = (t512 - t64) / max(i512 - i64, 1)
print(f"\n# Up to 768 instances are ONE round of resident workgroups, so a launch lasts as long as the dependency chain of its slowest instance, and that")
print(f"# follows the LARGEST active-set iteration count in the batch -- every batch size here is its own random draw -- not the batch size:")
print("# " + ", ".join(f"b{nb} (max {it} iterations) {ms:.3f} ms" for nb, (ms, it) in sorted(one_round.items())) + ".")
print(f"# The line through b64 and b512: {1e3 * slope:.1f} us per iteration of the slowest instance on top of {t64 - slope * i64:.3f} ms of assembly + inverse + block start;")
print(f"# it predicts {t64 + slope * (i256 - i64):.3f} ms for b256's {i256} iterations (measured {t256:.3f} ms).")
